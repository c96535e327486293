#!/bin/bash
# Builds pathtracer_amd/variants/libptk_<name>.so from the sources of a git revision (default HEAD): the "before" side
# of a same-box A/B run against the working tree.  Usage: tools/build_base_variant.sh [name] [revision] [flags]
set -e
cd "$(dirname "$0")/.."
NAME=${1:-base}; REV=${2:-HEAD}; FLAGS=${3:-}
T=$(mktemp -d)
git archive $REV pathtracer_amd/csrc include | tar -x -C $T
mkdir -p pathtracer_amd/variants
( cd $T/pathtracer_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -fno-slp-vectorize -mllvm --sink-insts-to-avoid-spills -mllvm --amdgpu-schedule-relaxed-occupancy $FLAGS \
    -o $OLDPWD/pathtracer_amd/variants/libptk_${NAME}.so ptk_kernels.hip ptk_build.cpp ptk_lbvh.hip ptk_group.hip -ldl )
rm -rf $T
ls -la pathtracer_amd/variants/libptk_${NAME}.so
