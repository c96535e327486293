#!/bin/bash
# Builds tuning variants of libptk.so.  Each argument is name=FLAGS, e.g.
#   tools/build_variants.sh "base=" "w4=-DPTK_MIN_WAVES=4"
set -e
cd "$(dirname "$0")/../pathtracer_amd/csrc"
mkdir -p ../variants
for v in "$@"; do
  NAME=${v%%=*}; FLAGS=${v#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared ${BASEFLAGS--fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -fno-slp-vectorize -mllvm --sink-insts-to-avoid-spills -mllvm --amdgpu-schedule-relaxed-occupancy} $FLAGS \
    -o ../variants/libptk_${NAME}.so ptk_kernels.hip ptk_build.cpp ptk_lbvh.hip ptk_group.hip -ldl &
done
wait
ls ../variants
