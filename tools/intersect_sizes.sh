#!/bin/bash
# On the GPU box: time of one ray batch against its size -> gpurun_out/isizes.txt
cd $GRAFT_REPO_ROOT
: > gpurun_out/isizes.txt
for n in 4096 65536 393216 1048576 4194304 16777216 33554432 67108864; do
  timeout -k 10 120 python3 tools/bench_intersect.py --level 6 --reps 10 --rays $n 2>&1 | grep -v amdgpu.ids >> gpurun_out/isizes.txt || exit 1
done
cat gpurun_out/isizes.txt
