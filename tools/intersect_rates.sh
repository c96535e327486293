#!/bin/bash
# On the GPU box: the ray-batch entry point's rate on random and camera-like rays and against batch size
# -> gpurun_out/intersect_rates.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/intersect_rates.txt
: > $O
for mode in "" "--coherent" "--rays 67108864" "--rays 67108864 --coherent" "--level 7" "--level 7 --coherent"; do
  timeout -k 10 120 python3 tools/bench_intersect.py --level 6 --reps 10 $mode 2>&1 | grep -v amdgpu.ids >> $O || exit 1
done
for n in 4096 65536 393216 1048576 4194304 33554432; do
  timeout -k 10 120 python3 tools/bench_intersect.py --level 6 --reps 10 --rays $n 2>&1 | grep -v amdgpu.ids >> $O || exit 1
done
cat $O
