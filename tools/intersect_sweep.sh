#!/bin/bash
# On the GPU box: parity test of the ray-batch entry point, then its rate for every library under
# pathtracer_amd/variants/ on random and on camera-like rays -> gpurun_out/isweep.txt
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -u -m pytest tests/test_parity_intersect.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/isweep_test.log 2>&1 || { tail -20 gpurun_out/isweep_test.log; exit 1; }
tail -1 gpurun_out/isweep_test.log
: > gpurun_out/isweep.txt
for lib in pathtracer_amd/variants/*.so; do
  for mode in "" "--coherent" "--rays 67108864"; do
    echo -n "$(basename $lib) " >> gpurun_out/isweep.txt
    PTK_LIB=$PWD/$lib timeout -k 10 120 python3 tools/bench_intersect.py --level 6 --reps 10 $mode 2>&1 | grep -v amdgpu.ids >> gpurun_out/isweep.txt || { tail -5 gpurun_out/isweep.txt; exit 1; }
  done
done
cat gpurun_out/isweep.txt
