#!/bin/bash
# On the GPU box: everything the committed profiles/ files of a round are made from, in one go.
#   counter passes (tools/profile*.sh) for the three scenes of the default bench, the default bench itself, the other
#   workloads, the ray-batch rates.  Then, back in the container: tools/collect_final.sh bunny cornell bmw
cd $GRAFT_REPO_ROOT
for w in bunny cornell bmw; do
  A="--workload $w --spp 1024 --steps 1 --warmup 1 --no-cpu"
  bash tools/profile.sh r3_$w $A && bash tools/profile_mem.sh mem_$w $A && bash tools/profile_mix.sh mix_$w $A || exit 1
  echo "profiled $w"
done
timeout -k 10 900 python3 bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err || exit 1
echo "default bench done"
timeout -k 10 300 python3 bench.py --workload envmap --no-cpu > gpurun_out/final_bench_envmap.json 2>> gpurun_out/final_bench.err || exit 1
timeout -k 10 300 python3 bench.py --workload glass --no-cpu > gpurun_out/final_bench_glass.json 2>> gpurun_out/final_bench.err || exit 1
timeout -k 10 300 python3 bench.py --workload bunny --builder device --no-cpu > gpurun_out/final_bench_bunny_device.json 2>> gpurun_out/final_bench.err || exit 1
timeout -k 10 300 python3 bench.py --workload bmw --builder device --no-cpu > gpurun_out/final_bench_bmw_device.json 2>> gpurun_out/final_bench.err || exit 1
echo "workloads done"
bash tools/intersect_rates.sh > /dev/null || exit 1
PROFILE_CMD=tools/bench_intersect.py bash tools/profile.sh r3_intersect --level 6 --reps 5
