#!/bin/bash
# On the GPU box: the ray-batch rate for a few grid sizes around the number of resident waves -> gpurun_out/igrid.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/igrid.txt
: > $O
for g in default 6400; do
  for mode in "" "--coherent"; do
    echo -n "grid $g: " >> $O
    if [ $g = default ]; then
      timeout -k 10 120 python3 tools/bench_intersect.py --level 6 --reps 20 $mode 2>&1 | grep -v amdgpu.ids >> $O || exit 1
    else
      PTK_INTERSECT_GRID=$g timeout -k 10 120 python3 tools/bench_intersect.py --level 6 --reps 20 $mode 2>&1 | grep -v amdgpu.ids >> $O || exit 1
    fi
  done
done
cat $O
