#!/bin/bash
# On the GPU box: tools/bench_intersect.py once per library variant.  Usage: tools/sweep_intersect.sh <tag> [args]
TAG=${1:-isect}; shift
ARGS="${@:---level 5}"
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sweep_$TAG.txt
: > $OUT
for lib in pathtracer_amd/variants/*.so; do
  for mode in "" "--coherent"; do
    echo -n "$(basename $lib) " >> $OUT
    PTK_LIB=$PWD/$lib timeout -k 10 200 python3 tools/bench_intersect.py $ARGS $mode >> $OUT 2> gpurun_out/_sweep_one.err || { echo "FAILED $lib"; tail -3 gpurun_out/_sweep_one.err; exit 1; }
  done
done
cat $OUT
