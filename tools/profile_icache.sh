#!/bin/bash
# Runs on the GPU box: instruction-cache and scalar-cache counters of a short bench.py run (one counter pass each).
# Usage: tools/profile_icache.sh <tag> [bench args...]
set -u
TAG=${1:-icache}; shift || true
ARGS="${@:---workload bunny --spp 256 --steps 2 --warmup 1 --no-cpu}"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
i=0
for PMC in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -o pmc -- python3 bench.py $ARGS > $OUT/pmc${i}_bench.json 2> $OUT/pmc$i.err || { echo "pmc pass $i failed"; tail -5 $OUT/pmc$i.err; exit 1; }
done
echo "profile done"
