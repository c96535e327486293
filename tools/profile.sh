#!/bin/bash
# Runs on the GPU box: kernel trace + stats, then hardware-counter passes (one rocprofv3 run per pass),
# all on a short run of bench.py.  Usage: tools/profile.sh <tag> [bench args...]
# With PROFILE_CMD set, that program (a python script and its arguments) is profiled instead of bench.py.
set -u
TAG=${1:-r1}; shift || true
ARGS="${@:---workload bunny --spp 32 --steps 2 --warmup 1 --no-cpu}"
export TMPDIR=/tmp
CMD=${PROFILE_CMD:-bench.py}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $CMD $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err || { echo "trace failed"; tail -5 $OUT/trace.err; exit 1; }
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -o pmc -- python3 $CMD $ARGS > $OUT/pmc${i}_bench.json 2> $OUT/pmc$i.err || { echo "pmc pass $i failed"; tail -5 $OUT/pmc$i.err; exit 1; }
done
echo "profile done"
