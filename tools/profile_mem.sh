#!/bin/bash
# Runs on the GPU box: counter passes on the vector-memory path (address unit TA, data unit TD, L1 TCP) of a short
# bench.py run, one rocprofv3 run per pass.  Usage: tools/profile_mem.sh <tag> [bench args...]
set -u
TAG=${1:-mem}; shift || true
ARGS="${@:---workload bunny --spp 256 --steps 2 --warmup 1 --no-cpu}"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
i=0
for PMC in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TD_TD_BUSY_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum" \
           "TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -o pmc -- python3 bench.py $ARGS > $OUT/pmc${i}_bench.json 2> $OUT/pmc$i.err || { echo "pmc pass $i failed"; tail -5 $OUT/pmc$i.err; exit 1; }
done
echo "profile done"
