#!/bin/bash
# Memory-pipeline counter passes (texture addresser / L1 / data return) on a short bench.py run.
# Usage: tools/profile_mem.sh <tag> [bench args...]
set -u
TAG=${1:-mem}; shift || true
ARGS="${@:---spp 32 --steps 2 --warmup 1 --no-cpu}"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
i=0
for PMC in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TAGRAM0_REQ_sum TCP_LFIFO_STALL_CYCLES_sum SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -o pmc -- python3 bench.py $ARGS > $OUT/pmc${i}_bench.json 2> $OUT/pmc$i.err || { echo "pmc pass $i failed"; tail -5 $OUT/pmc$i.err; exit 1; }
done
python3 tools/summarize_prof.py $OUT > gpurun_out/prof_${TAG}_summary.json
echo "profile_mem done"
