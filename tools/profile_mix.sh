#!/bin/bash
# Runs on the GPU box: the vector instruction mix of a short bench.py run (one counter pass).
# Usage: tools/profile_mix.sh <tag> [bench args...]
set -u
TAG=${1:-mix}; shift || true
ARGS="${@:---workload bunny --spp 256 --steps 2 --warmup 1 --no-cpu}"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -o pmc -- python3 bench.py $ARGS > $OUT/pmc1_bench.json 2> $OUT/pmc1.err || { echo "pmc pass failed"; tail -5 $OUT/pmc1.err; exit 1; }
echo "profile done"
