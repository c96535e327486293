#!/bin/bash
# On the GPU box: run bench.py once per value of an environment variable.
#   tools/sweep_env.sh <tag> <VAR> "<v1 v2 ...>" [bench args]
TAG=$1; VAR=$2; VALUES=$3; shift 3
ARGS="${@:---workload bunny --spp 64 --steps 3 --warmup 1 --no-cpu}"
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sweep_$TAG.jsonl
: > $OUT
for v in $VALUES; do
  export $VAR=$v
  timeout -k 10 300 python3 bench.py $ARGS > gpurun_out/_sweep_one.json 2> gpurun_out/_sweep_one.err || { echo "FAILED $VAR=$v"; tail -3 gpurun_out/_sweep_one.err; exit 1; }
  python3 - "$VAR=$v" <<'PY' >> $OUT
import json,sys
d=json.loads(open('gpurun_out/_sweep_one.json').read().strip().splitlines()[-1])
print(json.dumps({"var":sys.argv[1],"workload":d["config"]["workload"].split(":")[0]+" "+d["config"]["workload"].split(";")[-1].strip(),"Mrays/s":round(d["value"],1),"ms_per_frame":round(d["ms_per_frame"],2),"kernel_ms":round(d["roofline"]["kernel_ms_avg"],2)}))
PY
done
cat $OUT
