#!/bin/bash
# On the GPU box: bench.py once per "LIB ENV=VALUE" line given on stdin (LIB = a file under pathtracer_amd/variants/
# or "-" for pathtracer_amd/libptk.so) -> gpurun_out/ab_env.txt.  Arguments: bench.py's.
cd $GRAFT_REPO_ROOT
ARGS="${@:---workload bunny --spp 256 --steps 3 --warmup 1 --no-cpu}"
O=gpurun_out/ab_env.txt
: > $O
while read -r lib envs; do
  [ -z "$lib" ] && continue
  L=$PWD/pathtracer_amd/libptk.so; [ "$lib" != "-" ] && L=$PWD/pathtracer_amd/variants/$lib
  ( export PTK_LIB=$L; for kv in $envs; do export "$kv"; done
    timeout -k 10 300 python3 bench.py $ARGS > gpurun_out/_ab_one.json 2> gpurun_out/_ab_one.err ) || { echo "FAILED $lib $envs"; tail -3 gpurun_out/_ab_one.err; exit 1; }
  python3 - "$lib $envs" <<'PY' >> $O
import json,sys
d=json.loads(open('gpurun_out/_ab_one.json').read().strip().splitlines()[-1])
print(sys.argv[1], "|", d["config"]["workload"].split(":")[0], d["config"]["workload"].split(";")[-1].strip(), "| Mrays/s", round(d["value"],1), "ms", round(d["ms_per_frame"],2))
PY
done
cat $O
