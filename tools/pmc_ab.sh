#!/bin/bash
# On the GPU box: one hardware-counter pass per library variant, counters per render launch printed side by side.
#   tools/pmc_ab.sh "<COUNTERS>" [bench args]      (variants = pathtracer_amd/variants/*.so)
set -u
PMC=$1; shift
ARGS="${@:---spp 128 --steps 2 --warmup 1 --no-cpu}"
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for lib in pathtracer_amd/variants/*.so; do
  name=$(basename $lib .so)
  OUT=gpurun_out/prof_ab_$name
  rm -rf $OUT; mkdir -p $OUT
  export PTK_LIB=$PWD/$lib
  timeout -s KILL 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc1 -o pmc -- python3 bench.py $ARGS > $OUT/pmc1_bench.json 2> $OUT/pmc1.err || { echo "pmc pass failed for $name"; tail -5 $OUT/pmc1.err; exit 1; }
  python3 tools/summarize_prof.py $OUT > $OUT/summary.json
  python3 - $name $OUT/summary.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2]))
for k,v in d["counters_per_dispatch"].items():
    if "Lb0E" in k or "<false>" in k:
        print(sys.argv[1], {a: float("%.4g" % b) for a,b in sorted(v.items())})
PY
done
