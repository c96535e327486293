#!/bin/bash
# On the GPU box: the device-built tree of each library named on stdin ("-" = pathtracer_amd/libptk.so, else a file
# under pathtracer_amd/variants/) against the host-built tree -> gpurun_out/ab_builder.txt
# (frame rate, box and primitive tests per ray, build time).  Arguments: workloads (default bunny bmw).
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_builder.txt
: > $O
LIBS=$(cat)
for w in ${@:-bunny bmw}; do
  for lib in host $LIBS; do
    L=$PWD/pathtracer_amd/libptk.so; B="--builder device"
    [ "$lib" = "host" ] && B="--builder host"
    [ "$lib" != "-" ] && [ "$lib" != "host" ] && L=$PWD/pathtracer_amd/variants/$lib
    PTK_LIB=$L timeout -k 10 300 python3 bench.py --workload $w --spp 256 --steps 3 --warmup 1 --no-cpu $B > gpurun_out/_abb.json 2> gpurun_out/_abb.err || { echo "FAILED $lib $w"; tail -3 gpurun_out/_abb.err; exit 1; }
    python3 - "$lib" <<'PY' >> $O
import json,sys
d=json.loads(open('gpurun_out/_abb.json').read().strip().splitlines()[-1])
r=d["roofline"]; c=d["config"]
print("%-18s %-8s Mrays/s %8.1f  box tests/ray %.2f  prim tests/ray %.2f  nodes %d depth %d form %d  build %.2f ms upload %.1f ms (%s)" % (sys.argv[1], c["workload"].split(":")[0], d["value"], r["box_tests_counted"], r["prim_tests_counted"], c["bvh_nodes"], c["bvh_depth"], c["traversal_kind"], c["device_build_ms"], c["scene_upload_ms"], c["bvh_builder"]))
PY
  done
done
cat $O
