#!/bin/bash
# On the GPU box: per-phase wave-step, lane and clock counts of the render kernel (a -DPTK_PHASE_STATS build of the
# library, see tools/build_variants.sh "phase=-DPTK_PHASE_STATS") on the benchmark scenes
# -> gpurun_out/<tag>_phase_stats_<workload>.txt.  Arguments: tag, then the workloads (default bunny cornell bmw).
cd $GRAFT_REPO_ROOT
TAG=${1:-r3}; shift
LIB=${PHASE_LIB:-pathtracer_amd/variants/libptk_phase.so}
for w in ${@:-bunny cornell bmw}; do
  O=gpurun_out/${TAG}_phase_stats_$w.txt
  echo "# bench.py --workload $w --spp 256 --stats-spp 64 (instrumented launch: 64 spp, library $LIB)" > $O
  PTK_LIB=$PWD/$LIB PTK_PHASE_STATS=1 timeout -k 10 300 python3 bench.py --workload $w --spp 256 --steps 1 --warmup 0 --no-cpu --stats-spp 64 \
    > gpurun_out/_phase_one.json 2> gpurun_out/_phase_one.err || { echo "FAILED $w"; tail -5 gpurun_out/_phase_one.err; exit 1; }
  grep "ptk phase" gpurun_out/_phase_one.err >> $O
  python3 - <<'PY' >> $O
import json
d=json.loads(open('gpurun_out/_phase_one.json').read().strip().splitlines()[-1])
r=d["roofline"]
print("box tests per ray %.2f, primitive tests per ray %.2f, rays per segment %.3f, Mrays/s (instrumented build, 256 spp) %.0f" % (r["box_tests_per_ray"], r["prim_tests_per_ray"], d["rays_per_segment"], d["value"]))
PY
  cat $O
done
