#!/bin/bash
# After tools/profile.sh / profile_mem.sh / profile_mix.sh have run for a workload (gpurun_out/prof_r3_<w>,
# prof_mem_<w>, prof_mix_<w>): merge the passes, write profiles/r3_final_<w>_* and the workload's profiles/traffic.json entry.
set -e
cd "$(dirname "$0")/.."
for w in "$@"; do
  if [ -d gpurun_out/prof_mem_$w ]; then for i in 1 2 3 4 5 6; do rm -rf gpurun_out/prof_r3_$w/pmc$((5+i)); cp -r gpurun_out/prof_mem_$w/pmc$i gpurun_out/prof_r3_$w/pmc$((5+i)); done; fi
  if [ -d gpurun_out/prof_mix_$w ]; then rm -rf gpurun_out/prof_r3_$w/pmc12; cp -r gpurun_out/prof_mix_$w/pmc1 gpurun_out/prof_r3_$w/pmc12; fi
  python3 tools/summarize_prof.py gpurun_out/prof_r3_$w > profiles/r3_final_${w}_summary.json
  cp gpurun_out/prof_r3_$w/trace/trace_kernel_stats.csv profiles/r3_final_${w}_kernel_stats.csv
  python3 tools/make_traffic.py profiles/r3_final_${w}_summary.json ${w}_1024x1024_1024spp > /dev/null
done
