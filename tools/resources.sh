#!/bin/bash
# Per-kernel register / spill / LDS / occupancy table of ptk_kernels.hip (compiler remarks).  Extra arguments: compiler flags.
cd "$(dirname "$0")/../pathtracer_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -fno-slp-vectorize -mllvm --sink-insts-to-avoid-spills -mllvm --amdgpu-schedule-relaxed-occupancy "$@" \
  -Rpass-analysis=kernel-resource-usage -c ptk_kernels.hip -o /dev/null 2>&1 | python3 -c '
import re,sys
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r"remark: [^ ]* *(Function Name|Name): (\S+)",line) or re.search(r"(Function Name|Name): (\S+)",line)
    if m:
        cur={"name":m.group(2)}; rows.append(cur); continue
    for key in ("VGPRs","AGPRs","SGPRs","ScratchSize [bytes/lane]","Occupancy [waves/SIMD]","SGPRs Spill","VGPRs Spill","LDS Size [bytes/block]"):
        m=re.search(re.escape(key)+r": (\d+)",line)
        if m and cur is not None and key not in cur: cur[key]=m.group(1)
import subprocess
for r in rows:
    n=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip().split("(")[0]
    print("%-52s vgpr %3s spill %3s sgpr-spill %3s scratch %4s lds %5s occ %s"%(n[:52],r.get("VGPRs"),r.get("VGPRs Spill"),r.get("SGPRs Spill"),r.get("ScratchSize [bytes/lane]"),r.get("LDS Size [bytes/block]"),r.get("Occupancy [waves/SIMD]")))
'
