#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE counters of a tools/profile.sh summary into the per-launch HBM traffic
figure that bench.py reports as roofline.traffic.

    tools/make_traffic.py <summary.json> <workload_WxH_SPPspp> [traffic.json]

rocprofv3 gives both counters in KiB per dispatch.  On gfx950 FETCH_SIZE tallies 128-byte requests as 64 bytes,
so reads are doubled (the correction measured for wide coalesced reads; the 64-byte node and 48-byte triangle
gathers of this kernel are not separately calibrated, which makes the figure an upper bound)."""
import json, sys

summary, key = json.load(open(sys.argv[1])), sys.argv[2]
path = sys.argv[3] if len(sys.argv) > 3 else "profiles/traffic.json"
per = [v for k, v in summary["counters_per_dispatch"].items() if "Lb0E" in k or "<false>" in k][0]
ms = [v for k, v in summary.get("kernel_ms", {}).items() if "render_kernel<false>" in k or "Lb0E" in k]
read_b, write_b = per["FETCH_SIZE"] * 1024.0 * 2.0, per["WRITE_SIZE"] * 1024.0
try:
    out = json.load(open(path))
except Exception:
    out = {}
out[key] = {"hbm_bytes_per_launch": read_b + write_b, "read_bytes": read_b, "write_bytes": write_b,
            "fetch_size_kib_raw": per["FETCH_SIZE"], "write_size_kib_raw": per["WRITE_SIZE"],
            "l2_hit_rate": per["TCC_HIT_sum"] / max(1.0, per["TCC_HIT_sum"] + per["TCC_MISS_sum"]),
            "kernel_ms_rocprof": ms[0]["avg_ms"] if ms else None, "source": summary.get("dir")}
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out[key]))
