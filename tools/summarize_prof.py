#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory into one JSON (per-kernel time stats + counter sums)."""
import csv, glob, json, os, sys
from collections import defaultdict

d = sys.argv[1]
out = {"dir": os.path.basename(d.rstrip("/"))}
stats = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    out["kernel_stats"] = [{k: r[k] for k in r} for r in rows if float(r.get("Percentage", 0) or 0) > 0.01]
trace = glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True)
if trace:
    per = defaultdict(list)
    for r in csv.DictReader(open(trace[0])):
        per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    out["kernel_ms"] = {k[:80]: {"n": len(v), "avg_ms": sum(v) / len(v), "min_ms": min(v), "max_ms": max(v)} for k, v in per.items() if sum(v) > 0.05}
cnt = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "render_kernel" not in k and "intersect_kernel" not in k:
            continue
        c = cnt[k[:80]][r["Counter_Name"]]
        c[0] += float(r["Counter_Value"]); c[1] += 1
out["counters_per_dispatch"] = {k: {c: v[0] / max(1, v[1]) for c, v in d2.items()} for k, d2 in cnt.items()}
out["dispatches"] = {k: max(v[1] for v in d2.values()) for k, d2 in cnt.items()}
for f in glob.glob(os.path.join(d, "*bench.json")):
    try:
        out.setdefault("bench", {})[os.path.basename(f)] = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        pass
print(json.dumps(out, indent=1))
