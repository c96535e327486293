#!/usr/bin/env python3
"""Turn the counters of a tools/profile.sh summary into the per-launch figures that bench.py reports under
"roofline" (traffic, l2_hit_rate, valu_lane_utilisation, valu_issue_slots_used, wait_fraction).

    tools/make_traffic.py <summary.json> <workload_WxH_SPPspp> [traffic.json]

rocprofv3 gives FETCH_SIZE / WRITE_SIZE in KiB per dispatch.  On gfx950 FETCH_SIZE tallies every request of the L2 to
the fabric at 64 bytes, and every such request is a whole 128-byte line - for wide coalesced reads AND for the 16-byte
per-lane gathers of this kernel (tools/ubench/fetch_calib.hip, profiles/r3_ubench_fetch_calib.txt: one request per
missed line whether a lane takes 16 bytes of it or both halves, no 32-byte requests, and the gather moves lines at the
streaming rate only if a line is 128 bytes), so reads are doubled.  The figure is what crosses from the fabric into
the L2s: hits in the 256 MB Infinity Cache are in it.
SQ cycle counters are in units of four cycles and summed over all waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import json, sys

summary, key = json.load(open(sys.argv[1])), sys.argv[2]
path = sys.argv[3] if len(sys.argv) > 3 else "profiles/traffic.json"
per = [v for k, v in summary["counters_per_dispatch"].items() if "render_kernel<false" in k][0]
ms = [v for k, v in summary.get("kernel_ms", {}).items() if "render_kernel<false" in k]
read_b, write_b = per["FETCH_SIZE"] * 1024.0 * 2.0, per["WRITE_SIZE"] * 1024.0
try:
    out = json.load(open(path))
except Exception:
    out = {}
kernel_cycles = per["GRBM_GUI_ACTIVE"] / 8.0
out[key] = {"hbm_bytes_per_launch": read_b + write_b, "read_bytes": read_b, "write_bytes": write_b,
            "fetch_size_kib_raw": per["FETCH_SIZE"], "write_size_kib_raw": per["WRITE_SIZE"],
            "l2_hit_rate": per["TCC_HIT_sum"] / max(1.0, per["TCC_HIT_sum"] + per["TCC_MISS_sum"]),
            "l1_hit_rate": 1.0 - per["TCP_TCC_READ_REQ_sum"] / max(1.0, per["TCP_TOTAL_CACHE_ACCESSES_sum"]),
            "valu_lane_utilisation": per["SQ_THREAD_CYCLES_VALU"] / max(1.0, 64.0 * per["SQ_ACTIVE_INST_VALU"]),
            "valu_issue_slots_used": per["SQ_INSTS_VALU"] * 2.0 / (1024.0 * kernel_cycles),
            "wait_fraction": per["SQ_WAIT_ANY"] / per["SQ_WAVE_CYCLES"],
            "issue_stall_fraction": per["SQ_WAIT_INST_ANY"] / per["SQ_WAVE_CYCLES"],
            "clock_ghz": kernel_cycles / (ms[0]["avg_ms"] * 1e-3) / 1e9 if ms else None,
            "valu_instructions": per["SQ_INSTS_VALU"], "salu_instructions": per["SQ_INSTS_SALU"],
            "vmem_read_instructions": per["SQ_INSTS_VMEM_RD"], "l1_accesses": per["TCP_TOTAL_CACHE_ACCESSES_sum"],
            "kernel_ms_rocprof": ms[0]["avg_ms"] if ms else None, "source": summary.get("dir")}
# passes of tools/profile_mem.sh / profile_mix.sh, when they were merged into the same directory
if "TA_TA_BUSY_sum" in per:
    out[key].update({"ta_busy": per["TA_TA_BUSY_sum"] / (256.0 * kernel_cycles), "td_busy": per["TD_TD_BUSY_sum"] / (256.0 * kernel_cycles),
                     "l1_latency_cycles": per["TCP_TCP_LATENCY_sum"] / max(1.0, per["TCP_TA_TCP_STATE_READ_sum"]),
                     "l1_pending_stall": per["TCP_PENDING_STALL_CYCLES_sum"] / (256.0 * kernel_cycles)})
if "SQ_INSTS_VALU_FMA_F32" in per:
    fast = per["SQ_INSTS_VALU_FMA_F32"] + per["SQ_INSTS_VALU_MUL_F32"] + per["SQ_INSTS_VALU_ADD_F32"]
    trans = per["SQ_INSTS_VALU_TRANS_F32"]
    rest = per["SQ_INSTS_VALU"] - fast - trans
    # measured issue cost per wave64 instruction (tools/ubench/valu_rate.hip): fma/mul/add 2 cycles, transcendentals 8,
    # everything else 4 at most (moves, and/or/xor and integer adds are 2: they cannot be told apart here, so this is an upper bound)
    out[key].update({"valu_fast_share": fast / per["SQ_INSTS_VALU"], "valu_pipe_occupancy_upper": min(1.0, (2.0 * fast + 4.0 * rest + 8.0 * trans) / (1024.0 * kernel_cycles)),
                     "valu_pipe_occupancy_lower": 2.0 * per["SQ_INSTS_VALU"] / (1024.0 * kernel_cycles)})
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out[key]))
