"""Shared comparison logic for the parity tests (GPU kernels or their host-compiled copy vs the oracle).

Closest-hit rule (first correctness check): primitive ids must match the oracle exactly and t must agree
within 1e-5 relative, except for the stated classes of excusable differences, each counted:

  tie        both hit, different ids, |dt| <= 1e-5 * t: the ray passes through a shared edge / two
             coincident surfaces, and which primitive wins depends on test order;
  near       one side reports a hit closer than 0.01 that the other does not: the reference tests node boxes
             from t = 0.01 (src/bvh.cpp:12), so whether it finds a primitive closer than that depends on how far
             ITS leaf box extends along the ray; the same class covers rays that start within float rounding
             of a surface (the reference applies no origin offset);
  ref_miss   the oracle misses but the kernel hits: the reference's own cracks (both edge tests can reject
             along a shared edge in double precision) and its rejection of zero-thickness boxes;
  cond       same primitive on both sides, but the ray meets it at grazing incidence, where the distance is
             ill-conditioned: |dt| <= 1e-5 t + 8 ulp32(origin) / cos(incidence);
  graze      the differing hit has |N.d| within fp32 rounding of the reference's 1e-5 parallel threshold, or
             lies within 1e-5 (barycentric) of a triangle edge: the accept/reject decision flips with rounding.
"""
from __future__ import annotations

import numpy as np

T_REL = 1e-5


def _tri_edge_distance(sd, ids, points):
    """Smallest barycentric coordinate of `points` w.r.t. triangles `ids` (negative = outside)."""
    ids = np.asarray(ids, np.int64)
    out = np.full(len(ids), np.inf)
    tri = ids < sd.n_triangles
    if tri.any():
        p = sd.tri_pos[ids[tri]]
        a, b, c = p[:, 0], p[:, 1], p[:, 2]
        v0, v1, v2 = b - a, c - a, points[tri] - a
        d00 = (v0 * v0).sum(1); d01 = (v0 * v1).sum(1); d11 = (v1 * v1).sum(1)
        d20 = (v2 * v0).sum(1); d21 = (v2 * v1).sum(1)
        inv = 1.0 / (d00 * d11 - d01 * d01)
        v = (d11 * d20 - d01 * d21) * inv
        w = (d00 * d21 - d01 * d20) * inv
        u = 1.0 - v - w
        out[tri] = np.minimum(np.minimum(u, v), w)
    return out


def _near_parallel(sd, ids, dirs):
    """|N.d| within fp32 rounding of the reference's 1e-5 rejection threshold (N unnormalised)."""
    ids = np.asarray(ids, np.int64)
    out = np.zeros(len(ids), bool)
    tri = ids < sd.n_triangles
    if tri.any():
        p = sd.tri_pos[ids[tri]]
        n = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
        nd = np.abs((n * dirs[tri]).sum(1))
        slack = 1e-6 * np.linalg.norm(n, axis=1) * np.linalg.norm(dirs[tri], axis=1)
        out[tri] = nd <= 1e-5 + slack
    return out


def _max_rel_t(rt, gt, t_floor, sel):
    if not sel.any():
        return 0.0
    return float((np.maximum(np.abs(rt - gt) - t_floor, 0.0)[sel] / np.maximum(rt[sel], 1e-30)).max())


def classify_hits(sd, rays, ref, got):
    """Returns dict of counts; 'unexplained' must be 0."""
    rays = np.asarray(rays, np.float64).reshape(-1, 6)
    n = len(rays)
    rh, gh = ref["hit"] == 1, got["hit"] == 1
    rt, gt = ref["t"].astype(np.float64), got["t"].astype(np.float64)
    both = rh & gh
    same = both & (ref["id"] == got["id"])
    # 1e-5 relative, with an absolute floor of 2 fp32 ulps of the largest origin coordinate: the kernels
    # take fp32 coordinates, and a hit much closer than the coordinates' magnitude cannot be resolved better.
    t_floor = np.abs(rays[:, :3]).max(axis=1) * 2.0 ** -22
    t_ok = np.abs(rt - gt) <= T_REL * np.abs(rt) + t_floor
    ok = (same & t_ok) | (~rh & ~gh)
    # same primitive, grazing incidence: dt = (distance error to the surface) / cos(incidence)
    nrm = np.array(ref["normal"], np.float64)
    dirs = rays[:, 3:]
    tri_hit = rh & (ref["id"] < sd.n_triangles)
    if tri_hit.any():  # geometric normal for triangles (the stored one is the interpolated shading normal)
        p = sd.tri_pos[ref["id"][tri_hit].astype(np.int64)]
        nrm[tri_hit] = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    cos_i = np.abs((nrm * dirs).sum(1)) / np.maximum(np.linalg.norm(nrm, axis=1) * np.linalg.norm(dirs, axis=1), 1e-300)
    cond = same & ~t_ok & (np.abs(rt - gt) <= T_REL * np.abs(rt) + 4.0 * t_floor / np.maximum(cos_i, 1e-6))
    ok = ok | cond
    res = dict(n=n, ok=int(ok.sum()), cond=int(cond.sum()), tie=0, near=0, ref_miss=0, graze=0, unexplained=0)
    bad = np.nonzero(~ok)[0]
    if len(bad) == 0:
        res["max_rel_t"] = _max_rel_t(rt, gt, t_floor, same & ~cond)  # grazing hits are counted in "cond" instead
        return res
    o, d = rays[bad, :3], rays[bad, 3:]
    b_rh, b_gh, b_rt, b_gt = rh[bad], gh[bad], rt[bad], gt[bad]
    b_both = b_rh & b_gh
    tie = b_both & (ref["id"][bad] != got["id"][bad]) & (np.abs(b_rt - b_gt) <= T_REL * b_rt + t_floor[bad])
    near = ~tie & ((b_rh & (b_rt < 1e-2)) | (b_gh & (b_gt < 1e-2)))
    ref_miss = ~tie & ~near & (~b_rh & b_gh)
    # grazing / edge cases: look at the hit that only one side has (or each side's own hit)
    rest = ~tie & ~near & ~ref_miss
    graze = np.zeros(len(bad), bool)
    if rest.any():
        idx = np.nonzero(rest)[0]
        for side_hit, side in ((b_rh, ref), (b_gh, got)):
            sel = idx[side_hit[idx]]
            if len(sel) == 0:
                continue
            ids = side["id"][bad][sel]
            t = side["t"][bad][sel].astype(np.float64)
            pts = o[sel] + d[sel] * t[:, None]
            edge = _tri_edge_distance(sd, ids, pts)
            graze[sel] |= (np.abs(edge) < 1e-5) | _near_parallel(sd, ids, d[sel])
    res["tie"] = int(tie.sum()); res["near"] = int(near.sum()); res["ref_miss"] = int(ref_miss.sum())
    res["graze"] = int((graze & rest).sum())
    res["unexplained"] = int((rest & ~graze).sum())
    res["unexplained_idx"] = bad[rest & ~graze][:10].tolist()
    res["max_rel_t"] = _max_rel_t(rt, gt, t_floor, same & ~cond)  # grazing hits are counted in "cond" instead
    return res


def check_hit_records(ref, got, atol=2e-5):
    """Where both sides agree on the primitive: normal, uv, inside flag, material, hit point."""
    m = (ref["hit"] == 1) & (got["hit"] == 1) & (ref["id"] == got["id"])
    out = dict(n=int(m.sum()))
    if not m.any():
        return out
    scale = np.maximum(1.0, np.abs(ref["point"][m]).max())
    out["point"] = float(np.abs(ref["point"][m] - got["point"][m]).max() / scale)
    # the side flip (inside) is decided by the sign of d.n; exclude rays with |d.n| ~ 0 from the strict check
    out["normal"] = float(np.median(np.abs(np.abs(ref["normal"][m]) - np.abs(got["normal"][m])).max(axis=1)))
    out["normal_max"] = float(np.abs(np.abs(ref["normal"][m]) - np.abs(got["normal"][m])).max())
    out["uv"] = float(np.abs(ref["uv"][m] - got["uv"][m]).max())
    out["inside_mismatch"] = int((ref["inside"][m] != got["inside"][m]).sum())
    out["material_mismatch"] = int((ref["material"][m] != got["material"][m]).sum())
    return out


def image_stats(ref_img, got_img, clamp=4.0):
    """Second correctness check: linear-buffer comparison after clamping to [0, clamp]."""
    a = np.asarray(ref_img, np.float64)
    b = np.asarray(got_img, np.float64)
    # pixels that are NaN/inf on either side (the reference's estimator can produce them, e.g. microfacet
    # terms below the horizon) are compared by mask only
    fa, fb = np.isfinite(a).all(axis=2), np.isfinite(b).all(axis=2)
    both = fa & fb
    a = np.clip(a[both], 0, clamp)
    b = np.clip(b[both], 0, clamp)
    diff = b - a
    mean_a, mean_b = a.mean(axis=0), b.mean(axis=0)
    return dict(rmse=float(np.sqrt((diff ** 2).mean())),
                mean_rel=float(np.max(np.abs(mean_b - mean_a) / np.maximum(mean_a, 1e-6))),
                frac_diff=float((np.abs(diff).max(axis=1) > 1e-2).mean()),
                frac_nonfinite_mismatch=float((fa != fb).mean()), frac_nonfinite=float((~fa).mean()),
                mean_ref=mean_a.tolist(), mean_got=mean_b.tolist())
