"""Seeded ray batches for the intersection parity tests."""
import numpy as np


def sphere_dirs(rng, n):
    z = 1.0 - 2.0 * rng.uniform(size=n)
    phi = 2.0 * np.pi * rng.uniform(size=n)
    s = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    return np.stack([s * np.cos(phi), s * np.sin(phi), z], 1)


def interior_rays(n, seed=42, lo=(-4.5, 0.5, -4.5), hi=(4.5, 9.5, 4.5)):
    """Origins uniform inside the box interior, directions uniform on the sphere."""
    rng = np.random.RandomState(seed)
    o = np.stack([rng.uniform(lo[k], hi[k], n) for k in range(3)], 1)
    return np.concatenate([o, sphere_dirs(rng, n)], 1)


def camera_like_rays(n, seed=43, position=(0.0, 5.0, 24.0), spread=0.26):
    """Pinhole rays from the camera position towards -z with seeded jitter (not a pixel lattice)."""
    rng = np.random.RandomState(seed)
    d = np.stack([rng.uniform(-spread, spread, n), rng.uniform(-spread, spread, n), -np.ones(n)], 1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.broadcast_to(np.asarray(position, np.float64), (n, 3))
    return np.concatenate([o, d], 1)


def secondary_rays(hits, seed=44):
    """Rays that start exactly on surfaces: origins = hit points, directions cosine-distributed about the
    hit normal (exercises the no-offset self-intersection rules)."""
    rng = np.random.RandomState(seed)
    m = hits["hit"] == 1
    p = hits["point"][m].astype(np.float64)
    nrm = hits["normal"][m].astype(np.float64)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    n = p.shape[0]
    r1, r2 = rng.uniform(size=n), rng.uniform(size=n)
    ct, st, phi = np.sqrt(r1), np.sqrt(1 - r1), 2 * np.pi * r2
    ref = np.where(np.abs(nrm[:, 2:3]) < 0.9, np.array([[0.0, 0.0, 1.0]]), np.array([[1.0, 0.0, 0.0]]))
    r = np.cross(ref, nrm)
    r /= np.linalg.norm(r, axis=1, keepdims=True)
    f = np.cross(nrm, r)
    d = r * (np.cos(phi) * st)[:, None] + nrm * ct[:, None] + f * (np.sin(phi) * st)[:, None]
    return np.concatenate([p, d], 1)
