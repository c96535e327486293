"""First correctness check: closest-hit primitive ids on fixed, seeded ray batches match the oracle exactly
(except for the stated, counted classes of rounding ties) and hit distances agree within 1e-5 relative.

Runs against the kernels' device functions compiled for the CPU ("hostcheck", no GPU needed) and against the
HIP kernels through the C ABI ("gpu")."""
import numpy as np
import pytest

from oracle import orc
from tests import parity, raygen
from tests.backends import BACKENDS, make_backend, scene


def _oracle(sd):
    return orc.OracleScene(sd)


def _f32_rays(rays):
    """Round ray data through float32 so that the oracle and the fp32 kernels see identical inputs."""
    return np.asarray(rays, np.float64).astype(np.float32).astype(np.float64)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("scene_name,n", [("cornell", 60000), ("mesh", 30000), ("glass", 30000), ("mirror_tex", 30000)])
def test_random_rays_ids_exact(backend, scene_name, n, request):
    sd = scene(scene_name)
    o = _oracle(sd)
    b = make_backend(backend, sd, request)
    rays = _f32_rays(np.concatenate([raygen.interior_rays(n, 42), raygen.camera_like_rays(n // 2, 43)]))
    ref, _ = o.intersect(rays)
    got, _ = b.intersect(rays)
    c = parity.classify_hits(sd, rays, ref, got)
    print(scene_name, c)
    assert c["unexplained"] == 0, c
    excused = c["tie"] + c["near"] + c["ref_miss"] + c["graze"] + c["cond"]
    assert excused <= max(3, 1e-4 * c["n"]), c  # total excused <= 1e-4 of rays on random batches
    assert c["max_rel_t"] <= 1e-5, c
    rec = parity.check_hit_records(ref, got)
    print(rec)
    # Sphere::texcoord goes through acos / atan2 of the unit hit vector: near the poles the fp32 error of the
    # polar angle grows like 1 / sin(theta), so textured spheres get a wider bar than interpolated triangle uvs
    assert rec["point"] < 1e-5 and rec["uv"] < (2e-3 if scene_name == "mirror_tex" else 1e-4) and rec["material_mismatch"] == 0
    assert rec["normal"] < 1e-5
    assert rec["inside_mismatch"] <= 1e-4 * rec["n"]
    if scene_name == "mirror_tex":  # the textured sphere is hit, and its texture coordinates span the map
        on_globe = (got["hit"] == 1) & (got["id"] == sd.n_triangles)
        assert on_globe.sum() > 500
        uv = got["uv"][on_globe]
        assert uv.min() >= 0.0 and uv.max() <= 1.0 and np.ptp(uv[:, 0]) > 0.9 and np.ptp(uv[:, 1]) > 0.8


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("scene_name", ["cornell", "mesh"])
def test_secondary_rays_start_on_surfaces(backend, scene_name, request):
    """Rays whose origin lies exactly on a surface (no offset, like the reference's bounce and shadow rays).
    Differences here are limited to hits within rounding distance of the origin ("near")."""
    sd = scene(scene_name)
    o = _oracle(sd)
    b = make_backend(backend, sd, request)
    first, _ = o.intersect(_f32_rays(raygen.interior_rays(20000, 7)))
    rays = _f32_rays(raygen.secondary_rays(first, 8))
    ref, _ = o.intersect(rays)
    got, _ = b.intersect(rays)
    c = parity.classify_hits(sd, rays, ref, got)
    print(scene_name, c)
    assert c["unexplained"] == 0, c
    assert c["tie"] + c["ref_miss"] + c["graze"] <= max(3, 2e-4 * c["n"]), c
    assert c["cond"] <= 1e-3 * c["n"], c
    assert c["near"] <= 2e-2 * c["n"], c


@pytest.mark.parametrize("backend", BACKENDS)
def test_edge_cases(backend, request):
    """Empty batch, axis-parallel rays (zero direction components), rays that miss everything, unnormalised
    directions, and a scene whose only primitives share one leaf."""
    sd = scene("cornell_pinhole")
    o = _oracle(sd)
    b = make_backend(backend, sd, request)
    got, _ = b.intersect(np.zeros((0, 6)))
    assert len(got) == 0
    rays = np.array([
        [0.3, 5.0, 2.0, 0.0, 0.0, -1.0],     # axis parallel, hits the back wall
        [0.3, 5.0, 2.0, 0.0, -1.0, 0.0],     # straight down onto the floor
        [0.3, 5.0, 2.0, 0.0, 0.0, 1.0],      # out through the open front: miss
        [0.3, 5.0, 2.0, 0.0, 0.0, -3.0],     # unnormalised direction: t scales by 1/3
        [0.1, 5.0, 24.0, 0.0, 0.0, -1.0],    # from beside the camera (the wall's centre is on its diagonal: a tie)
        [20.0, 20.0, 20.0, 1.0, 0.0, 0.0],   # outside, pointing away
        [0.25, 9.0, 0.5, 0.0, 1.0, 0.0],     # up into the light (off the quad's diagonal)
    ])
    ref, _ = o.intersect(rays)
    got, _ = b.intersect(rays)
    assert ref["hit"].tolist() == got["hit"].tolist() == [1, 1, 0, 1, 1, 0, 1]
    m = ref["hit"] == 1
    assert (ref["id"][m] == got["id"][m]).all()
    assert np.allclose(ref["t"][m], got["t"][m], rtol=1e-5)
    assert got["id"][6] in (8, 9)


@pytest.mark.parametrize("backend", BACKENDS)
def test_tiny_scenes(backend, request):
    """One triangle, one sphere, and an empty scene: the flattened tree degenerates to a single leaf."""
    from pathtracer_amd import scenes
    rays = _f32_rays(raygen.interior_rays(2000, 3, lo=(-2, -2, 3), hi=(2, 2, 6)))
    rays[:, 3:] = np.array([0.0, 0.0, -1.0]) + 0.2 * rays[:, 3:]
    for kind in ("tri", "sphere", "empty"):
        sd = scenes.SceneData()
        m = sd.add_material(scenes.Material())
        if kind == "tri":
            sd.add_triangles([[[-1, -1, 0], [1, -1, 0], [0, 1, 0]]], np.tile([0, 0, 1.0], (1, 3, 1)), np.zeros((1, 3, 2)), m)
        elif kind == "sphere":
            sd.add_sphere((0, 0, 0), 1.0, m)
        sd.camera = scenes.CameraDesc(16, 16)
        o = _oracle(sd)
        b = make_backend(backend, sd, request)
        ref, _ = o.intersect(rays)
        got, _ = b.intersect(rays)
        c = parity.classify_hits(sd, rays, ref, got)
        # a single flat triangle is a zero-thickness box in the reference's tree: it rejects it (ref_miss)
        assert c["unexplained"] == 0 and c["tie"] == 0 and c["graze"] <= 2, (kind, c)
        if kind == "sphere":
            assert c["ok"] == c["n"], c
        if kind == "empty":
            assert (got["hit"] == 0).all()


@pytest.mark.parametrize("backend", BACKENDS)
def test_fewer_tests_than_reference(backend, request):
    """The kernels' tree must not need more box or primitive tests per query than the reference's tree does
    (so a worse tree cannot inflate a bytes-per-ray roofline)."""
    for name in ("cornell", "mesh"):
        sd = scene(name)
        o = _oracle(sd)
        b = make_backend(backend, sd, request)
        rays = _f32_rays(raygen.interior_rays(20000, 11))
        _, cref = o.intersect(rays)
        _, cgot = b.intersect(rays)
        print(name, cref, cgot)
        assert cgot["prim_tests"] <= cref["prim_tests"]
        if name == "mesh":
            assert cgot["box_tests"] <= cref["box_tests"]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("scale,offset", [(100.0, (500.0, -300.0, 200.0)), (0.01, (0.0, 0.0, 0.0)), (1.0, (40.0, 40.0, -40.0))])
def test_scaled_and_offset_scene(backend, scale, offset, request):
    """The node boxes live on a grid over the scene bounds and the ray carries grid-relative quantities: a scene
    far from the origin, much larger or much smaller than the benchmark scenes must give the same ids."""
    import copy
    sd = copy.deepcopy(scene("mesh"))
    off = np.asarray(offset)
    sd.tri_pos = sd.tri_pos * scale + off
    sd.sphere_center = sd.sphere_center * scale + off
    sd.sphere_radius = sd.sphere_radius * scale
    o = _oracle(sd)
    b = make_backend(backend, sd, request)
    rays = np.concatenate([raygen.interior_rays(20000, 42), raygen.camera_like_rays(10000, 43)])
    rays[:, :3] = rays[:, :3] * scale + off
    rays = _f32_rays(rays)
    ref, _ = o.intersect(rays)
    got, _ = b.intersect(rays)
    c = parity.classify_hits(sd, rays, ref, got)
    print(scale, offset, c)
    assert c["unexplained"] == 0, c
    # "near" (hits closer than the reference's absolute 0.01 box entry distance, which the reference loses) is a
    # property of the scene's size, not of the kernels: a 100x smaller scene has many of them
    excused = c["tie"] + c["ref_miss"] + c["graze"] + c["cond"] + (c["near"] if scale >= 1.0 else 0)
    assert excused <= max(3, 3e-4 * c["n"]), c


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("factor", [30.0, 100.0, 300.0])
@pytest.mark.parametrize("scene_name", ["cornell_pinhole", "mesh"])
def test_origins_far_from_the_scene(backend, scene_name, factor, request):
    """A small object seen from far away: the slab distances of the box test are differences of numbers that
    grow with the origin's distance from the scene, so their rounding grows with it while the boxes' padding does
    not (a ray through the corner of a thin box can then be lost: about one ray in 10^4 at 300 scene extents).
    Rays that start far outside the scene's bounds are therefore advanced to just in front of the bounds before
    they are traversed; hits, hit records and distances (measured from the ORIGINAL origin) must still be the
    oracle's.  The box's walls lie in leaves as thin as the padding: the case that would be lost first."""
    sd = scene(scene_name)
    o = _oracle(sd)
    b = make_backend(backend, sd, request)
    rng = np.random.RandomState(17)
    n = 20000
    target = np.stack([rng.uniform(-4.5, 4.5, n), rng.uniform(0.5, 9.5, n), rng.uniform(-4.5, 4.5, n)], 1)
    away = raygen.sphere_dirs(rng, n)
    origin = target + away * (14.0 * factor)  # the scene is about 14 units across
    d = -away + 1e-4 * rng.normal(size=(n, 3))
    rays = _f32_rays(np.concatenate([origin, d], 1))
    ref, _ = o.intersect(rays)
    got, _ = b.intersect(rays)
    c = parity.classify_hits(sd, rays, ref, got)
    print(factor, c)
    assert c["ok"] + c["cond"] >= 0.99 * n and (ref["hit"] == 1).mean() > 0.9
    assert c["unexplained"] == 0, c
    # the reference itself loses precision out here (float d of its plane equation): its own misses are counted, not ours
    assert c["tie"] + c["near"] + c["graze"] <= max(3, 3e-4 * n), c
    rec = parity.check_hit_records(ref, got)
    # hit points: the product d * distance carries a few fp32 ulps of the distance travelled (up to 4200 units)
    assert rec["point"] < 4e-8 * 14.0 * factor and rec["material_mismatch"] == 0, rec


@pytest.mark.gpu
def test_batch_sizes_around_the_draw_boundaries(gpu_ctx, monkeypatch):
    """The ray-batch kernel's waves draw runs of rays from a counter (first run: the wave's own number).  Every ray of
    a batch must be answered exactly once whatever the batch size does to the run length, the number of workgroups
    and the last, partial run: the records of a batch cut into pieces are those of the whole batch, and a forced
    short run length and a forced small grid change nothing."""
    from tests.backends import GpuBackend
    sd = scene("mesh")
    b = GpuBackend(gpu_ctx, sd)
    rays = _f32_rays(np.concatenate([raygen.interior_rays(150000, 7), raygen.camera_like_rays(50000, 8)]))
    whole, _ = b.intersect(rays)
    assert (whole["hit"] == 1).mean() > 0.5
    for n in (1, 2, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 4097, 65535, 65537, 131073):
        part, _ = b.intersect(rays[:n])
        for key in ("hit", "id", "t", "point", "normal", "uv", "material", "inside"):
            assert np.array_equal(part[key], whole[key][:n]), (n, key)
    for run in ("64", "192", "7"):  # read at every launch
        monkeypatch.setenv("PTK_INTERSECT_RUN", run)
        part, _ = b.intersect(rays[:50001])
        monkeypatch.delenv("PTK_INTERSECT_RUN")
        for key in ("hit", "id", "t"):
            assert np.array_equal(part[key], whole[key][:50001]), (run, key)
    from pathtracer_amd import native
    for grid in ("7", "100000"):  # read when a context launches its first batch
        monkeypatch.setenv("PTK_INTERSECT_GRID", grid)
        ctx = native.Context(0)
        try:
            other = GpuBackend(ctx, sd)
            part, _ = other.intersect(rays[:50001])
        finally:
            ctx.close()
            monkeypatch.delenv("PTK_INTERSECT_GRID")
        for key in ("hit", "id", "t"):
            assert np.array_equal(part[key], whole[key][:50001]), (grid, key)
