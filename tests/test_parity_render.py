"""Second correctness check: rendered images (linear accumulation buffer) match the oracle.

The kernels and the oracle draw the same counter-based random numbers, so for most pixels the two follow the
same paths and differ only by fp32 rounding; a small fraction of paths takes a different branch somewhere
(a hit across an edge, a Russian-roulette or Fresnel decision within rounding of its threshold) and then the
pixel differs by Monte-Carlo noise.  Three bars, all stated here:

  * per-channel image mean within 1 % of the oracle's (bias check);
  * RMSE (after clamping to [0,4]) <= 1.5 x RMSE0, where RMSE0 is oracle-vs-oracle at the same sample
    count with another seed (the noise floor the reference itself has from run to run);
  * at most 6 % of pixels differ by more than 1e-2 in any channel (same random numbers => same paths).
"""
import numpy as np
import pytest

from oracle import orc
from pathtracer_amd import native, scenes
from tests import parity
from tests.backends import BACKENDS, GpuBackend, make_backend, scene

CASES = [
    # scene, samples, max_bounce
    ("cornell", 32, 3),          # the reference's own CPU-runnable configuration, scaled down
    ("cornell_pinhole", 16, 5),  # Russian roulette active (depth > 3)
    ("mesh", 16, 3),             # mesh + glass sphere + mirror sphere
    ("glass", 24, 8),            # dielectric mesh, 8 bounces
    ("envmap", 24, 3),           # environment map + textured floor + depth of field
    ("sky", 24, 4),              # sky gradient, fuzzy specular, microfacet, sphere light
    ("tiled", 16, 3),            # floor texture coordinates in [-1, 2]
    ("mirror_tex", 24, 5),       # textured sphere, MIRROR-type sphere and wall
    ("envmap_full", 16, 3),      # environment map of 3584 x 1792 texels
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,spp,bounces", CASES)
def test_image_matches_oracle(backend, name, spp, bounces, request):
    sd = scene(name)
    o = orc.OracleScene(sd)
    b = make_backend(backend, sd, request)
    ref, cref = o.render(spp, bounces, seed=11)
    ref2, _ = o.render(spp, bounces, seed=12)
    got, cgot = b.render(spp, bounces, seed=11)
    assert np.isfinite(got).all() == np.isfinite(ref).all()
    st = parity.image_stats(ref, got)
    noise = parity.image_stats(ref, ref2)
    print(name, st, "noise rmse", noise["rmse"])
    assert st["mean_rel"] <= 0.01, st
    assert st["rmse"] <= 1.5 * noise["rmse"], (st, noise)
    assert st["frac_diff"] <= 0.06, st
    # same paths => nearly the same amount of work
    assert cgot["camera_rays"] == cref["camera_rays"]
    assert abs(cgot["segments"] - cref["segments"]) <= 0.005 * cref["segments"], (cgot, cref)
    assert cgot["queries"] <= cref["queries"]  # shadow rays from a light to itself are not traced
    assert cgot["queries"] >= 0.97 * cref["queries"], (cgot, cref)


@pytest.mark.parametrize("backend", BACKENDS)
def test_converged_image(backend, request):
    """Higher sample count on the small Cornell box: tonemapped 8-bit output within 2 levels on average."""
    sd = scene("cornell")
    o = orc.OracleScene(sd)
    b = make_backend(backend, sd, request)
    ref, _ = o.render(256, 3, seed=3)
    got, _ = b.render(256, 3, seed=3)
    a8 = orc.tonemap_u8(ref).astype(np.int32)
    b8 = orc.tonemap_u8(got).astype(np.int32)
    mad = np.abs(a8 - b8).mean()
    print("mean abs 8-bit diff", mad)
    assert mad <= 2.0
    st = parity.image_stats(ref, got)
    assert st["mean_rel"] <= 0.01, st


@pytest.mark.parametrize("backend", BACKENDS)
def test_flags(backend, request):
    """Each config switch (src/config.h:3-8) changes the kernels the same way it changes the oracle."""
    sd = scene("sky")
    o = orc.OracleScene(sd)
    b = make_backend(backend, sd, request)
    variants = {
        "debug_normal": dict(debug_normal=1),
        "no_rr": dict(russian_roulette=0),
        "no_direct": dict(direct_light_sampling=0),
        "no_indirect": dict(indirect_light_sampling=0),
        "no_importance": dict(importance_sample=0),
        "no_hemisphere_check": dict(check_hemisphere=0),
    }
    for label, kw in variants.items():
        fo = orc.lib().orc_default_flags()
        fk = native.PtkFlags.defaults()
        for k, v in kw.items():
            setattr(fo, k, v)
            setattr(fk, k, v)
        ref, _ = o.render(12, 6, seed=5, flags=fo)
        got, _ = b.render(12, 6, seed=5, flags=fk)
        st = parity.image_stats(ref, got)
        print(label, st)
        assert st["mean_rel"] <= 0.02, (label, st)
        assert st["frac_diff"] <= 0.08, (label, st)
        assert st["frac_nonfinite_mismatch"] <= 0.02, (label, st)
        if label == "debug_normal":
            assert st["rmse"] < 1e-3


@pytest.mark.parametrize("backend", BACKENDS)
def test_frame_splits_give_the_same_samples(backend, request):
    """The same call twice gives bit-identical frames.  A frame split by rows, by sample index (stride 2, as
    two GPUs would) or into successive batches (Renderer::render called repeatedly, src/main.cpp:242-252) is the
    same sample set, equal up to fp32 summation order (a pixel's samples are summed in the order its tile's
    lanes finish them, which depends on where the tiles start)."""
    sd = scene("cornell")
    b = make_backend(backend, sd, request)
    h = sd.camera.height
    full, _ = b.render(8, 3, seed=9)
    again, _ = b.render(8, 3, seed=9)
    assert np.array_equal(full, again)
    top, _ = b.render(8, 3, seed=9, row_begin=0, row_end=h // 2 + 3)
    both, _ = b.render(8, 3, seed=9, row_begin=h // 2 + 3, row_end=h, reset=False)
    assert np.allclose(both, full, rtol=1e-5, atol=1e-6)
    assert np.allclose(top[: h // 2 + 3], full[: h // 2 + 3], rtol=1e-5, atol=1e-6) and not top[h // 2 + 3:].any()
    b.render(4, 3, seed=9, sample_first=0, sample_stride=2)
    strided, _ = b.render(4, 3, seed=9, sample_first=1, sample_stride=2, reset=False)
    assert np.allclose(strided, full, rtol=1e-5, atol=1e-6)
    b.render(5, 3, seed=9, sample_first=0)
    batched, _ = b.render(3, 3, seed=9, sample_first=5, reset=False)
    assert np.allclose(batched, full, rtol=1e-5, atol=1e-6)
    # interleaved bands of 8 rows, as three GPUs would share them (one launch each)
    first, _ = b.render(8, 3, seed=9, band_rows=8, band_first=0, band_stride=3)
    rows = np.arange(h)
    assert first[(rows // 8) % 3 == 0].any() and not first[(rows // 8) % 3 != 0].any()
    b.render(8, 3, seed=9, band_rows=8, band_first=1, band_stride=3, reset=False)
    banded, cnt = b.render(8, 3, seed=9, band_rows=8, band_first=2, band_stride=3, reset=False)
    assert np.allclose(banded, full, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("backend", BACKENDS)
def test_zero_work(backend, request):
    sd = scene("cornell")
    b = make_backend(backend, sd, request)
    img, cnt = b.render(0, 3, seed=1)
    assert not img.any() and cnt["segments"] == 0
    img, cnt = b.render(4, 0, seed=1)  # max_bounce 0: trace_ray returns black at once (src/renderer.cpp:59-61)
    assert not img.any() and cnt["segments"] == 0 and cnt["camera_rays"] == 4 * sd.camera.width * sd.camera.height


@pytest.mark.gpu
def test_display_bytes_match_the_reference_post_process(gpu_ctx):
    """ptk_read_display_u8 = Renderer::save_image's per-pixel loop (src/renderer.cpp:174-192) applied to the
    buffer ptk_read_accum returns: the oracle's restatement of that loop gives the same bytes (a value within
    1e-13 of a byte boundary could differ by one level between two pow implementations; none may differ by more)."""
    from oracle import orc
    sd = scene("envmap")
    b = GpuBackend(gpu_ctx, sd)
    b.render(16, 3, seed=5)
    mean, total = gpu_ctx.read_accum()
    got, total8 = gpu_ctx.read_display_u8()
    want = orc.tonemap_u8(mean.astype(np.float64))
    assert total == total8 == 16
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-4
    assert got.max() > 100 and got.min() < 50  # a real image, not a constant


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("width,height", [(37, 23), (8, 8), (1, 1), (5, 130)])
def test_image_sizes_that_do_not_fill_tiles(backend, width, height, request):
    """Widths and heights that are not multiples of the 8 x 8 tile, down to a single pixel; also as interleaved
    bands whose last band is cut off by the image edge."""
    from pathtracer_amd import scenes
    sd = scenes.cornell_box(width, height)
    o = orc.OracleScene(sd)
    b = make_backend(backend, sd, request)
    ref, cref = o.render(32, 3, seed=17)
    got, cgot = b.render(32, 3, seed=17)
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert cgot["camera_rays"] == cref["camera_rays"] == 32 * width * height
    st = parity.image_stats(ref, got)
    assert st["frac_diff"] <= 0.08 and st["mean_rel"] <= 0.02, st
    b.render(32, 3, seed=17, band_rows=8, band_first=0, band_stride=2)
    banded, _ = b.render(32, 3, seed=17, band_rows=8, band_first=1, band_stride=2, reset=False)
    assert np.allclose(banded, got, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "cornell_pinhole", "cornell_plain_floor"])
def test_tiny_scene_form_renders_the_same_frame(name, gpu_ctx, monkeypatch):
    """A scene of a few dozen primitives is rendered by the kernel form that keeps the whole scene in LDS
    (ptk_layout.h: kTinyNodes ...).  Where the records are fetched from must not change the frame:
    PTK_NO_TINY=1 at upload time selects the ordinary form for the same scene."""
    sd = scene(name)
    tiny = GpuBackend(gpu_ctx, sd)
    assert tiny.info()["n_nodes"] <= 16 and tiny.info()["n_prims"] <= 24
    img_tiny, cnt_tiny = tiny.render(64, 3, seed=5, collect_stats=False)
    img_tiny_stats, cnt_tiny_stats = tiny.render(64, 3, seed=5, collect_stats=True)
    monkeypatch.setenv("PTK_NO_TINY", "1")
    plain = GpuBackend(gpu_ctx, sd)
    img_plain, cnt_plain = plain.render(64, 3, seed=5, collect_stats=False)
    _, cnt_plain_stats = plain.render(64, 3, seed=5, collect_stats=True)
    # Same arithmetic, same random numbers, same order of work: the frames agree to rounding (the compiler contracts
    # product-sums differently in the kernel's instantiations), except where a path took the other side of a
    # comparison because of that.  On the textured floor a last-bit difference in a hit's texture coordinate is a
    # difference of 3e-5 in the filter weight (coordinate times 511), and the filter truncates to 8 bits after every
    # blend as the reference does: about one textured sample in twenty then lands one level (1/255) off, so with a
    # texture the per-pixel band is a level's worth of one sample in 64 instead of rounding.
    band = 1e-5 if name == "cornell_plain_floor" else 1e-2
    for other in (img_plain, img_tiny_stats):
        differing = np.any(np.abs(img_tiny - other) > band * np.maximum(np.abs(img_tiny), np.abs(other)) + 1e-7, axis=-1).mean()
        assert differing <= 0.02, differing
        assert abs(img_tiny.mean() - other.mean()) <= 1e-3 * img_tiny.mean()
    for key in ("queries", "segments"):
        assert abs(cnt_tiny[key] - cnt_plain[key]) <= 1e-4 * cnt_plain[key]
    for key in ("box_tests", "prim_tests"):
        assert abs(cnt_tiny_stats[key] - cnt_plain_stats[key]) <= 1e-4 * cnt_plain_stats[key]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mesh", "sky", "mirror_tex"])
def test_table_forms_render_the_same_frame(name, gpu_ctx, monkeypatch):
    """Scenes whose material and light tables fit 768 bytes are rendered by the kernel forms that keep those tables
    in LDS beside a 19-entry stack (ptk_layout.h: kTableBytes).  The frame must not depend on that: PTK_NO_TABLES=1
    at upload time selects the forms that fetch them from global memory, PTK_FORCE_DEEP=1 the forms with overflow
    stack entries, and a scene with too many materials for the tables falls back by itself."""
    import copy
    sd = scene(name)
    monkeypatch.setenv("PTK_NO_TINY", "1")  # (two of the scenes are small enough for the whole-scene-in-LDS form)
    with_tables = GpuBackend(gpu_ctx, sd)
    assert with_tables.info()["form"] in (4, 5), with_tables.info()
    frames = {"tables": with_tables.render(64, 3, seed=9, collect_stats=False)}
    monkeypatch.setenv("PTK_FORCE_DEEP", "1")
    b = GpuBackend(gpu_ctx, sd)
    assert b.info()["form"] == 5
    frames["tables, overflow stack"] = b.render(64, 3, seed=9, collect_stats=False)
    monkeypatch.delenv("PTK_FORCE_DEEP")
    monkeypatch.setenv("PTK_NO_TABLES", "1")
    b = GpuBackend(gpu_ctx, sd)
    assert b.info()["form"] in (0, 3)
    frames["global memory"] = b.render(64, 3, seed=9, collect_stats=False)
    monkeypatch.delenv("PTK_NO_TABLES")
    crowded = copy.deepcopy(sd)
    crowded.materials = list(crowded.materials) + [crowded.materials[0]] * 16  # unused, but the table no longer fits
    b = GpuBackend(gpu_ctx, crowded)
    assert b.info()["form"] in (0, 3)
    frames["too many materials"] = b.render(64, 3, seed=9, collect_stats=False)
    img0, cnt0 = frames["tables"]
    for what, (img, cnt) in frames.items():
        differing = np.any(np.abs(img0 - img) > 1e-5 * np.maximum(np.abs(img0), np.abs(img)) + 1e-7, axis=-1).mean()
        assert differing <= 0.02, (what, differing)
        assert abs(img0.mean() - img.mean()) <= 1e-3 * img0.mean(), what
        for key in ("queries", "segments"):
            assert abs(cnt0[key] - cnt[key]) <= 1e-4 * cnt0[key], (what, key)


@pytest.mark.gpu
def test_persistent_launch_renders_the_same_bits(gpu_ctx, monkeypatch):
    """A frame with more work units (tile x sample chunk) than fit the GPU at once is rendered by as many workgroups as
    fit, which draw the units from counters.  Which workgroup renders a unit changes nothing: the frame is the one that a
    launch with one workgroup per unit (PTK_NO_PERSISTENT=1) gives, bit for bit, and so is a second persistent launch."""
    sd = scenes.cornell_with_mesh(4, 256, 256)  # 1 024 tiles x 8 chunks of 16 samples = 8 192 units
    b = GpuBackend(gpu_ctx, sd)
    first, cnt_first = b.render(128, 3, seed=21, collect_stats=False)
    again, _ = b.render(128, 3, seed=21, collect_stats=False)
    monkeypatch.setenv("PTK_NO_PERSISTENT", "1")
    plain, cnt_plain = b.render(128, 3, seed=21, collect_stats=False)
    assert np.array_equal(first, again)
    assert np.array_equal(first, plain)
    assert cnt_first == cnt_plain
    # the same with the test-counting form of the kernel (per-lane counts are folded into the wave's totals after every
    # unit a persistent workgroup draws): same frame, and the same counts as one workgroup per unit
    plain_counted, cnt_plain_counted = b.render(128, 3, seed=21, collect_stats=True)
    monkeypatch.delenv("PTK_NO_PERSISTENT")
    counted, cnt_counted = b.render(128, 3, seed=21, collect_stats=True)
    assert cnt_counted == cnt_plain_counted and cnt_counted["box_tests"] > 0 and cnt_counted["prim_tests"] > 0
    assert {k: cnt_counted[k] for k in ("camera_rays", "segments", "queries")} == {k: cnt_first[k] for k in ("camera_rays", "segments", "queries")}
    assert np.array_equal(counted, plain_counted)
