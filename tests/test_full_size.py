"""The benchmark's own configuration (Cornell box + 81,920-triangle mesh, 1024 x 1024) checked on the GPU
through properties that do not need the oracle to render the whole frame:

  * every pixel receives exactly the requested number of samples and a finite value;
  * the same call twice gives the same bits;
  * a frame rendered as two batches of sample indices (Renderer::render called twice, src/main.cpp:242-252)
    equals the one-call frame up to fp32 summation order;
  * a band of rows of the full-size frame matches the oracle's render of the same rows (same bars as
    tests/test_parity_render.py: mean within 1 %, RMSE <= 1.5 x the oracle's own seed-to-seed RMSE);
  * closest hits of random and camera-like rays into the full scene equal the oracle's (ids exact but for
    the counted tie classes, distances within 1e-5 relative).
"""
import numpy as np
import pytest

from oracle import orc
from pathtracer_amd import scenes
from tests import parity, raygen
from tests.backends import GpuBackend

pytestmark = pytest.mark.gpu

W = H = 1024
SPP = 16
BOUNCES = 3


@pytest.fixture(scope="module")
def full(gpu_ctx):
    sd = scenes.cornell_with_mesh(6, W, H)
    return sd, GpuBackend(gpu_ctx, sd)


def test_every_pixel_gets_its_samples_and_the_frame_is_reproducible(full, gpu_ctx):
    sd, b = full
    img, cnt = b.render(SPP, BOUNCES, seed=21)
    _, total = gpu_ctx.read_accum()
    assert total == SPP
    assert np.isfinite(img).all() and img.min() >= 0.0
    assert cnt["camera_rays"] == W * H * SPP
    assert 2.0 < cnt["segments"] / cnt["camera_rays"] < 3.0  # up to 3 segments per path, most paths use them all
    again, cnt2 = b.render(SPP, BOUNCES, seed=21)
    assert np.array_equal(img, again) and cnt == cnt2
    other, _ = b.render(SPP, BOUNCES, seed=22)
    assert not np.array_equal(img, other)
    # two batches of sample indices = one call, up to summation order
    b.render(SPP // 2, BOUNCES, seed=21, sample_first=0)
    batched, _ = b.render(SPP - SPP // 2, BOUNCES, seed=21, sample_first=SPP // 2, reset=False)
    assert np.allclose(batched, img, rtol=2e-5, atol=1e-6)


def test_a_band_of_the_full_frame_matches_the_oracle(full):
    sd, b = full
    o = orc.OracleScene(sd)
    y0, y1 = 496, 528  # through the mesh, the walls and the floor
    ref, cref = o.render(SPP, BOUNCES, seed=31, rows=(y0, y1))
    ref2, _ = o.render(SPP, BOUNCES, seed=32, rows=(y0, y1))
    got, cgot = b.render(SPP, BOUNCES, seed=31, row_begin=y0, row_end=y1)
    assert not got[:y0].any() and not got[y1:].any()
    st = parity.image_stats(ref[y0:y1], got[y0:y1])
    noise = parity.image_stats(ref[y0:y1], ref2[y0:y1])
    print(st, "noise rmse", noise["rmse"])
    assert st["mean_rel"] <= 0.01, st
    assert st["rmse"] <= 1.5 * noise["rmse"], (st, noise)
    assert st["frac_diff"] <= 0.06, st
    assert cgot["camera_rays"] == cref["camera_rays"]
    assert abs(cgot["segments"] - cref["segments"]) <= 0.005 * cref["segments"], (cgot, cref)


def test_rays_into_the_full_scene_hit_what_the_oracle_hits(full):
    sd, b = full
    o = orc.OracleScene(sd)
    rays = np.concatenate([raygen.interior_rays(40000, 52), raygen.camera_like_rays(40000, 53)])
    rays = np.asarray(rays, np.float64).astype(np.float32).astype(np.float64)  # both sides see the same inputs
    ref, _ = o.intersect(rays)
    got, _ = b.intersect(rays)
    c = parity.classify_hits(sd, rays, ref, got)
    print(c)
    assert c["unexplained"] == 0, c
    excused = c["tie"] + c["near"] + c["ref_miss"] + c["graze"] + c["cond"]
    assert excused <= max(3, 1e-4 * c["n"]), c
    assert c["max_rel_t"] <= 1e-5, c
