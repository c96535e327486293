"""ptk_render_static, the render entry point that allocates no device memory (its partial sums live in a static
device array), gives the bits of ptk_render: two calls in a row, the second on a smaller image, each compared
with the allocating entry point; a frame that does not fit the array is refused with a message."""
import numpy as np
import pytest

from pathtracer_amd import native, scenes

pytestmark = pytest.mark.gpu


def _frame(ctx, static, spp, bounces, seed, **kw):
    ctx.reset_accum()
    ctx.render(spp, bounces, seed=seed, static=static, **kw)
    img, total = ctx.read_accum()
    return img, total


def test_static_scratch_render_matches_the_allocating_entry_point():
    ctx = native.Context(0)
    try:
        big = scenes.cornell_with_mesh(3, 200, 136)
        scenes.add_json_spheres(big)
        ctx.upload_scene(big)
        ctx.set_camera(big.camera)
        # 64 spp in 4 sample chunks per tile: the partial sums are in use
        a, na = _frame(ctx, True, 64, 4, 11)
        b, nb = _frame(ctx, False, 64, 4, 11)
        assert na == nb == 64 and np.isfinite(a).all() and a.max() > 0.1
        assert np.array_equal(a, b)
        # second call, smaller input (fewer pixels, fewer chunks): stale contents of the shared array must not leak in
        small = scenes.cornell_box(72, 40)
        ctx.upload_scene(small)
        ctx.set_camera(small.camera)
        c, nc = _frame(ctx, True, 32, 3, 12)
        d, nd = _frame(ctx, False, 32, 3, 12)
        assert nc == nd == 32 and np.array_equal(c, d)
        # row bands and accumulation over calls go through the same path
        ctx.reset_accum()
        for first in (0, 1):
            ctx.render(32, 3, seed=13, band_rows=8, band_first=first, band_stride=2, static=True)
        e, _ = ctx.read_accum()
        ctx.reset_accum()
        for first in (0, 1):
            ctx.render(32, 3, seed=13, band_rows=8, band_first=first, band_stride=2, static=False)
        f, _ = ctx.read_accum()
        assert np.array_equal(e, f)
        # beyond the static array: 2048 x 2048 pixels x 32 chunks x 16 bytes = 2 GiB
        huge = native.make_camera(big.camera)
        huge.width, huge.height = 2048, 2048
        ctx.set_camera(huge)
        with pytest.raises(native.PtkError, match="static buffer holds"):
            ctx.render(1024, 3, static=True)
        ctx.render(16, 3, static=True)  # one chunk per tile needs no partial sums: any image size
        ctx.synchronize()
    finally:
        ctx.close()
