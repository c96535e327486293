"""ptk_group: several GPUs behind one handle.  On a one-GPU machine the group logic (sharing a render call by
sample index or by 16-row bands, accumulation across calls on every device, the merge on device 0, the display
bytes of the merged frame) runs with one device named several times; with two or more GPUs present the same
checks run over distinct devices, where the merge is the RCCL reduce."""
import numpy as np
import pytest

from oracle import orc
from pathtracer_amd import native, scenes
from tests.backends import scene

pytestmark = pytest.mark.gpu


def _device_lists():
    import torch
    n = torch.cuda.device_count()
    lists = [[0, 0, 0]]
    if n >= 2:
        lists.append(list(range(min(n, 4))))
    return lists


def _single(sd, calls, bounces, seed, builder="host"):
    ctx = native.Context(0)
    try:
        ctx.upload_scene(sd, builder)
        ctx.set_camera(sd.camera)
        for first, n in calls:
            ctx.render(n, bounces, seed=seed, sample_first=first)
        img, total = ctx.read_accum()
        _single.stats = ctx.stats()
        return img, total
    finally:
        ctx.close()


@pytest.mark.parametrize("devices", _device_lists(), ids=lambda d: "gpus" + "".join(map(str, d)))
@pytest.mark.parametrize("split", ["spp", "tile"])
def test_group_frame_equals_the_single_gpu_frame(devices, split):
    sd = scene("mesh")
    calls = [(0, 20), (20, 7)]  # two Renderer::render calls: the second accumulates onto the first
    want, want_total = _single(sd, calls, 4, seed=6)
    g = native.Group(devices)
    try:
        assert g.size() == len(devices)
        g.upload_scene(sd)
        g.set_camera(sd.camera)
        g.render(20, 4, seed=6, sample_first=0, split=split)
        first, total = g.read_accum()  # reading must not disturb the per-device sums
        assert total == 20
        g.render(7, 4, seed=6, sample_first=20, split=split)
        got, total = g.read_accum()
        assert total == want_total == 27
        assert np.allclose(got, want, rtol=3e-5, atol=1e-6)
        assert not np.array_equal(got, first)
        # the group's counters are the whole run's, equal to what one device counts for the same paths
        whole = g.stats()
        for key in ("camera_rays", "segments", "queries"):
            assert whole[key] == _single.stats[key], (key, whole[key], _single.stats[key])
        shares = [g.context_stats(i)["camera_rays"] for i in range(len(devices))]
        assert sum(shares) == 27 * sd.camera.width * sd.camera.height
        assert min(shares) > 0.2 * max(shares)  # every device took a real share
        rgb8, total8 = g.read_display_u8()
        assert total8 == 27
        ref8 = orc.tonemap_u8(got.astype(np.float64))
        assert np.abs(rgb8.astype(int) - ref8.astype(int)).max() <= 1
        g.reset_accum()
        blank, total = g.read_accum()
        assert total == 0 and not blank.any()
    finally:
        g.close()


def test_group_of_one_and_device_built_tree():
    sd = scene("mesh")
    want, _ = _single(sd, [(0, 16)], 3, seed=2, builder="device")
    for devices in ([0], [0, 0]):
        g = native.Group(devices)
        try:
            g.upload_scene(sd, "device")
            g.set_camera(sd.camera)
            g.render(16, 3, seed=2)
            got, total = g.read_accum()
            assert total == 16 and np.allclose(got, want, rtol=3e-5, atol=1e-6)
        finally:
            g.close()


def test_group_argument_errors():
    import torch
    with pytest.raises(native.PtkError, match="device index"):
        native.Group([0, torch.cuda.device_count()])
    if torch.cuda.device_count() >= 2:
        with pytest.raises(native.PtkError, match="every device once"):
            native.Group([0, 1, 1])
    with pytest.raises(native.PtkError, match="1 to 64"):
        native.Group([])
    g = native.Group([0, 0])
    try:
        with pytest.raises(native.PtkError, match="camera"):
            g.read_accum()
        sd = scenes.cornell_box(32, 32)
        g.upload_scene(sd)
        g.set_camera(sd.camera)
        p = native.PtkRenderParams(4, 3, 1, 0, 1, 0, 16, 0, 0, 0, 1)
        assert g._L.ptk_group_render(g.h, p, 1) != 0 and b"band split" in g._L.ptk_group_last_error(g.h)
        assert g._L.ptk_group_render(g.h, p, 7) != 0
    finally:
        g.close()
