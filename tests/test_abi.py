"""The C-ABI library loads and exports every symbol include/ptk.h declares; argument checking works without a
GPU; and on a machine without a GPU the product path fails loudly instead of falling back to anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from pathtracer_amd import native, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "ptk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptk_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_functions() == sorted(native.EXPORTS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(native.LIB_PATH), "build libptk.so first (__graft_entry__.build())"
    lib = C.CDLL(native.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(lib, name), name


def test_struct_layouts_match_the_header():
    # sizes implied by the C declarations (natural alignment)
    assert C.sizeof(native.PtkMaterial) == 8 + 6 * 8 + 3 * 8
    assert C.sizeof(native.PtkTexture) == 16
    assert C.sizeof(native.PtkCamera) == 8 + 12 * 8 + 3 * 8
    assert C.sizeof(native.PtkFlags) == 24
    assert C.sizeof(native.PtkRenderParams) == 11 * 4
    assert C.sizeof(native.PtkStats) == 5 * 8 + 2 * 8 + 8
    assert C.sizeof(native.PtkSceneInfo) == 6 * 4 + 4 * 8 + 2 * 4 + 2 * 8
    assert native.RAY_DTYPE.itemsize == 48 and native.HIT_DTYPE.itemsize == 52


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="this check is for machines without a GPU")
def test_no_gpu_means_a_loud_failure_not_a_fallback():
    with pytest.raises(native.PtkError) as e:
        native.Context(0)
    assert "ptk_create" in str(e.value)
    lib = native.load_library()
    assert lib.ptk_render(None, None) != 0 and lib.ptk_upload_scene(None, None) != 0
    assert lib.ptk_last_error(None)  # the reason is available


def test_missing_library_is_reported(monkeypatch, tmp_path):
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "LIB_PATH", str(tmp_path / "libptk.so"))
    with pytest.raises(native.PtkError) as e:
        native.load_library()
    assert "no CPU fallback" in str(e.value)


@pytest.mark.gpu
def test_argument_errors_are_reported(gpu_ctx):
    ctx = gpu_ctx
    L = native.load_library()
    sd = scenes.cornell_box(32, 32)
    ctx.upload_scene(sd)
    ctx.set_camera(sd.camera)
    with pytest.raises(native.PtkError, match="sample_stride"):
        ctx.render(4, 3, sample_stride=0)
    with pytest.raises(native.PtkError, match="row range"):
        ctx.render(4, 3, row_begin=10, row_end=40)
    bad = scenes.cornell_box(32, 32)
    bad.tri_mat = bad.tri_mat + 100
    with pytest.raises(native.PtkError, match="material index"):
        ctx.upload_scene(bad)
    # the failed upload leaves no scene behind: rendering is refused until a good one arrives
    with pytest.raises(native.PtkError, match="upload a scene"):
        ctx.render(1, 1)
    nan = scenes.cornell_box(32, 32)
    nan.tri_pos = nan.tri_pos.copy()
    nan.tri_pos[3, 1, 2] = np.nan
    with pytest.raises(native.PtkError, match="non-finite"):
        ctx.upload_scene(nan)
    ctx.upload_scene(sd)
    cam = native.make_camera(sd.camera)
    cam.width = 0
    with pytest.raises(native.PtkError, match="image size"):
        ctx.set_camera(cam)
    ctx.set_camera(sd.camera)
    ctx.render(1, 1)
    ctx.synchronize()
    assert L.ptk_read_accum(ctx.h, None, None) != 0
