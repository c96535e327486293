"""The N > 1 path on CPU: two processes, `gloo` backend, each rendering its share of a frame with the
host-compiled kernel logic, merged with the same reduce the GPU run uses (pathtracer_amd.parallel).

Checks the sharding contract: the union of the ranks' work is exactly the single-process frame --
bit-identical for the tile split, equal up to fp32 summation order for the spp split."""
import os
import socket

import numpy as np
import pytest

from pathtracer_amd import parallel, scenes


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, split, spp, bounces, out_path):
    import torch
    import torch.distributed as dist
    from tests.hostcheck import hc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = scenes.cornell_with_mesh(2, 40, 40)
    k = hc.HostCheck(sd)
    accum = np.zeros((40, 40, 4), np.float32)
    for c in parallel.frame_plan(split, rank, world, spp, 40, batch=3):
        k.render(c.samples, bounces, seed=4, sample_first=c.sample_first, sample_stride=c.sample_stride,
                 row_begin=c.row_begin, row_end=c.row_end, accum=accum,
                 band_rows=c.band_rows, band_first=c.band_first, band_stride=c.band_stride)
    t = torch.from_numpy(accum)
    merge = parallel.FrameMerge(split, rank, world, t)
    assert merge.bytes_per_rank() == (t.numel() * 4 if split == "spp" else 24 * 40 * 16)  # the largest share: bands 0 and 2 = rows 0..15 and 32..39
    merge.run()
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("split", ["tile", "spp"])
def test_two_ranks_reproduce_the_single_process_frame(split, tmp_path):
    import torch.multiprocessing as mp
    from tests.hostcheck import hc
    hc.build()
    world, spp, bounces = 2, 5, 3
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), split, spp, bounces, out), nprocs=world, join=True)
    merged = np.load(out)
    sd = scenes.cornell_with_mesh(2, 40, 40)
    k = hc.HostCheck(sd)
    total = spp if split == "tile" else spp * world
    single, _ = k.render(total, bounces, seed=4)
    assert (merged[..., 3] == total).all()
    if split == "tile":
        # each pixel is rendered by exactly one rank, in the same batches: same sums bit for bit
        ref = np.zeros_like(single)
        for c in parallel.frame_plan("tile", 0, 1, spp, 40, batch=3):
            k.render(c.samples, bounces, seed=4, sample_first=c.sample_first, sample_stride=c.sample_stride,
                     row_begin=c.row_begin, row_end=c.row_end, accum=ref,
                     band_rows=c.band_rows, band_first=c.band_first, band_stride=c.band_stride)
        assert np.array_equal(merged, ref)
    assert np.allclose(merged[..., :3], single[..., :3], rtol=2e-5, atol=1e-5)


def test_frame_plan_covers_every_sample_once():
    for split in ("spp", "tile"):
        for world in (1, 2, 3, 8):
            height, spp = 70, 7
            seen = {}
            for rank in range(world):
                for c in parallel.frame_plan(split, rank, world, spp, height, batch=4):
                    for y in c.rows(height):
                        for s in range(c.samples):
                            key = (y, c.sample_first + s * c.sample_stride)
                            seen[key] = seen.get(key, 0) + 1
            total = spp * world if split == "spp" else spp
            assert set(seen) == {(y, s) for y in range(height) for s in range(total)}
            assert set(seen.values()) == {1}
    with pytest.raises(ValueError):
        parallel.frame_plan("rows", 0, 1, 4, 16)
    with pytest.raises(ValueError):
        parallel.frame_plan("spp", 2, 2, 4, 16)


def _gpu_worker(rank, world, port, split, spp, bounces, out_path, one_gpu=False):
    import torch
    import torch.distributed as dist
    from pathtracer_amd import native
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    device = 0 if one_gpu else rank
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    if one_gpu:  # RCCL refuses two ranks on one device: the collective goes through gloo
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    sd = scenes.cornell_with_mesh(3, 96, 72)
    ctx = native.Context(device)
    ctx.upload_scene(sd)
    ctx.set_camera(sd.camera)
    accum = torch.zeros((72, 96, 4), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx.bind_accum(accum.data_ptr())
    merge = parallel.FrameMerge(split, rank, world, accum)
    for c in parallel.frame_plan(split, rank, world, spp, 72, batch=16):
        ctx.render(c.samples, bounces, seed=4, sample_first=c.sample_first, sample_stride=c.sample_stride,
                   row_begin=c.row_begin, row_end=c.row_end, band_rows=c.band_rows, band_first=c.band_first,
                   band_stride=c.band_stride)
    ctx.synchronize()  # the library renders on its own stream; the collective runs on torch's
    merge.run()
    torch.cuda.synchronize()
    if rank == 0:
        np.save(out_path, accum.cpu().numpy())
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


def _device_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


def _two_ranks_against_one(split, tmp_path, one_gpu):
    import torch.multiprocessing as mp
    from pathtracer_amd import native
    world, spp, bounces = 2, 32, 3
    out = str(tmp_path / "frame.npy")
    mp.spawn(_gpu_worker, args=(world, _free_port(), split, spp, bounces, out, one_gpu), nprocs=world, join=True)
    merged = np.load(out)
    sd = scenes.cornell_with_mesh(3, 96, 72)
    ctx = native.Context(0)
    try:
        ctx.upload_scene(sd)
        ctx.set_camera(sd.camera)
        total = spp if split == "tile" else spp * world
        if split == "tile":
            for r in range(world):
                for c in parallel.frame_plan("tile", r, world, spp, 72, batch=16):
                    ctx.render(c.samples, bounces, seed=4, sample_first=c.sample_first, sample_stride=c.sample_stride,
                               band_rows=c.band_rows, band_first=c.band_first, band_stride=c.band_stride)
        else:
            ctx.render(total, bounces, seed=4)
        single, n = ctx.read_accum()
    finally:
        ctx.close()
    assert n == total and (merged[..., 3] == total).all()
    mean = merged[..., :3] * (np.float32(1.0) / merged[..., 3:4])  # as ptk_read_accum forms it: fp32 reciprocal, fp32 product
    if split == "tile":
        assert np.array_equal(mean, single)
    assert np.allclose(mean, single, rtol=2e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("split", ["tile", "spp"])
def test_two_gpus_reproduce_the_single_gpu_frame(split, tmp_path):
    """The real N > 1 sequence: libptk renders into a torch tensor bound as the accumulation buffer on each
    GPU, then one RCCL collective merges the frame on rank 0.  Tile split: bit-identical to one GPU rendering
    the same bands; spp split: equal up to fp32 summation order."""
    _two_ranks_against_one(split, tmp_path, one_gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize("split", ["tile", "spp"])
def test_two_ranks_on_one_gpu_reproduce_the_single_gpu_frame(split, tmp_path):
    """The same sequence on a box with one GPU: two processes, each with its own libptk context on GPU 0 and its own
    bound accumulation tensor, the merge through gloo.  Everything but the RCCL transport is what two GPUs would run."""
    _two_ranks_against_one(split, tmp_path, one_gpu=True)
