"""Splitting one frame over several GPUs (one process per GPU) and putting it back together.

Every (pixel, sample) path is independent (the reference's only parallel loop is over image rows,
src/renderer.cpp:31-32), so a frame shards without any exchange while rendering; the one exchange step is the
sum of the accumulation buffers at the end of the frame, which maps onto a single reduce (RCCL over xGMI when
the process group's backend is "nccl").

Two ways to shard, both with GLOBAL sample indices so that the union over ranks is exactly the sample set a
single GPU would render:

  "spp"   rank r renders sample indices r, r + world, r + 2*world, ... of every pixel;
  "tile"  rank r renders all samples of the 16-row bands b with b % world == r (interleaved for balance),
          all of them in one launch per batch.

The accumulation buffer holds (sum r, sum g, sum b, sample count) per pixel.  Interleaved samples are merged
by a SUM reduce of the whole buffer onto rank 0 (W*H*16 bytes per rank).  Disjoint bands need no arithmetic:
each rank sends only its own bands, packed, and rank 0 puts them in place (a gather of W*H*16/world bytes per
rank).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

BAND_ROWS = 16


@dataclass
class RenderCall:
    samples: int
    sample_first: int
    sample_stride: int
    row_begin: int
    row_end: int
    band_rows: int = 0    # > 0: the interleaved bands band_first, band_first + band_stride, ... of this many rows
    band_first: int = 0
    band_stride: int = 1

    def rows(self, height: int):
        """The image rows this call renders."""
        if self.band_rows <= 0:
            return range(self.row_begin, self.row_end if self.row_end > 0 else height)
        return [y for y in range(height)
                if y // self.band_rows >= self.band_first and (y // self.band_rows - self.band_first) % self.band_stride == 0]


def frame_plan(split: str, rank: int, world: int, samples_per_rank: int, height: int, sample_base: int = 0,
               batch: int = 0) -> List[RenderCall]:
    """The ptk_render calls rank `rank` makes for one frame.

    spp split: each rank renders `samples_per_rank` samples (the frame gets world x samples_per_rank).
    tile split: each rank renders `samples_per_rank` samples of its own bands (the frame gets samples_per_rank).
    `batch` > 0 cuts the samples into launches of at most that many (Renderer::render called per batch,
    src/main.cpp:242-252)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    if split not in ("spp", "tile"):
        raise ValueError("split must be 'spp' or 'tile'")
    chunks = []
    done = 0
    while done < samples_per_rank:
        n = min(batch, samples_per_rank - done) if batch > 0 else samples_per_rank
        chunks.append((done, n))
        done += n
    calls: List[RenderCall] = []
    if split == "spp":
        for done, n in chunks:
            calls.append(RenderCall(n, sample_base + rank + done * world, world, 0, height))
    else:
        n_bands = (height + BAND_ROWS - 1) // BAND_ROWS
        if rank < n_bands:
            for done, n in chunks:
                calls.append(RenderCall(n, sample_base + done, 1, 0, 0, BAND_ROWS, rank, world))
    return calls


def reduce_frame(accum, world: int, dst: int = 0):
    """Sum the ranks' accumulation buffers onto rank `dst` (a torch tensor of shape (H, W, 4)).
    One collective per frame.  Needs an initialised torch.distributed process group when world > 1."""
    if world > 1:
        import torch.distributed as dist
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


class FrameMerge:
    """The frame's one exchange step, set up once per image size.

    spp split: dist.reduce(SUM) of the whole buffer.  tile split: every rank packs the rows of its own bands
    into a contiguous tensor (padded to the largest share), one dist.gather brings the shares to rank `dst`,
    which copies them into its buffer; 1/world of the buffer moves per rank instead of all of it."""

    def __init__(self, split: str, rank: int, world: int, accum, dst: int = 0):
        self.split, self.rank, self.world, self.accum, self.dst = split, rank, world, accum, dst
        self.rows = None
        if world > 1 and split == "tile":
            import torch
            height = accum.shape[0]
            shares = [RenderCall(0, 0, 1, 0, 0, BAND_ROWS, r, world).rows(height) for r in range(world)]
            self.most = max(len(s) for s in shares)
            self.rows = [torch.as_tensor(list(s), dtype=torch.long, device=accum.device) for s in shares]
            self.send = torch.zeros((self.most,) + tuple(accum.shape[1:]), dtype=accum.dtype, device=accum.device)
            self.recv = ([torch.zeros_like(self.send) for _ in range(world)] if rank == dst else None)

    def bytes_per_rank(self) -> int:
        if self.world <= 1:
            return 0
        t = self.send if self.rows is not None else self.accum
        return t.numel() * t.element_size()

    def describe(self) -> str:
        if self.rows is not None:
            return f"one RCCL gather of each rank's own bands per frame ({self.bytes_per_rank()} bytes per rank)"
        return f"one RCCL sum-reduce of the accumulation buffer per frame ({self.bytes_per_rank()} bytes per rank)"

    def run(self):
        if self.world <= 1:
            return self.accum
        import torch.distributed as dist
        if self.rows is None:
            dist.reduce(self.accum, dst=self.dst, op=dist.ReduceOp.SUM)
            return self.accum
        mine = self.rows[self.rank]
        if len(mine):
            self.send[: len(mine)] = self.accum[mine]
        dist.gather(self.send, self.recv, dst=self.dst)
        if self.rank == self.dst:
            for r in range(self.world):
                if r != self.rank and len(self.rows[r]):
                    self.accum[self.rows[r]] = self.recv[r][: len(self.rows[r])]
        return self.accum
