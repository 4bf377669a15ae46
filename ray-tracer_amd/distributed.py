"""Multi-GPU combination of the float accumulators (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The reference is single-GPU; this follows BASELINE.json: the image partitions by tile rows (RCCL gather) or by
sample slices (RCCL sum-reduce).  The scene is replicated, nothing is exchanged inside the render loop: ONE
collective per frame on the raw-sum accumulators.

* rows:    image rows are cut into bands of `band_rows`; rank r renders bands b with b % world == r into a
           band-packed accumulator (rt_render_params.band_*).  Keyed RNG depends only on (pixel, sample), so the
           gathered image is bit-identical to a 1-GPU render.
* samples: rank r renders samples [r*spp, (r+1)*spp) of every pixel; sum-reduce to dst.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

BAND_ROWS = 16


def partition_params(params, rank: int, world: int, mode: str, spp_per_rank: int, band_rows: int = BAND_ROWS):
    """Fill the partition fields of an rt_render_params (in place) for this rank."""
    if mode == "samples":
        params.sample_begin = rank * spp_per_rank
        params.sample_count = spp_per_rank
        params.band_rows, params.band_stride, params.band_index = 0, 1, 0
    elif mode == "rows":
        params.sample_begin = 0
        params.sample_count = spp_per_rank
        params.band_rows, params.band_stride, params.band_index = band_rows, world, rank
    else:
        raise ValueError(f"unknown partition mode {mode!r}")
    return params


def rows_of_rank(height: int, rank: int, world: int, band_rows: int = BAND_ROWS):
    """Image rows (ascending) owned by `rank` under the band interleave."""
    return [y for y in range(height) if (y // band_rows) % world == rank]


def max_rows(height: int, world: int, band_rows: int = BAND_ROWS) -> int:
    return max(len(rows_of_rank(height, r, world, band_rows)) for r in range(world))


def combine_samples(accum: torch.Tensor, dst: int = 0) -> torch.Tensor:
    """Sum-reduce of sample-sliced accumulators (in place on dst)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


class RowGather:
    """Pre-allocated buffers for the band-packed gather + de-interleave (no allocation per frame)."""

    def __init__(self, width: int, height: int, rank: int, world: int, device, band_rows: int = BAND_ROWS, dst: int = 0):
        self.width, self.height, self.rank, self.world, self.dst = width, height, rank, world, dst
        self.band_rows = band_rows
        self.my_rows = rows_of_rank(height, rank, world, band_rows)
        self.pad_rows = max_rows(height, world, band_rows)
        n = self.pad_rows * width * 3
        self.send = torch.zeros(n, dtype=torch.float32, device=device)
        self.recv = [torch.empty(n, dtype=torch.float32, device=device) for _ in range(world)] if rank == dst else None
        self.full = torch.zeros(height, width, 3, dtype=torch.float32, device=device) if rank == dst else None
        self.index = None
        if rank == dst:
            self.index = [torch.tensor(rows_of_rank(height, r, world, band_rows), dtype=torch.long, device=device)
                          for r in range(world)]

    def __call__(self, accum: torch.Tensor):
        """accum: this rank's band-packed raw sums (len(my_rows)*W*3 floats).  Returns the full [H,W,3] image
        on dst, None elsewhere."""
        n = len(self.my_rows) * self.width * 3
        self.send[:n].copy_(accum[:n])
        if self.world > 1:
            dist.gather(self.send, self.recv, dst=self.dst)
            parts = self.recv
        else:
            parts = [self.send]
        if self.rank != self.dst:
            return None
        for r in range(self.world):
            k = self.index[r].numel()
            self.full.index_copy_(0, self.index[r], parts[r][: k * self.width * 3].view(k, self.width, 3))
        return self.full
