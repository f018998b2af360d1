"""Multi-GPU sharding of one PairHMM batch: one process per GPU, reads split into contiguous
ranges, haplotypes replicated, ONE exchange step (gather of the double results to rank 0).

The reference has no distributed path (its only parallelism is an OpenMP loop over pairs,
src/main/native/pairhmm/IntelPairHmm.cc:151-154); (read, haplotype) pairs are independent, so
the shard needs no data-path collective -- only the final gather (RCCL over xGMI when the
backend is "nccl"; "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np

from .batch import FlatBatch


def partition_reads(read_lens: Sequence[int], n_parts: int) -> List[int]:
    """Boundaries b[0]=0 <= ... <= b[n_parts]=n_reads of contiguous read ranges with balanced
    work.  Work of a read is its length (cells = rslen * sum(haplen), and every shard sees all
    haplotypes), so this balances cells, not read counts."""
    lens = np.asarray(read_lens, dtype=np.int64)
    n = lens.size
    if n_parts <= 0:
        raise ValueError("n_parts must be positive")
    cum = np.concatenate([[0], np.cumsum(lens)])
    total = int(cum[-1])
    bounds = [0]
    for p in range(1, n_parts):
        target = total * p / n_parts
        i = int(np.searchsorted(cum, target, side="left"))
        # choose the closer of the two neighbouring cut points
        if i > 0 and (i > n or abs(cum[i - 1] - target) <= abs(cum[min(i, n)] - target)):
            i -= 1
        bounds.append(min(max(i, bounds[-1]), n))
    bounds.append(n)
    return bounds


def shard_batch(batch: FlatBatch, rank: int, world: int):
    """The slice of `batch` rank `rank` computes, plus all boundaries."""
    bounds = partition_reads(batch.read_lens, world)
    return batch.read_slice(bounds[rank], bounds[rank + 1]), bounds


def gather_to_root(local, rows_per_rank: Sequence[int], n_haps: int, dist_module=None, root: int = 0):
    """Gather each rank's [rows_g * n_haps] double results on `root`, in rank order.

    `local` is a 1-D torch float64 tensor (CPU for gloo, CUDA for nccl/RCCL).  Slices are
    padded to the largest shard so one fixed-size gather serves (8 B x pairs/G per rank; at
    1 M pairs over 8 GPUs that is 1 MB per xGMI link, far below the per-link bandwidth).
    Returns the concatenated tensor on root, None elsewhere."""
    import torch
    dist = dist_module
    if dist is None:
        import torch.distributed as dist
    world = dist.get_world_size()
    rank = dist.get_rank()
    if world == 1:
        return local
    max_n = max(rows_per_rank) * n_haps
    buf = local
    if local.numel() != max_n:
        buf = torch.zeros(max_n, dtype=local.dtype, device=local.device)
        buf[:local.numel()] = local
    if rank == root:
        parts = [torch.empty(max_n, dtype=local.dtype, device=local.device) for _ in range(world)]
        dist.gather(buf, gather_list=parts, dst=root)
        return torch.cat([p[: rows_per_rank[g] * n_haps] for g, p in enumerate(parts)])
    dist.gather(buf, gather_list=None, dst=root)
    return None


class PipelinedGather:
    """The same exchange step for a STREAM of equal-shaped batches: the gather of step k is issued
    asynchronously and runs (on RCCL's own stream) while step k+1 computes; `depth` result buffers
    rotate, and a buffer is only written again after the gather that read it has completed.

        g = PipelinedGather(rows_per_rank, n_haps, device, dist)
        for k in range(steps):
            out = g.buffer(k)          # waits (stream-level for RCCL) for the gather of step k-depth
            compute_into(out)
            g.submit(k)
        full = g.finish()              # last step's concatenated result on root, None elsewhere
    """

    def __init__(self, rows_per_rank: Sequence[int], n_haps: int, device, dist_module=None, root: int = 0,
                 depth: int = 2, always_collective: bool = False):
        import torch
        self.dist = dist_module
        if self.dist is None:
            import torch.distributed as dist
            self.dist = dist
        self.rows, self.n_haps, self.root, self.depth = list(rows_per_rank), n_haps, root, depth
        self.world, self.rank = self.dist.get_world_size(), self.dist.get_rank()
        self.n_local = self.rows[self.rank] * n_haps
        self.max_n = max(self.rows) * n_haps
        # local buffers are allocated at the padded size; the compute writes the first n_local entries
        self.local = [torch.zeros(self.max_n, dtype=torch.float64, device=device) for _ in range(depth)]
        # always_collective: issue the gather even in a one-rank group (tests of the real backend on a one-GPU box)
        self.collective = self.world > 1 or always_collective
        self.parts = None
        if self.rank == root and self.collective:
            self.parts = [[torch.empty(self.max_n, dtype=torch.float64, device=device) for _ in range(self.world)]
                          for _ in range(depth)]
        self.works = [None] * depth
        self.last = -1

    def buffer(self, k: int):
        w = self.works[k % self.depth]
        if w is not None:
            w.wait()
            self.works[k % self.depth] = None
        return self.local[k % self.depth][: self.n_local]

    def submit(self, k: int) -> None:
        self.last = k
        if not self.collective:
            return
        slot = k % self.depth
        self.works[slot] = self.dist.gather(self.local[slot], gather_list=self.parts[slot] if self.parts else None,
                                            dst=self.root, async_op=True)

    def finish(self):
        import torch
        for i, w in enumerate(self.works):
            if w is not None:
                w.wait()
                self.works[i] = None
        if self.last < 0:
            return None
        slot = self.last % self.depth
        if not self.collective:
            return self.local[slot][: self.n_local]
        if self.rank != self.root:
            return None
        return torch.cat([p[: self.rows[g] * self.n_haps] for g, p in enumerate(self.parts[slot])])


def compute_sharded(batch: FlatBatch, compute_local: Callable, device: str = "cpu",
                    dist_module=None) -> Optional[np.ndarray]:
    """Shard `batch` over the initialised process group, run `compute_local(shard) -> float64
    array/tensor of shard.n_pairs` on every rank, gather on rank 0 and return the full
    r-major result there (None on other ranks)."""
    import torch
    dist = dist_module
    if dist is None:
        import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    shard, bounds = shard_batch(batch, rank, world)
    rows = [bounds[g + 1] - bounds[g] for g in range(world)]
    local = compute_local(shard)
    if not torch.is_tensor(local):
        local = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64))
    local = local.to(device)
    full = gather_to_root(local, rows, batch.n_haps, dist)
    if full is None:
        return None
    return full.cpu().numpy()
