"""The N>1 path on CPU: partitioning + a world_size-2 gloo run of the shard/gather logic,
with the oracle standing in for the per-rank compute (the GPU kernels are covered by
test_gpu_parity; this covers what is distributed: slicing, ordering, padding, gather)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gkl_amd.shard import PipelinedGather, compute_sharded, partition_reads, shard_batch
from gkl_amd.synth import make_batch


def test_partition_balances_work_and_covers_all_reads():
    rng = np.random.RandomState(0)
    lens = rng.randint(50, 251, size=1000)
    for parts in (1, 2, 3, 4, 8):
        b = partition_reads(lens, parts)
        assert b[0] == 0 and b[-1] == 1000 and all(b[i] <= b[i + 1] for i in range(parts))
        work = [lens[b[i]:b[i + 1]].sum() for i in range(parts)]
        assert max(work) - min(work) <= 2 * 250
    assert partition_reads([5], 4) == [0, 0, 0, 1, 1] or partition_reads([5], 4)[-1] == 1


def test_shards_reassemble_the_batch():
    b = make_batch("hc", 37, 5, seed=2)
    rows = 0
    for r in range(3):
        s, bounds = shard_batch(b, r, 3)
        lo, hi = bounds[r], bounds[r + 1]
        assert s.n_reads == hi - lo and s.n_haps == b.n_haps
        assert s.read_bases.tobytes() == b.read_bases[b.read_off[lo]:b.read_off[hi]].tobytes()
        assert np.array_equal(s.read_off, b.read_off[lo:hi + 1] - b.read_off[lo])
        rows += s.n_reads
    assert rows == b.n_reads


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.oracle import Oracle
        oracle = Oracle()
        batch = make_batch("hc", 23, 4, seed=9)  # odd count: shards have different sizes
        full = compute_sharded(batch, lambda s: oracle.batch(s, n_threads=1), device="cpu")
        if rank == 0:
            expect = oracle.batch(batch, n_threads=1)
            ret["ok"] = bool(full is not None and full.tobytes() == expect.tobytes())
        else:
            assert full is None
        # the pipelined variant bench.py uses: 5 steps, 2 rotating buffers, every step a different batch
        shard, bounds = shard_batch(batch, rank, world)
        rows = [bounds[g + 1] - bounds[g] for g in range(world)]
        g = PipelinedGather(rows, batch.n_haps, "cpu", dist)
        for k in range(5):
            out = g.buffer(k)
            out.copy_(torch.from_numpy(oracle.batch(shard, n_threads=1)) + float(k))
            g.submit(k)
        last = g.finish()
        if rank == 0:
            ret["ok_pipelined"] = bool(last is not None and
                                       np.array_equal(last.numpy(), oracle.batch(batch, n_threads=1) + 4.0))
        else:
            assert last is None
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world_size_2_gloo_gather_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret.get("ok") is True
    assert ret.get("ok_pipelined") is True


def _gpu_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gkl_amd import native
        batch = make_batch("hc", 301, 9, seed=11)
        with native.PairHmmContext() as ctx:
            def local(shard):
                out = np.empty(shard.n_pairs)
                ctx.compute(shard, out)
                return out
            full = compute_sharded(batch, local, device="cpu")
        if rank == 0:
            from oracle.oracle import Oracle
            ret["ok_gpu"] = bool(full is not None and full.tobytes() == Oracle().batch(batch, n_threads=4).tobytes())
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_world_size_2_shards_on_the_gpu_match_the_oracle():
    # two ranks share the one GPU of the test box (the driver's multi-GPU run gives each rank its own): every rank
    # runs the HIP path on its read range, rank 0 gathers (gloo here, RCCL in bench.py) and compares bit for bit
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gpu_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret.get("ok_gpu") is True


def _rccl_worker(rank, world, port, ret):
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm
    try:
        from gkl_amd import native
        from gkl_amd.shard import PipelinedGather
        batch = make_batch("hc", 200, 8, seed=12)
        db = native.DeviceBatch.upload(batch, dev)
        streams = torch.cuda.Stream(dev)
        with native.PairHmmContext(device=0) as ctx:
            g = PipelinedGather([batch.n_reads], batch.n_haps, dev, dist, always_collective=True)
            for k in range(4):   # more steps than buffers: a buffer is reused only after its gather has completed
                with torch.cuda.stream(streams):
                    ctx.compute_device(db, g.buffer(k), streams)
                    g.submit(k)
            full = g.finish()
            torch.cuda.synchronize(dev)
            ref = ctx.compute_device(db)
            torch.cuda.synchronize(dev)
        ret["ok_rccl"] = bool(full is not None and torch.equal(full, ref))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_pipelined_gather_on_the_rccl_backend():
    # bench.py's exchange step on the REAL backend (torch "nccl" = RCCL): CUDA tensors, asynchronous gather on its own
    # stream, buffer reuse.  The test box has one GPU and RCCL refuses two ranks on one device, so the group has one
    # rank; the N-rank case differs only in the number of peers and is covered on CPU with gloo.
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rccl_worker, args=(1, port, ret), nprocs=1, join=True)
    assert ret.get("ok_rccl") is True
