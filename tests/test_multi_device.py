"""Multi-device sharding INSIDE the library (behind the C ABI and the JNI symbols): contiguous read ranges balanced by
cells, one per device of the context's list, results gathered -- what the reference's OpenMP loop over pairs
(IntelPairHmm.cc:151-154) is to its cores.  The one-GPU test box lists device 0 twice (two shards, two streams, the
peer-copy gather); the RCCL calls of the real gather are exercised on a one-device communicator."""
import numpy as np
import pytest

from gkl_amd import shard
from gkl_amd.synth import make_batch, random_batch
from tests import mockjni


def test_library_partition_rule_equals_the_python_one():
    # the torch.distributed harness (gkl_amd/shard.py) and the library must cut a batch at the same reads
    from gkl_amd import native
    rng = np.random.RandomState(5)
    for _ in range(200):
        n = int(rng.randint(1, 400))
        lens = rng.randint(1, 300, size=n)
        off = np.zeros(n + 1, np.int64)
        off[1:] = np.cumsum(lens)
        for parts in (1, 2, 3, 4, 8, 13):
            assert native.partition_reads(off, parts) == shard.partition_reads(lens, parts), (n, parts)
    # fewer reads than parts: empty ranges, still monotone and complete
    off = np.array([0, 10, 30], np.int64)
    b = native.partition_reads(off, 8)
    assert b[0] == 0 and b[-1] == 2 and all(x <= y for x, y in zip(b, b[1:]))


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


@pytest.mark.gpu
@pytest.mark.parametrize("devices,use_double", [([0, 0], False), ([0, 0, 0], False), ([0, 0], True)])
def test_two_shards_on_one_gpu_host_path_bit_exact(oracle, devices, use_double):
    from gkl_amd import native
    b = make_batch("hc", 700, 24, seed=31)
    with native.PairHmmContext(devices=devices, use_double=use_double) as c:
        assert c.n_devices == len(devices) and c.gather_backend == "peer"
        out = c.compute(b)
        r32, r64, u = c.raw(b.n_pairs)
        st = c.stats()
    oo, o32, o64, ou = oracle.batch(b, use_double=use_double, want_raw=True, n_threads=8)
    assert np.array_equal(u, ou)
    if not use_double:
        assert np.array_equal(bits(r32), bits(o32))
    assert np.array_equal(bits(r64[u == 1]), bits(o64[ou == 1]))
    assert np.array_equal(bits(out), bits(oo)), "host-finalised likelihoods are not bit-identical"
    assert st["n_pairs"] == b.n_pairs and st["n_fallback"] == int(ou.sum())
    # fewer reads than devices: empty shards are skipped
    tiny = random_batch(np.random.RandomState(1), 2, 3)
    with native.PairHmmContext(devices=[0, 0, 0, 0]) as c:
        assert np.array_equal(bits(c.compute(tiny)), bits(oracle.batch(tiny, n_threads=2)))


@pytest.mark.gpu
def test_two_shards_on_one_gpu_device_resident_path(oracle):
    import torch
    from gkl_amd import native
    b = make_batch("hc", 900, 32, seed=32)
    db = native.DeviceBatch.upload(b, "cuda:0")
    expd = oracle.batch(b, use_double=True, n_threads=8)
    with native.PairHmmContext(device=0) as one, native.PairHmmContext(devices=[0, 0]) as two:
        ref = one.compute_device(db)
        torch.cuda.synchronize()
        out = torch.full((b.n_pairs,), float("nan"), dtype=torch.float64, device="cuda:0")
        for _ in range(3):  # back-to-back calls: the shards' streams and the caller's must stay ordered
            two.compute_device(db, out)
        torch.cuda.synchronize()
        r32, r64, u = two.raw(b.n_pairs)
    # sharding changes which wavefront computes a pair, never the arithmetic: identical to the single-device result
    assert np.array_equal(bits(out.cpu().numpy()), bits(ref.cpu().numpy()))
    assert np.max(np.abs(out.cpu().numpy() - expd) / np.abs(expd)) < 1e-5
    _, o32, _, ou = oracle.batch(b, want_raw=True, n_threads=8)
    assert np.array_equal(u, ou) and np.array_equal(bits(r32), bits(o32))


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["init", "group", "twice"])
def test_rccl_failure_falls_back_to_peer_copies_bit_exact(oracle, monkeypatch, where):
    """RCCL that cannot be had -- communicator creation failing, a group call failing, a device listed twice --
    must degrade to the peer-copy gather, say so, and change no bit (GKL_HIP_RCCL_FAIL forces the first two; the
    third is real: ncclCommInitAll cannot hold device 0 twice, which is also all a one-GPU box can offer)."""
    import torch
    from gkl_amd import native
    monkeypatch.setenv("GKL_HIP_GATHER", "rccl")
    monkeypatch.setenv("GKL_HIP_QUIET", "1")
    if where != "twice":
        monkeypatch.setenv("GKL_HIP_RCCL_FAIL", where)
    b = make_batch("hc", 600, 20, seed=34)
    db = native.DeviceBatch.upload(b, "cuda:0")
    with native.PairHmmContext(device=0) as one, native.PairHmmContext(devices=[0, 0, 0]) as three:
        ref = one.compute_device(db)
        torch.cuda.synchronize()
        assert three.gather_backend == "rccl"   # what the first call will try
        out = torch.full((b.n_pairs,), float("nan"), dtype=torch.float64, device="cuda:0")
        for _ in range(2):
            three.compute_device(db, out)
        torch.cuda.synchronize()
        assert three.gather_backend == "peer-after-rccl-failure" and three.gather_note
        # the host path of the same context never gathers and is unaffected
        host = three.compute(b)
    assert np.array_equal(bits(out.cpu().numpy()), bits(ref.cpu().numpy()))
    assert np.array_equal(bits(host), bits(oracle.batch(b, n_threads=8)))


@pytest.mark.gpu
def test_config4_batch_through_eight_shards_bit_exact(oracle):
    """BASELINE config 4's sharding arithmetic at N = 8 on one GPU: the 1M-pair batch (8000 reads x 125 haplotypes)
    through a device list of eight entries -- eight read ranges, eight engines, eight gathers -- must equal the
    single-device result bit for bit (device-resident path) and the oracle on a sample (host path)."""
    import torch
    from gkl_amd import native
    b = make_batch("hc", 8000, 125, seed=4)
    db = native.DeviceBatch.upload(b, "cuda:0")
    with native.PairHmmContext(device=0) as one, native.PairHmmContext(devices=[0] * 8) as eight:
        ref = one.compute_device(db)
        out = eight.compute_device(db)
        torch.cuda.synchronize()
        bounds = native.partition_reads(b.read_off, 8)
        assert len(set(bounds)) == 9, "eight non-empty read ranges"
        host = eight.compute(b)
        host1 = one.compute(b)
    assert np.array_equal(bits(out.cpu().numpy()), bits(ref.cpu().numpy()))
    assert np.array_equal(bits(host), bits(host1))
    for lo, hi in ((0, 40), (3990, 4030), (7960, 8000)):   # oracle on the first, a middle and the last read range
        exp = oracle.batch(b.read_slice(lo, hi), n_threads=8)
        assert np.array_equal(bits(host.reshape(b.n_reads, b.n_haps)[lo:hi].ravel()), bits(exp))


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, [0, 0]])
def test_one_context_serves_two_streams_with_an_engine_each(oracle, devices):
    """Consecutive device-resident calls on two streams through ONE context run on two engine sets (the second is
    created by the first call that arrives on another stream), so they overlap instead of waiting for each other's
    scratch; results stay bit-identical, also when a third stream shows up and when the batch changes between calls."""
    import torch
    from gkl_amd import native
    b1 = make_batch("hc", 500, 24, seed=41)
    b2 = make_batch("hc", 300, 16, seed=42)
    d1, d2 = native.DeviceBatch.upload(b1, "cuda:0"), native.DeviceBatch.upload(b2, "cuda:0")
    kw = dict(devices=devices) if devices else dict(device=0)
    with native.PairHmmContext(device=0) as one, native.PairHmmContext(**kw) as c:
        r1, r2 = one.compute_device(d1).clone(), one.compute_device(d2).clone()
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream("cuda:0") for _ in range(3)]
        outs = [(k, torch.full((b1.n_pairs if k % 2 == 0 else b2.n_pairs,), float("nan"), dtype=torch.float64, device="cuda:0"))
                for k in range(9)]
        torch.cuda.synchronize()
        for k, out in outs:
            st = streams[k % 2] if k < 6 else streams[k % 3]
            with torch.cuda.stream(st):
                c.compute_device(d1 if k % 2 == 0 else d2, out, st)
        torch.cuda.synchronize()
        for k, out in outs:
            assert torch.equal(out, r1 if k % 2 == 0 else r2), k
        # and the host path of the same context still answers like the oracle
        assert np.array_equal(bits(c.compute(b2)), bits(oracle.batch(b2, n_threads=8)))


@pytest.mark.gpu
def test_jni_path_shards_over_the_device_list(oracle, monkeypatch):
    # computeLikelihoodsNative itself scales: GKL_HIP_DEVICES is read by initNative
    monkeypatch.setenv("GKL_HIP_DEVICES", "0,0")
    b = make_batch("hc", 300, 12, seed=33)
    rc, out, cls, msg, _ = mockjni.run(b)
    assert rc == 0, (cls, msg)
    assert out.tobytes() == oracle.batch(b, n_threads=8).tobytes()


@pytest.mark.gpu
def test_rccl_send_recv_group_runs():
    # the gather's RCCL calls (dlopen, ncclCommInitAll, grouped ncclSend/ncclRecv) on a one-device communicator
    from gkl_amd import native
    native.rccl_selftest(0)


@pytest.mark.gpu
def test_bad_device_list_is_rejected(monkeypatch):
    from gkl_amd import native
    from gkl_amd.errors import IllegalArgumentException
    with pytest.raises(IllegalArgumentException):
        native.PairHmmContext(devices=[0, 99])
    monkeypatch.setenv("GKL_HIP_DEVICES", "0;1")
    with pytest.raises(IllegalArgumentException):
        native.PairHmmContext()
