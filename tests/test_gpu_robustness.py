"""Shapes and usage patterns beyond the parity suites: BASELINE config 4 size (1 M pairs), single read /
single haplotype, many tiny haplotypes, repeated calls with changing sizes on one context, concurrent
callers (GATK Spark calls computeLikelihoods from several Java threads)."""
import threading

import numpy as np
import pytest

from gkl_amd.synth import make_batch, random_batch

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


@pytest.fixture(scope="module")
def native():
    from gkl_amd import native as n
    return n


def test_config4_one_million_pairs(native, oracle):
    b = make_batch("hc", 8000, 125, seed=20250418)  # 1 M pairs
    with native.PairHmmContext(record_events=True) as c:
        out = c.compute(b).reshape(8000, 125)
        st = c.stats()
        assert st["n_pairs"] == 1_000_000 and np.isfinite(out).all() and (out < 0).all()
        # a random sample of reads against the oracle, bit-exact
        pick = np.sort(np.random.RandomState(1).choice(8000, 24, replace=False))
        for r in pick[:8]:
            one = b.read_slice(int(r), int(r) + 1)
            assert np.array_equal(bits(oracle.batch(one, n_threads=4)), bits(out[r]))


@pytest.mark.parametrize("shape", [(1, 1), (1, 300), (300, 1), (3, 2)])
def test_degenerate_shapes(native, oracle, shape):
    rng = np.random.RandomState(sum(shape))
    b = random_batch(rng, shape[0], shape[1], read_len=(1, 120), hap_len=(1, 200))
    with native.PairHmmContext() as c:
        assert np.array_equal(bits(c.compute(b)), bits(oracle.batch(b, n_threads=4)))
    with native.PairHmmContext(use_double=True) as c:
        assert np.array_equal(bits(c.compute(b)), bits(oracle.batch(b, use_double=True, n_threads=4)))


def test_many_tiny_haplotypes_and_reads(native, oracle):
    rng = np.random.RandomState(8)
    b = random_batch(rng, 200, 300, read_len=(1, 6), hap_len=(1, 5))
    with native.PairHmmContext() as c:
        assert np.array_equal(bits(c.compute(b)), bits(oracle.batch(b, n_threads=8)))


def test_context_reuse_with_changing_sizes(native, oracle):
    rng = np.random.RandomState(21)
    with native.PairHmmContext() as c:
        for n_reads, n_haps in [(5, 3), (400, 20), (2, 90), (150, 7), (1, 1), (60, 60)]:
            b = make_batch("hc", n_reads, n_haps, seed=n_reads * 1000 + n_haps) if n_reads > 30 else \
                random_batch(rng, n_reads, n_haps)
            assert np.array_equal(bits(c.compute(b)), bits(oracle.batch(b, n_threads=8)))


def test_concurrent_callers_on_one_context(native, oracle):
    batches = [make_batch("hc", 120, 10, seed=s) for s in range(6)]
    expect = [oracle.batch(b, n_threads=2) for b in batches]
    results = [None] * len(batches)
    with native.PairHmmContext() as c:
        def work(i):
            for _ in range(3):
                results[i] = c.compute(batches[i])
        threads = [threading.Thread(target=work, args=(i,)) for i in range(len(batches))]
        [t.start() for t in threads]
        [t.join() for t in threads]
    for r, e in zip(results, expect):
        assert np.array_equal(bits(r), bits(e))


def test_two_contexts_do_not_interfere(native, oracle):
    b1, b2 = make_batch("hc", 200, 12, seed=1), make_batch("region", 150, 9, seed=2)
    with native.PairHmmContext() as c1, native.PairHmmContext(use_double=True) as c2:
        o1a = c1.compute(b1)
        o2 = c2.compute(b2)
        o1b = c1.compute(b1)
    assert np.array_equal(bits(o1a), bits(o1b)) and np.array_equal(bits(o1a), bits(oracle.batch(b1, n_threads=8)))
    assert np.array_equal(bits(o2), bits(oracle.batch(b2, use_double=True, n_threads=8)))


def test_deferred_event_ring_reports_each_pipelined_call(native, oracle):
    # record_events=2: no call synchronises; gklhip_get_step_times reads the HIP-event times afterwards
    import torch
    from gkl_amd.errors import IllegalArgumentException
    b = make_batch("hc", 400, 16, seed=21)
    db = native.DeviceBatch.upload(b)
    outs = [torch.empty(b.n_pairs, dtype=torch.float64, device="cuda") for _ in range(5)]
    with native.PairHmmContext(record_events=2) as c:
        for o in outs:
            c.compute_device(db, o)
        times = [c.step_times(k) for k in range(5)]
        with pytest.raises(IllegalArgumentException):
            c.step_times(5)
    assert all(t[0] > 0 and t[2] >= t[0] + t[1] - 1e-3 for t in times)
    ref = outs[0].cpu().numpy()
    assert all(np.array_equal(o.cpu().numpy(), ref) for o in outs[1:])
    exp = oracle.batch(b, use_double=True, n_threads=4)
    assert np.max(np.abs(ref - exp) / np.abs(exp)) < 1e-5   # device finalisation: tolerance of the north star
    with native.PairHmmContext(record_events=1) as c:
        with pytest.raises(IllegalArgumentException):
            c.step_times(0)


def test_lds_reads_beyond_the_allocation_return_zero(tmp_path):
    """The fp32 general step of the asm programs takes a separator lane's prior rows from beyond the workgroup's LDS
    allocation and relies on the hardware returning 0 there (ISA manuals since GCN3).  tools/ubench_lds_oob.hip asks the
    chip: four address classes, 262 144 reads each, with neighbouring workgroups' allocations right behind ours."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "lds_oob")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(root, "tools", "ubench_lds_oob.hip")],
                   check=True, capture_output=True, timeout=600)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout
    lines = [ln for ln in out.splitlines() if ln.startswith("address class")]
    assert len(lines) == 4 and all(" 0 non-zero" in ln for ln in lines), out


@pytest.mark.gpu
def test_forced_lds_selftest_failure_keeps_the_fp32_general_steps_in_cxx(oracle, monkeypatch, capfd):
    """gklhip_init asks the device once whether a DS read beyond the LDS allocation returns 0 (lds_oob_selftest_kernel) --
    the fp32 whole-job programs fetch a separator lane's priors from there.  GKLHIP_SELFTEST_FAIL=lds_oob makes a context
    behave as if the answer had been no: one line on stderr, the fp32 general steps stay in C++ (the round-3 arrangement,
    what GKLHIP_ASM_GENERAL=0 selects), the fp64 programs -- which do not depend on it -- stay, and every bit is the oracle's:
    bench shape (planned fp64 pass), long reads (wide kernel), a tiny call (fused kernel)."""
    from gkl_amd import native
    from gkl_amd.synth import make_batch, random_batch
    rng = np.random.RandomState(515)
    batches = [make_batch("hc", 1400, 50, seed=19), make_batch("hc", 100, 10, seed=3),
               random_batch(rng, 70, 10, read_len=(60, 500), hap_len=(80, 700), alphabet=b"ACGTNacgtRY"),
               random_batch(rng, 6, 5, read_len=(700, 1300), hap_len=(900, 1500), qual_range=(25, 45))]
    with native.PairHmmContext() as c:   # (the real self-test runs with the first context of the process and must pass here)
        base = [c.compute(b).copy() for b in batches]
    capfd.readouterr()
    monkeypatch.setenv("GKLHIP_SELFTEST_FAIL", "lds_oob")
    with native.PairHmmContext() as c:
        err = capfd.readouterr().err
        assert "GKLHIP_SELFTEST_FAIL=lds_oob" in err and "stay in C++" in err
        for b, ref in zip(batches, base):
            out = c.compute(b)
            r32, r64, u = c.raw(b.n_pairs)
            oo, o32, o64, ou = oracle.batch(b, want_raw=True, n_threads=8)
            assert np.array_equal(u, ou) and r32.tobytes() == o32.tobytes()
            assert r64[u == 1].tobytes() == o64[ou == 1].tobytes()
            assert out.tobytes() == oo.tobytes() == ref.tobytes()


def test_an_idle_context_gives_back_its_streams_and_twin_engines_and_works_on(native, oracle):
    """gklhip_release_idle (what the JNI library's janitor calls for a slot unused for a second): after a 410k-pair host
    call -- twin engines, upload / copy streams -- and a device-resident call on a second stream (a second engine set) the
    context gives them back, and the calls after that -- small, big, device-resident -- return the same bits as before."""
    import torch
    big = make_batch("hc", 3200, 128, seed=3)
    small = make_batch("hc", 100, 10, seed=4)
    with native.PairHmmContext() as c:
        ref_big = c.compute(big).copy()
        ref_small = c.compute(small).copy()
        assert np.array_equal(bits(ref_small), bits(oracle.batch(small, n_threads=4)))
        db = native.DeviceBatch.upload(small, "cuda:0")
        s2 = torch.cuda.Stream()
        dev0 = c.compute_device(db)                    # the caller's current stream: first engine set
        with torch.cuda.stream(s2):
            dev1 = c.compute_device(db, None, s2)      # another stream while the first set is that stream's: second engine set
        torch.cuda.synchronize()
        assert torch.equal(dev0, dev1)
        first = c.release_idle()
        assert first >= 1, first                       # at least the twin engine's stream
        assert c.release_idle() == 0                   # nothing left to give
        assert np.array_equal(bits(c.compute(small)), bits(ref_small))
        assert np.array_equal(bits(c.compute(big)), bits(ref_big))     # (makes the twin engine again)
        with torch.cuda.stream(s2):
            dev2 = c.compute_device(db, None, s2)
        torch.cuda.synchronize()
        assert torch.equal(dev1, dev2)
        assert c.release_idle() >= 1
