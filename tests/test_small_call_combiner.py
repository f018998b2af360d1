"""Concurrent small host-buffer calls (one GATK region each, the reference's re-entrant computeLikelihoodsNative,
SURVEY 8b "Threading"): calls that meet on the device leave in ONE set of launches (pairhmm_api.hip SmallCombiner,
prep_multi_kernel / fwd_stream_multi_kernel / pair_policy_multi_kernel).  Whatever is launched together, every caller
must get the bits the oracle computes for its own batch."""
import threading

import numpy as np
import pytest

from gkl_amd.synth import make_batch, random_batch
from tests import mockjni


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def caller_batches():
    """Shapes that take different variants inside one combined launch: 2 / 4 / 8 rows per lane in the fp32 kernel,
    2 / 4 / 6 rows in the per-pair policy kernel, with and without underflowed pairs, one read, one haplotype."""
    out = [
        make_batch("hc", 100, 10, seed=3),
        make_batch("hc", 40, 6, seed=4, read_len=(20, 90), hap_len=(60, 120)),
        make_batch("hc", 64, 8, seed=5, read_len=(130, 250), hap_len=(200, 400)),
        make_batch("hc", 30, 4, seed=6, read_len=(260, 380), hap_len=(300, 500)),
        make_batch("mixed", 150, 12, seed=7),
        make_batch("hc", 1, 1, seed=8),
        make_batch("hc", 1, 9, seed=9),
        make_batch("hc", 33, 1, seed=10),
    ]
    rng = np.random.RandomState(17)
    out += [random_batch(rng, int(rng.randint(1, 60)), int(rng.randint(1, 12))) for _ in range(8)]
    return out


@pytest.mark.gpu
def test_sixteen_threads_of_small_calls_are_bit_exact_and_get_combined(oracle):
    from gkl_amd import native
    batches = caller_batches()
    want = [oracle.batch(b, n_threads=4) for b in batches]
    native.small_call_counts(0, reset=True)
    iters = 40
    errors = []
    start = threading.Barrier(len(batches))

    def caller(i):
        try:
            with native.PairHmmContext() as c:
                c.compute(batches[i])  # (allocations)
                start.wait()
                for k in range(iters):
                    got = c.compute(batches[i])
                    if not np.array_equal(bits(got), bits(want[i])):
                        errors.append((i, k, int((bits(got) != bits(want[i])).sum())))
                        return
                assert c.stats()["n_fallback"] == int(oracle.batch(batches[i], want_raw=True, n_threads=2)[3].sum())
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))
            try:
                start.abort()
            except Exception:  # noqa: BLE001
                pass

    th = [threading.Thread(target=caller, args=(i,)) for i in range(len(batches))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors[:4]
    calls, combined, sets = native.small_call_counts(0)
    assert calls >= (len(batches) - 2) * iters  # (a call whose plan block exceeds 256 KB keeps the copy-engine path)
    assert combined > 0 and sets < calls, (calls, combined, sets)


@pytest.mark.gpu
def test_combined_calls_through_the_jni_symbols(oracle):
    b = make_batch("hc", 16 * 60, 10, seed=77)
    rc, out, cls, msg, _ = mockjni.run_concurrent(b, n_threads=16, iters=25, max_threads=1)
    assert rc == 0, (cls, msg)
    assert np.array_equal(bits(out), bits(oracle.batch(b, n_threads=8)))


@pytest.mark.gpu
def test_a_lone_caller_is_not_combined(oracle):
    from gkl_amd import native
    b = make_batch("hc", 100, 10, seed=3)
    want = oracle.batch(b, n_threads=4)
    native.small_call_counts(0, reset=True)
    with native.PairHmmContext() as c:
        for _ in range(5):
            assert np.array_equal(bits(c.compute(b)), bits(want))
    calls, combined, sets = native.small_call_counts(0)
    assert (calls, combined, sets) == (5, 0, 5)
    # a call that wants step times keeps the stream-ordered path
    with native.PairHmmContext(record_events=1) as c:
        assert np.array_equal(bits(c.compute(b)), bits(want))
        assert c.stats()["ms_total_device"] > 0
    assert native.small_call_counts(0)[0] == 5


@pytest.mark.gpu
def test_speculated_fp64_beside_fp32_for_a_lone_tiny_call(oracle, monkeypatch):
    """GKLHIP_SPECULATE_FP64=1: a tiny call that is alone on the device runs two wavefronts per pair (fp32 and fp64 at the
    same time, pairhmm_pair_spec_kernel) and lets the policy pick; same bits, same flags, on batches with failing pairs,
    without any, with N / odd bytes and with reads of every rows-per-lane class."""
    from gkl_amd import native
    from gkl_amd.synth import make_batch, random_batch
    monkeypatch.setenv("GKLHIP_SPECULATE_FP64", "1")
    rng = np.random.RandomState(12)
    batches = [make_batch("hc", 100, 10, seed=3), make_batch("region", 80, 12, seed=4),
               random_batch(rng, 40, 7, read_len=(1, 380), hap_len=(1, 500), alphabet=b"ACGTNacgtRY"),
               random_batch(rng, 30, 20, read_len=(130, 383), hap_len=(400, 600), related=False)]
    with native.PairHmmContext() as c:
        for b in batches:
            for _ in range(2):
                out = c.compute(b)
                r32, r64, u = c.raw(b.n_pairs)
                oo, o32, o64, ou = oracle.batch(b, want_raw=True, n_threads=8)
                assert np.array_equal(u, ou)
                assert np.array_equal(r64[u == 1].view(np.uint64), o64[ou == 1].view(np.uint64))
                assert out.tobytes() == oo.tobytes()
