"""Pin the oracle (oracle/pairhmm_oracle.c) before anything trusts it.

* against the reference's golden file (104 cases, abs tol 1e-5, both precisions:
  PairHmmUnitTest.java:171-234) and the inline simpleTest vector (:55-89);
* bit-for-bit against vectors produced by the reference's own kernel objects
  (tests/golden/ref_vectors.json, generator committed next to it);
* live against oracle/_ref when it is present on this machine.
"""
import numpy as np
import pytest

from gkl_amd.batch import FlatBatch, HaplotypeDataHolder, ReadDataHolder
from gkl_amd.synth import make_batch, random_batch
from tests.golden_io import batch_from_vector, load_ref_vectors


def test_golden_file_both_precisions(oracle, golden_cases):
    assert len(golden_cases) == 104
    for fma_mode in (0, 1):
        for c in golden_cases:
            r32, r64 = oracle.pair_raw(c["read"], c["q"], c["i"], c["d"], c["c"], c["hap"], fma_mode)
            assert r32 >= 1e-28, "no golden case takes the fp64 fallback"
            assert abs(oracle.finalize_f32(r32) - c["expected"]) <= 1e-5
            assert abs(oracle.finalize_f64(r64) - c["expected"]) <= 1e-5


def test_simple_test_vector(oracle):
    # PairHmmUnitTest.java:55-89: expected -6.022797e-01, abs tol 1e-5
    b = FlatBatch.from_holders([ReadDataHolder(b"ACGT", b"++++", b"++++", b"++++", b"++++")],
                               [HaplotypeDataHolder(b"ACGT")])
    for fma_mode in (0, 1):
        assert abs(oracle.batch(b, fma_mode=fma_mode)[0] - (-6.022797e-01)) <= 1e-5
        assert abs(oracle.batch(b, use_double=True, fma_mode=fma_mode)[0] - (-6.022797e-01)) <= 1e-5


def test_bit_identical_to_reference_vectors(oracle):
    vs = load_ref_vectors()["vectors"]
    assert len(vs) >= 40
    n_pairs = n_fallback = 0
    for v in vs:
        b = batch_from_vector(v)
        for eng, e in v["engines"].items():
            fma_mode = 1 if eng == "2" else 0
            out, r32, r64, u = oracle.batch(b, fma_mode=fma_mode, want_raw=True, n_threads=4)
            assert [int(x) for x in r32.view(np.uint32)] == e["raw32"], v["name"]
            assert [int(x) for x in u] == e["used64"], v["name"]
            fb = np.where(u == 1, r64, 0.0)
            assert [int(x) for x in fb.view(np.uint64)] == e["raw64_fallback"], v["name"]
            assert [int(x) for x in out.view(np.uint64)] == e["out"], v["name"]
            outd, _, r64d, _ = oracle.batch(b, use_double=True, fma_mode=fma_mode, want_raw=True, n_threads=4)
            assert [int(x) for x in r64d.view(np.uint64)] == e["raw64_all"], v["name"]
            assert [int(x) for x in outd.view(np.uint64)] == e["out_double"], v["name"]
            n_pairs += u.size
            n_fallback += int(u.sum())
    assert n_fallback > 50 and n_pairs - n_fallback > 200  # both branches of the policy are pinned


def test_tables_match_reference_live(oracle, reference):
    for which in range(4):
        for dt in (np.float32, np.float64):
            a, b = oracle.table(which, dt), reference.table(which, dt)
            assert a.tobytes() == b.tobytes(), (which, dt)


def test_live_reference_random_batches(oracle, reference):
    rng = np.random.RandomState(99)
    engines = [1, 2] if reference.has_avx512() else [1]
    for trial in range(6):
        kw = [dict(), dict(alphabet=b"ACGTN"), dict(alphabet=b"ACGTNacgtXRY*", qual_range=(0, 255)),
              dict(read_len=(1, 300), hap_len=(1, 520)), dict(related=False), dict(read_len=(60, 70))][trial]
        b = random_batch(rng, 16, 6, **kw)
        for eng in engines:
            reference.set_engine(eng)
            ro, r32, r64, u = reference.batch(b, want_raw=True)
            oo, o32, o64, ou = oracle.batch(b, fma_mode=reference.fma_mode, want_raw=True, n_threads=4)
            assert r32.tobytes() == o32.tobytes() and r64.tobytes() == o64.tobytes()
            assert ro.tobytes() == oo.tobytes() and u.tobytes() == ou.tobytes()
    reference.set_engine(0)


def test_hc_generator_is_deterministic_and_shaped():
    a, b = make_batch("hc", 64, 8), make_batch("hc", 64, 8)
    assert a.read_bases.tobytes() == b.read_bases.tobytes() and a.hap_bases.tobytes() == b.hap_bases.tobytes()
    assert a.read_lens.min() >= 50 and a.read_lens.max() <= 250
    assert a.hap_lens.min() >= 95 and a.hap_lens.max() <= 500
    assert a.cells == int(a.read_lens.sum()) * int(a.hap_lens.sum())


def test_policy_uses_fallback_on_hc(oracle):
    b = make_batch("hc", 24, 6, seed=7)
    out, r32, r64, u = oracle.batch(b, want_raw=True, n_threads=4)
    assert 0 < u.sum() < u.size
    assert np.all(r32[u == 1] < 1e-28) and np.all(r32[u == 0] >= 1e-28)
    # fp32 policy output stays within 1e-5 relative of the fp64 path (north_star tolerance)
    outd = oracle.batch(b, use_double=True, n_threads=4)
    assert np.max(np.abs(out - outd) / np.abs(outd)) < 1e-5


def test_empty_inputs_are_nan_not_crash(oracle):
    # the reference does not guard rslen==0 / haplen==0; the oracle returns NaN
    r32, r64 = oracle.pair_raw(b"", b"", b"", b"", b"", b"ACGT")
    assert np.isnan(r32) and np.isnan(r64)
