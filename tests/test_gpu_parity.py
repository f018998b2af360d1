"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle.

Bars (written here, checked below):
* raw scaled sums (fp32 and fp64 kernels): BIT-EXACT against the oracle / the
  reference-generated vectors, for both arithmetic patterns the reference ships;
* host-finalised log10 likelihoods (the JNI path): bit-exact against the reference's
  policy output, and within 1e-5 absolute of the reference tests' stored expectations
  (PairHmmUnitTest.java:88,221);
* device-finalised likelihoods (F64LOG mode): within 1e-5 RELATIVE of the fp64 path
  (north_star tolerance).
"""
import numpy as np
import pytest

from gkl_amd.batch import FlatBatch, HaplotypeDataHolder, ReadDataHolder
from gkl_amd.synth import make_batch, random_batch
from tests.golden_io import batch_from_vector, load_ref_vectors

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5  # north_star: within 1e-5 relative of GKL's double-precision path
ABS_TOL = 1e-5  # reference tests: abs 1e-5 against stored expectations


@pytest.fixture(scope="module")
def native():
    from gkl_amd import native as n
    return n


@pytest.fixture(scope="module")
def ctx32(native):
    with native.PairHmmContext(use_double=False, record_events=True) as c:
        yield c


@pytest.fixture(scope="module")
def ctx64(native):
    with native.PairHmmContext(use_double=True, record_events=True) as c:
        yield c


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def check_against_oracle(ctx, oracle, b, fma_mode=1, use_double=False):
    out = ctx.compute(b)
    r32, r64, u = ctx.raw(b.n_pairs)
    oo, o32, o64, ou = oracle.batch(b, use_double=use_double, fma_mode=fma_mode, want_raw=True, n_threads=8)
    assert np.array_equal(u, ou), "fallback flags differ"
    if not use_double:
        assert np.array_equal(bits(r32), bits(o32)), "raw fp32 sums are not bit-identical"
    fb = u == 1
    assert np.array_equal(bits(r64[fb]), bits(o64[fb])), "raw fp64 sums are not bit-identical"
    assert np.array_equal(bits(out), bits(oo)), "host-finalised likelihoods are not bit-identical"
    return out, u


def test_golden_file_both_precisions(ctx32, ctx64, oracle, golden_cases):
    reads = [ReadDataHolder(c["read"], c["q"], c["i"], c["d"], c["c"]) for c in golden_cases]
    # one call per case, 1 read x 1 hap, like dataFileTest (PairHmmUnitTest.java:171-234)
    for c, r in zip(golden_cases, reads):
        b = FlatBatch.from_holders([r], [HaplotypeDataHolder(c["hap"])])
        for ctx in (ctx32, ctx64):
            assert abs(ctx.compute(b)[0] - c["expected"]) <= ABS_TOL
    # and all 104 in a few calls grouped by haplotype (bit-exact vs oracle)
    by_hap = {}
    for c, r in zip(golden_cases, reads):
        by_hap.setdefault(c["hap"], []).append((c, r))
    for hap, items in by_hap.items():
        b = FlatBatch.from_holders([r for _, r in items], [HaplotypeDataHolder(hap)])
        out, _ = check_against_oracle(ctx32, oracle, b)
        outd, _ = check_against_oracle(ctx64, oracle, b, use_double=True)
        exp = np.array([c["expected"] for c, _ in items])
        assert np.max(np.abs(out - exp)) <= ABS_TOL and np.max(np.abs(outd - exp)) <= ABS_TOL


def test_simple_test_vector(ctx32, ctx64):
    b = FlatBatch.from_holders([ReadDataHolder(b"ACGT", b"++++", b"++++", b"++++", b"++++")],
                               [HaplotypeDataHolder(b"ACGT")])
    assert abs(ctx32.compute(b)[0] - (-6.022797e-01)) <= ABS_TOL
    assert abs(ctx64.compute(b)[0] - (-6.022797e-01)) <= ABS_TOL


@pytest.mark.parametrize("engine", ["2", "1"])
def test_reference_vectors_bit_exact(native, engine):
    """Vectors produced by the reference's own objects: AVX-512 build (FMA) and AVX build."""
    fma_mode = 1 if engine == "2" else 0
    vs = [v for v in load_ref_vectors()["vectors"] if engine in v["engines"]]
    assert len(vs) >= 40
    with native.PairHmmContext(use_double=False, fma_mode=fma_mode) as c32, \
            native.PairHmmContext(use_double=True, fma_mode=fma_mode) as c64:
        for v in vs:
            b = batch_from_vector(v)
            e = v["engines"][engine]
            out = c32.compute(b)
            r32, r64, u = c32.raw(b.n_pairs)
            assert [int(x) for x in bits(r32)] == e["raw32"], v["name"]
            assert [int(x) for x in u] == e["used64"], v["name"]
            assert [int(x) for x in bits(np.where(u == 1, r64, 0.0))] == e["raw64_fallback"], v["name"]
            assert [int(x) for x in bits(out)] == e["out"], v["name"]
            outd = c64.compute(b)
            _, r64d, _ = c64.raw(b.n_pairs)
            assert [int(x) for x in bits(r64d)] == e["raw64_all"], v["name"]
            assert [int(x) for x in bits(outd)] == e["out_double"], v["name"]


@pytest.mark.parametrize("kw", [dict(), dict(alphabet=b"ACGTN"),
                                dict(alphabet=b"ACGTNacgtXRY*", qual_range=(0, 255)),
                                dict(read_len=(1, 300), hap_len=(1, 520)), dict(related=False),
                                dict(read_len=(60, 70), hap_len=(1, 40))])
def test_random_batches_bit_exact(ctx32, ctx64, oracle, kw):
    rng = np.random.RandomState(1234)
    b = random_batch(rng, 37, 11, **kw)
    check_against_oracle(ctx32, oracle, b)
    check_against_oracle(ctx64, oracle, b, use_double=True)


def test_hc_batch_policy_and_tolerance(native, ctx32, ctx64, oracle):
    b = make_batch("hc", 300, 24, seed=11)
    out, u = check_against_oracle(ctx32, oracle, b)
    assert 0 < u.sum() < u.size  # both branches of the policy ran
    outd, _ = check_against_oracle(ctx64, oracle, b, use_double=True)
    assert np.max(np.abs(out - outd) / np.abs(outd)) < REL_TOL
    # device-resident path, double-log finalisation
    import torch
    db = native.DeviceBatch.upload(b)
    dev = ctx32.compute_device(db)
    torch.cuda.synchronize()
    dev = dev.cpu().numpy()
    assert np.max(np.abs(dev - outd) / np.abs(outd)) < REL_TOL
    st = ctx32.stats()
    assert st["n_fallback"] == int(u.sum()) and st["ms_fwd_main"] > 0
    # device emulation of the reference's float formula: within one float ulp of log10f(2^120)
    with native.PairHmmContext(finalize=native.FINALIZE_DEVICE_REF32) as c:
        ref32 = c.compute_device(db).cpu().numpy()
    assert np.max(np.abs(ref32 - out)) <= 3.9e-6
    assert np.mean(ref32 == out) > 0.9


@pytest.mark.parametrize("shape", [(300, 8), (230, 60), (800, 30), (500, 128)])
def test_mid_size_calls_policy_in_two_launches(native, oracle, shape):
    """2049 .. 65536 pairs: the per-pair policy runs as a flag + compaction launch and a recomputation launch whose
    wavefronts sit at the front of the grid (pairhmm_pair_flag_kernel); one launch below, the planned fp64 pass above."""
    n_reads, n_haps = shape
    b = make_batch("hc", n_reads, n_haps, seed=5 + n_reads)
    assert 2048 < b.n_pairs <= 65536
    for fma in (1, 0):
        with native.PairHmmContext(fma_mode=fma) as c:
            out, u = check_against_oracle(c, oracle, b, fma_mode=fma)
            assert c.stats()["n_fallback"] == int(u.sum()) > 0
    # the device finalisation modes write through the flag kernel too
    import torch
    db = native.DeviceBatch.upload(b)
    with native.PairHmmContext(finalize=native.FINALIZE_DEVICE_F64) as c:
        dev = c.compute_device(db)
        torch.cuda.synchronize()
        assert np.max(np.abs(dev.cpu().numpy() - out) / np.abs(out)) < REL_TOL
    with native.PairHmmContext(finalize=native.FINALIZE_DEVICE_REF32) as c:
        ref32 = c.compute_device(db).cpu().numpy()
    assert np.max(np.abs(ref32 - out)) <= 3.9e-6
    if n_reads == 300:
        # more failing pairs than recomputing blocks (the grid is half the pairs: blocks take several), and none at all
        rng = np.random.RandomState(3)
        noisy = random_batch(rng, 120, 30, read_len=(200, 250), hap_len=(250, 300), qual_range=(20, 40), related=False)  # unrelated reads: every pair underflows in fp32
        with native.PairHmmContext() as c:
            _, u = check_against_oracle(c, oracle, noisy)
            assert u.mean() > 0.5 and noisy.n_pairs > 2048
        clean = make_batch("region", 250, 12, seed=8)
        with native.PairHmmContext() as c:
            _, u = check_against_oracle(c, oracle, clean)
            assert clean.n_pairs > 2048
    # reads of 384 bases or more do not fit the one-pair-per-wavefront kernel: such a call takes the planned pass
    long_b = make_batch("hc", 120, 30, seed=9, read_len=(300, 450), hap_len=(400, 600))
    with native.PairHmmContext() as c:
        check_against_oracle(c, oracle, long_b)


@pytest.mark.parametrize("rpl", [4, 8])
def test_every_fp32_kernel_variant_bit_exact(native, oracle, rpl):
    """rows_per_lane 8 = the 8-row kernel (what auto picks for big batches), 4 = the 4-row kernel (what auto
    picks for small batches)."""
    b = make_batch("hc", 150, 12, seed=77)
    rng = np.random.RandomState(abs(rpl))
    b2 = random_batch(rng, 30, 7, read_len=(1, 250), hap_len=(1, 90), alphabet=b"ACGTN")
    with native.PairHmmContext(rows_per_lane=rpl) as c:
        assert c.stats()["rows_per_lane"] == 0
        check_against_oracle(c, oracle, b)
        assert c.stats()["rows_per_lane"] == abs(rpl)
        check_against_oracle(c, oracle, b2)


def test_two_rows_per_lane_and_direct_fp64_jobs(native, oracle):
    """Small calls whose reads have at most 127 bases run two rows per lane (fp32 pass, and the fp64 pass when every
    flagged pair is its own job); `rows_per_lane=2` with longer reads falls back to 4."""
    rng = np.random.RandomState(22)
    short = random_batch(rng, 40, 9, read_len=(1, 127), hap_len=(20, 200), alphabet=b"ACGTN")
    with native.PairHmmContext() as c:
        _, u = check_against_oracle(c, oracle, short)
        assert c.stats()["rows_per_lane"] == 2 and 0 < u.sum() < u.size
    for fma in (0, 1):
        with native.PairHmmContext(rows_per_lane=2, fma_mode=fma) as c:
            check_against_oracle(c, oracle, short, fma_mode=fma)
            assert c.stats()["rows_per_lane"] == 2
            check_against_oracle(c, oracle, make_batch("hc", 60, 8, seed=3), fma_mode=fma)   # reads up to 250 bases
            assert c.stats()["rows_per_lane"] == 4
    # direct jobs with six rows per lane (reads longer than 127 bases), N in haplotypes, and a long read that
    # switches the call back to the packed planner
    mixed = random_batch(rng, 30, 7, read_len=(100, 380), hap_len=(50, 400), alphabet=b"ACGTN")
    longb = random_batch(rng, 12, 6, read_len=(300, 500), hap_len=(200, 500), related=False, qual_range=(25, 45))
    with native.PairHmmContext() as c:
        check_against_oracle(c, oracle, mixed)
        _, u = check_against_oracle(c, oracle, longb)
        assert u.all()


def test_auto_picks_the_kernel_by_batch_size(native, oracle):
    with native.PairHmmContext() as c:
        check_against_oracle(c, oracle, make_batch("hc", 100, 10, seed=5))      # one active region
        assert c.stats()["rows_per_lane"] == 4
        check_against_oracle(c, oracle, make_batch("hc", 1500, 40, seed=6))     # enough jobs to fill the chip
        assert c.stats()["rows_per_lane"] == 8


@pytest.mark.parametrize("rpl", [0, 4])
def test_long_reads_striped_kernel(native, oracle, rpl):
    """Reads longer than a chunk (fp32: > 511 bases at 8 rows/lane, > 255 at 4; fp64: > 255) take the
    striped kernel with the carry row in memory; mixed with short reads in one batch, with and
    without the fp64 fallback, they must stay bit-exact."""
    rng = np.random.RandomState(2024)
    short = random_batch(rng, 20, 5, read_len=(30, 250), hap_len=(40, 700), qual_range=(10, 45))
    longb = random_batch(rng, 7, 5, read_len=(256, 1400), hap_len=(300, 1500), qual_range=(10, 45))
    edge = random_batch(rng, 4, 5, read_len=(511, 513), hap_len=(60, 70), qual_range=(10, 45))
    # one batch: short + long reads against the long batch's haplotypes
    def cat(*bs):
        lens = np.concatenate([b.read_lens for b in bs])
        off = np.zeros(lens.size + 1, np.int64)
        off[1:] = np.cumsum(lens)
        j = lambda name: np.concatenate([getattr(b, name) for b in bs])  # noqa: E731
        return FlatBatch(lens.size, longb.n_haps, off, longb.hap_off, j("read_bases"), j("read_quals"),
                         j("ins_gop"), j("del_gop"), j("gcp"), longb.hap_bases)
    b = cat(short, longb, edge)
    with native.PairHmmContext(rows_per_lane=rpl, record_events=True) as c:
        out, u = check_against_oracle(c, oracle, b)
        assert c.stats()["n_long_pairs"] > 0
        assert u[20 * 5:].any() and not u[20 * 5:].all()  # long reads on both sides of the policy
    with native.PairHmmContext(use_double=True) as c:
        check_against_oracle(c, oracle, b, use_double=True)
    # unrelated long reads: every long pair underflows and is recomputed by the striped fp64 kernel
    unrelated = random_batch(rng, 3, 4, read_len=(600, 900), hap_len=(500, 800), related=False, qual_range=(25, 45))
    with native.PairHmmContext(rows_per_lane=rpl) as c:
        out, u = check_against_oracle(c, oracle, unrelated)
        assert u.all()


def test_full_size_batch_properties(native, oracle):
    """BASELINE config 2/3 at full size (10 000 reads x 128 haps, 1.28 M pairs): too big for the
    oracle as a whole, so check size-independent properties bit-for-bit: (1) every pair's result
    is independent of what else is in the batch (a sub-batch reproduces the full batch's entries),
    (2) permuting reads and haplotypes permutes the outputs, (3) a random sub-batch agrees with
    the oracle, (4) the fp32 policy stays within 1e-5 relative of the fp64 path."""
    b = make_batch("hc")
    rng = np.random.RandomState(42)
    with native.PairHmmContext(record_events=True) as c:
        full = c.compute(b).reshape(b.n_reads, b.n_haps)
        st = c.stats()
        assert st["n_fallback"] > 100000 and st["n_chunks"] > 2000  # packed fp64 fallback path in use
        r32, r64, u = c.raw(b.n_pairs)
        u = u.reshape(b.n_reads, b.n_haps)
        # (1)+(3): 48 random reads against all haplotypes
        pick = np.sort(rng.choice(b.n_reads, 48, replace=False))
        subs = [b.read_slice(int(r), int(r) + 1) for r in pick]
        lens = np.array([s.read_lens[0] for s in subs])
        off = np.zeros(len(subs) + 1, np.int64)
        off[1:] = np.cumsum(lens)
        j = lambda name: np.concatenate([getattr(s, name) for s in subs])  # noqa: E731
        sb = FlatBatch(len(subs), b.n_haps, off, b.hap_off, j("read_bases"), j("read_quals"), j("ins_gop"),
                       j("del_gop"), j("gcp"), b.hap_bases)
        sub = c.compute(sb).reshape(len(subs), b.n_haps)
        assert np.array_equal(bits(sub), bits(full[pick]))
        exp = oracle.batch(sb, n_threads=8).reshape(len(subs), b.n_haps)
        assert np.array_equal(bits(sub), bits(exp))
        # (2): reversed reads and haplotypes
        rb = FlatBatch.from_holders(*[list(reversed(x)) for x in b.read_slice(0, 2000).to_holders()])
        rev = c.compute(rb).reshape(2000, b.n_haps)
        assert np.array_equal(bits(rev[::-1, ::-1]), bits(full[:2000]))
    with native.PairHmmContext(use_double=True) as c64:
        fulld = c64.compute(b).reshape(b.n_reads, b.n_haps)
    # BASELINE config 3 ("bit-matched to GKL double-precision"): the same 48-read sample of the all-fp64 run against
    # the oracle's useDoublePrecision path (IntelPairHmm.cc:153-156), bit for bit
    expd = oracle.batch(sb, use_double=True, n_threads=8).reshape(len(subs), b.n_haps)
    assert np.array_equal(bits(fulld[pick]), bits(expd))
    assert np.max(np.abs(full - fulld) / np.abs(fulld)) < REL_TOL
    assert np.array_equal(bits(full[u == 1]), bits(fulld[u == 1]))  # fallback pairs ARE the fp64 path


def test_asm_fast_loop_matches_the_cxx_build_and_the_oracle(native, oracle):
    """The fp32 / 8-row jobs run as generated asm programs (pairhmm_fwd_asm.h); libgklhip_pairhmm_cxxfast.so is the
    same library with the C++ steps in their place.  Both must give the oracle's bits -- long haplotypes so that most
    columns run in the unrolled loop, N and odd bytes included, one read overflowing next to healthy ones."""
    import os
    cxx = os.path.join(os.path.dirname(native.LIB_PATH), "libgklhip_pairhmm_cxxfast.so")
    assert os.path.exists(cxx), "make -C gkl_amd/csrc builds it"
    rng = np.random.RandomState(808)
    batches = [make_batch("hc", 400, 24, seed=9), make_batch("region", 300, 20, seed=10),
               random_batch(rng, 60, 9, read_len=(1, 500), hap_len=(1, 700), alphabet=b"ACGTNacgtRY"),
               random_batch(rng, 40, 6, read_len=(100, 511), hap_len=(400, 900), qual_range=(0, 255))]
    bad = random_batch(rng, 24, 4, read_len=(150, 250), hap_len=(300, 500), qual_range=(20, 40))
    lo, hi = int(bad.read_off[3]), int(bad.read_off[4])
    bad.ins_gop[lo:hi] = 0
    bad.del_gop[lo:hi] = 0
    bad.gcp[lo:hi] = 60
    with native.PairHmmContext(rows_per_lane=8) as a, native.PairHmmContext(rows_per_lane=8, lib_path=cxx) as c:
        for b in batches:
            check_against_oracle(a, oracle, b)
            ra = [x.copy() for x in a.raw(b.n_pairs)]
            c.compute(b)
            rc = c.raw(b.n_pairs)
            assert np.array_equal(bits(ra[0]), bits(rc[0])) and np.array_equal(ra[2], rc[2])
        oa = a.compute(bad)
        ra32 = a.raw(bad.n_pairs)[0].copy()
        oc = c.compute(bad)
        oo, o32, _, _ = oracle.batch(bad, want_raw=True, n_threads=8)
        others = np.ones(bad.n_pairs, bool)
        others[3 * bad.n_haps:4 * bad.n_haps] = False
        assert np.array_equal(bits(ra32[others]), bits(o32[others])) and np.array_equal(bits(oa[others]), bits(oo[others]))
        assert np.array_equal(bits(oa[others]), bits(oc[others]))


@pytest.mark.parametrize("fma_mode", [1, 0], ids=["avx512-arith", "avx-arith"])
@pytest.mark.parametrize("use_double", [False, True])
def test_whole_job_asm_programs_match_the_cxx_build(native, oracle, use_double, fma_mode, monkeypatch):
    """Round 4: whole jobs -- fill, columns, separator windows, drain -- run as ONE generated asm program per haplotype
    (tools/gen_fwd_asm.py), fp32 at 8 rows per lane and fp64 at 10 (the packed recomputation pass and the all-fp64 mode).
    Three builds / modes must give the same bits as the oracle: the asm programs (default), the round-3 arrangement
    (GKLHIP_ASM_GENERAL=0: asm fast blocks inside C++ general steps; fp64 all C++) and the all-C++ cross-check library.
    Batches: the bench shape; haplotypes shorter than the array is deep (several separators in flight: until round 5 such
    a job took the C++ steps inside the asm build, now the programs look the output column up per lane); haplotypes with N (fp64: four prior planes); lower case and odd bytes;
    reads of one base up to the longest a chunk holds; single-lane reads (every lane feeds the separator itself).
    Round 5: the same for the UNFUSED arithmetic of the reference's AVX translation unit (fma_mode 0: the "...n" programs,
    12 operations per cell) -- what GKL computes on a host without AVX-512."""
    import os
    cxx = os.path.join(os.path.dirname(native.LIB_PATH), "libgklhip_pairhmm_cxxfast.so")
    rng = np.random.RandomState(4242)
    batches = [make_batch("hc", 1400, 50, seed=19),   # > 65 536 pairs: the planned fp64 pass (pairhmm_fwd_jobs_kernel)
               make_batch("mixed", 300, 24, seed=20),
               random_batch(rng, 80, 12, read_len=(1, 500), hap_len=(1, 60), alphabet=b"ACGT"),          # short haplotypes
               random_batch(rng, 70, 10, read_len=(60, 500), hap_len=(80, 700), alphabet=b"ACGTNacgtRY"),  # N, odd bytes
               random_batch(rng, 200, 8, read_len=(1, 7), hap_len=(10, 300)),                            # one lane per read
               random_batch(rng, 30, 5, read_len=(480, 511), hap_len=(520, 900), qual_range=(0, 255))]
    res = {}
    for mode in ("asm", "round3", "cxx"):
        if mode == "round3":
            monkeypatch.setenv("GKLHIP_ASM_GENERAL", "0")
        else:
            monkeypatch.delenv("GKLHIP_ASM_GENERAL", raising=False)
        with native.PairHmmContext(use_double=use_double, rows_per_lane=8, fma_mode=fma_mode, lib_path=cxx if mode == "cxx" else None) as c:
            for i, b in enumerate(batches):
                out = c.compute(b)
                r32, r64, u = c.raw(b.n_pairs)
                res[mode, i] = (out.copy(), r32.copy(), r64.copy(), u.copy())
    for i, b in enumerate(batches):
        oo, o32, o64, ou = oracle.batch(b, use_double=use_double, fma_mode=fma_mode, want_raw=True, n_threads=8)
        for mode in ("asm", "round3", "cxx"):
            out, r32, r64, u = res[mode, i]
            assert np.array_equal(u, ou), (mode, i)
            if not use_double:
                assert np.array_equal(bits(r32), bits(o32)), (mode, i)
            assert np.array_equal(bits(r64[u == 1]), bits(o64[ou == 1])), (mode, i)
            assert np.array_equal(bits(out), bits(oo)), (mode, i)


@pytest.mark.parametrize("fma_mode", [1, 0], ids=["avx512-arith", "avx-arith"])
@pytest.mark.parametrize("use_double", [False, True])
def test_short_haplotype_jobs_in_the_asm_programs(native, oracle, use_double, fma_mode, monkeypatch):
    """Round 5: a job with a haplotype no longer than the array is deep (several separators in flight) runs in the asm
    programs too.  Calls big enough for the planned kernels (> 65 536 pairs: job lists, the packed fp64 pass): haplotypes
    of 1 .. 63 bases only, and of 1 .. 200 (jobs that start with short haplotypes and go on with long ones), reads of one
    base up to the longest a chunk holds; against the oracle bit for bit, and the same bits from the arrangement without
    the general-step programs."""
    rng = np.random.RandomState(6363)
    batches = [random_batch(rng, 560, 120, read_len=(1, 500), hap_len=(1, 63), alphabet=b"ACGT"),
               random_batch(rng, 700, 100, read_len=(1, 300), hap_len=(1, 200), alphabet=b"ACGTN")]
    for b in batches:
        with native.PairHmmContext(use_double=use_double, fma_mode=fma_mode) as c:
            out, _ = check_against_oracle(c, oracle, b, fma_mode=fma_mode, use_double=use_double)
        monkeypatch.setenv("GKLHIP_ASM_GENERAL", "0")
        with native.PairHmmContext(use_double=use_double, fma_mode=fma_mode) as c:
            assert np.array_equal(bits(c.compute(b)), bits(out))
        monkeypatch.delenv("GKLHIP_ASM_GENERAL")


def test_region_batch_no_fallback(ctx32, oracle):
    b = make_batch("region", 200, 16, seed=5)
    out, u = check_against_oracle(ctx32, oracle, b)
    assert u.sum() == 0


def test_isolation_between_pairs_nan_inf(ctx32, oracle):
    """Absurd quals (0) make one read overflow to inf/nan; neighbours in the same wavefront
    must still match the oracle bit for bit."""
    rng = np.random.RandomState(5)
    b = random_batch(rng, 24, 4, read_len=(150, 250), hap_len=(300, 500), qual_range=(20, 40))
    lo, hi = int(b.read_off[3]), int(b.read_off[4])
    for arr in (b.ins_gop, b.del_gop):
        arr[lo:hi] = 0
    b.gcp[lo:hi] = 60
    out = ctx32.compute(b)
    r32, r64, u = ctx32.raw(b.n_pairs)
    oo, o32, o64, ou = oracle.batch(b, want_raw=True, n_threads=8)
    others = np.ones(b.n_pairs, bool)
    others[3 * b.n_haps:4 * b.n_haps] = False
    assert np.array_equal(bits(r32[others]), bits(o32[others]))
    assert np.array_equal(bits(out[others]), bits(oo[others]))


def test_rejects_empty_read_and_null(native, ctx32):
    b = make_batch("hc", 4, 2, seed=3)
    bad = FlatBatch(b.n_reads, b.n_haps, b.read_off.copy(), b.hap_off, b.read_bases, b.read_quals,
                    b.ins_gop, b.del_gop, b.gcp, b.hap_bases)
    bad.read_off[2] = bad.read_off[1]  # empty read
    with pytest.raises(native.IllegalArgumentException):
        ctx32.compute(bad)


def test_empty_batch_is_noop(ctx32):
    b = make_batch("hc", 4, 2, seed=3)
    e = FlatBatch(0, b.n_haps, np.zeros(1, np.int64), b.hap_off, np.zeros(0, np.uint8), np.zeros(0, np.uint8),
                  np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(0, np.uint8), b.hap_bases)
    assert ctx32.compute(e).size == 0
