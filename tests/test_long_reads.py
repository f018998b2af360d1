"""Reads longer than one wavefront holds (fp32: > 511 bases at 8 rows per lane, fp64: > 639 at 10).

Round 4: such a read spans the wavefronts of ONE workgroup -- the reference's stripes with their carry row
(avx-pairhmm-template.h:249,291-323) side by side, the row handed on through a ring in LDS -- and every wavefront runs
the generated asm program (pairhmm_fwd_wide_kernel, tools/gen_fwd_asm.py).  Jobs that fail the program's preconditions
(fp64: a haplotype with an N) are striped through memory by one wavefront as before; round 5: more than four wavefronts'
worth of rows run in super-stripes, and haplotypes no longer than a wavefront is deep no longer disqualify a job.  Everything must stay bit-exact against the oracle, in both
precisions, mixed with short reads, on both sides of the precision policy."""
import numpy as np
import pytest

from gkl_amd.batch import FlatBatch
from gkl_amd.synth import random_batch

pytestmark = pytest.mark.gpu


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


@pytest.fixture(scope="module")
def native():
    from gkl_amd import native as n
    return n


def check(ctx, oracle, b, use_double=False, fma_mode=1):
    out = ctx.compute(b)
    r32, r64, u = ctx.raw(b.n_pairs)
    oo, o32, o64, ou = oracle.batch(b, use_double=use_double, fma_mode=fma_mode, want_raw=True, n_threads=8)
    assert np.array_equal(u, ou), "fallback flags differ"
    if not use_double:
        assert np.array_equal(bits(r32), bits(o32)), "raw fp32 sums are not bit-identical"
    assert np.array_equal(bits(r64[u == 1]), bits(o64[ou == 1])), "raw fp64 sums are not bit-identical"
    assert np.array_equal(bits(out), bits(oo)), "host-finalised likelihoods are not bit-identical"
    return out, u


def cat_reads(haps, *bs):
    lens = np.concatenate([b.read_lens for b in bs])
    off = np.zeros(lens.size + 1, np.int64)
    off[1:] = np.cumsum(lens)
    j = lambda name: np.concatenate([getattr(b, name) for b in bs])  # noqa: E731
    return FlatBatch(lens.size, haps.n_haps, off, haps.hap_off, j("read_bases"), j("read_quals"), j("ins_gop"), j("del_gop"),
                     j("gcp"), haps.hap_bases)


def related_batch(rng, hap_lens, read_lens, err=0.01):
    """reads cut out of the haplotypes (a few substitutions): pairs whose likelihood mass runs along a diagonal that
    crosses every wavefront boundary of the read -- unrelated long pairs underflow to 0 in both precisions and would
    agree with the oracle whatever the hand-off does"""
    haps = [rng.choice(list(b"ACGT"), size=h).astype(np.uint8) for h in hap_lens]
    reads = []
    for i, rl in enumerate(read_lens):
        h = haps[i % len(haps)]
        rl = min(rl, h.size)
        off = int(rng.randint(0, h.size - rl + 1))
        r = h[off:off + rl].copy()
        for k in rng.choice(rl, size=max(1, int(rl * err)), replace=False):
            r[k] = ord("A") if r[k] != ord("A") else ord("G")
        reads.append(r)
    lens = np.array([r.size for r in reads], np.int64)
    roff = np.zeros(lens.size + 1, np.int64)
    roff[1:] = np.cumsum(lens)
    hoff = np.zeros(len(haps) + 1, np.int64)
    hoff[1:] = np.cumsum([h.size for h in haps])
    n = int(lens.sum())
    return FlatBatch(lens.size, len(haps), roff, hoff, np.concatenate(reads), rng.randint(20, 41, n).astype(np.uint8),
                     rng.randint(35, 46, n).astype(np.uint8), rng.randint(35, 46, n).astype(np.uint8), np.full(n, 10, np.uint8),
                     np.concatenate(haps))


@pytest.mark.parametrize("use_double", [False, True])
def test_reads_of_two_three_four_wavefronts_and_beyond(native, oracle, use_double):
    rng = np.random.RandomState(77)
    # 2 / 3 / 4 wavefronts' worth of rows in either precision, the last sizes the wide kernel takes, and beyond (striped)
    read_lens = [100, 400, 512, 520, 640, 700, 1000, 1030, 1279, 1281, 1500, 1919, 1921, 2040, 2047, 2048, 2548, 2549, 2560, 3000]
    b = related_batch(rng, [3200, 2900, 3100, 600, 64, 65], read_lens)
    with native.PairHmmContext(use_double=use_double, record_events=True) as c:
        out, u = check(c, oracle, b, use_double)
        assert c.stats()["n_long_pairs"] > 0
    if not use_double:
        assert u.any() and not u.all()   # long reads on both sides of the precision policy
    oo = oracle.batch(b, use_double=use_double, n_threads=8).reshape(b.n_reads, b.n_haps)
    assert (oo[np.arange(b.n_reads), np.arange(b.n_reads) % 6][4:] > -400).all(), "the related pairs carry real likelihoods"


def test_wide_and_striped_kernels_agree_and_fallbacks_hold(native, oracle, monkeypatch):
    rng = np.random.RandomState(78)
    # haplotypes no longer than a wavefront is deep next to long ones; N and lower case in the haplotypes; related reads so that
    # some long pairs pass the fp32 policy, unrelated ones so that the fp64 pass meets long reads too
    haps = random_batch(rng, 1, 9, read_len=(10, 20), hap_len=(20, 1800), alphabet=b"ACGTNacgt", qual_range=(10, 45))
    longs = related_batch(rng, [1800, 1500, 1750, 900, 1000, 700, 1650, 1200, 800], [600, 700, 1000, 1300, 1650, 1700, 900, 1100])
    other = random_batch(rng, 4, 9, read_len=(700, 1300), hap_len=(700, 900), related=False, qual_range=(25, 45))
    b = cat_reads(longs, longs, other)          # reads cut out of their haplotypes: both sides of the policy
    b2 = cat_reads(haps, longs, other)          # short / N haplotypes: jobs that take the striped path inside the wide kernel
    res = {}
    for mode in ("wide", "striped"):
        if mode == "striped":
            monkeypatch.setenv("GKLHIP_ASM_GENERAL", "0")   # (the wide kernel's jobs then all take the one-wavefront stripes)
        else:
            monkeypatch.delenv("GKLHIP_ASM_GENERAL", raising=False)
        for dbl in (False, True):
            with native.PairHmmContext(use_double=dbl) as c:
                res[mode, dbl, 0] = check(c, oracle, b, dbl)
                res[mode, dbl, 1] = check(c, oracle, b2, dbl)
    u = res["wide", False, 0][1]
    assert u.any() and not u.all()


@pytest.mark.parametrize("use_double", [False, True])
def test_reads_of_5kb_and_15kb_in_super_stripes(native, oracle, use_double, monkeypatch):
    """Round 5: a read that needs more wavefronts than one workgroup holds runs in SUPER-STRIPES (pairhmm_fwd_super_kernel): the
    wide kernel's wavefronts with their LDS rings inside a super-stripe, the carry row between super-stripes through HBM
    by a helper wavefront that plays producer and consumer on the open ends of the workgroup's array.  5 kb and 15 kb
    reads cut out of their haplotypes (related: the likelihood mass crosses every boundary), sizes on both sides of a
    super-stripe's edge (5 x 512 rows in fp32, 7 x 512 in fp64), against the oracle bit for bit in both precisions; and
    the same call with the super-stripes switched off (one-wavefront stripes through memory) gives the same bits."""
    rng = np.random.RandomState(4711)
    read_lens = [5000, 2559, 2560, 3583, 3584, 5119, 5121, 7168, 15000, 12000, 300]
    b = related_batch(rng, [15500, 16000, 5200, 7400], read_lens, err=0.005)
    with native.PairHmmContext(use_double=use_double, record_events=True) as c:
        out, u = check(c, oracle, b, use_double)
        assert c.stats()["n_long_pairs"] > 0
    oo = oracle.batch(b, use_double=use_double, n_threads=8).reshape(b.n_reads, b.n_haps)
    assert (oo[np.arange(b.n_reads), np.arange(b.n_reads) % 4] > -3000).all(), "the related pairs carry real likelihoods"
    monkeypatch.setenv("GKLHIP_SUPER_LONG", "0")
    with native.PairHmmContext(use_double=use_double) as c:
        assert np.array_equal(bits(c.compute(b)), bits(out))


@pytest.mark.parametrize("use_double", [False, True])
def test_long_reads_in_the_unfused_arithmetic(native, oracle, use_double):
    """fma_mode 0 (the AVX translation unit's arithmetic) through the wide and the super-stripe kernels: the "...n" programs
    (round 5; before, this arithmetic was striped by one wavefront in C++).  Reads of 2 .. 4 wavefronts and beyond,
    related to their haplotypes, against the oracle's unfused arithmetic bit for bit."""
    rng = np.random.RandomState(909)
    b = related_batch(rng, [3300, 2900, 700, 64], [100, 520, 1000, 1500, 2047, 2048, 3000, 4200])
    with native.PairHmmContext(use_double=use_double, fma_mode=0, record_events=True) as c:
        out, u = check(c, oracle, b, use_double, fma_mode=0)
        assert c.stats()["n_long_pairs"] > 0
    # (the two arithmetics differ in the last bits of the raw sums: the test would not pass on the contracted programs)
    with native.PairHmmContext(use_double=use_double, fma_mode=0) as c0, native.PairHmmContext(use_double=use_double, fma_mode=1) as c1:
        c0.compute(b)
        c1.compute(b)
        r0, r1 = c0.raw(b.n_pairs), c1.raw(b.n_pairs)
        k = 1 if use_double else 0
        assert not np.array_equal(bits(r0[k]), bits(r1[k]))


@pytest.mark.parametrize("use_double", [False, True])
def test_long_reads_against_haplotypes_shorter_than_a_wavefront(native, oracle, use_double, monkeypatch):
    """Haplotypes of 1 .. 63 bases put SEVERAL separators into a 64-lane array at once; since round 5 the asm programs take
    such jobs too (a storing lane looks its output column up by the index its separator carries): reads of up to four
    wavefronts run them in the wide kernel instead of the one-wavefront stripes (longer reads keep the stripes for
    such jobs: short streams against a very deep array).  Long reads of two wavefronts up to super-stripe sizes against
    short, medium and long haplotypes in one call (every job's first haplotype is a short one: stream order is by
    ascending length), both precisions; the arrangement without the programs gives the same bits."""
    rng = np.random.RandomState(606)
    haps = random_batch(rng, 1, 14, read_len=(10, 20), hap_len=(1, 63), alphabet=b"ACGT", qual_range=(10, 45))
    mixed = random_batch(rng, 1, 9, read_len=(10, 20), hap_len=(1, 1400), alphabet=b"ACGT", qual_range=(10, 45))
    longs = related_batch(rng, [1800, 1500, 5200], [600, 700, 1000, 1300, 1650, 2047, 2600, 5000])
    for hb in (haps, mixed):
        b = cat_reads(hb, longs)
        with native.PairHmmContext(use_double=use_double, record_events=True) as c:
            out, _ = check(c, oracle, b, use_double)
            assert c.stats()["n_long_pairs"] > 0
        monkeypatch.setenv("GKLHIP_ASM_GENERAL", "0")
        with native.PairHmmContext(use_double=use_double) as c:
            assert np.array_equal(bits(c.compute(b)), bits(out))
        monkeypatch.delenv("GKLHIP_ASM_GENERAL")
