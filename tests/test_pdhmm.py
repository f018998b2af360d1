"""PDHMM (BASELINE config 5): oracle pins on CPU, HIP parity on the GPU.

Tolerances: the reference's own bar is abs 1e-4 against the stored expectations
(IntelPDHMMUnitTest.java:33); the HIP path is additionally BIT-EXACT against the oracle's "vector"
arithmetic, which is bit-identical to GKL's AVX2 kernels."""
import numpy as np
import pytest

from gkl_amd.pdhmm_batch import PdhmmBatch
from tests.golden_io import load_pdhmm_file, load_pdhmm_holders_file

FILES = ["pdhmm_syn_990_1_2.txt", "pdhmm_syn_199_68_51.txt", "pdhmm_syn_1412_129_223.txt.gz"]
TOL = 1e-4


@pytest.fixture(scope="module")
def pd_oracle():
    from oracle.pdhmm import PdhmmOracle
    return PdhmmOracle()


@pytest.fixture(scope="module")
def pd_reference():
    from oracle.pdhmm import PdhmmReference
    if not PdhmmReference.available():
        pytest.skip("oracle/_ref/libgkl_ref_pdhmm.so not built")
    return PdhmmReference()


def random_pd_batch(rng, n, read_len=(1, 60), hap_len=(1, 80), flag_rate=0.15, lower=True, with_n=True, odd_haps=0.0):
    """odd_haps: share of the haplotypes that carry bases outside ACGTN (lower case, IUPAC codes, 'n') -- the reference
    compares raw bytes (pdhmm.h:256-262), so 'a' != 'A' but 'R' == 'R'."""
    pairs = []
    alpha = np.frombuffer(b"ACGT" + (b"acgt" if lower else b"") + (b"N" if with_n else b"") + (b"RYn" if odd_haps else b""),
                          dtype=np.int8)
    odd_alpha = np.frombuffer(b"acgtnRYKM*", dtype=np.int8)
    for _ in range(n):
        H = int(rng.randint(hap_len[0], hap_len[1] + 1))
        R = int(rng.randint(read_len[0], read_len[1] + 1))
        hap = np.frombuffer(b"ACGT", dtype=np.int8)[rng.randint(0, 4, H)].copy()
        if with_n and H > 3 and rng.random_sample() < 0.3:
            hap[rng.randint(0, H)] = ord("N")
        if odd_haps and rng.random_sample() < odd_haps:
            k = int(rng.randint(1, 4))
            hap[rng.randint(0, H, k)] = odd_alpha[rng.randint(0, odd_alpha.size, k)]
        pd = np.zeros(H, np.int8)
        for j in range(H):
            u = rng.random_sample()
            if u < flag_rate * 0.4:
                pd[j] = 1 | int(rng.randint(1, 16)) << 3          # SNP with a random allele set
            elif u < flag_rate * 0.6:
                pd[j] = 2                                          # DEL_START
            elif u < flag_rate * 0.8:
                pd[j] = 4                                          # DEL_END
            elif u < flag_rate:
                pd[j] = int(rng.randint(0, 128))                  # arbitrary combination
        if R <= H and rng.random_sample() < 0.7:
            off = int(rng.randint(0, H - R + 1))
            read = hap[off:off + R].copy()
            flip = rng.random_sample(R) < 0.05
            read[flip] = alpha[rng.randint(0, alpha.size, int(flip.sum()))]
        else:
            read = alpha[rng.randint(0, alpha.size, R)]
        q = lambda lo, hi: rng.randint(lo, hi + 1, R).astype(np.int8)  # noqa: E731
        pairs.append((hap, pd, read, q(2, 60), q(5, 70), q(5, 70), q(3, 40)))
    return PdhmmBatch.from_pairs(pairs)


# ------------------------------------------------------------------ oracle pins (CPU)
def test_pdhmm_oracle_matches_expected_files(pd_oracle):
    for f in FILES:
        b, exp = load_pdhmm_file(f)
        for sem in (0, 1, 2):
            st, out = pd_oracle.compute(b, semantics=sem)
            assert st == 0 and np.max(np.abs(out - exp)) <= TOL, (f, sem)


def test_pdhmm_oracle_bit_identical_to_reference_kernels(pd_oracle, pd_reference):
    assert all(np.array_equal(pd_oracle.table(w), pd_reference.table(w)) for w in (0, 1))
    rng = np.random.RandomState(3)
    batches = [load_pdhmm_file(f)[0] for f in FILES[:2]] + [random_pd_batch(rng, 64, read_len=(1, 90), hap_len=(1, 120))]
    for b in batches:
        _, vec = pd_oracle.compute(b, semantics=0)
        _, ser = pd_oracle.compute(b, semantics=1)
        st, ref = pd_reference.compute(b, engine=1)  # AVX2: vector kernel for full groups of 4, scalar for the tail
        assert st == 0
        nv = (b.batch // pd_reference.simd_width(1)) * pd_reference.simd_width(1)
        assert ref[:nv].tobytes() == vec[:nv].tobytes()
        assert ref[nv:].tobytes() == ser[nv:].tobytes()
        st, ref0 = pd_reference.compute(b, engine=0)  # scalar engine = serial semantics everywhere
        assert st == 0 and ref0.tobytes() == ser.tobytes()
        if pd_reference.has_avx512():
            st, ref2 = pd_reference.compute(b, engine=2)
            nv2 = (b.batch // pd_reference.simd_width(2)) * pd_reference.simd_width(2)
            _, vec_fma = pd_oracle.compute(b, semantics=2)  # gcc contracts FMAs in that TU: semantics 2
            assert st == 0 and ref2[:nv2].tobytes() == vec_fma[:nv2].tobytes()
            assert ref2[nv2:].tobytes() == ser[nv2:].tobytes()


def test_pdhmm_oracle_fma_pattern_is_the_avx512_objects(pd_oracle, pd_reference):
    # 4000 random pairs: semantics 2 (fma(c, d, a*b) for a*b + c*d) reproduces GKL's AVX-512 object bit for
    # bit, the unfused semantics 0 does not (and reproduces the AVX2 object instead)
    if not pd_reference.has_avx512():
        pytest.skip("host CPU has no AVX-512")
    b = random_pd_batch(np.random.RandomState(5), 4000, read_len=(20, 150), hap_len=(20, 200))
    _, ref512 = pd_reference.compute(b, engine=2)
    _, ref2 = pd_reference.compute(b, engine=1)
    _, fused = pd_oracle.compute(b, semantics=2)
    _, plain = pd_oracle.compute(b, semantics=0)
    assert fused.tobytes() == ref512.tobytes()      # 4000 = 500 full groups of 8: no scalar tail
    assert plain.tobytes() == ref2.tobytes()
    assert plain.tobytes() != fused.tobytes()


def holders_fixture_batch():
    reads, haps, exp = load_pdhmm_holders_file()
    pairs = [(h[0], h[1], r[0], r[1], r[2], r[3], r[4]) for r in reads for h in haps]  # read-major, JavaData.h:190
    return reads, haps, PdhmmBatch.from_pairs(pairs), exp


def test_pdhmm_oracle_matches_holders_fixture(pd_oracle):
    # pdhmm_new.txt: 276 reads x 48 haplotypes, 13 248 expectations (the Java test keeps its asserts commented
    # out, IntelPDHMMUnitTest.java:546-552; the values do pin the arithmetic: max |diff| 4.8e-6)
    reads, haps, b, exp = holders_fixture_batch()
    assert (len(reads), len(haps), exp.size) == (276, 48, 13248)
    st, vec = pd_oracle.compute(b, semantics=0)
    assert st == 0 and np.max(np.abs(vec - exp)) <= TOL


def test_pdhmm_oracle_negative_quality_is_input_error(pd_oracle):
    b = random_pd_batch(np.random.RandomState(1), 4)
    b.read_ins_qual[0] = -3
    st, _ = pd_oracle.compute(b)
    assert st == 2  # PDHMM_INPUT_DATA_ERROR


def test_pdhmm_library_tables_match_oracle(pd_oracle):
    from gkl_amd import native
    for w in (0, 1):
        assert native.pdhmm_host_table(w).tobytes() == pd_oracle.table(w).tobytes()


def test_pdhmm_no_gpu_fails_loudly():
    import torch
    from gkl_amd import native
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(native.RuntimeException):
        native.PdhmmContext()


def expected_per_position(pd_oracle, b, sem, width):
    """What GKL returns for every POSITION of a paired batch: its vector arithmetic (oracle semantics `sem`) for the
    full groups of `width`, its scalar engine's (semantics 1) for the last batch mod width pairs (pdhmm.h:1264-1270)."""
    nv = b.batch - b.batch % width
    st, vec = pd_oracle.compute(b, semantics=sem)
    if nv == b.batch:
        return st, vec
    st1, ser = pd_oracle.compute(b.subset(list(range(nv, b.batch))), semantics=1)   # (its input checks apply to the tail only)
    return max(st, st1), np.concatenate([vec[:nv], ser])


def test_pdhmm_tail_fixtures_pin_the_oracle_per_position(pd_oracle):
    # reference-generated vectors (tests/golden/make_pdhmm_tail_fixtures.py): batches whose size is not a multiple of
    # the SIMD width, haplotypes that end inside / right after a deletion
    from tests.golden_io import load_pdhmm_tail_vectors
    vectors = load_pdhmm_tail_vectors()
    assert len(vectors) >= 10 and {b.batch % 8 for b, _, _ in vectors} >= set(range(1, 8))
    differ = 0
    for b, e512, e2 in vectors:
        st, got512 = expected_per_position(pd_oracle, b, 2, 8)
        assert st == 0 and got512.tobytes() == e512.tobytes()
        st, got2 = expected_per_position(pd_oracle, b, 0, 4)
        assert st == 0 and got2.tobytes() == e2.tobytes()
        _, vec = pd_oracle.compute(b, semantics=2)
        differ += int((vec.view(np.uint64) != e512.view(np.uint64)).sum())
    assert differ > 0   # the tail positions really are a different arithmetic


# ------------------------------------------------------------------ HIP parity (GPU)
SEMANTICS_OF_FMA_MODE = {1: 2, 0: 0}  # fma_mode of the HIP path -> oracle semantics (AVX-512 / AVX2 arithmetic)


@pytest.fixture(scope="module", params=[1, 0], ids=["avx512-arith", "avx2-arith"])
def pd_ctx(request):
    from gkl_amd import native
    # the position-independent mode (vector arithmetic for every pair; the default is GKL's position-dependent one)
    with native.PdhmmContext(fma_mode=request.param, reference_tail=False) as c:
        c.sem = SEMANTICS_OF_FMA_MODE[request.param]
        c.fma_mode = request.param
        yield c


@pytest.mark.gpu
@pytest.mark.parametrize("fname", FILES)
def test_pdhmm_gpu_fixture_files(pd_ctx, pd_oracle, fname):
    b, exp = load_pdhmm_file(fname)
    out = pd_ctx.compute(b)
    assert np.max(np.abs(out - exp)) <= TOL                      # the reference's own bar
    _, vec = pd_oracle.compute(b, semantics=pd_ctx.sem)
    assert out.tobytes() == vec.tobytes()                         # and bit-exact vs GKL's AVX-512 / AVX2 arithmetic


@pytest.mark.gpu
@pytest.mark.parametrize("fname", FILES)
def test_pdhmm_gpu_fixture_files_default_mode_is_gkl_at_every_position(pd_oracle, fname):
    # the library's default: the AVX-512 object's arithmetic with the scalar engine's tail, i.e. what GKL's
    # computePDHMMNative returns position by position (all 1412 vectors of the big file included)
    from gkl_amd import native
    b, exp = load_pdhmm_file(fname)
    with native.PdhmmContext() as c:
        out = c.compute(b)
    st, ref = pd_oracle.compute_reference(b, fma_mode=1)
    assert st == 0 and out.tobytes() == ref.tobytes()
    assert np.max(np.abs(out - exp)) <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(flag_rate=0.5), dict(read_len=(200, 520), hap_len=(100, 400)),
                                dict(read_len=(1, 8), hap_len=(1, 6), flag_rate=0.6),
                                dict(read_len=(255, 257), hap_len=(60, 70)), dict(read_len=(317, 323), hap_len=(60, 70)),
                                dict(odd_haps=0.3), dict(odd_haps=1.0, flag_rate=0.4, read_len=(20, 120), hap_len=(30, 160))])
def test_pdhmm_gpu_random_batches_bit_exact(pd_ctx, pd_oracle, kw):
    rng = np.random.RandomState(77)
    b = random_pd_batch(rng, 96, **kw)
    out = pd_ctx.compute(b)
    st, vec = pd_oracle.compute(b, semantics=pd_ctx.sem)
    assert st == 0 and out.tobytes() == vec.tobytes()
    # (the reference's scalar engine keeps the deletion state across rows and can differ from its own
    #  vector kernels by 0.2 on such random flag patterns; parity is defined against the vector kernels)


@pytest.mark.gpu
@pytest.mark.parametrize("fma_mode", [1, 0])
def test_pdhmm_gpu_reference_tail_mode_matches_gkl_at_every_position(pd_oracle, fma_mode):
    """GKL_HIP_PDHMM_TAIL=reference / gklhip_pdhmm_set_tail_mode(1): the last batch mod 8 (AVX-512 arithmetic) or
    mod 4 (AVX2) pairs take the scalar engine's arithmetic, so every position is bit-identical to what GKL returns."""
    from gkl_amd import native
    from tests.golden_io import load_pdhmm_tail_vectors
    width = 8 if fma_mode else 4
    with native.PdhmmContext(fma_mode=fma_mode) as c:         # the default mode
        for b, e512, e2 in load_pdhmm_tail_vectors():        # generated by the reference's own objects
            exp = e512 if fma_mode else e2
            assert c.compute(b).tobytes() == exp.tobytes()
        rng = np.random.RandomState(91)
        for n, kw in [(37, dict(with_n=False)), (70, dict(with_n=False, flag_rate=0.5)), (6, dict(with_n=False, read_len=(250, 300), hap_len=(10, 60))),
                      (64, dict(with_n=False)), (5, dict(with_n=False, read_len=(1, 4), hap_len=(1, 3), flag_rate=0.7))]:
            b = random_pd_batch(rng, n, **kw)
            st, exp = expected_per_position(pd_oracle, b, SEMANTICS_OF_FMA_MODE[fma_mode], width)
            assert st == 0 and c.compute(b).tobytes() == exp.tobytes(), (n, kw)
        # the scalar engine rejects a non-ACGT read base under a SNP column (pdhmm-serial.cc:222-248): only when such a
        # pair sits in the tail
        bad = random_pd_batch(rng, 9, with_n=False, read_len=(10, 20), hap_len=(20, 30), flag_rate=0.0)
        bad.read_bases.reshape(9, bad.max_read_len)[8, 0] = ord("N")
        bad.hap_pdbases.reshape(9, bad.max_hap_len)[8, 3] = 1 | 8
        if width == 8:
            with pytest.raises(native.IllegalArgumentException):
                c.compute(bad)
        moved = bad.subset([8, 0, 1, 2, 3, 4, 5, 6, 7])      # the same pair at a vectorised position: fine
        st, exp = expected_per_position(pd_oracle, moved, SEMANTICS_OF_FMA_MODE[fma_mode], width)
        assert st == 0 and c.compute(moved).tobytes() == exp.tobytes()
    # opt-in: the vector arithmetic everywhere (position-independent results)
    with native.PdhmmContext(fma_mode=fma_mode, reference_tail=False) as c:
        b = random_pd_batch(np.random.RandomState(92), 13, with_n=False)
        assert c.compute(b).tobytes() == pd_oracle.compute(b, semantics=SEMANTICS_OF_FMA_MODE[fma_mode])[1].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("fma_mode", [1, 0])
def test_pdhmm_gpu_cross_product_replays_the_reference_batches(pd_oracle, fma_mode, monkeypatch):
    """computeLikelihoodsNative in the reference expands reads x haplotypes into read-major pairs, in batches of
    min(total, maxMemoryInMB / memoryPerPair) pairs, and every batch ends in its own scalar tail (JavaData.h:83-101,
    177-242; pdhmm.h:1264-1268).  The default mode of the cross entry point puts the tails where GKL puts them."""
    from gkl_amd import native
    rng = np.random.RandomState(17)
    reads = random_pd_batch(rng, 37, with_n=False, read_len=(20, 120), hap_len=(1, 2))
    haps = random_pd_batch(rng, 7, with_n=False, read_len=(1, 2), hap_len=(60, 200), flag_rate=0.1)
    pairs = []
    for r in range(reads.batch):
        R = int(reads.read_lengths[r])
        rr = lambda a: a.reshape(reads.batch, reads.max_read_len)[r, :R]  # noqa: E731
        for h in range(haps.batch):
            H = int(haps.hap_lengths[h])
            hh = lambda a: a.reshape(haps.batch, haps.max_hap_len)[h, :H]  # noqa: E731
            pairs.append((hh(haps.hap_bases), hh(haps.hap_pdbases), rr(reads.read_bases), rr(reads.read_qual),
                          rr(reads.read_ins_qual), rr(reads.read_del_qual), rr(reads.gcp)))
    expanded = PdhmmBatch.from_pairs(pairs)
    total = reads.batch * haps.batch   # 259 = 32 * 8 + 3
    with native.PdhmmContext(fma_mode=fma_mode) as c:
        for ref_batch in (0, total, 100, 13, 8, 5):
            st, exp = pd_oracle.compute_reference(expanded, fma_mode=fma_mode, ref_batch=ref_batch)
            assert st == 0 and c.compute_cross(reads, haps, ref_batch).tobytes() == exp.tobytes(), ref_batch
        # the same pairs through the paired entry point: one batch, one tail
        st, exp = pd_oracle.compute_reference(expanded, fma_mode=fma_mode)
        assert c.compute(expanded).tobytes() == exp.tobytes()
    # the batch size the JNI shim derives from maxMemoryInMB is the reference's formula
    per_pair = reads.max_read_len * 5 + haps.max_hap_len * 2 + 8 + 16
    assert native.pdhmm_reference_batch_pairs(1, reads.max_read_len, haps.max_hap_len, 10**9) == 1024 * 1024 // per_pair
    assert native.pdhmm_reference_batch_pairs(512, reads.max_read_len, haps.max_hap_len, total) == total


def cross_product(rng, n_reads, n_haps, read_len, hap_len):
    """reads x haplotypes, read-major: the batches IntelPDHMM.computeLikelihoods builds (IntelPDHMM.java:83-145)."""
    src = random_pd_batch(rng, n_reads, read_len=read_len, hap_len=(1, 2))
    haps = random_pd_batch(rng, n_haps, read_len=(1, 2), hap_len=hap_len)
    pairs = []
    for r in range(n_reads):
        R = int(src.read_lengths[r])
        rr = lambda a: a.reshape(n_reads, src.max_read_len)[r, :R]  # noqa: E731
        for h in range(n_haps):
            H = int(haps.hap_lengths[h])
            hh = lambda a: a.reshape(n_haps, haps.max_hap_len)[h, :H]  # noqa: E731
            pairs.append((hh(haps.hap_bases), hh(haps.hap_pdbases), rr(src.read_bases), rr(src.read_qual),
                          rr(src.read_ins_qual), rr(src.read_del_qual), rr(src.gcp)))
    return PdhmmBatch.from_pairs(pairs)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(40, 6, (1, 60), (1, 90)), (90, 3, (100, 151), (150, 260)),
                                   (12, 4, (200, 600), (50, 300))])
def test_pdhmm_gpu_cross_product_shares_haplotypes(pd_ctx, pd_oracle, shape):
    # whole pairs (any haplotype) ride side by side in one wavefront; reads of 320 rows or more run striped
    n_reads, n_haps, rl, hl = shape
    b = cross_product(np.random.RandomState(n_reads), n_reads, n_haps, rl, hl)
    got = pd_ctx.compute(b)
    _, vec = pd_oracle.compute(b, semantics=pd_ctx.sem)
    assert got.tobytes() == vec.tobytes()
    # permuting the pairs must not change any result (packing must not leak between lanes)
    perm = np.random.RandomState(1).permutation(b.batch)
    assert pd_ctx.compute(b.subset(perm)).tobytes() == vec[perm].tobytes()


@pytest.mark.gpu
def test_pdhmm_gpu_cross_entry_point_with_odd_haplotype_bases(pd_ctx, pd_oracle):
    # some haplotypes carry bases outside ACGTN: their jobs take the byte-comparing steps, the others the bit-test ones
    rng = np.random.RandomState(5)
    reads = random_pd_batch(rng, 70, read_len=(30, 151), hap_len=(1, 2), odd_haps=0.5)
    haps = random_pd_batch(rng, 9, read_len=(1, 2), hap_len=(100, 260), flag_rate=0.08, odd_haps=0.5)
    got = pd_ctx.compute_cross(reads, haps)
    pairs = []
    for r in range(reads.batch):
        R = int(reads.read_lengths[r])
        rr = lambda a: a.reshape(reads.batch, reads.max_read_len)[r, :R]  # noqa: E731
        for h in range(haps.batch):
            H = int(haps.hap_lengths[h])
            hh = lambda a: a.reshape(haps.batch, haps.max_hap_len)[h, :H]  # noqa: E731
            pairs.append((hh(haps.hap_bases), hh(haps.hap_pdbases), rr(reads.read_bases), rr(reads.read_qual),
                          rr(reads.read_ins_qual), rr(reads.read_del_qual), rr(reads.gcp)))
    _, vec = pd_oracle.compute(PdhmmBatch.from_pairs(pairs), semantics=pd_ctx.sem)
    assert got.tobytes() == vec.tobytes()


@pytest.mark.gpu
def test_pdhmm_gpu_cross_entry_point_equals_paired(pd_ctx, pd_oracle):
    # gklhip_pdhmm_compute_cross walks reads x haplotypes on the device; the paired entry point gets the same
    # cross product expanded on the host
    rng = np.random.RandomState(31)
    reads = random_pd_batch(rng, 37, read_len=(1, 400), hap_len=(1, 2))       # includes reads of 320 bases or more (striped)
    haps = random_pd_batch(rng, 11, read_len=(1, 2), hap_len=(1, 260), flag_rate=0.05)
    got = pd_ctx.compute_cross(reads, haps)
    pairs = []
    for r in range(reads.batch):
        R = int(reads.read_lengths[r])
        rr = lambda a: a.reshape(reads.batch, reads.max_read_len)[r, :R]  # noqa: E731
        for h in range(haps.batch):
            H = int(haps.hap_lengths[h])
            hh = lambda a: a.reshape(haps.batch, haps.max_hap_len)[h, :H]  # noqa: E731
            pairs.append((hh(haps.hap_bases), hh(haps.hap_pdbases), rr(reads.read_bases), rr(reads.read_qual),
                          rr(reads.read_ins_qual), rr(reads.read_del_qual), rr(reads.gcp)))
    b = PdhmmBatch.from_pairs(pairs)
    _, vec = pd_oracle.compute(b, semantics=pd_ctx.sem)
    assert got.tobytes() == vec.tobytes()
    assert pd_ctx.compute(b).tobytes() == vec.tobytes()


def expand_cross(reads, haps):
    pairs = []
    for r in range(reads.batch):
        R = int(reads.read_lengths[r])
        rr = lambda a: a.reshape(reads.batch, reads.max_read_len)[r, :R]  # noqa: E731
        for h in range(haps.batch):
            H = int(haps.hap_lengths[h])
            hh = lambda a: a.reshape(haps.batch, haps.max_hap_len)[h, :H]  # noqa: E731
            pairs.append((hh(haps.hap_bases), hh(haps.hap_pdbases), rr(reads.read_bases), rr(reads.read_qual),
                          rr(reads.read_ins_qual), rr(reads.read_del_qual), rr(reads.gcp)))
    return PdhmmBatch.from_pairs(pairs)


@pytest.mark.gpu
def test_pdhmm_gpu_table_kernel_routing_and_parity(pd_ctx, pd_oracle):
    # A haplotype whose columns fall into at most six classes of (base, SNP alleles, 'N') takes the kernel that
    # fetches its match priors from an LDS table; more classes -> the predicate kernel; a base outside ACGTN -> the
    # byte-comparing one.  Same bits whichever kernel computes a pair.
    rng = np.random.RandomState(77)
    one = np.zeros(1, np.int8)
    acgt = np.frombuffer(b"ACGT", dtype=np.int8)

    def hap(n_snp_kinds, with_n=False, odd=False, deletion=True):
        H = int(rng.randint(120, 300))
        b = acgt[rng.randint(0, 4, H)].copy()
        pd = np.zeros(H, np.int8)
        kinds = [(int(rng.randint(0, 4)), int(rng.randint(1, 16))) for _ in range(n_snp_kinds)]
        for k, (base, alleles) in enumerate(kinds):           # every kind at least once, some twice
            for j in rng.choice(np.arange(5, H - 5), 2, replace=False):
                b[j] = acgt[base]
                pd[j] = 1 | alleles << 3
        if with_n:
            b[int(rng.randint(0, H))] = ord("N")
        if odd:
            b[int(rng.randint(0, H))] = ord("a")
        if deletion:
            j = int(rng.randint(10, H - 20))
            pd[j] |= 2
            pd[j + int(rng.randint(1, 6))] |= 4
        return b, pd

    haps_l = [hap(0), hap(1), hap(2), hap(1, with_n=True),            # 4, 5, 6, 6 classes: table
              hap(0, deletion=False), hap(2, deletion=False),
              hap(3), hap(6), hap(2, with_n=True),                      # 7+ classes (all four bases occur in 120+ random columns): predicate
              hap(1, odd=True), hap(0, odd=True)]                       # byte-comparing
    haps = PdhmmBatch.from_pairs([(b, pd, one, one, one, one, one) for b, pd in haps_l])
    reads = random_pd_batch(rng, 150, read_len=(1, 200), hap_len=(1, 2))   # lower-case and 'N' read bases included
    got = pd_ctx.compute_cross(reads, haps)
    tab, pred, odd = pd_ctx.last_routing()
    assert odd == 2 and tab >= 6 and tab + pred + odd == len(haps_l), (tab, pred, odd)
    _, vec = pd_oracle.compute(expand_cross(reads, haps), semantics=pd_ctx.sem)
    assert got.tobytes() == vec.tobytes()
    # the reference's own reads x haplotypes fixture: every haplotype has five or six classes
    r2, h2, b2, _ = holders_fixture_batch()
    src = PdhmmBatch.from_pairs([(one, one, *r) for r in r2])
    hp = PdhmmBatch.from_pairs([(x[0], x[1], one, one, one, one, one) for x in h2])
    got = pd_ctx.compute_cross(src, hp)
    assert pd_ctx.last_routing() == (len(h2), 0, 0)
    assert got.tobytes() == pd_oracle.compute(b2, semantics=pd_ctx.sem)[1].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_pdhmm_gpu_table_kernel_flags_anywhere(pd_ctx, pd_oracle, seed):
    # The table kernel's whole-job program switches between plain and general steps by the haplotype's next-special-
    # column table and runs idle lanes on an all-zero prior class: deletion flags on the first and last columns (no
    # lead-in possible), both flags on one column, dense and unbalanced flags, haplotypes shorter than the wavefront is
    # deep, one-base reads, many haplotypes per group -- all with at most six column classes, so that every
    # haplotype takes the table kernel.
    rng = np.random.RandomState(1000 + seed)
    one = np.zeros(1, np.int8)
    acgt = np.frombuffer(b"ACGT", dtype=np.int8)
    haps_l = []
    for k in range(40):
        H = int(rng.choice([1, 2, 3, 5, 17, 40, 64, 65, 130, 257])) if k < 20 else int(rng.randint(1, 300))
        b = acgt[rng.randint(0, 4, H)].copy()
        pd = np.zeros(H, np.int8)
        rate = [0.0, 0.02, 0.1, 0.4][k % 4]
        for j in range(H):
            if rng.rand() < rate:
                pd[j] |= int(rng.choice([2, 4, 6]))
        if k % 3 == 0:
            pd[0] |= int(rng.choice([2, 4, 6]))
        if k % 5 == 0:
            pd[H - 1] |= int(rng.choice([2, 4, 6]))
        if k % 7 == 0 and H > 1:
            pd[1] |= 4
        if k % 2 == 0 and H > 6:                                  # one SNP kind (a fifth class), twice
            for j in rng.choice(np.arange(H), 2, replace=False):
                b[j] = acgt[1]
                pd[j] |= 1 | (5 << 3)
        haps_l.append((b, pd))
    haps = PdhmmBatch.from_pairs([(b, pd, one, one, one, one, one) for b, pd in haps_l])
    reads = random_pd_batch(rng, 120, read_len=(1, 130), hap_len=(1, 2))
    got = pd_ctx.compute_cross(reads, haps)
    tab, pred, odd = pd_ctx.last_routing()
    assert tab == len(haps_l), (tab, pred, odd)
    _, vec = pd_oracle.compute(expand_cross(reads, haps), semantics=pd_ctx.sem)
    assert got.tobytes() == vec.tobytes()
    # the all-C++ cross-check build of the same kernel (libgklhip_pdhmm_cxx.so: step loops, ballots, idle tests)
    import os
    from gkl_amd import native
    cxx = os.path.join(os.path.dirname(native.PDHMM_LIB_PATH), "libgklhip_pdhmm_cxx.so")
    assert os.path.exists(cxx), "make -C gkl_amd/csrc builds it"
    with native.PdhmmContext(fma_mode=pd_ctx.fma_mode, reference_tail=False, lib_path=cxx) as c:
        assert c.compute_cross(reads, haps).tobytes() == vec.tobytes()


@pytest.mark.gpu
def test_pdhmm_gpu_table_kernel_haplotype_groups(pd_ctx, pd_oracle):
    # With enough work per wavefront the table launch hands out GROUPS of up to six haplotypes per chunk of reads (rows
    # set up once, matrices restarted per haplotype, the LDS table kept while the class list stays the same) and packs
    # the reads in windows of 2048 -- none of which a small call reaches.  The reference's fixture with its reads
    # replicated 24 times (6624 reads x 48 haplotypes) must give every replica the bits of the single call, which is
    # checked against the oracle; then the same with haplotypes whose class lists differ (five classes each, seven in
    # their union: no common list, so the table is rebuilt inside a group).
    one = np.zeros(1, np.int8)
    r2, h2, b2, _ = holders_fixture_batch()
    hp = PdhmmBatch.from_pairs([(x[0], x[1], one, one, one, one, one) for x in h2])
    src1 = PdhmmBatch.from_pairs([(one, one, *r) for r in r2])
    base = pd_ctx.compute_cross(src1, hp)
    assert base.tobytes() == pd_oracle.compute(b2, semantics=pd_ctx.sem)[1].tobytes()
    reps = 24
    srcn = PdhmmBatch.from_pairs([(one, one, *r) for r in r2] * reps)
    got = pd_ctx.compute_cross(srcn, hp).reshape(reps, -1)
    assert pd_ctx.last_routing() == (len(h2), 0, 0)
    for k in range(reps):
        assert got[k].tobytes() == base.tobytes(), k
    # class lists that differ from haplotype to haplotype
    rng = np.random.RandomState(99)
    acgt = np.frombuffer(b"ACGT", dtype=np.int8)
    haps_l = []
    for k in range(30):
        H = int(rng.randint(150, 340))
        b = acgt[rng.randint(0, 4, H)].copy()
        pd = np.zeros(H, np.int8)
        base_k, alleles = [(1, 5), (3, 3), (2, 9)][k % 3]
        for j in rng.choice(np.arange(5, H - 5), 3, replace=False):
            b[j] = acgt[base_k]
            pd[j] = 1 | alleles << 3
        j = int(rng.randint(10, H - 30))
        pd[j] |= 2
        pd[j + int(rng.randint(1, 9))] |= 4
        haps_l.append((b, pd))
    hq = PdhmmBatch.from_pairs([(b, pd, one, one, one, one, one) for b, pd in haps_l])
    reads1 = random_pd_batch(rng, 220, read_len=(20, 120), hap_len=(1, 2))
    n1 = reads1.batch
    rr = lambda a, r: a.reshape(n1, reads1.max_read_len)[r, :int(reads1.read_lengths[r])]  # noqa: E731
    rows = [(one, one, rr(reads1.read_bases, r), rr(reads1.read_qual, r), rr(reads1.read_ins_qual, r), rr(reads1.read_del_qual, r),
             rr(reads1.gcp, r)) for r in range(n1)]
    base = pd_ctx.compute_cross(PdhmmBatch.from_pairs(rows), hq)
    assert pd_ctx.last_routing()[0] == len(haps_l)
    _, vec = pd_oracle.compute(expand_cross(PdhmmBatch.from_pairs(rows), hq), semantics=pd_ctx.sem)
    assert base.tobytes() == vec.tobytes()
    got = pd_ctx.compute_cross(PdhmmBatch.from_pairs(rows * 40), hq).reshape(40, -1)
    assert pd_ctx.last_routing()[0] == len(haps_l)
    for k in range(40):
        assert got[k].tobytes() == base.tobytes(), k


@pytest.mark.gpu
def test_pdhmm_gpu_paired_table_route_and_parity(pd_ctx, pd_oracle, monkeypatch):
    # computePDHMMNative's layout (every pair its own haplotype item) through the table kernel: the column classes, the
    # special columns and the routing of the packed jobs are found on the DEVICE.  Pairs over eligible haplotypes (at
    # most six classes), ineligible ones (seven or more: predicate kernel) and ones with bases outside ACGTN
    # (byte-comparing kernel), interleaved so that jobs of all three kinds arise -- and one job never mixes kernels
    # silently: the bits are the oracle's whichever launch computed a pair.
    rng = np.random.RandomState(4242)
    acgt = np.frombuffer(b"ACGT", dtype=np.int8)

    def hap(kind):
        H = int(rng.randint(60, 260))
        b = acgt[rng.randint(0, 4, H)].copy()
        pd = np.zeros(H, np.int8)
        n_snp = {"tab": int(rng.randint(0, 3)), "many": 4, "odd": 1}[kind]
        for k in range(n_snp):
            alleles = [3, 5, 9, 6][k]
            for j in rng.choice(np.arange(2, H - 2), 2, replace=False):
                b[j] = acgt[k % 4]
                pd[j] = 1 | alleles << 3
        if kind == "odd":
            b[int(rng.randint(0, H))] = ord("r")
        if rng.random_sample() < 0.8:
            j = int(rng.randint(3, H - 12))
            pd[j] |= 2
            pd[j + int(rng.randint(1, 9))] |= 4
        if rng.random_sample() < 0.2:
            pd[0] |= int(rng.choice([2, 4]))
        if rng.random_sample() < 0.2:
            pd[H - 1] |= int(rng.choice([2, 4, 6]))
        return b, pd

    src = random_pd_batch(rng, 600, read_len=(1, 150), hap_len=(1, 2))
    n1 = src.batch
    rr = lambda a, r: a.reshape(n1, src.max_read_len)[r, :int(src.read_lengths[r])]  # noqa: E731
    # blocks of 120 pairs of one kind: the packing window (192 pairs, sorted by haplotype length) then yields jobs that
    # are purely of one kind as well as mixed ones
    kinds = ["tab"] * 240 + ["many"] * 60 + ["tab"] * 120 + ["odd"] * 60 + ["tab"] * 120
    pairs = []
    for r, kind in enumerate(kinds):
        hb, hp = hap(kind)
        pairs.append((hb, hp, rr(src.read_bases, r), rr(src.read_qual, r), rr(src.read_ins_qual, r), rr(src.read_del_qual, r), rr(src.gcp, r)))
    b = PdhmmBatch.from_pairs(pairs)
    got = pd_ctx.compute(b)
    tab, pred, full = pd_ctx.last_routing()
    assert tab > 0 and pred > 0 and full > 0, (tab, pred, full)
    st, vec = pd_oracle.compute(b, semantics=pd_ctx.sem)
    assert st == 0 and got.tobytes() == vec.tobytes()
    # only eligible haplotypes: every packed job is the table kernel's
    only = b.subset(np.array([i for i, k in enumerate(kinds) if k == "tab"]))
    got = pd_ctx.compute(only)
    tab, pred, full = pd_ctx.last_routing()
    assert tab > 0 and pred == 0 and full == 0, (tab, pred, full)
    assert got.tobytes() == vec[[i for i, k in enumerate(kinds) if k == "tab"]].tobytes()
    # the same pairs with the table route switched off (GKL_HIP_PDHMM_TABLE=0: predicate kernel) and through the all-C++
    # cross-check build of the table kernel (ballots on the lanes' own entries instead of the job's next-special-step table)
    import os
    from gkl_amd import native
    cxx = os.path.join(os.path.dirname(native.PDHMM_LIB_PATH), "libgklhip_pdhmm_cxx.so")
    with native.PdhmmContext(fma_mode=pd_ctx.fma_mode, reference_tail=False, lib_path=cxx) as c:
        assert c.compute(b).tobytes() == vec.tobytes()
        assert c.last_routing()[0] > 0
    monkeypatch.setenv("GKL_HIP_PDHMM_TABLE", "0")
    with native.PdhmmContext(fma_mode=pd_ctx.fma_mode, reference_tail=False) as c:
        assert c.compute(b).tobytes() == vec.tobytes()
        assert c.last_routing()[0] == 0 and c.last_routing()[1] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
def test_pdhmm_gpu_paired_table_route_flags_anywhere(pd_ctx, pd_oracle, seed):
    # The whole-job program on jobs whose lanes sit on DIFFERENT haplotypes: it switches between plain and general
    # steps on the job's next-special-step table, so special columns of one pair force general steps on its wavefront
    # mates.  Deletion flags on first / last columns, both on one column, dense and unbalanced, haplotypes of 1..300
    # columns next to each other, one-base reads; at most six classes everywhere (all table jobs).
    rng = np.random.RandomState(2000 + seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.int8)
    src = random_pd_batch(rng, 500, read_len=(1, 130), hap_len=(1, 2))
    n1 = src.batch
    rr = lambda a, r: a.reshape(n1, src.max_read_len)[r, :int(src.read_lengths[r])]  # noqa: E731
    pairs = []
    for k in range(n1):
        H = int(rng.choice([1, 2, 3, 5, 17, 40, 64, 65, 130, 257])) if k % 3 == 0 else int(rng.randint(1, 300))
        hb = acgt[rng.randint(0, 4, H)].copy()
        pd = np.zeros(H, np.int8)
        rate = [0.0, 0.02, 0.1, 0.4][k % 4]
        for j in range(H):
            if rng.rand() < rate:
                pd[j] |= int(rng.choice([2, 4, 6]))
        if k % 3 == 0:
            pd[0] |= int(rng.choice([2, 4, 6]))
        if k % 5 == 0:
            pd[H - 1] |= int(rng.choice([2, 4, 6]))
        if k % 7 == 0 and H > 1:
            pd[1] |= 4
        if k % 2 == 0 and H > 6:
            for j in rng.choice(np.arange(H), 2, replace=False):
                hb[j] = acgt[1]
                pd[j] |= 1 | (5 << 3)
        pairs.append((hb, pd, rr(src.read_bases, k), rr(src.read_qual, k), rr(src.read_ins_qual, k), rr(src.read_del_qual, k), rr(src.gcp, k)))
    b = PdhmmBatch.from_pairs(pairs)
    got = pd_ctx.compute(b)
    tab, pred, full = pd_ctx.last_routing()
    assert tab > 0 and pred == 0 and full == 0, (tab, pred, full)
    st, vec = pd_oracle.compute(b, semantics=pd_ctx.sem)
    assert st == 0 and got.tobytes() == vec.tobytes()


@pytest.mark.gpu
def test_pdhmm_gpu_paired_large_batch_by_replication(pd_ctx, pd_oracle):
    # The paired entry point at a size no other test reaches: inputs past the pinned staging block (direct copies from a
    # helper thread), the packing expanded and routed on the device, job arrays past their staging block, threaded
    # log10.  The holders fixture as 13 248 explicit pairs, then nine copies of it in one call (119 232 pairs): every
    # copy must carry the bits of the single call, which is checked against the oracle.
    _, _, b1, _ = holders_fixture_batch()
    base = pd_ctx.compute(b1)
    assert base.tobytes() == pd_oracle.compute(b1, semantics=pd_ctx.sem)[1].tobytes()
    reps = 9
    big = b1.subset(np.tile(np.arange(b1.batch), reps))
    got = pd_ctx.compute(big).reshape(reps, -1)
    tab, pred, full = pd_ctx.last_routing()
    assert tab > 0 and pred == 0 and full == 0, (tab, pred, full)   # real PD haplotypes: every job is the table kernel's
    for k in range(reps):
        assert got[k].tobytes() == base.tobytes(), k


@pytest.mark.gpu
def test_pdhmm_gpu_paired_sliced_call_equals_the_unsliced_one(pd_ctx, pd_oracle, monkeypatch):
    # A big paired call (>= 65 536 pairs and >= 64 MB of padded input) is cut into slices of consecutive pairs whose
    # kernels run while the later slices still cross PCIe: per-slice packing, entry / expand / collect / special / table
    # launches over ranges of items and listed jobs, the striped, odd-base, ineligible and tail jobs at the end.  The
    # holders fixture six times over with 3 000 random pairs of every other kind spread through it (reads of up to 420
    # bases: striped jobs; haplotypes with bases outside ACGTN; with seven or more column classes) and a pair count that
    # leaves a tail: same bits as the same call unsliced (GKL_HIP_PDHMM_PIPELINE=0), and the oracle's on the random pairs.
    from gkl_amd import native
    _, _, b1, _ = holders_fixture_batch()
    rng = np.random.RandomState(6060)
    extra = random_pd_batch(rng, 3003, read_len=(1, 420), hap_len=(20, 330), odd_haps=0.2)
    base = b1.pairs() * 6
    at = np.sort(rng.choice(len(base), extra.batch, replace=False))
    pairs, e = [], extra.pairs()
    nxt = 0
    for i, pr in enumerate(base):
        if nxt < len(at) and at[nxt] == i:
            pairs.append(e[nxt])
            nxt += 1
        pairs.append(pr)
    big = PdhmmBatch.from_pairs(pairs)
    assert big.batch >= 65536 and big.batch % 8 != 0 and big.batch * (2 * big.max_hap_len + 5 * big.max_read_len) >= 64 << 20
    got = pd_ctx.compute(big)
    tab, pred, full = pd_ctx.last_routing()
    assert tab > 0 and pred > 0 and full > 0, (tab, pred, full)
    monkeypatch.setenv("GKL_HIP_PDHMM_PIPELINE", "0")
    with native.PdhmmContext(fma_mode=pd_ctx.fma_mode, reference_tail=False) as c:
        assert c.compute(big).tobytes() == got.tobytes()
        assert c.last_routing() != (0, 0, 0)
    # ... and sliced with the table route switched off (the predicate launch then walks each slice's range of listed jobs)
    monkeypatch.delenv("GKL_HIP_PDHMM_PIPELINE")
    monkeypatch.setenv("GKL_HIP_PDHMM_TABLE", "0")
    with native.PdhmmContext(fma_mode=pd_ctx.fma_mode, reference_tail=False) as c:
        assert c.compute(big).tobytes() == got.tobytes()
        assert c.last_routing()[0] == 0 and c.last_routing()[1] > 0
    monkeypatch.delenv("GKL_HIP_PDHMM_TABLE")
    monkeypatch.setenv("GKL_HIP_PDHMM_PIPELINE", "0")
    # the random pairs sit at positions at[k] + k
    pos = at + np.arange(len(at))
    st, vec = pd_oracle.compute(extra, semantics=pd_ctx.sem)
    assert st == 0 and got[pos].tobytes() == vec.tobytes()
    # the default mode (GKL's scalar tail on the last `batch mod 8` pairs) through the sliced call
    monkeypatch.delenv("GKL_HIP_PDHMM_PIPELINE")
    sub = big.subset(np.arange(big.batch - 70003, big.batch))   # (70 003 = 3 mod 8: a scalar tail)
    with native.PdhmmContext(fma_mode=1) as c:
        a1 = c.compute(sub)
    monkeypatch.setenv("GKL_HIP_PDHMM_PIPELINE", "0")
    with native.PdhmmContext(fma_mode=1) as c:
        assert c.compute(sub).tobytes() == a1.tobytes()


@pytest.mark.gpu
def test_pdhmm_gpu_paired_sliced_call_of_striped_reads_only(pd_oracle, monkeypatch):
    # A sliced paired call (>= 65 536 pairs, >= 64 MB) in which EVERY read needs more than one wavefront's lanes (384
    # bases or more: 64 lanes x 6 rows): no packed chunk exists, the call falls to the branch the cross layout uses, and that branch must
    # meet all seven slice uploads before its first kernel (r05 advisor finding: it launched while the helper thread was
    # still sending).  64 random pairs x 1025 copies, three times over: every copy carries the oracle's bits.
    from gkl_amd import native
    rng = np.random.RandomState(7171)
    uniq = random_pd_batch(rng, 64, read_len=(384, 460), hap_len=(40, 200))
    st, vec = pd_oracle.compute(uniq, semantics=2)
    assert st == 0
    reps = 1025
    big = uniq.subset(np.tile(np.arange(uniq.batch), reps))
    assert big.batch >= 65536 and big.batch * (2 * big.max_hap_len + 5 * big.max_read_len) >= 64 << 20
    with native.PdhmmContext(fma_mode=1, reference_tail=False) as c:
        for _ in range(3):
            got = c.compute(big).reshape(reps, -1)
            assert c.last_routing()[0] == 0
            bad = [k for k in range(reps) if got[k].tobytes() != vec.tobytes()]
            assert not bad, bad[:5]
    monkeypatch.setenv("GKL_HIP_PDHMM_PIPELINE", "0")
    with native.PdhmmContext(fma_mode=1, reference_tail=False) as c:
        assert c.compute(big).reshape(reps, -1)[reps - 1].tobytes() == vec.tobytes()


@pytest.mark.gpu
def test_pdhmm_gpu_buffers_shrink_again_after_a_big_call(pd_oracle):
    # a context's buffers grow with its biggest call; when the last 16 calls each needed less than a quarter of a buffer
    # above 32 MB AND the buffer has not grown for 64 calls (hipFree synchronises the whole device: a workload that
    # alternates one big call with a few small ones keeps its buffers) it is given back (one 120k-pair call holds ~0.7 GB
    # of streams and tables) -- and the calls after that still return the oracle's bits
    from gkl_amd import native
    _, _, b1, _ = holders_fixture_batch()
    small = b1.subset(np.arange(300))
    exp = pd_oracle.compute(small, semantics=2)[1]
    with native.PdhmmContext(fma_mode=1, reference_tail=False) as c:
        c.compute(b1.subset(np.tile(np.arange(b1.batch), 9)))
        big = c.buffer_bytes()
        assert big > 300 << 20
        for _ in range(62):
            assert c.compute(small).tobytes() == exp.tobytes()
        assert c.buffer_bytes() >= big                       # sixty-two small calls: nothing is given back yet (the small calls add their staging blocks)
        for _ in range(4):
            assert c.compute(small).tobytes() == exp.tobytes()
        assert c.buffer_bytes() < big // 8, (big, c.buffer_bytes())


@pytest.mark.gpu
def test_pdhmm_gpu_argument_errors(pd_ctx):
    from gkl_amd import native
    b = random_pd_batch(np.random.RandomState(5), 8)
    bad = random_pd_batch(np.random.RandomState(5), 8)
    bad.gcp[3] = -1
    with pytest.raises(native.IllegalArgumentException):
        pd_ctx.compute(bad)
    bad2 = random_pd_batch(np.random.RandomState(5), 8)
    bad2.hap_lengths[2] = 0
    with pytest.raises(native.IllegalArgumentException):
        pd_ctx.compute(bad2)
    assert np.isfinite(pd_ctx.compute(b)).all()  # the context survives errors


# ------------------------------------------------------------------ JNI shim (mock JNIEnv) + plugin mirror
def test_pdhmm_jni_exports_and_field_errors():
    import ctypes as C
    from tests import mockjni
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(mockjni.PD_JNI_LIB)
    for s in ("initNative", "computeLikelihoodsNative", "computePDHMMNative", "doneNative"):
        assert hasattr(lib, "Java_com_intel_gkl_pdhmm_IntelPDHMM_" + s)
    b = random_pd_batch(np.random.RandomState(2), 4)
    rc, _, cls, msg = mockjni.run_pdhmm(b, flags=mockjni.PD_DROP_PDBASES_FIELD)
    assert rc == 1 and cls == "java/lang/IllegalArgumentException" and msg == "Unable to get field ID"
    rc, _, cls, msg = mockjni.run_pdhmm(b, flags=mockjni.PD_SKIP_INIT)
    assert rc == 2 and cls == "java/lang/RuntimeException"


def test_pdhmm_mirror_argument_validation():
    # IntelPDHMMUnitTest.java:109-159: null -> NPE, wrong sizes -> IAE, before any native call
    from gkl_amd.errors import IllegalArgumentException, NullPointerException
    from gkl_amd.pdhmm import IntelPDHMM
    hmm = IntelPDHMM()
    with pytest.raises(NullPointerException):
        hmm.computeLikelihoods(None, [], np.zeros(0))
    with pytest.raises(IllegalArgumentException):
        hmm.computeLikelihoods([object()], [object()], np.zeros(3))
    z = np.zeros(4, np.int8)
    with pytest.raises(NullPointerException):
        hmm.computePDHMM(None, z, z, z, z, z, z, np.ones(2), np.ones(2), 2, 2, 2)
    with pytest.raises(IllegalArgumentException):
        hmm.computePDHMM(z, z, z, z, z, z, np.zeros(3, np.int8), np.ones(2), np.ones(2), 2, 2, 2)


@pytest.mark.gpu
def test_pdhmm_jni_flat_and_holder_paths(pd_oracle):
    from tests import mockjni
    b, exp = load_pdhmm_file(FILES[1])
    rc, out, cls, msg = mockjni.run_pdhmm(b)
    assert rc == 0 and msg == "", (cls, msg)   # (no exception, and nothing -Xcheck:jni would flag: the mock reports the first violation here)
    _, ref = pd_oracle.compute_reference(b, fma_mode=1)   # GKL's computePDHMMNative, position by position
    assert out.tobytes() == ref.tobytes() and np.max(np.abs(out - exp)) <= TOL
    # holders: 61 reads x 41 haplotypes, read-major cross product (staged once each, crossed on the device); with
    # maxMemoryInMB = 1 the reference cuts the 2501 pairs into two batches, each with its own scalar tail
    from gkl_amd import native
    rng = np.random.RandomState(9)
    nr, nh = 61, 41
    src = random_pd_batch(rng, nr, with_n=False, read_len=(20, 60), hap_len=(30, 80))
    haps = random_pd_batch(rng, nh, with_n=False, read_len=(1, 2), hap_len=(30, 80))
    rc, out, cls, msg = mockjni.run_pdhmm(None, holders=(src, haps), max_memory_mb=1)
    assert rc == 0 and msg == "", (cls, msg)   # (no exception, and nothing -Xcheck:jni would flag: the mock reports the first violation here)
    pairs = []
    for r in range(nr):
        for h in range(nh):
            R, H = int(src.read_lengths[r]), int(haps.hap_lengths[h])
            rr = lambda a: a.reshape(nr, src.max_read_len)[r, :R]  # noqa: E731
            hh = lambda a: a.reshape(nh, haps.max_hap_len)[h, :H]  # noqa: E731
            pairs.append((hh(haps.hap_bases), hh(haps.hap_pdbases), rr(src.read_bases), rr(src.read_qual),
                          rr(src.read_ins_qual), rr(src.read_del_qual), rr(src.gcp)))
    max_r, max_h = int(src.read_lengths.max()), int(haps.hap_lengths.max())
    ref_batch = native.pdhmm_reference_batch_pairs(1, max_r, max_h, nr * nh)
    assert 0 < ref_batch < nr * nh, "two reference batches"
    _, ref = pd_oracle.compute_reference(PdhmmBatch.from_pairs(pairs), fma_mode=1, ref_batch=ref_batch)
    assert out.tobytes() == ref.tobytes()
    bad = random_pd_batch(np.random.RandomState(2), 4)
    bad.read_del_qual[1] = -7
    rc, _, cls, msg = mockjni.run_pdhmm(bad)
    assert rc == 2 and cls == "java/lang/IllegalArgumentException" and "aren't valid" in msg


@pytest.mark.gpu
def test_pdhmm_mirror_compute_likelihoods_on_holders_fixture(pd_oracle):
    # newPDHMMTest (IntelPDHMMUnitTest.java:446-556) with the asserts switched on
    from gkl_amd.pdhmm import IntelPDHMM, PDHaplotypeDataHolder, ReadDataHolder
    reads, haps, b, exp = holders_fixture_batch()
    hmm = IntelPDHMM()
    assert hmm.load(None)
    hmm.initialize(None)
    rd = []
    for r in reads:
        h = ReadDataHolder()
        h.readBases, h.readQuals, h.insertionGOP, h.deletionGOP, h.overallGCP = r
        rd.append(h)
    hd = []
    for x in haps:
        h = PDHaplotypeDataHolder()
        h.haplotypeBases, h.haplotypePDBases = x
        hd.append(h)
    out = np.zeros(len(rd) * len(hd))
    hmm.computeLikelihoods(rd, hd, out)
    hmm.done()
    # 13 248 pairs = 1656 groups of 8: one reference batch at the default 512 MB and no scalar tail at all
    _, vec = pd_oracle.compute_reference(b, fma_mode=1)
    assert vec.tobytes() == pd_oracle.compute(b, semantics=2)[1].tobytes()
    assert np.max(np.abs(out - exp)) <= TOL
    assert out.tobytes() == vec.tobytes()
    # the same holders through the JNI symbol computeLikelihoodsNative (mock JNIEnv), default memory budget
    from tests import mockjni
    one = np.zeros(1, np.int8)
    src = PdhmmBatch.from_pairs([(one, one, *r) for r in reads])
    hp = PdhmmBatch.from_pairs([(x[0], x[1], one, one, one, one, one) for x in haps])
    rc, jout, cls, msg = mockjni.run_pdhmm(None, holders=(src, hp))
    assert rc == 0 and msg == "", (cls, msg)   # (no exception, and nothing -Xcheck:jni would flag: the mock reports the first violation here)
    assert jout.tobytes() == vec.tobytes()


@pytest.mark.gpu
def test_pdhmm_mirror_like_reference_unit_test():
    # pdhmmPerformanceTest (IntelPDHMMUnitTest.java:161-257): computePDHMM on each data file, abs tol 1e-4
    from gkl_amd.pdhmm import IntelPDHMM
    hmm = IntelPDHMM()
    assert hmm.load(None)
    hmm.initialize(None)
    for f in FILES:
        b, exp = load_pdhmm_file(f)
        out = hmm.computePDHMM(b.hap_bases, b.hap_pdbases, b.read_bases, b.read_qual, b.read_ins_qual,
                               b.read_del_qual, b.gcp, b.hap_lengths, b.read_lengths, b.batch, b.max_hap_len,
                               b.max_read_len)
        assert np.max(np.abs(out - exp)) <= TOL
    hmm.done()
