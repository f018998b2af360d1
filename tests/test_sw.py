"""Smith-Waterman: oracle pins on CPU, HIP parity on the GPU.  Integer work: every
comparison is exact (CIGAR bytes, their count, the alignment offset)."""
import numpy as np
import pytest

from oracle.sw import IGNORE, INDEL, LEADING_INDEL, SOFTCLIP, STRATEGIES

PARAM_SETS = [(200, -150, -260, -11),   # SmithWatermanUnitTest.java:44 (GATK's haplotype-to-reference set)
              (3, -1, -4, -3), (25, -50, -110, -6), (10, -5, -10, -10), (1, -1, -1, -1), (5, -4, 0, 0),
              (64 * 1024, -5, -10, -10)]  # MAXIMUM_SW_MATCH_VALUE


@pytest.fixture(scope="module")
def sw_oracle():
    from oracle.sw import SwOracle
    return SwOracle()


@pytest.fixture(scope="module")
def sw_reference():
    from oracle.sw import SwReference
    try:
        return SwReference()
    except RuntimeError as e:
        pytest.skip(str(e))


def mutate(rng, s, rate):
    out = bytearray()
    for c in s:
        u = rng.rand()
        if u < rate / 3:
            continue
        if u < 2 * rate / 3:
            out.append(int(rng.choice(list(b"ACGT"))))
            out.append(c)
            continue
        if u < rate:
            out.append(int(rng.choice(list(b"ACGT"))))
            continue
        out.append(c)
    return bytes(out) or b"A"


def random_pairs(rng, n, lengths=(1, 2, 3, 5, 8, 17, 33, 64, 100, 150, 300)):
    """Related and unrelated sequence pairs: point mutations, a window of the reference, random, with a tail."""
    pairs = []
    for _ in range(n):
        L = int(rng.choice(lengths))
        ref = bytes(rng.choice(list(b"ACGT"), size=L).tolist())
        kind = rng.randint(4)
        if kind == 0:
            alt = mutate(rng, ref, 0.1)
        elif kind == 1:
            alt = mutate(rng, ref[L // 4:L // 4 + max(1, L // 2)], 0.2)
        elif kind == 2:
            alt = bytes(rng.choice(list(b"ACGT"), size=max(1, int(L * rng.uniform(0.3, 1.7)))).tolist())
        else:
            alt = mutate(rng, ref, 0.02) + bytes(rng.choice(list(b"ACGT"), size=rng.randint(0, 20)).tolist())
        pairs.append((ref, alt))
    return pairs


def golden_vectors():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sw_vectors.json")) as f:
        return json.load(f)["vectors"]


# ------------------------------------------------------------------ oracle pins (CPU)
def test_sw_oracle_matches_reference_generated_vectors(sw_oracle):
    # tests/golden/sw_vectors.json: answers of GKL's AVX2 and AVX-512 objects (generator beside it)
    vs = golden_vectors()
    assert len(vs) >= 250
    for v in vs:
        st, cig, cnt, off = sw_oracle.align(v["ref"].encode(), v["alt"].encode(), v["params"], v["strategy"],
                                            cigar_len=v["cigar_len"])
        assert st == 0 and (cig.decode(), cnt, off) == (v["cigar"], v["count"], v["offset"]), v


def test_sw_oracle_reference_unit_test_cases(sw_oracle):
    # singleElementSequencesAlignmentTest / twoElementSequencesAlignmentTest (SmithWatermanUnitTest.java:171-205)
    assert sw_oracle.align(b"C", b"C", (3, -2, -2, -1), IGNORE)[1] == b"1M"
    assert sw_oracle.align(b"AD", b"AT", (3, -5, -2, -1), IGNORE)[1] == b"1M1I"


def test_sw_oracle_bit_identical_to_reference_objects(sw_oracle, sw_reference):
    rng = np.random.RandomState(1)
    engines = (1, 2) if sw_reference.has_avx512() else (1,)
    n = 0
    for ref, alt in random_pairs(rng, 400):
        p = PARAM_SETS[rng.randint(len(PARAM_SETS))]
        for s in STRATEGIES:
            mine = sw_oracle.align(ref, alt, p, s)
            for eng in engines:
                assert mine == sw_reference.align(ref, alt, p, s, engine=eng), (eng, s, p, ref, alt)
                n += 1
    assert n >= 3200
    # longer than one 1024-wide matrix of the reference (D_MAX_SEQ_LEN grows, PairWiseSW.h:463-469), and a
    # CIGAR buffer too small for the text (elements that do not fit are skipped, :442-447)
    ref = bytes(rng.choice(list(b"ACGT"), size=1500).tolist())
    alt = mutate(rng, ref[100:1400], 0.05)
    for s in STRATEGIES:
        assert sw_oracle.align(ref, alt, PARAM_SETS[0], s) == sw_reference.align(ref, alt, PARAM_SETS[0], s)
        for cl in (1, 2, 3, 5, 9):
            assert sw_oracle.align(ref, alt, PARAM_SETS[0], s, cigar_len=cl) == \
                sw_reference.align(ref, alt, PARAM_SETS[0], s, cigar_len=cl)


def test_sw_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gkl_amd import native
    from gkl_amd.errors import RuntimeException
    with pytest.raises(RuntimeException):
        native.SwContext()


# ------------------------------------------------------------------ HIP parity (GPU)
@pytest.fixture(scope="module")
def sw_ctx():
    from gkl_amd import native
    with native.SwContext() as c:
        yield c


@pytest.mark.gpu
def test_sw_gpu_reference_unit_test_cases(sw_ctx):
    assert sw_ctx.align(b"C", b"C", (3, -2, -2, -1), IGNORE)[0] == b"1M"
    assert sw_ctx.align(b"AD", b"AT", (3, -5, -2, -1), IGNORE)[0] == b"1M1I"


@pytest.mark.gpu
def test_sw_gpu_matches_reference_generated_vectors(sw_ctx):
    for v in golden_vectors():
        cig, cnt, off = sw_ctx.align(v["ref"].encode(), v["alt"].encode(), v["params"], v["strategy"],
                                     cigar_len=v["cigar_len"])
        assert (cig.decode(), cnt, off) == (v["cigar"], v["count"], v["offset"]), v


@pytest.mark.gpu
@pytest.mark.parametrize("strategy", STRATEGIES)
def test_sw_gpu_batch_bit_exact(sw_ctx, sw_oracle, strategy):
    rng = np.random.RandomState(100 + strategy)
    pairs = random_pairs(rng, 300)
    for p in PARAM_SETS[:4]:
        cig, cnt, off = sw_ctx.align_batch([r for r, _ in pairs], [a for _, a in pairs], p, strategy)
        stride = 2 * max(max(len(r), len(a)) for r, a in pairs)
        for k, (r, a) in enumerate(pairs):
            st, ecig, ecnt, eoff = sw_oracle.align(r, a, p, strategy, cigar_len=stride)
            assert st == 0 and (cig[k], int(cnt[k]), int(off[k])) == (ecig, ecnt, eoff), (k, p, r, a)


@pytest.mark.gpu
def test_sw_gpu_batch_larger_than_the_grid(sw_ctx, sw_oracle):
    # 4096 persistent wavefronts pull pairs longest-first and reuse one scratch slab each: only a batch of more pairs
    # than that sends a wavefront round its loop a second time (back-track words, maxima and operations of the
    # previous pair still in its slab).  300 distinct pairs, 24 shuffled copies of each = 7200 pairs in one call;
    # every copy must be the oracle's answer for its original.
    rng = np.random.RandomState(4242)
    pairs = random_pairs(rng, 300, lengths=(1, 3, 17, 64, 100, 150, 300, 520))
    p, strategy = PARAM_SETS[0], STRATEGIES[0]
    stride = 2 * max(max(len(r), len(a)) for r, a in pairs)
    exp = [sw_oracle.align(r, a, p, strategy, cigar_len=stride) for r, a in pairs]
    order = rng.permutation(np.repeat(np.arange(len(pairs)), 24))
    cig, cnt, off = sw_ctx.align_batch([pairs[k][0] for k in order], [pairs[k][1] for k in order], p, strategy)
    for i, k in enumerate(order):
        st, ecig, ecnt, eoff = exp[k]
        assert st == 0 and (cig[i], int(cnt[i]), int(off[i])) == (ecig, ecnt, eoff), (i, k)
    for strategy in STRATEGIES[1:]:
        exp = [sw_oracle.align(r, a, p, strategy, cigar_len=stride) for r, a in pairs]
        cig, cnt, off = sw_ctx.align_batch([pairs[k][0] for k in order], [pairs[k][1] for k in order], p, strategy)
        for i, k in enumerate(order):
            st, ecig, ecnt, eoff = exp[k]
            assert st == 0 and (cig[i], int(cnt[i]), int(off[i])) == (ecig, ecnt, eoff), (strategy, i, k)


@pytest.mark.gpu
def test_sw_gpu_single_pair_long_and_striped(sw_ctx, sw_oracle):
    # more than 256 rows -> several stripes with the boundary row carried through HBM; odd / even widths
    rng = np.random.RandomState(7)
    for L, M in ((257, 300), (300, 257), (513, 64), (1000, 999), (1500, 1300), (64, 1500), (255, 256), (256, 255)):
        ref = bytes(rng.choice(list(b"ACGT"), size=L).tolist())
        alt = mutate(rng, (ref * 2)[L // 7:L // 7 + M], 0.06)
        for s in STRATEGIES:
            st, ecig, ecnt, eoff = sw_oracle.align(ref, alt, PARAM_SETS[0], s)
            assert sw_ctx.align(ref, alt, PARAM_SETS[0], s) == (ecig, ecnt, eoff), (L, M, s)


@pytest.mark.gpu
def test_sw_gpu_small_cigar_buffer_and_argument_errors(sw_ctx, sw_oracle):
    from gkl_amd.errors import IllegalArgumentException
    ref, alt = b"ACGTACGTTTGACCA" * 5, b"ACGTACGTGACCAACG" * 4
    for cl in (1, 2, 3, 4, 6, 11):
        st, ecig, ecnt, eoff = sw_oracle.align(ref, alt, PARAM_SETS[1], SOFTCLIP, cigar_len=cl)
        assert sw_ctx.align(ref, alt, PARAM_SETS[1], SOFTCLIP, cigar_len=cl) == (ecig, ecnt, eoff)
    with pytest.raises(IllegalArgumentException):
        sw_ctx.align(b"", b"AC", PARAM_SETS[1], IGNORE, cigar_len=4)          # emptyReferenceSequence...Test
    with pytest.raises(IllegalArgumentException):
        sw_ctx.align(b"AC", b"", PARAM_SETS[1], IGNORE, cigar_len=4)
    with pytest.raises(IllegalArgumentException):
        sw_ctx.align(b"A" * 32768, b"TCCG", (10, -5, -10, -10), IGNORE)       # ...SequenceLengthTooLong
    with pytest.raises(IllegalArgumentException):
        sw_ctx.align(b"ACCG", b"TCCG", (64 * 1024 + 1, -5, -10, -10), IGNORE)  # ...MatchValueGreaterThanMaxAllowed
    with pytest.raises(IllegalArgumentException):
        sw_ctx.align(b"ACCG", b"TCCG", (3, -1, -4, -3), 13)
    assert sw_ctx.align(b"ACCG", b"TCCG", (3, -1, -4, -3), IGNORE)[0]         # the context survives errors


# ------------------------------------------------------------------ JNI shim (mock JNIEnv) + plugin mirror
def test_sw_jni_exports_and_errors_without_a_call():
    import ctypes as C
    from tests import mockjni
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(mockjni.SW_JNI_LIB)
    for s in ("initNative", "alignNative", "doneNative"):
        assert hasattr(lib, "Java_com_intel_gkl_smithwaterman_IntelSmithWaterman_" + s)
    rc, _, _, cls, msg, _ = mockjni.run_sw(b"ACGT", b"ACGT", (3, -1, -4, -3), IGNORE, flags=mockjni.SW_SKIP_INIT)
    assert rc == 2 and cls == "java/lang/RuntimeException"
    rc, _, _, cls, msg, _ = mockjni.run_sw(b"ACGT", b"ACGT", (3, -1, -4, -3), IGNORE,
                                           flags=mockjni.SW_SKIP_INIT | mockjni.SW_NULL_REF)
    assert rc == 2 and cls == "java/lang/IllegalArgumentException" and msg == "Arrays aren't valid."


@pytest.mark.gpu
def test_sw_jni_batch_entry_point(sw_oracle):
    # alignBatchNative (not in the reference: the entry point a batching caller binds, include/gkl_sw_jni.h)
    from tests import mockjni
    rng = np.random.RandomState(44)
    pairs = random_pairs(rng, 150, lengths=(1, 9, 40, 130, 257, 300, 520))
    refs, alts = [r for r, _ in pairs], [a for _, a in pairs]
    for strategy in STRATEGIES:
        rc, ret, cigs, offs, cls, msg = mockjni.run_sw_batch(refs, alts, PARAM_SETS[0], strategy)
        assert rc == 0 and ret == len(pairs) and msg == "", (cls, msg)
        stride = 2 * max(max(len(r), len(a)) for r, a in pairs)
        for k, (r, a) in enumerate(pairs):
            _, ecig, _, eoff = sw_oracle.align(r, a, PARAM_SETS[0], strategy, cigar_len=stride)
            assert (cigs[k], int(offs[k])) == (ecig, eoff), (k, strategy)


def test_sw_mirror_argument_validation():
    # SmithWatermanUnitTest.java:33-166: null -> NPE; too long / match too large / empty -> IAE, before any native call
    from gkl_amd.errors import IllegalArgumentException, NullPointerException
    from gkl_amd.smithwaterman import IntelSmithWaterman, SWOverhangStrategy, SWParameters
    sw = IntelSmithWaterman()
    prm = SWParameters(200, -150, -260, -11)
    for args in ((None, b"AC", prm, SWOverhangStrategy.SOFTCLIP), (b"AC", None, prm, SWOverhangStrategy.SOFTCLIP),
                 (b"AC", b"AC", None, SWOverhangStrategy.SOFTCLIP), (b"AC", b"AC", prm, None)):
        with pytest.raises(NullPointerException):
            sw.align(*args)
    too_long = b"A" * (IntelSmithWaterman.MAX_SW_SEQUENCE_LENGTH + 1)
    small = SWParameters(10, -5, -10, -10)
    with pytest.raises(IllegalArgumentException):
        sw.align(too_long, b"TCCG", small, SWOverhangStrategy.IGNORE)
    with pytest.raises(IllegalArgumentException):
        sw.align(b"TCCG", too_long, small, SWOverhangStrategy.IGNORE)
    with pytest.raises(IllegalArgumentException):
        sw.align(b"ACCG", b"TCCG", SWParameters(IntelSmithWaterman.MAXIMUM_SW_MATCH_VALUE + 1, -5, -10, -10),
                 SWOverhangStrategy.IGNORE)
    with pytest.raises(IllegalArgumentException):
        sw.align(b"", b"AC", SWParameters(3, -2, -2, -1), SWOverhangStrategy.IGNORE)
    with pytest.raises(IllegalArgumentException):
        sw.align(b"AC", b"", SWParameters(3, -2, -2, -1), SWOverhangStrategy.IGNORE)


@pytest.mark.gpu
def test_sw_jni_and_mirror_paths(sw_oracle):
    from gkl_amd.smithwaterman import IntelSmithWaterman, SWOverhangStrategy, SWParameters
    from tests import mockjni
    rng = np.random.RandomState(12)
    sw = IntelSmithWaterman()
    assert sw.load(None)
    # singleElementSequencesAlignmentTest / twoElementSequencesAlignmentTest
    assert sw.align(b"C", b"C", SWParameters(3, -2, -2, -1), SWOverhangStrategy.IGNORE).cigar == "1M"
    assert sw.align(b"AD", b"AT", SWParameters(3, -5, -2, -1), SWOverhangStrategy.IGNORE).cigar == "1M1I"
    for ref, alt in random_pairs(rng, 12, lengths=(20, 90, 260, 400)):
        for strat in SWOverhangStrategy:
            st, ecig, _, eoff = sw_oracle.align(ref, alt, PARAM_SETS[0], strat.value)
            res = sw.align(ref, alt, SWParameters(*PARAM_SETS[0]), strat)
            assert (res.cigar, res.alignment_offset) == (ecig.decode(), eoff)
            rc, jcig, joff, cls, msg, _ = mockjni.run_sw(ref, alt, PARAM_SETS[0], strat.value)
            assert rc == 0 and msg == "", (cls, msg)   # (no exception, and nothing -Xcheck:jni would flag: the mock reports the first violation here)
            assert (jcig, joff) == (ecig, eoff)
    sw.close()
    # maxSequenceFullAlignmentTest (disabled in the reference: 32767 x 32767 with match 65536) at a tenth of the size
    n = 3277
    ref = bytes(rng.choice(list(b"ACGT"), size=n).tolist())
    sw = IntelSmithWaterman()
    assert sw.load(None)
    assert sw.align(ref, ref, SWParameters(IntelSmithWaterman.MAXIMUM_SW_MATCH_VALUE, -5, -10, -10),
                    SWOverhangStrategy.IGNORE).cigar == f"{n}M"
    sw.close()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_sw_gpu_maximum_sequence_length(sw_ctx, sw_oracle):
    # maxSequenceFullAlignmentTest (SmithWatermanUnitTest.java:207-229, disabled in the reference): 32 767 x 32 767 with
    # the largest match value; 64 stripes of 512 rows, ~1.07e9 cells in one wavefront
    n = 32 * 1024 - 1
    rng = np.random.RandomState(99)
    ref = bytes(rng.choice(list(b"ACGT"), size=n).tolist())
    got = sw_ctx.align(ref, ref, (64 * 1024, -5, -10, -10), IGNORE)
    assert got == (f"{n}M".encode(), len(f"{n}M"), 0)
    alt = mutate(rng, ref[1000:31000], 0.01)
    exp = sw_oracle.align(ref, alt, PARAM_SETS[0], SOFTCLIP)[1:]
    assert sw_ctx.align(ref, alt, PARAM_SETS[0], SOFTCLIP) == exp
