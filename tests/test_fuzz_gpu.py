"""Shape fuzzing of the three HIP paths against their oracles (hypothesis picks the shapes, numpy the contents):
tiny and ragged batches, single-base sequences, lengths around the lane / stripe boundaries.  Everything is compared
bit for bit (PairHMM host-finalised doubles, PDHMM doubles, Smith-Waterman CIGAR + offset)."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

pytestmark = pytest.mark.gpu

EDGE = [1, 2, 3, 4, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 383, 384, 385, 511, 512, 513]
length = st.one_of(st.sampled_from(EDGE), st.integers(1, 600))
COMMON = dict(deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)


@pytest.fixture(scope="module")
def ctxs():
    from gkl_amd import native
    from oracle.oracle import Oracle
    from oracle.pdhmm import PdhmmOracle
    from oracle.sw import SwOracle
    with native.PairHmmContext() as ph, native.PdhmmContext(reference_tail=False) as pd, native.SwContext() as sw:
        yield ph, pd, sw, Oracle(), PdhmmOracle(), SwOracle()


@settings(max_examples=150, **COMMON)
@given(n_reads=st.integers(1, 9), n_haps=st.integers(1, 7), rl=st.lists(length, min_size=9, max_size=9),
       hl=st.lists(length, min_size=7, max_size=7), seed=st.integers(0, 2**31 - 1), use_double=st.booleans())
def test_pairhmm_shapes(ctxs, n_reads, n_haps, rl, hl, seed, use_double):
    from gkl_amd import native
    from gkl_amd.batch import FlatBatch, HaplotypeDataHolder, ReadDataHolder
    ph, _, _, oracle, _, _ = ctxs
    rng = np.random.RandomState(seed)
    bases = np.frombuffer(b"ACGTN", dtype=np.uint8)
    reads = []
    for r in range(n_reads):
        n = rl[r]
        reads.append(ReadDataHolder(rng.choice(bases, n, p=[.24, .24, .24, .24, .04]).tobytes(),
                                    rng.randint(0, 64, n).astype(np.uint8).tobytes(),
                                    rng.randint(0, 64, n).astype(np.uint8).tobytes(),
                                    rng.randint(0, 64, n).astype(np.uint8).tobytes(),
                                    rng.randint(0, 64, n).astype(np.uint8).tobytes()))
    haps = [HaplotypeDataHolder(rng.choice(bases, hl[h], p=[.24, .24, .24, .24, .04]).tobytes()) for h in range(n_haps)]
    b = FlatBatch.from_holders(reads, haps)
    exp = oracle.batch(b, use_double=use_double, n_threads=2)
    if use_double:
        with native.PairHmmContext(use_double=True) as c:
            got = c.compute(b)
    else:
        got = ph.compute(b)
    assert got.tobytes() == exp.tobytes()


@settings(max_examples=100, **COMMON)
@given(n=st.integers(1, 12), rl=st.lists(length, min_size=12, max_size=12), hl=st.lists(length, min_size=12, max_size=12),
       seed=st.integers(0, 2**31 - 1), flag_rate=st.sampled_from([0.0, 0.02, 0.3]))
def test_pdhmm_shapes(ctxs, n, rl, hl, seed, flag_rate):
    from tests.test_pdhmm import random_pd_batch
    _, pd, _, _, pd_oracle, _ = ctxs
    rng = np.random.RandomState(seed)
    from gkl_amd.pdhmm_batch import PdhmmBatch
    pairs = []
    for k in range(n):
        one = random_pd_batch(rng, 1, read_len=(rl[k], rl[k]), hap_len=(hl[k], hl[k]), flag_rate=flag_rate)
        R, H = int(one.read_lengths[0]), int(one.hap_lengths[0])
        pairs.append((one.hap_bases[:H], one.hap_pdbases[:H], one.read_bases[:R], one.read_qual[:R],
                      one.read_ins_qual[:R], one.read_del_qual[:R], one.gcp[:R]))
    b = PdhmmBatch.from_pairs(pairs)
    st_, exp = pd_oracle.compute(b, semantics=2)
    assert st_ == 0 and pd.compute(b).tobytes() == exp.tobytes()


@settings(max_examples=300, **COMMON)
@given(rl=length, al=length, seed=st.integers(0, 2**31 - 1), strategy=st.sampled_from([9, 10, 11, 12]),
       params=st.sampled_from([(200, -150, -260, -11), (3, -1, -4, -3), (1, -1, -1, -1), (5, -4, 0, 0), (10, -5, -10, -10)]),
       related=st.booleans())
def test_smith_waterman_shapes(ctxs, rl, al, seed, strategy, params, related):
    from tests.test_sw import mutate
    _, _, sw, _, _, sw_oracle = ctxs
    rng = np.random.RandomState(seed)
    ref = bytes(rng.choice(list(b"ACGT"), size=rl).tolist())
    if related:
        alt = mutate(rng, (ref * (al // rl + 2))[:al], 0.08)
    else:
        alt = bytes(rng.choice(list(b"ACGT"), size=al).tolist())
    exp = sw_oracle.align(ref, alt, params, strategy)[1:]
    assert sw.align(ref, alt, params, strategy) == exp


def test_sixty_second_slice_of_the_randomised_fuzz():
    """tests/fuzz.py (every kernel variant, both arithmetics, one- and multi-shard contexts, PDHMM paired and cross,
    Smith-Waterman) for 60 s of wall clock, seed fixed: any mismatch fails."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz.py"), "--seconds", "60", "--seed", "20250418"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
