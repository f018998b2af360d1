#!/usr/bin/env python3
"""Generates tests/golden/sw_vectors.json from the REFERENCE's own Smith-Waterman objects
(oracle/_ref/libgkl_ref_sw.so = src/main/native/smithwaterman/{avx2,avx512}_impl.cc + smithwaterman_common.cc,
built by `make -C oracle ref`).  Run in the build container, where /root/reference exists:

    python tests/golden/make_sw_fixtures.py

Each vector: ref, alt (ASCII), params [match, mismatch, open, extend], strategy (9..12), cigar_len, and the
reference's answer: cigar text, cigar count, alignment offset (identical for both engines, asserted here)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.sw import STRATEGIES, SwReference  # noqa: E402
from tests.test_sw import PARAM_SETS, mutate, random_pairs  # noqa: E402


def main():
    ref = SwReference()
    engines = (1, 2) if ref.has_avx512() else (1,)
    rng = np.random.RandomState(20250418)
    cases = [(b"C", b"C", (3, -2, -2, -1), None), (b"AD", b"AT", (3, -5, -2, -1), None),
             (b"ACGT", b"TTTT", (3, -1, -4, -3), None), (b"A", b"ACGTACGT", (3, -1, -4, -3), None),
             (b"ACGTACGTACGT", b"G", (3, -1, -4, -3), None), (b"acgtACGTNNNN", b"ACGTacgtNNNN", (3, -1, -4, -3), None)]
    for r, a in random_pairs(rng, 60, lengths=(1, 2, 7, 16, 33, 65, 130, 257, 300)):
        cases.append((r, a, PARAM_SETS[rng.randint(len(PARAM_SETS))], None))
    big = bytes(rng.choice(list(b"ACGT"), size=700).tolist())
    cases.append((big, mutate(rng, big[50:650], 0.05), PARAM_SETS[0], None))
    cases.append((big[:300], mutate(rng, big[:300], 0.1), PARAM_SETS[1], 7))   # CIGAR buffer too small
    cases.append((big[:120], big[20:100], PARAM_SETS[0], 3))
    out = []
    for r, a, p, cl in cases:
        for s in STRATEGIES:
            res = [ref.align(r, a, p, s, cigar_len=cl, engine=e) for e in engines]
            assert all(x == res[0] for x in res), (r, a, p, s)
            st, cig, cnt, off = res[0]
            assert st == 0
            out.append({"ref": r.decode("ascii"), "alt": a.decode("ascii"), "params": list(p), "strategy": s,
                        "cigar_len": cl if cl is not None else 2 * max(len(r), len(a)),
                        "cigar": cig.decode("ascii"), "count": cnt, "offset": off})
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sw_vectors.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_sw_fixtures.py", "engines": list(engines), "vectors": out}, f, indent=0)
    print(len(out), "vectors")


if __name__ == "__main__":
    main()
