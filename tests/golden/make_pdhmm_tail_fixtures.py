#!/usr/bin/env python3
"""Generates tests/golden/pdhmm_tail_vectors.json from the REFERENCE's own PDHMM objects (oracle/_ref, built from
/root/reference by oracle/Makefile): paired batches whose size is not a multiple of the SIMD width, so that the
reference finishes them with its scalar engine (pdhmm.h:1264-1270).  Stored: the inputs and the bit patterns of what
computePDHMM returns per position for the AVX-512 engine (needs an AVX-512 host) and the AVX2 engine.
Run here (the GPU box has no /root/reference):  python tests/golden/make_pdhmm_tail_fixtures.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pdhmm import PdhmmReference  # noqa: E402
from tests.test_pdhmm import random_pd_batch  # noqa: E402


def end_in_deletion(b, rng):
    """Make some haplotypes END inside / right after a deletion: the scalar engine then starts rows >= 2 in that state."""
    pd = b.hap_pdbases.reshape(b.batch, b.max_hap_len)      # a view: edits land in the batch
    for p in range(b.batch):
        H = int(b.hap_lengths[p])
        u = rng.random_sample()
        if u < 0.35:
            pd[p, H - 1] = 4                                 # DEL_END on the last column -> AFTER_DEL
        elif u < 0.7 and H > 2:
            k = int(rng.randint(max(0, H - 6), H))
            pd[p, k:H] = 0
            pd[p, k] = 2                                     # DEL_START with nothing behind it -> INSIDE_DEL
    return b


def main():
    ref = PdhmmReference()
    assert ref.has_avx512(), "needs an AVX-512 host for the engine-2 expectations"
    rng = np.random.RandomState(20250928)
    out = []
    short = dict(read_len=(5, 70), hap_len=(5, 90), with_n=False)
    longr = dict(read_len=(240, 300), hap_len=(30, 120), with_n=False, flag_rate=0.3)   # reads over 255 bases: striped jobs
    # n % 8 in 1..7 and n % 4 in 1..3 (and some with n % 4 == 0)
    for n, kw in [(n, short) for n in (1, 3, 7, 9, 12, 13, 18, 23, 30, 33)] + [(3, longr), (10, longr)]:
        if True:
            b = end_in_deletion(random_pd_batch(rng, n, **kw), rng)
            st2, r2 = ref.compute(b, engine=2)
            st1, r1 = ref.compute(b, engine=1)
            assert st2 == 0 and st1 == 0
            rec = {k: np.ascontiguousarray(getattr(b, k), np.int8).tobytes().hex()   # [batch][max_len] int8, as hex
                   for k in ("hap_bases", "hap_pdbases", "read_bases", "read_qual", "read_ins_qual", "read_del_qual", "gcp")}
            rec.update(hap_lengths=b.hap_lengths.tolist(), read_lengths=b.read_lengths.tolist())
            rec.update(batch=int(b.batch), max_hap_len=int(b.max_hap_len), max_read_len=int(b.max_read_len),
                       avx512_bits=[format(int(x), "016x") for x in r2.view(np.uint64)],
                       avx2_bits=[format(int(x), "016x") for x in r1.view(np.uint64)])
            out.append(rec)
    path = os.path.join(ROOT, "tests", "golden", "pdhmm_tail_vectors.json")
    json.dump({"generator": "tests/golden/make_pdhmm_tail_fixtures.py", "vectors": out}, open(path, "w"))
    print(path, len(out), "batches", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
