"""Deterministic synthetic HaplotypeCaller-shaped PairHMM batches.

There is no dataset in the reference for batch-sized runs (its golden file holds
104 single pairs), so the bench / parity-at-scale workloads are generated:

* ``hc``     -- primary: one 500-base window; haplotypes are prefixes W[0:L],
                L~U[100,500], with 0..5 edits; reads are drawn from a random
                haplotype (R~U[50,250], clipped), with Phred-consistent
                substitution errors.  A sizeable fraction of pairs underflows
                fp32 (reads from the tail of a long haplotype scored against a
                short one) and exercises the fp64 fallback.
* ``region`` -- all haplotypes span the same 400-base window: fallback ~0 %.
* ``mixed``  -- haplotypes start at random offsets of a 520-base window (stress).

All randomness comes from numpy's MT19937 ``RandomState`` (default seed 20250418).
"""
from __future__ import annotations

import numpy as np

from .batch import FlatBatch

DEFAULT_SEED = 20250418
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _edit(rng, seq: np.ndarray, k: int) -> np.ndarray:
    seq = seq.copy()
    for _ in range(k):
        if seq.size < 3:
            break
        pos = int(rng.randint(0, seq.size))
        kind = int(rng.randint(0, 3))
        if kind == 0:  # SNP
            seq[pos] = _ACGT[(int(np.searchsorted(_ACGT, seq[pos])) + 1 + int(rng.randint(0, 3))) % 4]
        elif kind == 1:  # 1-bp deletion
            seq = np.delete(seq, pos)
        else:  # 1-bp insertion
            seq = np.insert(seq, pos, _ACGT[int(rng.randint(0, 4))])
    return seq


def make_batch(kind: str = "hc", n_reads: int = 10000, n_haps: int = 128, seed: int = DEFAULT_SEED,
               read_len=(50, 250), hap_len=(100, 500), max_edits: int = 5,
               read_seed: int | None = None) -> FlatBatch:
    """``seed`` fixes the window and the haplotypes; ``read_seed`` (default: continue the same
    stream) draws the reads, so several shards can share one haplotype set."""
    if kind not in ("hc", "region", "mixed"):
        raise ValueError(f"unknown synthetic workload {kind!r}")
    rng = np.random.RandomState(seed)
    hmin, hmax = hap_len
    rmin, rmax = read_len
    wlen = hmax if kind != "mixed" else hmax + 20
    window = _ACGT[rng.randint(0, 4, size=wlen)]

    haps = []
    for _ in range(n_haps):
        k = int(rng.randint(0, max_edits + 1))
        if kind == "hc":
            L = int(rng.randint(hmin, hmax + 1))
            base = window[:L]
        elif kind == "region":
            L = min(400, hmax)
            base = window[:L]
        else:
            L = int(rng.randint(hmin, hmax + 1))
            off = int(rng.randint(0, wlen - L + 1))
            base = window[off:off + L]
        h = _edit(rng, base, k)
        if h.size > hmax:
            h = h[:hmax]
        haps.append(h)

    if read_seed is not None:
        rng = np.random.RandomState(read_seed)
    hap_pick = rng.randint(0, n_haps, size=n_reads)
    want_len = rng.randint(rmin, rmax + 1, size=n_reads)
    rb, rq, ri, rd, lens = [], [], [], [], []
    for r in range(n_reads):
        h = haps[int(hap_pick[r])]
        R = int(min(want_len[r], h.size))
        off = int(rng.randint(0, h.size - R + 1))
        bases = h[off:off + R].copy()
        q = rng.randint(6, 41, size=R).astype(np.uint8)
        err = rng.random_sample(R) < np.power(10.0, -q.astype(np.float64) / 10.0)
        if err.any():
            idx = np.nonzero(err)[0]
            cur = np.searchsorted(_ACGT, bases[idx])
            bases[idx] = _ACGT[(cur + 1 + rng.randint(0, 3, size=idx.size)) % 4]
        rb.append(bases)
        rq.append(q)
        ri.append(rng.randint(30, 46, size=R).astype(np.uint8))
        rd.append(rng.randint(30, 46, size=R).astype(np.uint8))
        lens.append(R)

    read_off = np.zeros(n_reads + 1, np.int64)
    read_off[1:] = np.cumsum(lens)
    hap_off = np.zeros(n_haps + 1, np.int64)
    hap_off[1:] = np.cumsum([h.size for h in haps])
    total = int(read_off[-1])
    return FlatBatch(n_reads, n_haps, read_off, hap_off,
                     np.concatenate(rb).astype(np.uint8), np.concatenate(rq).astype(np.uint8),
                     np.concatenate(ri).astype(np.uint8), np.concatenate(rd).astype(np.uint8),
                     np.full(total, 10, np.uint8), np.concatenate(haps).astype(np.uint8),
                     {"kind": kind, "seed": seed, "n_reads": n_reads, "n_haps": n_haps})


def random_batch(rng, n_reads: int, n_haps: int, read_len=(1, 60), hap_len=(1, 80),
                 alphabet: bytes = b"ACGT", qual_range=(0, 60), related: bool = True) -> FlatBatch:
    """Small unstructured batches for edge-case parity tests (any bytes, any quals)."""
    alpha = np.frombuffer(alphabet, dtype=np.uint8)
    haps = [alpha[rng.randint(0, alpha.size, size=int(rng.randint(hap_len[0], hap_len[1] + 1)))]
            for _ in range(n_haps)]
    rb, rq, ri, rd, rc, lens = [], [], [], [], [], []
    for _ in range(n_reads):
        R = int(rng.randint(read_len[0], read_len[1] + 1))
        h = haps[int(rng.randint(0, n_haps))]
        if related and h.size >= R and rng.random_sample() < 0.7:
            off = int(rng.randint(0, h.size - R + 1))
            b = h[off:off + R].copy()
            flip = rng.random_sample(R) < 0.03
            b[flip] = alpha[rng.randint(0, alpha.size, size=int(flip.sum()))]
        else:
            b = alpha[rng.randint(0, alpha.size, size=R)]
        rb.append(b)
        for dst in (rq, ri, rd, rc):
            dst.append(rng.randint(qual_range[0], qual_range[1] + 1, size=R).astype(np.uint8))
        lens.append(R)
    read_off = np.zeros(n_reads + 1, np.int64)
    read_off[1:] = np.cumsum(lens)
    hap_off = np.zeros(n_haps + 1, np.int64)
    hap_off[1:] = np.cumsum([h.size for h in haps])
    cat = lambda xs: np.concatenate(xs).astype(np.uint8)  # noqa: E731
    return FlatBatch(n_reads, n_haps, read_off, hap_off, cat(rb), cat(rq), cat(ri), cat(rd), cat(rc),
                     cat(haps), {"kind": "random"})
