#!/usr/bin/env python3
"""Measurement script (lives under tests/ because it times / checks against oracle/, which only tests may use): Smith-Waterman throughput (batch C ABI) and per-call latency (single-pair C ABI = what alignNative
pays) on GATK-shaped inputs, with the reference's own AVX2 / AVX-512 objects timed on one host core beside it."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_pairs(rng, n, ref_len, alt_len):
    from tests.test_sw import mutate
    pairs = []
    for _ in range(n):
        L = int(rng.randint(ref_len[0], ref_len[1] + 1))
        M = int(rng.randint(alt_len[0], alt_len[1] + 1))
        ref = bytes(rng.choice(list(b"ACGT"), size=L).tolist())
        start = int(rng.randint(0, max(1, L - M + 1)))
        alt = mutate(rng, ref[start:start + M], 0.03)
        pairs.append((ref, alt))
    return pairs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from gkl_amd import native
    SOFTCLIP = native.SW_SOFTCLIP
    params = (200, -150, -260, -11)
    rng = np.random.RandomState(3)
    shapes = {"haplotype-to-reference (ref 300-600, alt 250-600)": ((300, 600), (250, 600)),
              "read-to-haplotype (ref 300-500, alt 100-250)": ((300, 500), (100, 250))}
    ctx = native.SwContext()
    ref_eng = None
    try:
        from oracle.sw import SwReference
        ref_eng = SwReference()
    except Exception as e:  # noqa: BLE001
        print(f"(reference unavailable: {e})")
    for name, (rl, al) in shapes.items():
        pairs = make_pairs(rng, a.pairs, rl, al)
        refs, alts = [r for r, _ in pairs], [x for _, x in pairs]
        pk = ctx.pack(refs, alts, cigar_stride=256)   # the text of these shapes is far shorter than 2 * max(len)
        cells = pk["cells"]
        ctx.align_packed(pk, params, SOFTCLIP)
        best_k, best_w = 1e9, 1e9
        for _ in range(a.reps):
            t0 = time.perf_counter()
            ctx.align_packed(pk, params, SOFTCLIP)
            best_w = min(best_w, time.perf_counter() - t0)
            best_k = min(best_k, ctx.last_kernel_ms())
        line = (f"{name}: {len(pairs)} pairs {cells:.3e} cells | batch kernel {best_k:.3f} ms = {cells / best_k / 1e6:.1f} GCUPS, "
                f"C ABI host-to-host {best_w * 1e3:.2f} ms = {cells / best_w / 1e9:.1f} GCUPS ({best_w / len(pairs) * 1e6:.2f} us/pair)")
        # single-pair latency, the alignNative pattern
        sub = pairs[:200]
        for r, x in sub[:20]:
            ctx.align(r, x, params, SOFTCLIP)
        t0 = time.perf_counter()
        for r, x in sub:
            ctx.align(r, x, params, SOFTCLIP)
        one = (time.perf_counter() - t0) / len(sub)
        line += f" | single-pair call {one * 1e6:.0f} us"
        if ref_eng is not None:
            eng = 2 if ref_eng.has_avx512() else 1
            for r, x in sub[:20]:
                ref_eng.align(r, x, params, SOFTCLIP, engine=eng)
            t0 = time.perf_counter()
            for r, x in sub:
                ref_eng.align(r, x, params, SOFTCLIP, engine=eng)
            cpu = (time.perf_counter() - t0) / len(sub)
            sub_cells = sum(len(r) * len(x) for r, x in sub)
            line += (f" | reference ({'AVX-512' if eng == 2 else 'AVX2'}, 1 thread) {cpu * 1e6:.0f} us/call = "
                     f"{sub_cells / (cpu * len(sub)) / 1e9:.2f} GCUPS")
        print(line, flush=True)


if __name__ == "__main__":
    main()
