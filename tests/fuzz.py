#!/usr/bin/env python3
"""Measurement/verification script (lives under tests/ because it checks against oracle/, which only tests may use):
randomised parity fuzz of the three HIP paths against their oracles for a wall-clock budget.

    python tests/fuzz.py --seconds 300 [--seed 1]

Every round draws new shapes (read / haplotype counts and lengths from 1 up to the striped-kernel range, alphabets with
N / lower case / arbitrary bytes, qualities 0..255, related and unrelated reads, both arithmetics, every rows-per-lane
variant, single- and multi-shard contexts) and compares bit for bit: PairHMM raw fp32 / fp64 sums, fallback flags and
host-finalised doubles; PDHMM doubles (vector arithmetic, paired and cross entry points); Smith-Waterman CIGAR bytes
and offsets.  Prints one line per mismatch (with the seed that reproduces it) and a summary; exit code 1 on any."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import torch  # noqa: F401
    from gkl_amd import native
    from gkl_amd.pdhmm_batch import PdhmmBatch
    from gkl_amd.synth import make_batch, random_batch
    from oracle.oracle import Oracle
    from oracle.pdhmm import PdhmmOracle
    from oracle.sw import SwOracle
    from tests.test_pdhmm import cross_product, expand_cross, random_pd_batch

    oracle, pdo, swo = Oracle(), PdhmmOracle(), SwOracle()
    ctxs = {}

    def ctx(use_double, fma, rpl, devices=None):
        k = (use_double, fma, rpl, devices)
        if k not in ctxs:
            ctxs[k] = native.PairHmmContext(use_double=use_double, fma_mode=fma, rows_per_lane=rpl, devices=devices)
        return ctxs[k]

    pd_ctx = {m: native.PdhmmContext(fma_mode=m, reference_tail=False) for m in (1, 0)}
    sw = native.SwContext()
    alphabets = [b"ACGT", b"ACGTN", b"ACGTNacgtXRY*", b"AC", b"N", bytes(range(1, 256))]
    t_end = time.time() + a.seconds
    n_rounds = n_bad = 0
    cells = 0
    rnd = 0
    while time.time() < t_end:
        seed = a.seed * 1000003 + rnd
        rnd += 1
        rng = np.random.RandomState(seed % (2 ** 31))
        try:
            # ---------------- PairHMM
            kind = rng.randint(0, 4)
            if kind == 0:
                b = random_batch(rng, int(rng.randint(1, 120)), int(rng.randint(1, 40)), read_len=(1, int(rng.randint(1, 320))),
                                 hap_len=(1, int(rng.randint(1, 540))), alphabet=alphabets[rng.randint(0, len(alphabets))],
                                 qual_range=(0, int(rng.choice([40, 60, 93, 255]))), related=bool(rng.randint(0, 2)))
            elif kind == 1:
                b = make_batch("hc", int(rng.randint(1, 700)), int(rng.randint(1, 48)), seed=int(rng.randint(0, 1 << 30)))
            elif kind == 2:   # long reads: the wide kernel (2-4 wavefronts per read), one time in three the super-stripe kernel
                              # (> 2047 bases) with its striped twin for the jobs it leaves (haplotypes of <= 63 bases, fp64: N haplotypes)
                top = int(rng.randint(2048, 4300)) if rng.randint(0, 3) == 0 else int(rng.randint(520, 1500))
                b = random_batch(rng, int(rng.randint(1, 12)), int(rng.randint(1, 6)), read_len=(200, top),
                                 hap_len=(int(rng.choice([20, 50, 70])), int(rng.randint(100, 900))), alphabet=b"ACGTN" if rng.randint(0, 2) else b"ACGT",
                                 related=bool(rng.randint(0, 2)))
            else:
                b = make_batch("mixed", int(rng.randint(1, 300)), int(rng.randint(1, 20)), seed=int(rng.randint(0, 1 << 30)),
                               read_len=(int(rng.randint(1, 40)), int(rng.randint(40, 260))), hap_len=(int(rng.randint(20, 90)), int(rng.randint(90, 500))))
            fma = int(rng.randint(0, 2))
            use_double = bool(rng.randint(0, 4) == 0)
            rpl = 0 if use_double else int(rng.choice([0, 0, 8, 4, 2]))
            devices = (0, 0) if rng.randint(0, 5) == 0 else None
            c = ctx(use_double, fma, rpl, devices)
            out = c.compute(b)
            oo, o32, o64, ou = oracle.batch(b, use_double=use_double, fma_mode=fma, want_raw=True, n_threads=8)
            ok = np.array_equal(bits(out), bits(oo))
            if devices is None:
                r32, r64, u = c.raw(b.n_pairs)
                ok = ok and np.array_equal(u, ou) and (use_double or np.array_equal(bits(r32), bits(o32))) and \
                    np.array_equal(bits(r64[u == 1]), bits(o64[ou == 1]))
            if not ok:
                n_bad += 1
                print(f"PAIRHMM MISMATCH seed={seed} kind={kind} reads={b.n_reads} haps={b.n_haps} fma={fma} double={use_double} rpl={rpl} devices={devices}", flush=True)
            cells += b.cells
            # ---------------- PDHMM
            m = int(rng.randint(0, 2))
            if rng.randint(0, 2):
                pb = random_pd_batch(rng, int(rng.randint(1, 150)), read_len=(1, int(rng.randint(2, 300))), hap_len=(1, int(rng.randint(2, 400))),
                                     flag_rate=float(rng.choice([0.0, 0.02, 0.15, 0.5])), odd_haps=float(rng.choice([0.0, 0.0, 0.3])))
            else:
                pb = cross_product(rng, int(rng.randint(1, 60)), int(rng.randint(1, 8)), (1, int(rng.randint(2, 200))), (1, int(rng.randint(2, 300))))
            got = pd_ctx[m].compute(pb)
            st, vec = pdo.compute(pb, semantics=2 if m == 1 else 0)
            if st != 0 or got.tobytes() != vec.tobytes():
                n_bad += 1
                print(f"PDHMM MISMATCH seed={seed} fma={m} batch={pb.batch}", flush=True)
            if rnd % 3 == 0:
                # the cross entry point (what computeLikelihoodsNative calls): few flags -> the LDS-table kernel, many
                # or odd bases -> the predicate / byte-comparing kernels, reads of 320+ bases -> striped jobs
                fr = float(rng.choice([0.0, 0.005, 0.02, 0.15]))
                prd = random_pd_batch(rng, int(rng.randint(1, 90)), read_len=(1, int(rng.randint(2, 400))), hap_len=(1, 2))
                phd = random_pd_batch(rng, int(rng.randint(1, 10)), read_len=(1, 2), hap_len=(1, int(rng.randint(2, 400))), flag_rate=fr,
                                      with_n=bool(rng.randint(0, 2)), odd_haps=float(rng.choice([0.0, 0.0, 0.3])))
                got = pd_ctx[m].compute_cross(prd, phd)
                st, vec = pdo.compute(expand_cross(prd, phd), semantics=2 if m == 1 else 0)
                if st != 0 or got.tobytes() != vec.tobytes():
                    n_bad += 1
                    print(f"PDHMM CROSS MISMATCH seed={seed} fma={m} reads={prd.batch} haps={phd.batch} routing={pd_ctx[m].last_routing()}", flush=True)
            # ---------------- Smith-Waterman
            n_ref, n_alt = int(rng.randint(1, 700)), int(rng.randint(1, 700))
            ref = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.randint(0, 4, n_ref)])
            if rng.randint(0, 2) and n_ref > 4:
                s0 = int(rng.randint(0, n_ref - 1))
                altb = bytearray(ref[s0:s0 + n_alt])
                for _ in range(int(rng.randint(0, 6))):
                    if altb:
                        altb[int(rng.randint(0, len(altb)))] = b"ACGT"[int(rng.randint(0, 4))]
                alt = bytes(altb) or b"A"
            else:
                alt = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.randint(0, 4, n_alt)])
            params = [(200, -150, -260, -11), (25, -50, -110, -6), (3, -1, -4, -3), (10, -15, -30, -5)][int(rng.randint(0, 4))]
            strategy = [9, 10, 11, 12][int(rng.randint(0, 4))]
            try:
                g = sw.align(ref, alt, params, strategy)
                e = swo.align(ref, alt, params, strategy)[1:]
                if g != e:
                    n_bad += 1
                    print(f"SW MISMATCH seed={seed} ref={n_ref} alt={len(alt)} params={params} strategy={strategy}", flush=True)
            except ValueError:
                pass  # a strategy code the mirror rejects
            n_rounds += 1
        except Exception as ex:  # an exception in a product call is a finding too
            n_bad += 1
            print(f"EXCEPTION seed={seed}: {ex!r}", flush=True)
    print(f"fuzz: {n_rounds} rounds in {a.seconds:.0f} s, {cells:.3e} PairHMM cells, {n_bad} mismatches", flush=True)
    for c in ctxs.values():
        c.close()
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
