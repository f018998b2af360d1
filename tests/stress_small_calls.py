#!/usr/bin/env python3
"""Verification script (lives under tests/ because it checks against oracle/, which only tests may use): concurrent
small host-buffer calls from many threads for a wall-clock budget -- every thread owns a context, draws a new small
batch every few calls (shapes across the 2 / 4 / 8-row fp32 kernels and the 2 / 4 / 6-row per-pair kernels, both
arithmetics, contexts closed and reopened while others are in flight, now and then a call too big for the small-call
path) and compares each result bit for bit with the oracle.

    python tests/stress_small_calls.py --seconds 120 --threads 16 [--seed 1]
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import torch  # noqa: F401
    from gkl_amd import native
    from gkl_amd.synth import make_batch, random_batch
    from oracle.oracle import Oracle

    lock = threading.Lock()
    bad, calls = [], [0]
    native.small_call_counts(0, reset=True)

    def draw(rng, seed):
        kind = rng.random_sample()
        if kind < 0.5:
            return make_batch("hc", int(rng.randint(1, 160)), int(rng.randint(1, 16)), seed=seed)
        if kind < 0.7:
            return make_batch("hc", int(rng.randint(1, 60)), int(rng.randint(1, 10)), seed=seed, read_len=(5, 120), hap_len=(20, 200))
        if kind < 0.85:
            return make_batch("hc", int(rng.randint(1, 30)), int(rng.randint(1, 8)), seed=seed, read_len=(200, 383), hap_len=(250, 500))
        if kind < 0.97:
            return random_batch(rng, int(rng.randint(1, 80)), int(rng.randint(1, 12)))
        return make_batch("hc", 2500, 40, seed=seed)  # 100k pairs: the stream-ordered path, between small calls

    # the batches and what the oracle says about them first (python holds the GIL while it draws and checks: the timed
    # part below is compute calls only, so that calls of different threads really meet on the device)
    plans = []
    for tid in range(a.threads):
        rng = np.random.RandomState(a.seed * 7919 + tid)
        oracle = Oracle()
        items = []
        for _ in range(10):
            seed = int(rng.randint(1, 1 << 30))
            b = draw(rng, seed)
            items.append((seed, b, {f: oracle.batch(b, n_threads=4, fma_mode=f) for f in (0, 1)}))
        plans.append((rng, items))
    t_end = time.time() + a.seconds
    start = threading.Barrier(a.threads)

    def worker(tid):
        rng, items = plans[tid]
        n_local = 0
        try:
            start.wait()
            while time.time() < t_end:
                fma = int(rng.randint(0, 2)) if rng.random_sample() < 0.2 else 1
                with native.PairHmmContext(fma_mode=fma) as c:
                    for _ in range(int(rng.randint(20, 400))):
                        seed, b, want = items[int(rng.randint(0, len(items)))]
                        got = c.compute(b)
                        n_local += 1
                        if got.tobytes() != want[fma].tobytes():
                            with lock:
                                bad.append((tid, seed, b.n_reads, b.n_haps, fma, int((bits(got) != bits(want[fma])).sum())))
                            return
        except Exception as e:  # noqa: BLE001
            with lock:
                bad.append((tid, repr(e)))
        finally:
            with lock:
                calls[0] += n_local

    th = [threading.Thread(target=worker, args=(i,)) for i in range(a.threads)]
    t0 = time.time()
    [t.start() for t in th]
    [t.join() for t in th]
    k = native.small_call_counts(0)
    print(f"{a.threads} threads, {time.time() - t0:.0f} s: {calls[0]} calls, small-call path {k[0]}, launched together {k[1]} in {k[2]} "
          f"launch sets, {len(bad)} mismatches / errors")
    for x in bad[:10]:
        print("  BAD", x)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
