#!/usr/bin/env python3
"""Dev tool: N threads, each with its own context, calling gklhip_compute on a 100 x 10 region back to back."""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from gkl_amd import native  # noqa: E402
from gkl_amd.synth import DEFAULT_SEED, make_batch  # noqa: E402

b = make_batch("hc", 100, 10, seed=DEFAULT_SEED)
iters = 300
for n in [int(x) for x in (sys.argv[1:] or ["1", "4", "16"])]:
    ctxs = [native.PairHmmContext() for _ in range(n)]
    pins = [native.PinnedBatch(b) for _ in range(n)]
    pbs = [p.batch for p in pins]
    outs = [np.empty(b.n_pairs) for _ in range(n)]
    for c, pb, o in zip(ctxs, pbs, outs):
        for _ in range(10):
            c.compute(pb, o)
    start = threading.Barrier(n + 1)

    def work(k):
        start.wait()
        for _ in range(iters):
            ctxs[k].compute(pbs[k], outs[k])
    th = [threading.Thread(target=work, args=(k,)) for k in range(n)]
    for t in th:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    print(f"C ABI, {n:2d} threads: {n * iters / dt:9.1f} calls/s, {b.cells * n * iters / dt / 1e9:7.1f} GCUPS aggregate, {dt / iters * 1e3:.3f} ms per call per thread", flush=True)
    for c, p in zip(ctxs, pins):
        p.close()
        c.close()
