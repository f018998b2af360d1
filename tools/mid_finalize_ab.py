#!/usr/bin/env python3
"""Host-call latency of mid-size regions against GKL_HIP_FINALIZE_MIN (pairs from which the one-pass host log10 is spread
over worker threads; read when the library is loaded): one process per setting.  usage: GKL_HIP_FINALIZE_MIN=n tools/mid_finalize_ab.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkl_amd import native
from gkl_amd.synth import make_batch
res = []
for nr, nh, mt in ((100, 10, 0), (150, 30, 0), (200, 40, 0), (400, 40, 0), (400, 40, 4), (400, 40, 1), (250, 128, 0)):
    b = make_batch("hc", nr, nh)
    out = np.empty(b.n_pairs)
    with native.PinnedBatch(b) as pb, native.PairHmmContext(device=0, max_threads=mt) as c:
        for _ in range(30):
            c.compute(pb, out)
        ts = []
        for _ in range(200):
            t = time.perf_counter(); c.compute(pb, out); ts.append(time.perf_counter() - t)
    res.append(f"{nr}x{nh}/mt{mt} {np.median(ts)*1e3:.4f}")
print(f"GKL_HIP_FINALIZE_MIN={os.environ.get('GKL_HIP_FINALIZE_MIN', 'default')}: " + "  ".join(res))
