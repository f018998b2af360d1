#!/usr/bin/env python3
"""Dev tool: mid-size host calls with the fused one-wavefront-per-pair kernel up to GKLHIP_FUSED_MAX_PAIRS pairs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkl_amd import native  # noqa: E402
from gkl_amd.synth import DEFAULT_SEED, make_batch  # noqa: E402
tag = os.environ.get("GKLHIP_FUSED_MAX_PAIRS", "2048")
for nr, nh in ((200, 10), (300, 10), (400, 10), (250, 16), (300, 16), (400, 20), (400, 40), (250, 128), (1000, 50)):
    b = make_batch("hc", nr, nh, seed=DEFAULT_SEED)
    out = np.empty(b.n_pairs)
    with native.PinnedBatch(b) as pb, native.PairHmmContext() as c:
        for _ in range(20):
            c.compute(pb, out)
        ts = []
        for _ in range(60):
            t = time.perf_counter(); c.compute(pb, out); ts.append(time.perf_counter() - t)
    ms = float(np.median(ts)) * 1e3
    print(f"fused<= {tag}: {nr}x{nh} ({b.n_pairs} pairs): {ms:.4f} ms = {b.cells / ms / 1e6:.0f} GCUPS", flush=True)
