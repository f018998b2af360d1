#!/usr/bin/env python3
"""Dev tool: host-buffer call latency of GATK-sized regions with the fused one-wavefront-per-pair kernel on and off
(GKLHIP_FUSED_PAIRS is read once per process: run with the variable set to 0 / 1), and 16 concurrent mock-JNI callers."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkl_amd import native  # noqa: E402
from gkl_amd.synth import DEFAULT_SEED, make_batch  # noqa: E402

tag = os.environ.get("GKLHIP_FUSED_PAIRS", "default")
for kind, nr, nh in (("hc", 100, 10), ("region", 100, 10), ("hc", 30, 5), ("hc", 200, 10), ("hc", 60, 32), ("hc", 400, 5), ("hc", 128, 16)):
    b = make_batch(kind, nr, nh, seed=DEFAULT_SEED)
    out = np.empty(b.n_pairs)
    with native.PinnedBatch(b) as pb, native.PairHmmContext() as c:
        for _ in range(30):
            c.compute(pb, out)
        ts = []
        for _ in range(200):
            t = time.perf_counter(); c.compute(pb, out); ts.append(time.perf_counter() - t)
    ms = float(np.median(ts)) * 1e3
    print(f"fused={tag} {kind} {nr}x{nh} ({b.n_pairs} pairs): {ms:.4f} ms per call = {b.cells / ms / 1e6:.0f} GCUPS", flush=True)
if len(sys.argv) > 1:
    from tests import mockjni
    os.environ["GKL_HIP_SLOTS"] = "16"
    for threads in (1, 4, 16):
        b = make_batch("hc", 100 * threads, 10, seed=DEFAULT_SEED)
        best = None
        for _ in range(3):
            t = []
            rc, _, cls, msg, wall = mockjni.run_concurrent(b, threads, iters=150, warm=20, timing=t)
            assert rc == 0, (cls, msg)
            best = wall if best is None else min(best, wall)
        print(f"fused={tag} {threads} callers: {b.cells * 150 / best / 1e6:.0f} GCUPS aggregate (best of 3)", flush=True)
