#!/usr/bin/env python3
"""Dev tool: host-buffer path of the full batch with 1, 2, 3, 4 shards on the same GPU (same box, interleaved)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkl_amd import native
from gkl_amd.synth import make_batch
b = make_batch("hc", int(sys.argv[1]) if len(sys.argv) > 1 else 10000, 128)
out = np.empty(b.n_pairs)
with native.PinnedBatch(b) as pb:
    ctxs = {n: native.PairHmmContext(devices=[0] * n) if n > 1 else native.PairHmmContext(device=0) for n in (1, 2, 3, 4)}
    for rep in range(3):
        for n, c in ctxs.items():
            for _ in range(2):
                c.compute(pb, out)
            ts = []
            for _ in range(6):
                t = time.perf_counter(); c.compute(pb, out); ts.append(time.perf_counter() - t)
            print(f"rep {rep} shards {n}: {np.median(ts)*1e3:.2f} ms", flush=True)
