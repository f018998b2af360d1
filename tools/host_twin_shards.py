#!/usr/bin/env python3
"""Host-buffer path of the C2 batch by number of twin engines (GKL_HIP_HOST_SHARDS, read when the context is made) and
maxNumberOfThreads: one process per setting.  usage: GKL_HIP_HOST_SHARDS=n tools/host_twin_shards.py max_threads"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkl_amd import native
from gkl_amd.synth import make_batch
mt = int(sys.argv[1]) if len(sys.argv) > 1 else 1
b = make_batch("hc", 10000, 128)
out = np.empty(b.n_pairs)
with native.PinnedBatch(b) as pb, native.PairHmmContext(device=0, max_threads=mt) as c:
    for _ in range(4):
        c.compute(pb, out)
    ts = []
    for _ in range(20):
        t = time.perf_counter(); c.compute(pb, out); ts.append(time.perf_counter() - t)
print(f"GKL_HIP_HOST_SHARDS={os.environ.get('GKL_HIP_HOST_SHARDS', '2 (default)')} max_threads={mt}: median {np.median(ts)*1e3:.2f} ms, p10 {np.percentile(ts,10)*1e3:.2f}, p90 {np.percentile(ts,90)*1e3:.2f}")
