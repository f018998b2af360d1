#!/usr/bin/env python3
"""Dev tool: per-call latency on GATK-sized batches (one active region each), device-resident and host-buffer paths."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkl_amd import native  # noqa: E402
from gkl_amd.synth import make_batch  # noqa: E402

for kind, nr, nh in (("hc", 100, 10), ("region", 100, 10), ("region", 300, 16), ("hc", 1000, 32), ("region", 2000, 64), ("hc", 1250, 128)):
    b = make_batch(kind, nr, nh)
    db = native.DeviceBatch.upload(b)
    out = torch.empty(b.n_pairs, dtype=torch.float64, device="cuda")
    host_out = np.empty(b.n_pairs)
    with native.PairHmmContext(record_events=True) as c:
        for _ in range(3):
            c.compute_device(db, out); torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            t = time.perf_counter(); c.compute_device(db, out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        st = c.stats()
        dev = float(np.median(ts))
    with native.PairHmmContext() as c:
        for _ in range(3):
            c.compute(b, host_out)
        ts = []
        for _ in range(20):
            t = time.perf_counter(); c.compute(b, host_out); ts.append(time.perf_counter() - t)
        host = float(np.median(ts))
    print(f"{kind} {nr}x{nh}: cells {b.cells:.2e} fb {st['n_fallback']/b.n_pairs:.2f} | device-resident {dev*1e3:.3f} ms "
          f"({b.cells/dev/1e9:.0f} GCUPS; main {st['ms_fwd_main']:.3f} fb {st['ms_fwd_fallback']:.3f} dev {st['ms_total_device']:.3f} "
          f"chunks {st['n_chunks']} groups {st['n_hap_groups']}) | host buffers {host*1e3:.3f} ms ({b.cells/host/1e9:.0f} GCUPS)", flush=True)
