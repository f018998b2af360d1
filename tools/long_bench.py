#!/usr/bin/env python3
"""Dev tool: long reads (beyond one wavefront's rows) -- device-resident step time and its split."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkl_amd import native  # noqa: E402
from gkl_amd.synth import DEFAULT_SEED, make_batch  # noqa: E402

for n, h, rl, hl in ((200, 16, (600, 1000), (900, 1100)), (1000, 32, (600, 1000), (900, 1100)), (1000, 32, (1000, 1000), (1200, 1400)),
                     (2000, 32, (1500, 2000), (2000, 2400)), (1000, 32, (520, 640), (700, 900)),
                     (200, 32, (4000, 6000), (5000, 7000)), (64, 32, (14000, 16000), (15000, 17000))):
    b = make_batch("hc", n, h, seed=DEFAULT_SEED, read_len=rl, hap_len=hl)
    db = native.DeviceBatch.upload(b)
    for dbl in (False, True):
        with native.PairHmmContext(use_double=dbl, record_events=True) as c:
            out = c.compute_device(db); torch.cuda.synchronize()
            ts = []
            for _ in range(6):
                t = time.perf_counter(); c.compute_device(db, out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
            st = c.stats()
        ms = float(np.median(ts)) * 1e3
        print(f"{n}x{h} reads {rl} haps {hl} double={dbl}: {ms:.2f} ms = {b.cells / ms / 1e6:.0f} GCUPS | main {st['ms_fwd_main']:.2f} fallback {st['ms_fwd_fallback']:.2f} "
              f"(fb {st['n_fallback'] / b.n_pairs:.2f}) long pairs {st['n_long_pairs']}", flush=True)
