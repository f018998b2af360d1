#!/usr/bin/env python3
"""Dev tool: time the device-resident path on a synthetic batch and print kernel stats."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkl_amd import native  # noqa: E402
from gkl_amd.synth import make_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="hc")
ap.add_argument("--reads", type=int, default=10000)
ap.add_argument("--haps", type=int, default=128)
ap.add_argument("--steps", type=int, default=40)  # short kernels need the clocks ramped up: measure the last of many
ap.add_argument("--double", action="store_true")
ap.add_argument("--rpl", type=int, default=0)
ap.add_argument("--lib", default=None)
a = ap.parse_args()

b = make_batch(a.kind, a.reads, a.haps)
db = native.DeviceBatch.upload(b)
ctx = native.PairHmmContext(use_double=a.double, record_events=True, rows_per_lane=a.rpl, lib_path=a.lib)
out = ctx.compute_device(db)
torch.cuda.synchronize()
for i in range(a.steps):
    t = time.time()
    out = ctx.compute_device(db)
    torch.cuda.synchronize()
    dt = time.time() - t
    st = ctx.stats()
    fb = st["n_fallback"]
    if i < a.steps - 1:
        continue
    print(f"{a.lib or 'default'}: {a.kind} {a.reads}x{a.haps} cells {b.cells:.3e} wall {dt*1e3:.2f} ms -> {b.cells/dt/1e9:.1f} GCUPS | "
          f"main {st['ms_fwd_main']:.2f} ms ({b.cells/st['ms_fwd_main']/1e6:.1f} GCUPS) fallback {st['ms_fwd_fallback']:.2f} ms "
          f"(n={fb}, {fb/b.n_pairs:.3f}) total_dev {st['ms_total_device']:.2f} ms chunks {st['n_chunks']} groups {st['n_hap_groups']} "
          f"rpl {st['rows_per_lane']} fill {st['lane_fill']:.3f}")
