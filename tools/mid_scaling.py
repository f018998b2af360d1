#!/usr/bin/env python3
"""Dev tool: where a mid-size host call's time goes (fp32 kernel, policy + fp64, everything else)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from gkl_amd import native  # noqa: E402
from gkl_amd.synth import DEFAULT_SEED, make_batch  # noqa: E402

for n, h in ((100, 10), (200, 20), (400, 10), (400, 40), (250, 128), (600, 64), (1000, 50)):
    b = make_batch("hc", n, h, seed=DEFAULT_SEED)
    out = np.empty(b.n_pairs)
    with native.PinnedBatch(b) as pb:
        with native.PairHmmContext() as c:
            for _ in range(8):
                c.compute(pb, out)
            ts = []
            for _ in range(30):
                t = time.perf_counter()
                c.compute(pb, out)
                ts.append(time.perf_counter() - t)
        with native.PairHmmContext(record_events=True) as c:
            for _ in range(6):
                c.compute(pb, out)
            s = c.stats()
    ms = np.median(ts) * 1e3
    print(f"{n:5d} x {h:3d} ({b.n_pairs:6d} pairs, {b.cells / 1e6:7.1f} Mcells): call {ms:.3f} ms = {b.cells / ms / 1e6:6.0f} GCUPS | fp32 {s['ms_fwd_main'] * 1e3:6.1f} us "
          f"(rpl {s['rows_per_lane']}, {s['n_chunks']} x {s['n_hap_groups']} jobs) policy+fp64 {s['ms_fwd_fallback'] * 1e3:6.1f} us device total {s['ms_total_device'] * 1e3:6.1f} us "
          f"fallback {s['n_fallback'] / b.n_pairs:.3f}", flush=True)
