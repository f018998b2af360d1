#!/usr/bin/env python3
"""Where a PDHMM region-sized call's time goes: the fixture's 276 reads x 48 haplotypes through gklhip_pdhmm_compute_cross
with GKLHIP_TIMING=1 (the library's own split on stderr) and the median call time.  usage: GKLHIP_TIMING=1 tools/pd_region_timing.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

c = bench._PdhmmRegionCaller(0)
for _ in range(50):
    c.compute()
ts = []
for _ in range(200):
    t = time.perf_counter()
    c.compute()
    ts.append(time.perf_counter() - t)
print("median %.4f ms, p10 %.4f, p90 %.4f; kernel %.4f ms" % (np.median(ts) * 1e3, np.percentile(ts, 10) * 1e3, np.percentile(ts, 90) * 1e3, c.ctx.last_kernel_ms()))
