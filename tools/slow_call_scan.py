#!/usr/bin/env python3
"""Which calls of a fresh process are slow?  One caller, N 100 x 10 host calls back to back from the very first one; prints
every call above 1 ms with its index (docs/NOTES.md 56: the ~500th call of a Python process costs 35-80 ms once).
usage: tools/slow_call_scan.py [calls]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkl_amd import native  # noqa: E402
from gkl_amd.synth import make_batch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
if os.environ.get("NO_GC") == "1":     # NOTES 56: the one slow call is the interpreter's full garbage collection, not the library
    import gc
    gc.collect()
    gc.disable()
b = make_batch("hc", 100, 10)
out = np.empty(b.n_pairs)
with native.PinnedBatch(b) as pb, native.PairHmmContext(device=0) as c:
    lat = []
    for _ in range(n):
        t = time.perf_counter()
        c.compute(pb, out)
        lat.append((time.perf_counter() - t) * 1e3)
lat = np.array(lat)
print("median %.3f ms; calls above 1 ms:" % np.median(lat), [(int(i), round(float(lat[i]), 2)) for i in np.nonzero(lat > 1.0)[0]])
