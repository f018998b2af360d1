#!/usr/bin/env python3
"""Dev tool: host-buffer path (gklhip_compute, pinned inputs) of two library builds alternating on one box:
the 10k x 128 batch and its first eighth.  usage: ab_host_path.py <libA.so> <libB.so>"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from gkl_amd import native  # noqa: E402
from gkl_amd.synth import make_batch  # noqa: E402

whole = make_batch("hc")
shapes = (("whole", whole, 12), ("eighth", whole.read_slice(0, whole.n_reads // 8), 40))
if len(sys.argv) > 3 and sys.argv[3] == "small":
    shapes = (("eighth", whole.read_slice(0, whole.n_reads // 8), 40), ("1000x32", make_batch("hc", 1000, 32), 60),
              ("300x16", make_batch("region", 300, 16), 100), ("100x10", make_batch("hc", 100, 10), 200))
for name, b, calls in shapes:
    out = np.empty(b.n_pairs)
    pin = native.PinnedBatch(b)
    for rnd in range(3):
        for lib in sys.argv[1:3]:
            with native.PairHmmContext(lib_path=lib) as c:
                for _ in range(4):
                    c.compute(pin.batch, out)
                ts = []
                for _ in range(calls):
                    t = time.perf_counter(); c.compute(pin.batch, out); ts.append(time.perf_counter() - t)
            print(f"{name:6s} {os.path.basename(lib):40s} {np.median(ts) * 1e3:8.3f} ms per call", flush=True)
    pin.close()
