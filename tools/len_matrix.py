#!/usr/bin/env python3
"""Dev tool: host-call throughput over read-length x haplotype-length ranges at a fixed 1000 x 32 shape -- looks for dips."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from gkl_amd import native  # noqa: E402
from gkl_amd.synth import DEFAULT_SEED, make_batch  # noqa: E402

n, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 32)
rls = [(30, 60), (70, 101), (100, 151), (140, 251), (240, 300), (300, 500), (480, 520), (600, 1000)]
hls = [(80, 150), (200, 400), (400, 800), (900, 1100), (1500, 2500)]
print(f"{n} x {h}; reads \\ haps " + " ".join(f"{str(x):>15s}" for x in hls))
with native.PairHmmContext(record_events=False) as c:
    for rl in rls:
        row = []
        for hl in hls:
            if rl[1] > hl[1] + 100:
                row.append("-")
                continue
            b = make_batch("hc", n, h, seed=DEFAULT_SEED, read_len=rl, hap_len=hl)
            out = np.empty(b.n_pairs)
            with native.PinnedBatch(b) as pb:
                for _ in range(3):
                    c.compute(pb, out)
                ts = []
                for _ in range(8):
                    t = time.perf_counter()
                    c.compute(pb, out)
                    ts.append(time.perf_counter() - t)
                st = c.stats()
            ms = np.median(ts) * 1e3
            row.append(f"{ms:5.2f}ms {b.cells / ms / 1e6:5.0f} r{st['rows_per_lane']}")
        print(f"{str(rl):>12s} " + " ".join(f"{x:>15s}" for x in row), flush=True)
