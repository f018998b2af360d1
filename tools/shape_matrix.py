#!/usr/bin/env python3
"""Dev tool: host-call throughput over a matrix of batch shapes (reads x haplotypes) -- looks for dips."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from gkl_amd import native  # noqa: E402
from gkl_amd.synth import DEFAULT_SEED, make_batch  # noqa: E402

reads = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "20,50,100,200,500,1000,2000,5000".split(","))]
haps = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "2,4,8,16,32,64,128".split(","))]
print("reads \\ haps " + " ".join(f"{h:>12d}" for h in haps))
with native.PairHmmContext() as c:
    for n in reads:
        row = []
        for h in haps:
            b = make_batch("hc", n, h, seed=DEFAULT_SEED)
            out = np.empty(b.n_pairs)
            with native.PinnedBatch(b) as pb:
                for _ in range(4):
                    c.compute(pb, out)
                ts = []
                for _ in range(12):
                    t = time.perf_counter()
                    c.compute(pb, out)
                    ts.append(time.perf_counter() - t)
            ms = np.median(ts) * 1e3
            row.append(f"{ms:5.2f}ms {b.cells / ms / 1e6:5.0f}")
        print(f"{n:12d} " + " ".join(f"{x:>12s}" for x in row), flush=True)
