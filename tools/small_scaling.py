#!/usr/bin/env python3
"""Dev tool: fp32 forward kernel and per-pair policy kernel time of ONE call of n x 10 pairs (n = 100 .. 1600) at forced
rows per lane -- how a combined launch of k GATK-sized calls should scale."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from gkl_amd import native  # noqa: E402
from gkl_amd.synth import DEFAULT_SEED, make_batch  # noqa: E402

for n in (100, 200, 400, 700, 1000, 1600):
    b = make_batch("hc", n, 10, seed=DEFAULT_SEED)
    row = []
    for rpl in (2, 4, 8):
        with native.PairHmmContext(record_events=True, rows_per_lane=rpl) as c:
            t = []
            for _ in range(12):
                c.compute(b)
                s = c.stats()
                t.append((s["ms_fwd_main"], s["ms_fwd_fallback"], s["n_chunks"] * s["n_hap_groups"], s["rows_per_lane"]))
            t = t[2:]
            row.append(f"rpl {t[0][3]}: {t[0][2]:5d} waves fwd {np.median([x[0] for x in t]) * 1e3:6.1f} us policy {np.median([x[1] for x in t]) * 1e3:6.1f} us")
    print(f"{n:5d} x 10 ({b.cells / 1e6:6.1f} Mcells): " + " | ".join(row), flush=True)
