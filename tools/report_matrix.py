#!/usr/bin/env python3
"""SURVEY.md 8(d) measurement matrix on one MI355X: for every synthetic workload print pairs, cells,
fallback fraction and GCUPS for
  (i)   the fp32 forward kernel over all pairs (device timestamps),
  (ii)  the fp64 forward kernel over all pairs (useDoublePrecision contexts),
  (iii) the precision policy end to end with inputs/outputs resident in HBM (what bench.py reports),
  (iv)  the host-buffer C ABI call gklhip_compute (H2D + kernels + D2H + host log10 = the JNI shim's cost).
Writes markdown to stdout (tools/profile.sh copies it to profiles/)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkl_amd import native  # noqa: E402
from gkl_amd.synth import make_batch  # noqa: E402

CASES = [("C1 hc 100x10", "hc", 100, 10), ("C2 hc 10000x128", "hc", 10000, 128),
         ("region 10000x128", "region", 10000, 128), ("mixed 10000x128", "mixed", 10000, 128),
         ("C4 hc 8000x125 (1 M pairs)", "hc", 8000, 125)]


def timed(fn, reps):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return float(np.median(ts))


def main():
    reps = 5
    print("| workload | pairs | cells | fp64 fallback | fp32 kernel, all pairs | fp64 kernel, all pairs | policy, HBM resident "
          "| policy, host buffers (C ABI) |")
    print("|---|---|---|---|---|---|---|---|")
    for name, kind, nr, nh in CASES:
        b = make_batch(kind, nr, nh)
        db = native.DeviceBatch.upload(b)
        out = torch.empty(b.n_pairs, dtype=torch.float64, device="cuda")
        row = {}
        with native.PairHmmContext(record_events=True) as c:
            def run():
                c.compute_device(db, out)
                torch.cuda.synchronize()
            run()
            wall = timed(run, reps)
            st = c.stats()
            row["fb"] = st["n_fallback"] / b.n_pairs
            row["k32"] = b.cells / st["ms_fwd_main"] / 1e6
            row["k32_ms"] = st["ms_fwd_main"]
            row["policy"] = b.cells / wall / 1e9
            row["policy_ms"] = wall * 1e3
        with native.PairHmmContext(use_double=True, record_events=True) as c:
            def run64():
                c.compute_device(db, out)
                torch.cuda.synchronize()
            run64()
            timed(run64, 3)
            st = c.stats()
            row["k64"] = b.cells / st["ms_fwd_main"] / 1e6
            row["k64_ms"] = st["ms_fwd_main"]
        with native.PairHmmContext() as c:
            host_out = np.empty(b.n_pairs)
            c.compute(b, host_out)
            wall = timed(lambda: c.compute(b, host_out), reps)
            row["host"] = b.cells / wall / 1e9
            row["host_ms"] = wall * 1e3
        print(f"| {name} | {b.n_pairs} | {b.cells:.3e} | {row['fb']:.3f} | {row['k32']:.0f} GCUPS ({row['k32_ms']:.3f} ms) | "
              f"{row['k64']:.0f} GCUPS ({row['k64_ms']:.3f} ms) | {row['policy']:.0f} GCUPS ({row['policy_ms']:.3f} ms) | "
              f"{row['host']:.0f} GCUPS ({row['host_ms']:.3f} ms) |", flush=True)


if __name__ == "__main__":
    main()
