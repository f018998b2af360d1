#!/usr/bin/env python3
"""Dev tool: kernel split of the host-buffer path vs the device-resident path on the same batch."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkl_amd import native
from gkl_amd.synth import make_batch
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
b = make_batch("hc", nr, 128)
out = np.empty(b.n_pairs)
for fin in (native.FINALIZE_REFERENCE_HOST, native.FINALIZE_DEVICE_F64):
    with native.PairHmmContext(record_events=True, finalize=fin) as c:
        for _ in range(6):
            t = time.perf_counter(); c.compute(b, out); dt = time.perf_counter() - t
        st = c.stats()
        print(f"host path finalize={fin}: {dt*1e3:.2f} ms main {st['ms_fwd_main']:.3f} fb {st['ms_fwd_fallback']:.3f} dev {st['ms_total_device']:.3f}")
db = native.DeviceBatch.upload(b)
o = torch.empty(b.n_pairs, dtype=torch.float64, device="cuda")
with native.PairHmmContext(record_events=True) as c:
    for _ in range(6):
        t = time.perf_counter(); c.compute_device(db, o); torch.cuda.synchronize(); dt = time.perf_counter() - t
    st = c.stats()
    print(f"device path: {dt*1e3:.2f} ms main {st['ms_fwd_main']:.3f} fb {st['ms_fwd_fallback']:.3f} dev {st['ms_total_device']:.3f}")
