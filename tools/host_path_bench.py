#!/usr/bin/env python3
"""Dev tool: time the host-buffer path (gklhip_compute: H2D + kernels + D2H + host log10), i.e. what
the JNI shim pays per call, for DESIGN.md's PCIe-inclusive figure."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkl_amd import native
from gkl_amd.synth import make_batch
b = make_batch("hc")
for threads, fin in ((1, native.FINALIZE_REFERENCE_HOST), (16, native.FINALIZE_REFERENCE_HOST), (1, native.FINALIZE_DEVICE_F64)):
    with native.PairHmmContext(max_threads=threads, finalize=fin) as c:
        out = np.empty(b.n_pairs)
        c.compute(b, out)
        ts = []
        for _ in range(5):
            t = time.time(); c.compute(b, out); ts.append(time.time() - t)
        print(f"host path threads={threads} finalize={fin}: {np.median(ts)*1e3:.2f} ms per 10k x 128 batch -> {b.cells/np.median(ts)/1e9:.0f} GCUPS")

# the JNI symbols through the mock JNIEnv: 1 caller, then 4 concurrent callers on quarter batches
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import mockjni  # noqa: E402
for thr in (1, 2, 4):
    rc, out, cls, msg, wall = mockjni.run_concurrent(b, n_threads=thr, iters=8, max_threads=4)
    assert rc == 0, (cls, msg)
    print(f"JNI shim, {thr} concurrent caller(s) x 8 calls over {b.n_reads // thr} reads x {b.n_haps} haps each: "
          f"{wall:.1f} ms -> {8 * b.cells / wall / 1e6:.0f} GCUPS aggregate")
