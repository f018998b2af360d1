#!/usr/bin/env python3
"""The JNI layer alone, no device: computeLikelihoodsNative of the C2 batch through the mock JVM with the stub C ABI
(tests/native/stub_gklhip.cpp, arithmetic skipped) -- what marshalling + write-back cost per call on this host, by
number of marshalling threads.  usage: tools/jni_marshal_cpu.py [reads haps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkl_amd.synth import make_batch  # noqa: E402
from tests import mockjni  # noqa: E402

reads, haps = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 128)
b = make_batch("hc", reads, haps)
stub = mockjni.build_stub()
stub.stub_reset()
stub.stub_skip_arithmetic(1)
for mt in (1, 2, 4, 8):
    t, calls, k = [], [], []
    rc, _, cls, msg, wall = mockjni.run_concurrent(b, 1, iters=30, warm=5, max_threads=mt, lib_path=mockjni.STUB_LIB, timing=t, calls=calls, counters=k)
    assert rc == 0, (cls, msg)
    ms = np.array([c[0] for c in calls])
    print(f"max_threads {mt}: call median {np.median(ms):.3f} ms (p10 {np.percentile(ms, 10):.3f}, p90 {np.percentile(ms, 90):.3f}), "
          f"caller marshal {t[0] / t[4] / 1e6:.3f} ms, write-back {t[2] / t[4] / 1e6:.3f} ms, JNI calls per read "
          f"{(k[mockjni.JNI_CALLS]) / 36 / reads:.2f}, helper share {k[mockjni.HELPER_JNI_CALLS] / max(1, k[mockjni.JNI_CALLS]):.2f}")
