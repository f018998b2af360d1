#!/usr/bin/env python3
"""Dev tool: computeLikelihoodsNative on the 10k x 128 batch through the mock JNIEnv: ms per call and the shim's split."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkl_amd.synth import DEFAULT_SEED, make_batch  # noqa: E402
from tests import mockjni  # noqa: E402
b = make_batch("hc", 10000, 128, seed=DEFAULT_SEED)
t = []
rc, _, cls, msg, wall = mockjni.run_concurrent(b, 1, iters=8, warm=3, timing=t)
assert rc == 0, (cls, msg)
c = max(t[4], 1)
print(f"{os.environ.get('GKL_HIP_JNI_RANGE_PAIRS', 'default')}: {wall / 8:.3f} ms per call (marshal {t[0] / c / 1e6:.2f} wait {t[1] / c / 1e6:.2f} write {t[2] / c / 1e6:.2f}) pipelined {t[5]}")
