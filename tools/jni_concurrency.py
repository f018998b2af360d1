#!/usr/bin/env python3
"""Dev tool: aggregate rate of N concurrent mock-JNI callers, each sending 100 x 10 regions through its own slot."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GKL_HIP_SLOTS", "16")
from gkl_amd.synth import DEFAULT_SEED, make_batch  # noqa: E402
from tests import mockjni  # noqa: E402

def throttled():
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("nr_throttled"):
                return int(line.split()[1])
    except OSError:
        pass
    return -1


for threads in [int(x) for x in (sys.argv[1:] or ["1", "4", "8", "16"])]:
    b = make_batch("hc", 100 * threads, 10, seed=DEFAULT_SEED)
    t = []
    iters = 200
    th0 = throttled()
    rc, _, cls, msg, wall = mockjni.run_concurrent(b, threads, iters=iters, warm=20, timing=t)
    assert rc == 0, (cls, msg)
    calls = max(t[4], 1)
    import ctypes as C
    k = (C.c_int64 * 3)()
    C.CDLL(mockjni.JNI_LIB).gklhip_small_call_counts(0, k, 1)
    print(f"callers {threads:2d}: {b.cells * iters / wall / 1e6:8.1f} GCUPS aggregate, {threads * iters / wall * 1e3:9.1f} calls/s, "
          f"per call {t[3] / calls / 1e6:.3f} ms (marshal {t[0] / calls / 1e6:.3f} compute {t[1] / calls / 1e6:.3f} write {t[2] / calls / 1e6:.4f}), cgroup throttled periods +{throttled() - th0}; small calls {k[0]}, combined {k[1]}, launch sets {k[2]}", flush=True)
