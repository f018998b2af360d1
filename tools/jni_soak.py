#!/usr/bin/env python3
"""Soak of the JNI layer's moving parts together: several Java threads, each with pipelined calls marshalled by helper
threads, while another IntelPairHmm instance keeps calling doneNative / initNative (MOCKJNI_CHURN) and the janitor
releases idle slots every few milliseconds (GKL_HIP_IDLE_RELEASE_MS).  Every round is checked bit for bit (against the
oracle on the GPU; with `stub` against the stub C ABI's checksums, CPU only) and for -Xcheck:jni violations.
usage: tools/jni_soak.py [seconds] [stub]      (run it in a fresh process: the janitor's period is read at the first initNative)"""
import os
import sys
import time

os.environ.setdefault("GKL_HIP_IDLE_RELEASE_MS", "3")
os.environ.setdefault("GKL_HIP_JNI_PIPELINE_PAIRS", "1")
os.environ.setdefault("MOCKJNI_CHURN", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from gkl_amd.synth import make_batch  # noqa: E402
from tests import mockjni  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
stub = len(sys.argv) > 2 and sys.argv[2] == "stub"
lib_path = mockjni.STUB_LIB if stub else mockjni.JNI_LIB
if stub:
    mockjni.build_stub().stub_reset()
    expected = mockjni.stub_expected
else:
    from oracle.oracle import Oracle
    orc = Oracle()
    expected = lambda b: orc.batch(b, n_threads=8)  # noqa: E731
rng = np.random.RandomState(606)
t0, rounds, calls = time.time(), 0, 0
while time.time() - t0 < seconds:
    reads, haps = int(rng.randint(64, 700)), int(rng.randint(2, 20))
    threads, mt = int(rng.randint(1, 6)), int(rng.randint(1, 5))
    os.environ["GKL_HIP_JNI_RANGE_PAIRS"] = str(int(rng.choice([200, 900, 4000, 100000])))
    b = make_batch("hc", reads, haps, seed=int(rng.randint(1 << 30)))
    k = []
    iters = int(rng.randint(1, 6))
    rc, out, cls, msg, _ = mockjni.run_concurrent(b, threads, iters=iters, max_threads=mt, lib_path=lib_path, counters=k)
    assert rc == 0, (rounds, cls, msg)
    assert out.tobytes() == expected(b).tobytes(), ("mismatch", rounds, reads, haps, threads, mt)
    assert k[mockjni.VIOLATIONS] == 0, (rounds, msg)
    assert k[mockjni.ATTACHES] == k[mockjni.DETACHES] and k[mockjni.GLOBALS_CREATED] == k[mockjni.GLOBALS_DELETED], k
    rounds += 1
    calls += threads * iters
print(f"jni soak ({'stub C ABI' if stub else 'GPU'}): {rounds} rounds, {calls} calls in {time.time() - t0:.0f} s, 0 mismatches, 0 violations")
