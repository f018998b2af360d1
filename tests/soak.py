#!/usr/bin/env python3
"""Measurement script (lives under tests/ because it times / checks against oracle/, which only tests may use): 20 s of GATK-sized calls through the host-buffer C ABI checked bit for bit against the oracle, then 300
back-to-back bench-sized batches on the device-resident path (no synchronisation in between)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gkl_amd import native
from gkl_amd.synth import make_batch
from oracle.oracle import Oracle
o = Oracle()
small = [make_batch("hc", r, h, seed=s) for s, (r, h) in enumerate([(5, 3), (60, 8), (130, 10), (300, 16), (17, 1), (1, 9)])]
exp = [o.batch(b, n_threads=4) for b in small]
t0 = time.time()
with native.PairHmmContext(max_threads=2) as c:
    n = 0
    while time.time() - t0 < 20:
        for b, e in zip(small, exp):
            assert c.compute(b).tobytes() == e.tobytes()
            n += 1
print("small-batch soak:", n, "calls bit-identical in", round(time.time() - t0, 1), "s")
big = make_batch("hc", 10000, 128)
db = native.DeviceBatch.upload(big)
out = torch.empty(big.n_pairs, dtype=torch.float64, device="cuda")
with native.PairHmmContext() as c:
    c.compute_device(db, out); torch.cuda.synchronize()
    ref = out.clone()
    t0 = time.time()
    for i in range(300):
        c.compute_device(db, out)
    torch.cuda.synchronize()
    dt = time.time() - t0
    assert torch.equal(out, ref)
print(f"300 back-to-back big batches: {dt/300*1e3:.2f} ms each, results stable")

# round 2: the same through a two-shard context on one GPU (worker threads, peer-copy gather), the host path with twin
# engines (big call), and concurrent JNI callers while another thread keeps calling initNative / doneNative
from tests import mockjni
mid = make_batch("hc", 4000, 128, seed=9)           # > 400 k pairs: twin engines on the host path
emid = o.batch(mid, n_threads=16)
t0 = time.time()
with native.PairHmmContext(devices=[0, 0]) as c2, native.PairHmmContext() as c1:
    dmid = native.DeviceBatch.upload(mid)
    omid = torch.empty(mid.n_pairs, dtype=torch.float64, device="cuda")
    ref1 = c1.compute_device(dmid).clone(); torch.cuda.synchronize()
    n = 0
    while time.time() - t0 < 15:
        for b, e in zip(small, exp):
            assert c2.compute(b).tobytes() == e.tobytes()
        assert c1.compute(mid).tobytes() == emid.tobytes()
        assert c2.compute(mid).tobytes() == emid.tobytes()
        c2.compute_device(dmid, omid); torch.cuda.synchronize()
        assert torch.equal(omid, ref1)
        n += 1
print("multi-shard / twin-engine soak:", n, "rounds bit-identical in", round(time.time() - t0, 1), "s")
b = make_batch("hc", 480, 12, seed=78)
eb = o.batch(b, n_threads=8)
for rep in range(10):
    rc, out_j, cls, msg, _ = mockjni.run_concurrent(b, n_threads=6, iters=6, max_threads=2)
    assert rc == 0 and out_j.tobytes() == eb.tobytes(), (rep, cls, msg)
print("JNI concurrent soak: 10 x 6 threads x 6 calls bit-identical")
