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
