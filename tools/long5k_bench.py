import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from gkl_amd import native
from gkl_amd.synth import DEFAULT_SEED, make_batch
for n, h, rl, hl in ((200, 32, (4000, 6000), (5000, 7000)), (64, 32, (14000, 16000), (15000, 17000)), (2000, 32, (1500, 2000), (2000, 2400))):
    b = make_batch("hc", n, h, seed=DEFAULT_SEED, read_len=rl, hap_len=hl)
    db = native.DeviceBatch.upload(b)
    for dbl in (False, True):
        with native.PairHmmContext(use_double=dbl, record_events=True) as c:
            out = c.compute_device(db); torch.cuda.synchronize()
            ts = []
            for _ in range(4):
                t = time.perf_counter(); c.compute_device(db, out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
            st = c.stats()
        ms = float(np.median(ts)) * 1e3
        print(f"{n}x{h} reads {rl} haps {hl} double={dbl}: {ms:.2f} ms = {b.cells / ms / 1e6:.0f} GCUPS | main {st['ms_fwd_main']:.2f} ({b.cells/st['ms_fwd_main']/1e6:.0f}) fallback {st['ms_fwd_fallback']:.2f}", flush=True)
