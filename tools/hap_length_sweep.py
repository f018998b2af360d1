import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from gkl_amd import native
from gkl_amd.synth import make_batch
for hl, nh in (((4000, 4000), 40), ((2000, 2000), 40), ((1000, 1000), 60), ((400, 400), 128), ((200, 200), 256)):
    b = make_batch("hc", 10000, nh, hap_len=hl)
    db = native.DeviceBatch.upload(b)
    with native.PairHmmContext(record_events=True) as c:
        for _ in range(3):
            c.compute_device(db); torch.cuda.synchronize()
        st = c.stats()
        print(f"H={hl[0]} haps={nh} cells {b.cells:.3e} main {st['ms_fwd_main']:.2f} ms -> {b.cells/st['ms_fwd_main']/1e6:.0f} GCUPS  groups {st['n_hap_groups']} chunks {st['n_chunks']} fill {st['lane_fill']:.3f} fallback {st['n_fallback']}")
