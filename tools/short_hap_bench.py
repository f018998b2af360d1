"""Jobs whose haplotypes are no longer than the 64-lane array is deep (several separators in flight): kernel time of the fp32
pass and the whole device-resident call, for the default arrangement and for GKLHIP_ASM_GENERAL=0 (the C++ general steps
that such jobs took before the asm programs learned to look the output column up per lane).
    python tools/short_hap_bench.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from gkl_amd import native
from gkl_amd.synth import make_batch

for hl, nh, rl in (((20, 63), 512, (50, 250)), ((40, 63), 512, (20, 60)), ((30, 120), 384, (50, 250)), ((100, 500), 128, (50, 250))):
    b = make_batch("hc", 10000, nh, hap_len=hl, read_len=rl)
    db = native.DeviceBatch.upload(b)
    row = []
    for env in (None, "0"):
        if env is None:
            os.environ.pop("GKLHIP_ASM_GENERAL", None)
        else:
            os.environ["GKLHIP_ASM_GENERAL"] = env
        with native.PairHmmContext(record_events=True) as c:
            for _ in range(4):
                c.compute_device(db); torch.cuda.synchronize()
            st = c.stats()
            row.append((st["ms_fwd_main"], st["ms_fwd_fallback"]))
    os.environ.pop("GKLHIP_ASM_GENERAL", None)
    print(f"haps {hl} x {nh}, reads {rl}: cells {b.cells:.3e}  fp32 pass {row[0][0]:.2f} ms = {b.cells / row[0][0] / 1e6:.0f} GCUPS "
          f"(C++ general steps: {row[1][0]:.2f} ms = {b.cells / row[1][0] / 1e6:.0f});  fp64 pass {row[0][1]:.2f} ms ({row[1][1]:.2f})", flush=True)
