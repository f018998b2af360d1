import os, sys, json
sys.path.insert(0, os.getcwd())
import torch
import bench
from gkl_amd import native
from gkl_amd.synth import make_batch
if os.environ.get("PARENT_TORCH", "1") == "1":
    torch.cuda.init(); x = torch.zeros(10, device="cuda")
ctxs = [native.PairHmmContext(device=0) for _ in range(int(os.environ.get("PARENT_CTX", "2")))]
b = make_batch("hc", 100, 10)
for c in ctxs: c.compute(b)
for rep in range(4):
    rec = bench.process_records(0, "hc", counts=tuple(int(x) for x in os.environ.get("COUNTS", "4,8,16").split(",")), duration_s=1.0)
    print(rep, {k: (v["aggregate_gcups"], v["calls"], v["p99_ms"], v["max_ms"], v["longest_child_s"]) for k, v in rec.items() if isinstance(v, dict)}, flush=True)
