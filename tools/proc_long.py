import os, sys
sys.path.insert(0, os.getcwd())
import bench
rec = bench.process_records(0, "hc", counts=(7, 8, 9, 16), duration_s=15.0)
for k, v in rec.items():
    if isinstance(v, dict): print(k, v["aggregate_gcups"], "p99", v["p99_ms"], "max", v["max_ms"], "longest", v["longest_child_s"], "calls", v["calls"], flush=True)
