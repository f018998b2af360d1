#!/usr/bin/env python3
"""Dev tool: P processes x one caller of 100 x 10 regions on one GPU (bench.py's process_records) over process counts and
environment settings: python tools/proc_scaling.py "1,2,4,8,16" [NAME=VALUE ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

counts = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8,16").split(","))
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    os.environ[k] = v
rec = bench.process_records(0, "hc", counts=counts, duration_s=1.0)
for n in counts:
    r = rec[f"processes_{n}"]
    print(f"{' '.join(sys.argv[2:]) or 'default'}: {n:2d} processes: {r['aggregate_gcups']:7.1f} GCUPS  {r['calls_per_s']:8.0f} calls/s  p50 {r['p50_ms']:.3f} ms  p99 {r['p99_ms']:.3f} ms")
