#!/usr/bin/env python3
"""Does an IDLE process that once ran a big batch disturb eight busy ones (docs/NOTES.md 49), and does giving its streams
back cure that?  The parent (no torch on the device) makes what a JNI slot holds after a pipelined call -- two contexts
that each ran a 410k-pair host call (own stream + upload / copy / padding streams + a twin engine) -- then stays idle while
P children loop 100 x 10 host calls (bench.process_records).  MODE=hold: as is; MODE=release: gklhip_release_idle on the
first context, the second closed (what the JNI library's janitor does after a second of idleness); MODE=reset: contexts closed and hipDeviceReset(); MODE=close: both contexts closed again (the process has used the device and holds no stream any more); MODE=none: no contexts.
usage: MODE=hold|release|none tools/idle_parent.py [counts]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gkl_amd import native  # noqa: E402
from gkl_amd.synth import make_batch  # noqa: E402

mode = os.environ.get("MODE", "hold")
counts = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8").split(","))
ctxs, released = [], None
if mode != "none":
    big = make_batch("hc", 3200, 128)
    out = np.empty(big.n_pairs)
    ctxs = [native.PairHmmContext(device=0) for _ in range(2)]
    for c in ctxs:
        for _ in range(3):
            c.compute(big, out)
    if mode == "release":
        released = ctxs[0].release_idle()
        ctxs[1].close()
        ctxs = ctxs[:1]
    if mode in ("close", "reset"):      # every context gone: the process keeps the library and the HIP runtime, no stream of its own
        for c in ctxs:
            c.close()
        ctxs = []
    if mode == "reset":      # ... and the runtime's own state on the device torn down (hipDeviceReset: everything this process holds there)
        import ctypes
        rc = ctypes.CDLL("libamdhip64.so").hipDeviceReset()
        released = f"hipDeviceReset -> {rc}"
stop_beat = False
if mode.startswith("heartbeat"):
    # the parent is not quite idle: a GATK-sized call every HEARTBEAT_MS (default 5) on its first context -- is "nine busy ones
    # share fine" (NOTES 49) true of a ninth that is only lightly busy?
    import threading
    small = make_batch("hc", 100, 10)
    sout = np.empty(small.n_pairs)
    period = float(os.environ.get("HEARTBEAT_MS", "5")) / 1e3

    def beat():
        import time
        while not stop_beat:
            ctxs[0].compute(small, sout)
            time.sleep(period)
    threading.Thread(target=beat, daemon=True).start()
for rep in range(int(os.environ.get('REPS', '3'))):
    rec = bench.process_records(0, "hc", counts=counts, duration_s=1.0)
    print(json.dumps({"mode": mode, "rep": rep, "streams_released": released,
                      **{k: {"gcups": v["aggregate_gcups"], "calls": v["calls"], "p99_ms": v["p99_ms"], "max_ms": v["max_ms"], "longest_child_s": v["longest_child_s"]}
                         for k, v in rec.items() if isinstance(v, dict)}}), flush=True)
stop_beat = True
if mode == "release":
    # the context still works, and makes again what it needs
    chk = np.empty(big.n_pairs)
    ctxs[0].compute(big, chk)
    print(json.dumps({"mode": mode, "after": "big call on the trimmed context", "same_bits": bool(np.array_equal(chk, out))}), flush=True)
