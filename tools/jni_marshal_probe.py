#!/usr/bin/env python3
"""Where computeLikelihoodsNative's time goes on the C2 batch, call by call (GPU box): per-call wall time with the CPU the
calling thread was on, the shim's marshal / wait / write-back split, by calling-thread placement (scheduler's choice, the
NUMA node of the process's main thread, the other node), by maxNumberOfThreads and by range schedule.  One JSON object per
line on stdout.  usage: tools/jni_marshal_probe.py [placement] [threads] [ranges]   (default: all three)"""
import ctypes as C
import glob
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkl_amd.synth import make_batch  # noqa: E402
from tests import mockjni  # noqa: E402


def cpulist(txt):
    out = []
    for part in txt.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            out += list(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out


nodes = {int(p.rsplit("node", 1)[1].split("/")[0]): cpulist(open(p).read()) for p in glob.glob("/sys/devices/system/node/node*/cpulist")}
libc = C.CDLL(None)
main_cpu = libc.sched_getcpu()
main_node = next((n for n, cpus in nodes.items() if main_cpu in cpus), None)
b = make_batch("hc", 10000, 128)


def measure(tag, iters=30, warm=6, max_threads=1, affinity=None, env=None):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = str(v)
    try:
        t, calls, k, clocks = [], [], [], []
        rc, _, cls, msg, wall = mockjni.run_concurrent(b, 1, iters=iters, warm=warm, max_threads=max_threads, timing=t, calls=calls, counters=k,
                                                       affinity=affinity, clocks=clocks, call_cost_ns=float(os.environ.get("PROBE_CALL_COST_NS", "0")))
    finally:
        for kk, v in old.items():
            if v is None:
                os.environ.pop(kk, None)
            else:
                os.environ[kk] = v
    if rc != 0:
        print(json.dumps({"tag": tag, "error": f"{cls} {msg}"}), flush=True)
        return None
    ms = np.array([c[0] for c in calls])
    cpus = sorted({c[2] for c in calls} | {c[3] for c in calls})
    rec = {"tag": tag, "max_threads": max_threads, "env": env or {}, "median_ms": round(float(np.median(ms)), 3),
           "p10_ms": round(float(np.percentile(ms, 10)), 3), "p90_ms": round(float(np.percentile(ms, 90)), 3), "max_ms": round(float(ms.max()), 3),
           "marshal_ms": round(t[0] / t[4] / 1e6, 3), "wait_ms": round(t[1] / t[4] / 1e6, 3), "writeback_ms": round(t[2] / t[4] / 1e6, 3),
           "core_ghz_before_call": round(float(np.median([c[0] for c in clocks])), 2), "core_ghz_after_call": round(float(np.median([c[1] for c in clocks])), 2),
           "caller_cpus": cpus, "caller_nodes": sorted({n for n, cl in nodes.items() for c in cpus if c in cl}),
           "helper_share_of_jni_calls": round(k[mockjni.HELPER_JNI_CALLS] / max(1, k[mockjni.JNI_CALLS]), 3), "violations": k[mockjni.VIOLATIONS]}
    print(json.dumps(rec), flush=True)
    return rec


what = sys.argv[1:] or ["placement", "threads", "ranges"]
print(json.dumps({"nodes": {n: f"{c[0]}..{c[-1]} ({len(c)})" for n, c in nodes.items()}, "main_cpu": main_cpu, "main_node": main_node,
                  "affinity": len(os.sched_getaffinity(0))}), flush=True)
if "placement" in what:
    for rep in range(4):
        measure(f"scheduler's choice #{rep}")
    if main_node is not None and len(nodes) > 1:
        other = next(n for n in nodes if n != main_node)
        for rep in range(2):
            measure(f"caller on the main thread's node {main_node} #{rep}", affinity=nodes[main_node])
            measure(f"caller on the other node {other} #{rep}", affinity=nodes[other])
        measure("caller on ONE cpu of the main node", affinity=[c for c in nodes[main_node] if c != main_cpu][:1])
if "numa" in what and len(nodes) > 1:
    # the holders are built (first touched) by the main thread inside every measure(): put IT on a node, then the caller on
    # the same / the other one; marshalling on the calling thread only, so marshal_ms is all of it
    everything = sorted(os.sched_getaffinity(0))
    for heap_node in sorted(nodes):
        os.sched_setaffinity(0, nodes[heap_node])
        for caller_node in sorted(nodes):
            for rep in range(3):
                measure(f"holders built on node {heap_node}, caller on node {caller_node} #{rep}", iters=20, warm=4, affinity=nodes[caller_node],
                        env={"GKL_HIP_JNI_MARSHAL_THREADS": 1})
    os.sched_setaffinity(0, everything)
if "clock" in what:
    # is marshal_ms the calling core's clock?  (a thread that works 4 ms of every 15 and sleeps the rest)
    for rep in range(8):
        measure(f"clock #{rep}", iters=20, warm=4, env={"GKL_HIP_JNI_MARSHAL_THREADS": 1})
    for rep in range(4):
        measure(f"clock, caller spins between calls #{rep}", iters=20, warm=4, env={"GKL_HIP_JNI_MARSHAL_THREADS": 1, "MOCKJNI_SPIN_BETWEEN_CALLS_US": 3000})
if "spin" in what:
    # do the library's own waiting threads (two engines in hipStreamSynchronize: HIP spins by default) slow the marshalling thread?
    # (one process per setting: the flag is read when the device is first opened -- run as  GKL_HIP_SCHEDULE=blocking tools/jni_marshal_probe.py spin)
    for rep in range(8):
        measure(f"GKL_HIP_SCHEDULE={os.environ.get('GKL_HIP_SCHEDULE', 'default (spin)')} #{rep}", iters=20, warm=4, env={"GKL_HIP_JNI_MARSHAL_THREADS": 1})
if "cost" in what:
    # schedules against what a JNI function costs (PROBE_CALL_COST_NS: every mock function at least that long)
    lists = ("4,32,32,32", "4,12,28,36,14,6", "3,6,12,24,30,17,8", "2,4,8,16,28,26,12,4")
    for rep in range(2):
        for mt in (1, 4):
            for sh in lists:
                measure(f"cost {os.environ.get('PROBE_CALL_COST_NS', '0')} ns, shares {sh}", max_threads=mt, iters=20, warm=4, env={"GKL_HIP_JNI_RANGE_SHARES": sh})
if "plain" in what:
    # nothing varied: the default call, a few times (compare processes started with different runtime settings, e.g. GPU_MAX_HW_QUEUES=8)
    for rep in range(3):
        for mt in (1, 4):
            measure(f"plain, GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}", max_threads=mt, iters=24, warm=4)
if "shards" in what:
    # ranges of >= 400k pairs are cut in two by the C ABI's twin engines (GKL_HIP_HOST_SHARDS, default 2): good or bad inside a pipelined call?
    for rep in range(3):
        for mt in (1, 4):
            for hs in (2, 1):
                measure(f"GKL_HIP_HOST_SHARDS={hs}", max_threads=mt, iters=24, warm=4, env={"GKL_HIP_HOST_SHARDS": hs})
if "shares" in what:
    lists = ("4,12,28,36,14,6", "6,12,26,36,14,6", "8,14,26,32,14,6", "2,6,14,28,30,14,6", "4,12,28,36,20")
    for rep in range(3):
        for mt in (1, 4):
            for sh in lists:
                measure(f"shares {sh}", max_threads=mt, iters=24, warm=4, env={"GKL_HIP_JNI_RANGE_SHARES": sh})
if "threads" in what:
    for mt in (1, 2, 4, 8):
        measure(f"max_threads {mt}", max_threads=mt)
        measure(f"max_threads {mt}, marshalling on the calling thread only", max_threads=mt, env={"GKL_HIP_JNI_MARSHAL_THREADS": 1})
if "ranges" in what:
    for mt in (1, 4):
        for first in (40000, 80000, 150000):
            for growth in (1.0, 1.5, 2.0):
                for last in (0, 60000):
                    measure("ranges", max_threads=mt, iters=16, warm=4,
                            env={"GKL_HIP_JNI_RANGE_PAIRS": first, "GKL_HIP_JNI_RANGE_GROWTH": growth, "GKL_HIP_JNI_RANGE_LAST": last})
