#!/usr/bin/env python3
"""PairHMM forward benchmark (BASELINE.json metric): GCUPS + likelihoods/s on the
HaplotypeCaller-shaped 10k-read x 128-haplotype batch, 1..N GPUs of one node.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W      # no launcher: bench.py starts the line above itself (free port)

A "step" is one full pass of the hot path over one batch with the inputs already resident
in HBM: plan -> fp32 forward kernel over all pairs -> precision policy + planning of the fp64
pass -> fp64 recomputation of the underflowed pairs -> log10 finalisation (doubles in HBM), and
for N>1 the gather of every rank's results on rank 0 over RCCL (issued asynchronously: it
overlaps the next step's kernels; all gathers complete inside the timed region).

N>1 is STRONG scaling by default (north_star: "a 10k-read x 128-haplotype batch at 1/2/4/8
GPUs"): the SAME 10k x 128 batch is cut into N contiguous read ranges balanced by cells (the
library's own sharding rule, gklhip_partition_reads), one per rank, haplotypes replicated.
`--weak` gives every rank its own 10k reads instead; `--config4` shards BASELINE config 4
(8000 x 125 = 1 M pairs).

The timed loop never synchronises inside: the host-side planning of step k+1 overlaps the
kernels of step k.  `value` = cells of K steps / wall time of K steps.  The latency of ONE call
(nothing to overlap with) is reported separately under "single_call" and "small_batch".
For N>1 every rank alternates consecutive steps between two contexts on two streams: a shard's
kernels are short, and their fill, drain and last-wave tails plus the planning kernel between the
passes leave holes that the neighbouring step's kernels fill (an eighth of the batch: 1.96 ->
1.73 ms per step; `config.step_overlap`, `--no-overlap` turns it off).  The N=1 line stays
single-stream -- a kernel's event-to-event time must be that of a kernel running alone for the
roofline -- and reports the two-stream rate of the whole batch as `two_callers` (-3 %);
`--overlap` makes it the timed region.  With two streams `kernels_ms` and `roofline` come from
the single-call probe after the timed region (`kernels_ms.from` says which).

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (the fp32 forward
kernel): 12 FLOP per cell x cells per launch / its HIP-event duration,
against the 157.3 TFLOP/s fp32 vector peak (no MFMA applies to a recurrence).
`cpu_baseline` times the reference's own AVX-512/AVX kernels (oracle/_ref, OpenMP dynamic,1
like IntelPairHmm.cc:151-154) on a bounded sample of the same workload on the host cores this
process may use (cgroup quota respected; `cores` = threads actually used).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_CELL = 12.0          # 8 mul + 4 add, avx-pairhmm-template.h:213-222
PEAK_FP32_VECTOR_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 256 flop/clk x 2.4 GHz


def cpu_baseline(batch, budget_s=6.0):
    """Reference kernels on host cores over a bounded read sample (rank 0, N=1 only)."""
    from oracle.oracle import Oracle, Reference
    try:
        eng = Reference()
        kind, run = "reference", lambda b, t: eng.batch(b, n_threads=t)
        isa = "avx512" if eng.engine == 2 else "avx"
    except Exception:  # oracle/_ref missing or no AVX: time our own scalar restatement
        eng = Oracle()
        kind, run = "port", lambda b, t: eng.batch(b, n_threads=t)
        isa = "scalar"
    # threads = the CPUs this process may really use: the cgroup quota when there is one (the GPU boxes expose 256
    # hardware threads but grant 16 CPUs; more threads than that only get throttled)
    avail, quota = _host_threads()
    threads = max(1, min(eng.max_threads(), avail))
    probe = batch.read_slice(0, min(batch.n_reads, 2 * threads))
    t0 = time.time()
    run(probe, threads)
    rate = probe.cells / max(time.time() - t0, 1e-4)
    n = int(min(batch.n_reads, max(2 * threads, rate * budget_s / (batch.cells / batch.n_reads))))
    sample = batch.read_slice(0, n)
    t0 = time.time()
    run(sample, threads)
    dt = time.time() - t0
    one = batch.read_slice(0, min(batch.n_reads, 96))      # also the single-thread rate
    t1 = time.time()
    run(one, 1)
    one_thread = one.cells / max(time.time() - t1, 1e-4) / 1e9
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    return {"value": round(sample.cells / dt / 1e9, 3), "unit": "GCUPS", "cores": threads, "kind": kind,
            "one_thread_gcups": round(one_thread, 3), "cpu_model": model, "host_hw_threads": os.cpu_count(),
            "cgroup_cpu_quota": quota,
            "isa": isa, "sample": f"first {n} reads x {batch.n_haps} haps of the same batch "
            f"({sample.cells:.3e} cells, {dt:.2f} s, fp32+fp64-fallback policy, OpenMP dynamic,1)",
            "likelihoods_per_s": round(sample.n_pairs / dt, 1)}


class _SingleRank:
    """Stands in for torch.distributed when WORLD_SIZE is 1 (no process group)."""
    @staticmethod
    def get_world_size():
        return 1

    @staticmethod
    def get_rank():
        return 0


def _median_ms(fn, calls, warm):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(calls):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e3


def host_call_record(native, batch, dev_index, calls=30, warm=15, max_threads=0):
    """One batch through gklhip_compute (what computeLikelihoodsNative calls after marshalling): host arrays in,
    host doubles out, H2D/D2H and the reference-exact host log10 included.  Latency of back-to-back single calls."""
    out = np.empty(batch.n_pairs)
    with native.PinnedBatch(batch) as pb:   # page-locked marshalling buffers, like the JNI shim's arenas
        with native.PairHmmContext(device=dev_index, max_threads=max_threads) as c:
            ms = _median_ms(lambda: c.compute(pb, out), calls, warm)
        with native.PairHmmContext(device=dev_index, max_threads=max_threads, record_events=True) as c:
            for _ in range(max(3, warm // 3)):
                c.compute(pb, out)
            st = c.stats()
    k = st["ms_fwd_main"] + st["ms_fwd_fallback"]
    return {"ms_per_call": round(ms, 4), "gcups": round(batch.cells / ms / 1e6, 1),
            "kernels_ms": round(k, 4), "over_kernels": round(ms / k - 1.0, 3) if k > 0 else None,
            "pairs": batch.n_pairs, "fallback_fraction": round(st["n_fallback"] / batch.n_pairs, 4)}


def jni_records(batch, c1, host_ms, host_ms_4=None):
    """computeLikelihoodsNative itself, driven through a mock JVM (tests/native/mock_jni.cpp: HotSpot's reference model --
    a jobject is a slot of the thread's handle arena -- with -Xcheck:jni's rules checked on every call; `mock_ns_per_jni_call`
    says what its functions cost, for scale): per call of the bench batch (C2, 36 timed calls after 6 warm ones: median, p10,
    p90) and of a GATK-sized region (C1), with the shim's own split -- marshalling on the calling thread, waiting for compute
    that marshalling did not cover, write-back -- and the aggregate rate of 1 / 4 / 16 concurrent Java threads each sending
    100 x 10 regions through their own slot (GKL_HIP_SLOTS raised to the thread count)."""
    import ctypes as C
    from tests import mockjni
    rec = {}

    def one(b, iters, warm, threads=1, max_threads=1, call_cost_ns=0.0):
        t, calls, k = [], [], []
        rc, _, cls, msg, wall = mockjni.run_concurrent(b, threads, iters=iters, warm=warm, timing=t, max_threads=max_threads, calls=calls, counters=k,
                                                       call_cost_ns=call_cost_ns)
        if rc != 0:
            raise RuntimeError(f"mock JNI run failed: {cls} {msg}")
        return wall, t, max(t[4], 1), calls, k

    def big(max_threads, host, call_cost_ns=0.0, iters=36):
        warm = 6   # (the first pipelined calls of a slot still grow its pinned arenas and start its helper threads)
        wall, t, calls, per_call, k = one(batch, iters, warm, max_threads=max_threads, call_cost_ns=call_cost_ns)
        ms = np.array([c[0] for c in per_call])
        med = float(np.median(ms))
        return {"ms_per_call": round(med, 3), "p10_ms": round(float(np.percentile(ms, 10)), 3), "p90_ms": round(float(np.percentile(ms, 90)), 3),
                "mean_ms": round(wall / iters, 3), "calls": iters, "gcups": round(batch.cells / med / 1e6, 1),
                "marshal_ms": round(t[0] / calls / 1e6, 3), "compute_wait_ms": round(t[1] / calls / 1e6, 3),
                "writeback_ms": round(t[2] / calls / 1e6, 3), "pipelined": bool(t[5]), "max_threads": max_threads,
                "jni_calls_per_read": round(k[mockjni.JNI_CALLS] / (iters + warm) / batch.n_reads, 2),
                "share_of_jni_calls_on_helper_threads": round(k[mockjni.HELPER_JNI_CALLS] / max(1, k[mockjni.JNI_CALLS]), 3),
                "xcheck_violations": k[mockjni.VIOLATIONS],
                "caller_cpus_seen": len({c[2] for c in per_call} | {c[3] for c in per_call}),
                "over_host_path": round(med / host, 3) if host else None}
    rec["c2"] = big(1, host_ms)
    rec["c2_max_threads_4"] = big(4, host_ms_4)
    # the same with every JNI function costing 25 ns more -- the order of a HotSpot function's thread-state transitions, which
    # the mock does not have: 131 000 calls = 3.3 ms more marshalling per call, on one thread or spread over four
    rec["c2_jni_calls_25ns_slower"] = {"max_threads_1": big(1, host_ms, call_cost_ns=25.0, iters=20), "max_threads_4": big(4, host_ms_4, call_cost_ns=25.0, iters=20)}
    wall, t, calls, _, _ = one(c1, 200, 30)
    ms = wall / 200
    rec["c1"] = {"ms_per_call": round(ms, 4), "gcups": round(c1.cells / ms / 1e6, 1),
                 "marshal_ms": round(t[0] / calls / 1e6, 4), "compute_wait_ms": round(t[1] / calls / 1e6, 4),
                 "writeback_ms": round(t[2] / calls / 1e6, 4)}
    try:
        lib = C.CDLL(mockjni.SO)
        lib.mockjni_selfbench.restype = C.c_double
        lib.mockjni_set_call_cost_ns.restype = C.c_double
        rec["mock_ns_per_jni_call"] = round(min(lib.mockjni_selfbench(4000, 150) for _ in range(3)), 1)
        lib.mockjni_set_call_cost_ns(C.c_double(25.0))
        rec["c2_jni_calls_25ns_slower"]["mock_ns_per_jni_call"] = round(min(lib.mockjni_selfbench(4000, 150) for _ in range(3)), 1)
        lib.mockjni_set_call_cost_ns(C.c_double(0.0))
    except (OSError, AttributeError):
        pass
    conc = {}
    os.environ["GKL_HIP_SLOTS"] = "16"
    try:
        from gkl_amd.synth import DEFAULT_SEED, make_batch
        for threads in (1, 4, 16):
            # every thread its own 100 reads (x 10 haplotypes): run_concurrent slices the reads by thread
            b = make_batch("hc", 100 * threads, 10, seed=DEFAULT_SEED)
            iters = 150
            runs = sorted((one(b, iters, 20, threads) for _ in range(3)), key=lambda r: r[0])
            wall, t, calls, _, _ = runs[1]  # the median of three (the aggregate of many short calls moves +-10 % run to run)
            conc[f"callers_{threads}"] = {"aggregate_gcups": round(b.cells * iters / wall / 1e6, 1),
                                          "calls_per_s": round(threads * iters / wall * 1e3, 1),
                                          "ms_per_call": round(t[3] / calls / 1e6, 4),
                                          "best_of_3_gcups": round(b.cells * iters / runs[0][0] / 1e6, 1)}
        try:
            k = (C.c_int64 * 3)()
            C.CDLL(mockjni.JNI_LIB).gklhip_small_call_counts(0, k, 0)
            conc["launched_together"] = {"small_calls": int(k[0]), "combined": int(k[1]), "launch_sets": int(k[2])}
        except (OSError, AttributeError):
            pass
    finally:
        os.environ.pop("GKL_HIP_SLOTS", None)
    rec["note"] = ("through Java_com_intel_gkl_pairhmm_IntelPairHmm_computeLikelihoodsNative with a mock JVM (no JDK exists in this image or on "
                   "the GPU boxes); ms_per_call = MEDIAN of the timed calls; big calls are pipelined (read ranges marshalled -- 13 JNI calls per "
                   "read -- while earlier ranges compute on the slot's two engines); initNative's maxNumberOfThreads caps the host threads per "
                   "engine's log10 pass AND the threads that marshal (the calling thread + helpers attached through the JavaVM): c2 = 1 (the "
                   "reference's default), c2_max_threads_4 = GATK's --native-pair-hmm-threads default")
    return rec, conc


def _host_threads():
    """CPUs this process may really use: the cgroup quota when there is one, else its affinity mask."""
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(avail, quota or avail)), quota


def _timed(fn, calls, warm, kernel_ms=None):
    """(median wall ms, best kernel ms) of `calls` back-to-back calls after `warm` untimed ones."""
    for _ in range(warm):
        fn()
    ts, ks = [], []
    for _ in range(calls):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
        if kernel_ms is not None:
            ks.append(kernel_ms())
    return float(np.median(ts)) * 1e3, (min(ks) if ks else None)


PEAK_FP64_VECTOR_TFLOPS = PEAK_FP32_VECTOR_TFLOPS / 2
PEAK_INT32_TIOPS = 256 * 64 * 2.4e9 / 1e12     # one 32-bit integer operation per lane and clock


def pdhmm_records(dev_index, fixture_x=32):
    """BASELINE config 5 in the driver's line: the reference's own reads x haplotypes fixture (tests/golden/pdhmm_new.txt,
    276 reads x 48 real GATK PD haplotypes) through the three ways IntelPDHMM is called -- `cross`: computeLikelihoodsNative's
    entry point on the fixture's reads x32 against its 48 haplotypes (what fills the chip), `paired`: computePDHMMNative's
    padded 1:1 layout on the same pairs (every pair its own haplotype item, 498 MB of input), `region_276x48_single_call`:
    ONE fixture-sized call, what a GATK region is -- with GKL's own AVX-512 (AVX2) PDHMM kernel on the host cores beside it
    (IntelPDHMM.cc:144-202 runs it under OpenMP).  12 flop per cell (pdhmm.h:427-443), fp64 vector peak."""
    from gkl_amd import native
    from gkl_amd.pdhmm_batch import PdhmmBatch
    from tests.golden_io import load_pdhmm_holders_file
    reads, haps, _ = load_pdhmm_holders_file()
    one = b"\0"
    r1 = PdhmmBatch.from_pairs([(one, one, r[0], r[1], r[2], r[3], r[4]) for r in reads])
    hb = PdhmmBatch.from_pairs([(h[0], h[1], one, one, one, one, one) for h in haps])
    rx = r1.subset(np.tile(np.arange(r1.batch), fixture_x))
    cells1 = int(r1.read_lengths.sum()) * int(hb.hap_lengths.sum())
    cells = cells1 * fixture_x

    def roof(kernel, k_ms, n_cells):
        ach = FLOP_PER_CELL * n_cells / (k_ms * 1e-3) / 1e12
        return {"bound": "mfma", "limiter": "valu-fp64 issue", "kernel": kernel, "flop_per_cell": FLOP_PER_CELL, "achieved": round(ach, 2),
                "peak": PEAK_FP64_VECTOR_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_FP64_VECTOR_TFLOPS, 4), "traffic": None}
    rec = {"data": "the reference's own fixture pdhmm_new.txt (276 reads x 48 PD haplotypes)", "dtype": "f64", "unit": "GCUPS"}
    with native.PdhmmContext(device=dev_index, fma_mode=1) as c:
        ms, k = _timed(lambda: c.compute_cross(rx, hb), 5, 2, c.last_kernel_ms)
        routing = c.last_routing()
        rec["cross"] = {"workload": f"IntelPDHMM.computeLikelihoods: {rx.batch} reads x {hb.batch} haplotypes (the fixture's reads x{fixture_x})",
                        "cells": cells, "kernel_ms": round(k, 4), "kernel_gcups": round(cells / k / 1e6, 1), "host_to_host_ms": round(ms, 3),
                        "gcups": round(cells / ms / 1e6, 1), "haplotypes_by_kernel": {"lds_prior_table": routing[0], "predicate": routing[1], "byte_comparing": routing[2]},
                        "roofline": roof("pdhmm_fwd_tab_kernel", k, cells)}
        ms1, k1 = _timed(lambda: c.compute_cross(r1, hb), 30, 10, c.last_kernel_ms)
        rec["region_276x48_single_call"] = {"workload": "ONE computeLikelihoodsNative call of the fixture: 276 reads x 48 haplotypes = 13 248 pairs",
                                            "cells": cells1, "ms_per_call": round(ms1, 4), "gcups": round(cells1 / ms1 / 1e6, 1),
                                            "kernel_ms": round(k1, 4), "roofline": roof("pdhmm_fwd_tab_kernel", k1, cells1)}
    # ... and through IntelPDHMM.computeLikelihoodsNative itself (mock JVM): marshalling of the 276 + 48 holders included
    try:
        from tests import mockjni
        _, jout, jms, jcalls = mockjni.time_pdhmm_holders(r1, hb, iters=60)
        _, _, jms25, _ = mockjni.time_pdhmm_holders(r1, hb, iters=40, call_cost_ns=25.0)
        rec["jni_region_276x48"] = {"ms_per_call": round(jms, 4), "jni_calls_per_call": round(jcalls, 1), "gcups": round(cells1 / jms / 1e6, 1),
                                    "ms_per_call_with_25ns_more_per_jni_call": round(jms25, 4),
                                    "note": "IntelPDHMM.computeLikelihoodsNative on the fixture's holders through the mock JVM: 13 JNI calls per read, "
                                            "7 per haplotype (r05: 40 and 16), one local frame per 32 holders"}
    except Exception as e:
        rec["jni_region_276x48"] = {"error": repr(e)}
    # the same pairs as padded 1:1 arrays (computePDHMMNative): pair (r, h) = read r with its own copy of haplotype h
    ri, hi = np.repeat(np.arange(r1.batch), hb.batch), np.tile(np.arange(hb.batch), r1.batch)
    hsub, rsub = hb.subset(hi), r1.subset(ri)
    p1 = PdhmmBatch(rsub.batch, hb.max_hap_len, r1.max_read_len, hsub.hap_bases, hsub.hap_pdbases, rsub.read_bases, rsub.read_qual,
                    rsub.read_ins_qual, rsub.read_del_qual, rsub.gcp, hsub.hap_lengths, rsub.read_lengths)
    px = p1.subset(np.tile(np.arange(p1.batch), fixture_x))
    with native.PdhmmContext(device=dev_index, fma_mode=1) as c:
        ms, k_sliced = _timed(lambda: c.compute(px), 4, 2, c.last_kernel_ms)
        routing = c.last_routing()
    os.environ["GKL_HIP_PDHMM_PIPELINE"] = "0"   # the call in one piece: the kernels' own time, without the slices' tails
    try:
        with native.PdhmmContext(device=dev_index, fma_mode=1) as c:
            ms_one, k = _timed(lambda: c.compute(px), 3, 1, c.last_kernel_ms)
    finally:
        os.environ.pop("GKL_HIP_PDHMM_PIPELINE", None)
    rec["paired"] = {"workload": f"IntelPDHMM.computePDHMM: the same {px.batch} (read, haplotype) pairs as padded 1:1 arrays "
                                 f"({px.batch * (2 * px.max_hap_len + 5 * px.max_read_len) / 1e6:.0f} MB of input)",
                     "cells": int(px.cells), "kernel_ms": round(k, 4), "kernel_gcups": round(px.cells / k / 1e6, 1), "kernel_ms_sliced": round(k_sliced, 4),
                     "host_to_host_ms": round(ms, 3), "gcups": round(px.cells / ms / 1e6, 1), "host_to_host_ms_in_one_piece": round(ms_one, 3),
                     "packed_jobs_by_kernel": {"lds_prior_table": routing[0], "predicate": routing[1], "byte_comparing": routing[2]},
                     "roofline": roof("pdhmm_fwd_tab_paired_kernel (+ pdhmm_job_special_kernel)", k, int(px.cells)),
                     "note": "kernel_ms = HIP events around the launches of the call in ONE piece (GKL_HIP_PDHMM_PIPELINE=0); by default a call of "
                             "this size is cut into seven slices whose kernels run while later slices cross PCIe (kernel_ms_sliced, host_to_host_ms)"}
    try:
        from oracle.pdhmm import PdhmmReference
        ref = PdhmmReference()
        eng = 2 if ref.simd_width(2) >= 8 else 1
        threads, quota = _host_threads()
        sample = p1.subset(np.tile(np.arange(p1.batch), 4))
        ref.compute(p1, engine=eng, threads=threads)
        t0 = time.perf_counter()
        st, _ = ref.compute(sample, engine=eng, threads=threads)
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        ref.compute(p1.subset(np.arange(1500)), engine=eng, threads=1)
        one_thread = p1.subset(np.arange(1500)).cells / (time.perf_counter() - t1) / 1e9
        rec["cpu_baseline"] = {"value": round(sample.cells / dt / 1e9, 3), "unit": "GCUPS", "cores": threads, "kind": "reference",
                               "one_thread_gcups": round(one_thread, 3), "isa": "avx512" if eng == 2 else "avx2", "cgroup_cpu_quota": quota,
                               "sample": f"the fixture's {p1.batch} pairs x4 ({sample.cells:.3e} cells, {dt:.2f} s), GKL's own PDHMM kernel under OpenMP"}
    except Exception as e:
        rec["cpu_baseline"] = {"value": None, "unit": "GCUPS", "cores": 0, "kind": "reference", "sample": f"failed: {e}"}
    return rec


def sw_records(dev_index, n_pairs=8192):
    """SURVEY 8 f4 in the driver's line: the batch entry point (gklhip_sw_align_batch) on haplotype-to-reference shaped pairs
    (ref 300-600, alt 250-600, GATK's 200/-150/-260/-11, SOFTCLIP) with GKL's own AVX-512 (AVX2) object -- one pair per call,
    one thread: IntelSmithWaterman.alignNative has no batch and no OpenMP (IntelSmithWaterman.cc:70-124) -- beside it.
    20 integer operations per cell (5 add, 5 max, compare + select for the score, 2 per back-track bit)."""
    from gkl_amd import native
    rng = np.random.RandomState(3)
    letters = np.frombuffer(b"ACGT", np.uint8)
    refs, alts = [], []
    for _ in range(n_pairs):
        L, M = int(rng.randint(300, 601)), int(rng.randint(250, 601))
        ref = letters[rng.randint(0, 4, L)]
        start = int(rng.randint(0, max(1, L - M + 1)))
        alt = ref[start:start + M].copy()
        u = rng.rand(alt.size)
        sub = u < 0.01
        alt[sub] = letters[rng.randint(0, 4, int(sub.sum()))]
        alt = np.delete(alt, np.nonzero((u >= 0.01) & (u < 0.02))[0])
        at = np.nonzero(rng.rand(alt.size) < 0.01)[0]
        alt = np.insert(alt, at, letters[rng.randint(0, 4, at.size)])
        refs.append(ref.tobytes())
        alts.append(alt.tobytes() or b"A")
    params = (200, -150, -260, -11)
    with native.SwContext(device=dev_index) as c:
        pk = c.pack(refs, alts, cigar_stride=256)
        cells = int(pk["cells"])
        ms, k = _timed(lambda: c.align_packed(pk, params, native.SW_SOFTCLIP), 6, 2, c.last_kernel_ms)
    ach = 20 * cells / (k * 1e-3) / 1e12
    rec = {"batch": {"workload": f"{n_pairs} pairs, haplotype-to-reference (ref 300-600, alt 250-600), GATK parameters 200/-150/-260/-11, SOFTCLIP, "
                                 "gklhip_sw_align_batch", "cells": cells, "kernel_ms": round(k, 4), "kernel_gcups": round(cells / k / 1e6, 1),
                     "host_to_host_ms": round(ms, 3), "gcups": round(cells / ms / 1e6, 1), "dtype": "int32", "data": "synthetic",
                     "roofline": {"bound": "mfma", "limiter": "valu-int32 issue", "kernel": "sw_align_kernel", "ops_per_cell": 20, "achieved": round(ach, 2),
                                  "peak": round(PEAK_INT32_TIOPS, 1), "unit": "Tiop/s", "frac": round(ach / PEAK_INT32_TIOPS, 4), "traffic": None}}}
    try:
        from oracle.sw import SwReference
        ref = SwReference()
        eng = 2 if ref.has_avx512() else 1
        sub = list(zip(refs, alts))[:300]
        for r, x in sub[:20]:
            ref.align(r, x, params, native.SW_SOFTCLIP, engine=eng)
        t0 = time.perf_counter()
        for r, x in sub:
            ref.align(r, x, params, native.SW_SOFTCLIP, engine=eng)
        per = (time.perf_counter() - t0) / len(sub)
        sc = sum(len(r) * len(x) for r, x in sub)
        rec["batch"]["cpu_baseline"] = {"value": round(sc / (per * len(sub)) / 1e9, 3), "unit": "GCUPS", "cores": 1, "kind": "reference",
                                        "us_per_call": round(per * 1e6, 1), "isa": "avx512" if eng == 2 else "avx2",
                                        "sample": f"first {len(sub)} pairs of the same batch, one alignNative-style call per pair on one thread "
                                                  "(the reference's entry point has no batch and no OpenMP)"}
    except Exception as e:
        rec["batch"]["cpu_baseline"] = {"value": None, "unit": "GCUPS", "cores": 0, "kind": "reference", "sample": f"failed: {e}"}
    return rec


class _PdhmmRegionCaller:
    """The PDHMM twin of a small_proc_worker's caller: ONE computeLikelihoodsNative-sized call of the reference's fixture
    (276 reads x 48 PD haplotypes) per compute()."""
    def __init__(self, dev_index):
        from gkl_amd import native
        from gkl_amd.pdhmm_batch import PdhmmBatch
        from tests.golden_io import load_pdhmm_holders_file
        reads, haps, _ = load_pdhmm_holders_file()
        one = b"\0"
        self.r = PdhmmBatch.from_pairs([(one, one, r[0], r[1], r[2], r[3], r[4]) for r in reads])
        self.h = PdhmmBatch.from_pairs([(h[0], h[1], one, one, one, one, one) for h in haps])
        self.cells = int(self.r.read_lengths.sum()) * int(self.h.hap_lengths.sum())
        self.ctx = native.PdhmmContext(device=dev_index, fma_mode=1)
        self.last = None

    def compute(self, *_):
        self.last = self.ctx.compute_cross(self.r, self.h)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.ctx.close()


def small_proc_worker(idx, dev_index, workload, duration_s):
    """Child-process mode of `process_records`: ONE caller with its own context (its own process: what a GATK
    HaplotypeCaller JVM is to the GPU), 100 x 10 regions through gklhip_compute back to back (workload "pdhmm": the PDHMM
    fixture's 276 x 48 region through gklhip_pdhmm_compute_cross).  Says "ready" when warm, starts on a line from the parent,
    prints one JSON object."""
    import contextlib
    from gkl_amd import native
    from gkl_amd.synth import DEFAULT_SEED, make_batch
    if workload == "pdhmm":
        b, out, pin_cm, ctx_cm = None, np.zeros(1), contextlib.nullcontext(), _PdhmmRegionCaller(dev_index)
    else:
        b = make_batch(workload, 100, 10, seed=DEFAULT_SEED + 17 * idx)   # every process its own region
        out = np.empty(b.n_pairs)
        pin_cm, ctx_cm = native.PinnedBatch(b), native.PairHmmContext(device=dev_index)
    with pin_cm as pb, ctx_cm as c:
        cells_per_call = int(b.cells) if b is not None else c.cells
        for _ in range(60):
            c.compute(pb, out)
        print("ready", flush=True)
        sys.stdin.readline()
        # What r05's 64-119 ms max_ms was (NOTES 56): every child's ~440th call, at the same moment in all of them -- the
        # Python interpreter's full garbage collection (generation 2 comes due after a fixed number of allocations; with
        # torch imported it walks ~1e6 objects: 35 ms alone, 60-85 ms with 4-16 interpreters doing it at once), not the
        # library and not the device (a C++ loop of the same launches never shows it; rocprofv3 --hip-trace shows a 51 ms
        # gap with no HIP call in it).  The collector is switched off for the timed loop.
        import gc
        gc.collect()
        gc.disable()
        t = time.perf_counter()
        c.compute(pb, out)
        first_ms = (time.perf_counter() - t) * 1e3
        for _ in range(3):
            c.compute(pb, out)
        lat, at = [], []
        t0 = time.perf_counter()
        t_end = t0 + duration_s
        while True:
            t = time.perf_counter()
            if t >= t_end:
                break
            c.compute(pb, out)
            lat.append(time.perf_counter() - t)
            at.append(t - t0)
        elapsed = time.perf_counter() - t0
    lat_a, at_a = np.array(lat), np.array(at)
    slow = np.nonzero(lat_a > 5e-3)[0]
    lat = np.sort(lat_a)
    if b is None:
        out = c.last
    print(json.dumps({"calls": int(lat.size), "elapsed_s": elapsed, "cells_per_call": cells_per_call,
                      "p50_ms": float(lat[lat.size // 2]) * 1e3, "p99_ms": float(lat[min(lat.size - 1, int(lat.size * 0.99))]) * 1e3,
                      "max_ms": float(lat[-1]) * 1e3, "first_call_after_idle_ms": first_ms,
                      # every call slower than 5 ms: (its index, when it started, how long it took)
                      "slow_calls": [[int(i), round(float(at_a[i]) * 1e3, 2), round(float(lat_a[i]) * 1e3, 2)] for i in slow[:8]],
                      "n_slow_calls": int(slow.size),
                      "checksum": float(out.sum())}), flush=True)


def process_records(dev_index, workload, counts=(4, 8, 16), duration_s=1.5):
    """P PROCESSES x one caller on one GPU -- the deployment GATK produces (HaplotypeCaller is one compute thread per JVM,
    scattered over many JVMs per node): each child owns a context and loops 100 x 10 host calls; all start together.
    Aggregate rate, median and 99th-percentile call latency over all children's calls.  The per-process SmallCombiner cannot
    combine across processes: what is measured here is the device's own scheduling of P independent HIP processes."""
    rec = {}
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    for n_proc in counts:
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--small-proc-worker", str(i), "--small-proc-device", str(dev_index),
                                   "--workload", workload, "--small-proc-seconds", str(duration_s)], stdin=subprocess.PIPE,
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for i in range(n_proc)]
        try:
            for p in procs:
                line = p.stdout.readline()
                if line.strip() != "ready":
                    raise RuntimeError(f"a worker process did not come up: {line!r}")
            for p in procs:
                p.stdin.write("go\n")
                p.stdin.flush()
            res = []
            for p in procs:
                out, _ = p.communicate(timeout=120)
                res.append(json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1]))
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        calls = sum(r["calls"] for r in res)
        rec[f"processes_{n_proc}"] = {
            "aggregate_gcups": round(sum(r["calls"] * r["cells_per_call"] / r["elapsed_s"] for r in res) / 1e9, 1),
            "calls_per_s": round(sum(r["calls"] / r["elapsed_s"] for r in res), 1),
            "p50_ms": round(float(np.median([r["p50_ms"] for r in res])), 4),
            "p99_ms": round(float(np.max([r["p99_ms"] for r in res])), 4),
            "max_ms": round(float(np.max([r["max_ms"] for r in res])), 3),
            "first_call_after_idle_ms": round(float(np.max([r["first_call_after_idle_ms"] for r in res])), 3),
            "calls_over_5ms": int(sum(r["n_slow_calls"] for r in res)),
            "slowest_calls": sorted((c for r in res for c in r["slow_calls"]), key=lambda c: -c[2])[:4],   # [index in its child, start ms, ms]
            "longest_child_s": round(float(np.max([r["elapsed_s"] for r in res])), 3),
            "calls": calls, "seconds": duration_s}
    rec["note"] = ("P processes, each ONE caller with its own context looping 100 x 10 regions through gklhip_compute (host arrays in, "
                   "host doubles out), all started together, nothing else on the GPU (measured after the process of the other records has gone); "
                   "p50 = median over processes of their median call, p99 / max_ms = the worst process's, longest_child_s = the longest "
                   "child's timed loop (seconds asked for: a stalled call shows here); the first call of a child after the start signal is reported "
                   "apart (first_call_after_idle_ms: the worst child's), calls_over_5ms / slowest_calls say what else stands out; the children's "
                   "garbage collector is off during the loop (r05's 64-119 ms max_ms was its generation-2 pass, docs/NOTES.md 56)")
    return rec


def in_library_probe(n_dev, reads, haps, workload, steps, warmup):
    """Child-process mode: ONE process drives n_dev devices through a multi-device context (GKL_HIP_DEVICES),
    i.e. what a JVM calling computeLikelihoodsNative gets.  Prints one JSON object."""
    import torch
    from gkl_amd import native
    from gkl_amd.synth import DEFAULT_SEED, make_batch
    b = make_batch(workload, reads, haps, seed=DEFAULT_SEED)
    res = {"devices": n_dev}
    # (GKL_BENCH_SAME_DEVICE=1: dry run on a one-GPU box, every shard on device 0)
    devices = [0] * n_dev if os.environ.get("GKL_BENCH_SAME_DEVICE") == "1" else list(range(n_dev))
    with native.PairHmmContext(devices=devices) as c:
        res["gather"] = c.gather_backend
        db = native.DeviceBatch.upload(b, "cuda:0")
        out = torch.empty(b.n_pairs, dtype=torch.float64, device="cuda:0")
        for _ in range(warmup):
            c.compute_device(db, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            c.compute_device(db, out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res["device_resident"] = {"ms_per_step": round(dt * 1e3, 3), "gcups": round(b.cells / dt / 1e9, 1)}
        res["gather"] = c.gather_backend          # after the calls: what really gathered (RCCL is created by the first one)
        res["gather_note"] = c.gather_note
        with native.PairHmmContext(device=0) as one:
            ref = one.compute_device(db)
            torch.cuda.synchronize()
        res["bit_identical_to_single_device"] = bool(torch.equal(ref, out))
        host_out = np.empty(b.n_pairs)
        ms = _median_ms(lambda: c.compute(b, host_out), 6, 2)
        res["host_path"] = {"ms_per_call": round(ms, 3), "gcups": round(b.cells / ms / 1e6, 1)}
    print(json.dumps(res), flush=True)


def self_launch(n_gpus):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves -- the same
    `torch.distributed.run --nnodes=1 --nproc-per-node N` command the module docstring shows, rendezvous on 127.0.0.1 and a
    port the kernel says is free -- hand the arguments through unchanged and leave with the launcher's exit status (rank 0
    prints the one JSON line on the inherited stdout; a rank that dies or a short-handed group stays a non-zero exit)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL across processes needs dmabuf IPC on these hosts
    env.setdefault("OMP_NUM_THREADS", "4")              # (what the launcher would otherwise set to 1, with a warning)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: --gpus {n_gpus} without a launcher; starting {n_gpus} ranks: {' '.join(cmd[1:9])} bench.py ...", file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="hc", choices=["hc", "region", "mixed", "pdhmm"], help='("pdhmm": only for the hidden per-process worker)')
    ap.add_argument("--reads", type=int, default=10000)
    ap.add_argument("--haps", type=int, default=128)
    ap.add_argument("--double", action="store_true", help="useDoublePrecision (BASELINE config 3)")
    ap.add_argument("--fma-mode", type=int, default=1, choices=[0, 1], help="1 (default): the arithmetic of GKL's AVX-512 object (gcc-contracted FMAs); "
                                                                           "0: of its AVX object (unfused: 12 operations per cell), what GKL computes on a host without AVX-512")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the single_call / small_batch / host_path / no_fallback sub-records")
    ap.add_argument("--weak", action="store_true", help="N>1: every rank owns its own --reads reads (weak scaling) instead of a "
                                                        "read range of the one batch")
    ap.add_argument("--config4", action="store_true", help="BASELINE config 4: ONE 8000 x 125 batch (1 M pairs), sharded")
    ap.add_argument("--strong", action="store_true", help="(default for N>1; kept for compatibility)")
    ap.add_argument("--overlap", action="store_true", help="N=1: two contexts on two streams alternate between consecutive steps (the "
                                                           "default for N>1, where a shard's kernels leave the chip under-filled at "
                                                           "their ends; at N=1 the default line stays single-stream so that the "
                                                           "roofline's kernel durations are those of kernels running alone)")
    ap.add_argument("--no-overlap", action="store_true", help="N>1: one context, one stream per rank")
    ap.add_argument("--dry-comm", action="store_true", help="N>1: only build the process group, run ONE gather of the real sizes and print the "
                                                            "line's `comm` object (who answered, gather time per rank) -- seconds, no kernels")
    ap.add_argument("--in-library-probe", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--jni-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--jni-host-ms", type=float, nargs=2, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--small-proc-worker", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--small-proc-device", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--small-proc-seconds", type=float, default=1.5, help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.jni_worker:
        # child-process mode: computeLikelihoodsNative through the mock JVM in a process of its own -- what a JVM is to the
        # device: the library's streams are the only ones the process has (in the process that measured everything else,
        # torch's and a dozen closed contexts' streams had been dealt onto the hardware queues first, and the slot's engines
        # shared queues: the same call measured 14.8 ms there and 13.3 ms here)
        from gkl_amd.synth import DEFAULT_SEED, make_batch
        b = make_batch(a.workload, a.reads, a.haps, seed=DEFAULT_SEED)
        c1 = make_batch(a.workload, 100, 10, seed=DEFAULT_SEED)
        hm = a.jni_host_ms or [None, None]
        jrec, conc = jni_records(b, c1, hm[0], hm[1])
        print(json.dumps({"jni_path": jrec, "concurrent": conc}), flush=True)
        return
    if a.small_proc_worker >= 0:
        return small_proc_worker(a.small_proc_worker, a.small_proc_device, a.workload, a.small_proc_seconds)
    if a.config4:
        a.reads, a.haps = 8000, 125
    if a.in_library_probe:
        return in_library_probe(a.in_library_probe, a.reads, a.haps, a.workload, a.steps, a.warmup)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        return self_launch(a.gpus)

    # P processes x one caller (small_batch.processes) wants the GPU to itself: an idle process that holds hardware queues of its
    # own (this one, once it has opened the device: torch's and several contexts') is part of what the device's scheduler shares
    # the chip among -- with exactly eight busy children it starves one of them for seconds (docs/NOTES.md 49) -- and measuring
    # the processes FIRST costs the headline 1-4 % (sixteen processes' worth of host activity just before the timed loop).  So
    # the N = 1 run with its extras is two steps: everything that needs this process on the GPU runs in a child (this same
    # command, GKL_BENCH_INNER=1), and when that child is gone the parent -- which never opened the device -- measures the processes
    # and prints the child's line with that record added.
    if (int(os.environ.get("WORLD_SIZE", "1")) == 1 and a.gpus == 1 and not a.no_extras and not a.double and not a.overlap and a.fma_mode == 1
            and os.environ.get("GKL_BENCH_INNER") != "1" and os.environ.get("GKL_BENCH_PROCESSES", "1") != "0"):
        p = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(os.environ, GKL_BENCH_INNER="1"),
                           stdout=subprocess.PIPE, text=True)
        line = next((ln for ln in reversed(p.stdout.splitlines()) if ln.startswith("{")), None)
        if p.returncode != 0 or line is None:
            sys.stdout.write(p.stdout)
            raise SystemExit(p.returncode or 1)
        res = json.loads(line)
        try:
            hp = res.get("host_path", {})
            cmd = [sys.executable, os.path.abspath(__file__), "--jni-worker", "--workload", a.workload, "--reads", str(a.reads), "--haps", str(a.haps)]
            if "max_threads_1" in hp and "max_threads_4" in hp:
                cmd += ["--jni-host-ms", str(hp["max_threads_1"]["ms_per_call"]), str(hp["max_threads_4"]["ms_per_call"])]
            pj = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                                env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
            jl = next((ln for ln in reversed(pj.stdout.splitlines()) if ln.startswith("{")), None)
            if pj.returncode != 0 or jl is None:
                raise RuntimeError((pj.stderr or pj.stdout)[-400:])
            jw = json.loads(jl)
            res["jni_path"] = jw["jni_path"]
            res["jni_path"]["measured_in"] = "a process of its own (like a JVM: the library's streams are the only ones it holds)"
            res.setdefault("small_batch", {})["concurrent"] = jw["concurrent"]
        except Exception as e:
            res["jni_path"] = {"error": repr(e)}
        try:
            rec = process_records(0 if os.environ.get("GKL_BENCH_SAME_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0")), a.workload)
        except Exception as e:
            rec = {"error": repr(e)}
        res.setdefault("small_batch", {})["processes"] = rec
        if isinstance(res.get("pdhmm"), dict) and "error" not in res["pdhmm"]:
            # ... and the same deployment for PDHMM: P processes, each one caller looping fixture-sized regions
            try:
                pr = process_records(0 if os.environ.get("GKL_BENCH_SAME_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0")), "pdhmm",
                                     counts=(4, 8), duration_s=1.0)
                pr["note"] = ("P processes, each ONE caller with its own PDHMM context looping the fixture's 276 x 48 region through "
                              "gklhip_pdhmm_compute_cross (= computeLikelihoodsNative after marshalling), all started together")
            except Exception as e:
                pr = {"error": repr(e)}
            res["pdhmm"]["region_processes"] = pr
        print(json.dumps(res), flush=True)
        return

    import torch
    import torch.distributed as dist
    from gkl_amd import native
    from gkl_amd.shard import PipelinedGather, shard_batch
    from gkl_amd.synth import DEFAULT_SEED, make_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    # GKL_BENCH_SAME_DEVICE=1 + GKL_BENCH_BACKEND=gloo: dry-run of the N>1 code path on a 1-GPU box
    same_device = os.environ.get("GKL_BENCH_SAME_DEVICE") == "1"
    backend = os.environ.get("GKL_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
    dev_index = 0 if same_device else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit(f"--gpus {a.gpus}: rank {rank} wants cuda:{dev_index} but this node shows {torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    comm_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # what the communicator itself says (the line's "comm" object): every rank contributes 1 to an all-reduce
    ranks_seen = 1
    if world > 1:
        one = torch.ones(1, dtype=torch.float64, device=comm_dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(one.item())))
        if dist.get_world_size() != a.gpus or ranks_seen != a.gpus:
            raise SystemExit(f"--gpus {a.gpus} but the process group has {dist.get_world_size()} ranks and {ranks_seen} answered")
    strong = world > 1 and not a.weak
    whole = None
    if a.weak and world > 1:
        # every rank: same haplotypes (seed), its own reads (read_seed)
        batch = make_batch(a.workload, a.reads, a.haps, seed=DEFAULT_SEED, read_seed=DEFAULT_SEED + 1 + rank)
        rows = [a.reads] * world
    else:
        # one global batch, this rank's contiguous read range of it (balanced by cells; shard.partition_reads is the
        # library's gklhip_partition_reads rule, tests/test_multi_device.py pins the two to each other)
        whole = make_batch(a.workload, a.reads, a.haps, seed=DEFAULT_SEED)
        batch, bounds = shard_batch(whole, rank, world)
        rows = [bounds[g + 1] - bounds[g] for g in range(world)]
    if a.dry_comm:
        # the exchange step alone: one gather of the real buffers, timed on every rank, nothing computed
        gather = PipelinedGather(rows, a.haps, comm_dev, dist if world > 1 else _SingleRank(), depth=2)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        gather.buffer(0)
        gather.submit(0)
        gather.finish()
        torch.cuda.synchronize(dev)
        mine = torch.tensor([float(rank), (time.perf_counter() - t0) * 1e3, float(batch.cells), float(batch.n_reads)], dtype=torch.float64, device=comm_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        if world > 1:
            dist.all_gather(allr, mine)
        else:
            allr = [mine]
        if rank == 0:
            cells = [float(t[2].item()) for t in allr]
            print(json.dumps({"comm": {"backend": ("rccl (torch.distributed 'nccl')" if backend == "nccl" else backend) if world > 1 else "none",
                                       "ranks_seen": ranks_seen, "world_size": world, "dry": True,
                                       "gather_bytes_per_rank": [int(r) * a.haps * 8 for r in rows],
                                       "per_rank": [{"rank": int(t[0].item()), "first_gather_ms": round(float(t[1].item()), 3), "cells": int(t[2].item()),
                                                     "reads": int(t[3].item())} for t in allr],
                                       "imbalance": round(max(cells) / (sum(cells) / len(cells)), 4)}}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    dbatch = native.DeviceBatch.upload(batch, dev)
    # record_events=2: kernels are bracketed with HIP events but no call synchronises, so the host-side planning of
    # step k+1 overlaps the kernels of step k; the event times are read after the timed region.
    n_ctx = 2 if (a.overlap or (world > 1 and not a.no_overlap)) else 1
    ctxs = [native.PairHmmContext(use_double=a.double, device=dev_index, record_events=2, fma_mode=a.fma_mode) for _ in range(n_ctx)]
    streams = [torch.cuda.Stream(dev) for _ in range(n_ctx)]
    # N>1: the gather of step k (RCCL, its own stream) overlaps the kernels of step k+1; result buffers rotate
    gather = PipelinedGather(rows, a.haps, comm_dev, dist if world > 1 else _SingleRank(), depth=max(2, n_ctx))
    dev_out = None if comm_dev == dev else [torch.empty(batch.n_pairs, dtype=torch.float64, device=dev) for _ in range(n_ctx)]
    counter = [0]

    def step():
        k = counter[0]
        counter[0] += 1
        i = k % n_ctx
        with torch.cuda.stream(streams[i]):
            out = gather.buffer(k)
            if dev_out is None:
                ctxs[i].compute_device(dbatch, out, streams[i])
            else:  # dry-run backend (gloo): results cross to the host first
                ctxs[i].compute_device(dbatch, dev_out[i], streams[i])
                out.copy_(dev_out[i])
            gather.submit(k)

    def drain():
        gather.finish()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    drain()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain()
    elapsed = time.perf_counter() - t0
    my_elapsed = elapsed          # this rank's own clock around the same K steps (the line's time is the max over ranks)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([float(batch.cells), float(batch.n_pairs)], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total_cells, total_pairs = float(c[0].item()), float(c[1].item())
    else:
        total_cells, total_pairs = float(batch.cells), float(batch.n_pairs)

    # HIP-event times of the timed steps (each context's ring holds its last 64 calls)
    times = []
    for i, c in enumerate(ctxs):
        # the timed steps of context i are its LAST calls
        n_mine = sum(1 for k in range(a.warmup, a.warmup + a.steps) if k % n_ctx == i)
        times += [c.step_times(k) for k in range(min(n_mine, 64))]
    ms_main, ms_fb, ms_dev = ([t[i] for t in times] for i in range(3))
    # Every rank's own numbers, for the line's comm.per_rank: when N = 8 lands under 6x this says whether it was one slow
    # rank (ms_per_step), an uneven split (cells), a slow kernel (fwd_main_ms / fwd_fp64_ms: HIP events of the timed steps;
    # with two contexts alternating they include the other stream's kernels) or everything around the kernels (fixed_cost_ms)
    mine = torch.tensor([float(rank), float(batch.cells), float(batch.n_pairs), float(batch.n_reads), my_elapsed / a.steps * 1e3,
                         float(np.mean(ms_main)) if ms_main else 0.0, float(np.mean(ms_fb)) if ms_fb else 0.0,
                         float(np.mean(ms_dev)) if ms_dev else 0.0], dtype=torch.float64, device=comm_dev)
    per_rank_t = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank_t, mine)
    else:
        per_rank_t = [mine]
    per_rank = []
    for t in per_rank_t:
        v = [float(x) for x in t.tolist()]
        # fixed cost of a rank's step: with one context = step - kernels; with alternating contexts a step's wall time is
        # shorter than its kernels' event-to-event times, so the figure can be negative (the other step's kernels fill the gaps)
        per_rank.append({"rank": int(v[0]), "cells": int(v[1]), "pairs": int(v[2]), "reads": int(v[3]), "ms_per_step": round(v[4], 3),
                         "fwd_main_ms": round(v[5], 3), "fwd_fp64_ms": round(v[6], 3), "device_total_ms": round(v[7], 3),
                         "fixed_cost_ms": round(v[4] - v[5] - v[6], 3), "gcups": round(v[1] / v[4] / 1e6, 1) if v[4] > 0 else None})
    if rank == 0:
        st = ctxs[0].stats()
        with native.PairHmmContext(use_double=a.double, device=dev_index, record_events=1, fma_mode=a.fma_mode) as probe:
            pout = torch.empty(batch.n_pairs, dtype=torch.float64, device=dev)
            for _ in range(2):
                probe.compute_device(dbatch, pout)
            torch.cuda.synchronize(dev)
            # latency of ONE call: nothing to overlap with (synchronised before and after)
            def one_call():
                probe.compute_device(dbatch, pout)
                torch.cuda.synchronize(dev)
            single_ms = _median_ms(one_call, 5, 1)
            pst = probe.stats()
            st["n_fallback"] = pst["n_fallback"]   # needs a synchronising call: outside the timed region
            # cells of the pairs the policy sent to the fp64 pass (flags of the probe call, read back outside the timed region)
            try:
                _, _, used = probe.raw(batch.n_pairs)
                flagged = used.reshape(batch.n_reads, batch.n_haps).astype(np.int64)
                cells_fp64 = int(batch.read_lens.astype(np.int64) @ flagged @ batch.hap_lens.astype(np.int64))
                mix64_cells_per_s = None if a.double else probe.issue_ceiling(use_double=True, ms_budget=30.0)[0]
            except Exception:
                cells_fp64, mix64_cells_per_s = None, None
            # the chip's issue ceiling for the recurrence's bare instruction mix (4 mul + 4 fma per cell, bank-clean
            # operands, four wavefronts per SIMD) and the clock it sustains there: 50 ms, outside the timed region
            try:
                mix_cells_per_s, mix_clock_ghz = probe.issue_ceiling(use_double=a.double, ms_budget=50.0)
            except Exception:
                mix_cells_per_s, mix_clock_ghz = None, None
        k_ms = float(np.mean(ms_main))
        fb_ms = float(np.mean(ms_fb))
        dev_ms = float(np.mean(ms_dev))
        kernel_times_from = "HIP events of the timed steps"
        if n_ctx > 1:
            # consecutive steps share the chip: a kernel's event-to-event time in the timed region includes the other
            # stream's kernels.  The durations of kernels running ALONE come from the probe call above.
            k_ms, fb_ms, dev_ms = pst["ms_fwd_main"], pst["ms_fwd_fallback"], pst["ms_total_device"]
            kernel_times_from = "the single-call probe after the timed region (in it consecutive steps overlap on two streams)"
        step_ms = elapsed / a.steps * 1e3
        achieved = FLOP_PER_CELL * batch.cells / (k_ms * 1e-3) / 1e12
        traffic_profile = None
        clock_profile = None
        pmc = os.path.join(ROOT, "profiles", "latest_pmc.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                clock_profile = round(j["effective_clock_ghz"], 3) if j.get("effective_clock_ghz") else None
                traffic_profile = {"hbm_bytes_per_launch": j.get("hbm_bytes_per_launch"), "profile": j.get("tag", "profiles/latest_pmc.json"),
                                   "note": "from the committed rocprofv3 --pmc passes of this command (profiles/), not measured in this run"}
            except Exception:
                traffic_profile = None
        peak = PEAK_FP32_VECTOR_TFLOPS / 2 if a.double else PEAK_FP32_VECTOR_TFLOPS
        res = {
            "metric": "pairhmm_gcups", "value": round(total_cells * a.steps / elapsed / 1e9, 2), "unit": "GCUPS",
            "likelihoods_per_s": round(total_pairs * a.steps / elapsed, 1),
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(step_ms, 3), "higher_is_better": True,
            "scaling": "weak" if (a.weak and world > 1) else "strong",
            "vs_baseline": None, "dtype": "f64" if a.double else "f32", "data": "synthetic",
            "config": {"workload": f"{a.workload}: {a.reads} reads x {a.haps} haps {'per GPU' if (a.weak and world > 1) else 'in total'} (reads 50-250 bp, haps "
                                   f"100-500 bp), {'fp64 all pairs' if a.double else 'fp32 + fp64 fallback policy'}, "
                                   f"inputs and log10 outputs resident in HBM",
                       "pairs_per_gpu": batch.n_pairs, "cells_per_gpu": batch.cells,   # rank 0's share
                       "fallback_fraction": round(st["n_fallback"] / batch.n_pairs, 4),
                       "parallelism": (f"one batch cut into {world} read ranges balanced by cells, one per rank; gather to rank 0 over RCCL, "
                                       f"overlapped with the next step" if strong else
                                       f"every rank its own reads x{world}; gather to rank 0" if world > 1 else "single GPU"),
                       "step_overlap": f"{n_ctx} contexts on {n_ctx} streams alternate between consecutive steps" if n_ctx > 1 else "none",
                       "finalize": "device log10 in double",
                       "arithmetic": "AVX-512 object's (gcc-contracted FMAs: 8 operations per cell)" if a.fma_mode else "AVX object's (unfused: 12 operations per cell)"},
            # "mfma" = the compute roofline of the bench contract, priced at the dense MFMA peak of the dtype (which
            # for fp32/fp64 equals the vector peak); the kernel itself is vector-ALU code, see `note`
            "roofline": {"bound": "mfma", "limiter": "valu-fp64 issue" if a.double else "valu-fp32 issue",
                         "kernel": "pairhmm_fwd_stream_kernel",
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4),
                         "traffic": None, "traffic_from_profile": traffic_profile,
                         "flop_per_cell": FLOP_PER_CELL, "kernel_ms": round(k_ms, 3),
                         "kernel_gcups": round(batch.cells / k_ms / 1e6, 1),
                         # what the chip issues when NOTHING but the recurrence's 4 mul + 4 fma per cell is in the way
                         # (gklhip_measure_issue_ceiling, measured in this run): the ceiling of any kernel with this arithmetic
                         "issue_ceiling_tflops": round(FLOP_PER_CELL * mix_cells_per_s / 1e12, 2) if mix_cells_per_s else None,
                         "frac_of_issue_ceiling": round(achieved / (FLOP_PER_CELL * mix_cells_per_s / 1e12), 4) if mix_cells_per_s else None,
                         # the same ceiling as cycles per wave64 VALU instruction per SIMD if the chip held its 2.4 GHz peak
                         # clock (8 instructions per cell, 1024 SIMDs x 64 lanes); 2.0 would be the datasheet rate
                         "issue_ceiling_cycles_per_instr_at_2p4ghz": round(1024 * 64 * 2.4e9 / (8 * mix_cells_per_s), 3) if mix_cells_per_s else None,
                         # GRBM_GUI_ACTIVE / 8 / time of this kernel in the committed PMC pass (profiles/<tag>_summary.md
                         # lists it per dispatch): the clock the chip sustains under THIS kernel -- it clocks to its power budget
                         "effective_clock_ghz_from_profile": clock_profile,
                         "note": "compute-bound recurrence priced at the dense fp32 (fp64 with --double) MFMA peak = the vector "
                                 "peak (SIMD-32: one wave64 FMA per 2 cycles); it has no contraction, so it runs on the vector ALUs "
                                 "and issues no MFMA. The recurrence needs 4 mul + 4 fma per cell (1.5 flop per instruction: 0.75 "
                                 "of peak at best) and a SIMD issues one such op per ~2.3 cycles, not 2: issue_ceiling_tflops is "
                                 "that bound, measured in this run (DESIGN.md section 3)"},
            # the second kernel of a step, priced the same way: the packed fp64 recomputation of the flagged pairs
            "fallback_pass": None if (a.double or not cells_fp64 or fb_ms <= 0) else {
                "kernel": "pairhmm_fwd_jobs_kernel<double, 10>", "cells": cells_fp64, "kernel_ms": round(fb_ms, 3),
                "kernel_gcups": round(cells_fp64 / fb_ms / 1e6, 1),
                "achieved": round(FLOP_PER_CELL * cells_fp64 / (fb_ms * 1e-3) / 1e12, 2), "peak": PEAK_FP32_VECTOR_TFLOPS / 2,
                "unit": "TFLOP/s", "frac": round(FLOP_PER_CELL * cells_fp64 / (fb_ms * 1e-3) / 1e12 / (PEAK_FP32_VECTOR_TFLOPS / 2), 4),
                "issue_ceiling_tflops": round(FLOP_PER_CELL * mix64_cells_per_s / 1e12, 2) if mix64_cells_per_s else None,
                "frac_of_issue_ceiling": round(cells_fp64 / (fb_ms * 1e-3) / mix64_cells_per_s, 4) if mix64_cells_per_s else None},
            # who took part in the exchange step (checked above: the run exits non-zero when ranks_seen != --gpus)
            "comm": {"backend": ("rccl (torch.distributed 'nccl')" if backend == "nccl" else backend) if world > 1 else "none",
                     "ranks_seen": ranks_seen, "world_size": dist.get_world_size() if world > 1 else 1,
                     "gather": "dist.gather to rank 0, asynchronous, overlapped with the next step" if world > 1 else "none",
                     "gather_bytes_per_rank": [int(r) * a.haps * 8 for r in rows] if world > 1 else [],
                     # what to read first when N GPUs scale badly: per_rank[*].ms_per_step against the line's
                     # ms_per_step (= the slowest rank), then imbalance (cells: max / mean over ranks; 1.0 = even), then
                     # time_imbalance (ms_per_step: max / mean), then each rank's kernels against its fixed cost
                     "per_rank": per_rank,
                     "imbalance": round(max(r["cells"] for r in per_rank) / (sum(r["cells"] for r in per_rank) / len(per_rank)), 4),
                     "time_imbalance": round(max(r["ms_per_step"] for r in per_rank) / (sum(r["ms_per_step"] for r in per_rank) / len(per_rank)), 4),
                     "in_library_gather": None},   # filled from the in-library child below (N>1)
            "kernels_ms": {"fwd_main": round(k_ms, 3), "fwd_fp64_fallback": round(fb_ms, 3),
                           "device_total": round(dev_ms, 3), "from": kernel_times_from},
            # what a step costs beyond its two forward kernels (planning, policy, log10, launches, gaps); with
            # overlapping steps it can be negative (the other step's kernels fill the gaps)
            "fixed_cost_ms": round(step_ms - k_ms - fb_ms, 3),
            "single_call": {"ms_per_call": round(single_ms, 3), "gcups": round(batch.cells / single_ms / 1e6, 1),
                            "kernels_ms": {"fwd_main": round(pst["ms_fwd_main"], 3), "fwd_fp64_fallback": round(pst["ms_fwd_fallback"], 3)},
                            "fixed_cost_ms": round(single_ms - pst["ms_fwd_main"] - pst["ms_fwd_fallback"], 3),
                            "note": "one device-resident call, synchronised before and after (host planning included)"},
            "plan": {"chunks": st["n_chunks"], "hap_groups": st["n_hap_groups"], "rows_per_lane": st["rows_per_lane"],
                     "lane_fill": round(st["lane_fill"], 4)},
        }
        if world == 1 and not a.no_extras and not a.double and n_ctx == 1 and a.fma_mode == 1:
            try:
                # two callers on one GPU (what the JNI shim's slots give concurrent GATK threads, and what every rank of
                # an N>1 run does): two contexts on two streams alternate between consecutive steps
                c2 = [native.PairHmmContext(device=dev_index, record_events=0) for _ in range(2)]
                s2 = [torch.cuda.Stream(dev) for _ in range(2)]
                o2 = [torch.empty(batch.n_pairs, dtype=torch.float64, device=dev) for _ in range(2)]
                def two(k):
                    with torch.cuda.stream(s2[k % 2]):
                        c2[k % 2].compute_device(dbatch, o2[k % 2], s2[k % 2])
                for k in range(4):
                    two(k)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for k in range(a.steps):
                    two(k)
                torch.cuda.synchronize(dev)
                e2 = time.perf_counter() - t1
                for c in c2:
                    c.close()
                res["two_callers"] = {"ms_per_step": round(e2 / a.steps * 1e3, 3), "gcups": round(batch.cells * a.steps / e2 / 1e9, 1),
                                      "note": "same batch, two contexts on two streams alternating between consecutive steps: the "
                                              "tails of one step's kernels and its planning kernel are filled by the other's"}
            except Exception as e:
                res["two_callers"] = {"error": repr(e)}
            try:
                # end to end through gklhip_compute (H2D, D2H, reference-exact host log10)
                res["host_path"] = {
                    "max_threads_1": host_call_record(native, batch, dev_index, calls=6, warm=2, max_threads=1),
                    "max_threads_4": host_call_record(native, batch, dev_index, calls=6, warm=2, max_threads=4),
                    "max_threads_auto": host_call_record(native, batch, dev_index, calls=6, warm=2, max_threads=0),
                    "note": "gklhip_compute on host arrays = what computeLikelihoodsNative runs after marshalling. "
                            "maxNumberOfThreads is honoured as a cap on the host threads of the reference-exact log10 pass: 1 "
                            "(PairHMMNativeArguments' default) = ONE thread, 4 = GATK HaplotypeCaller's default "
                            "--native-pair-hmm-threads, auto (C ABI max_threads <= 0) = min(cores, 8)"}
                # per-call cost on the sizes GATK and an 8-GPU shard really send
                c1 = make_batch(a.workload, 100, 10, seed=DEFAULT_SEED)
                eighth = whole.read_slice(0, whole.n_reads // 8)
                res["small_batch"] = {"c1_100x10": host_call_record(native, c1, dev_index, calls=60, warm=30),
                                      f"eighth_{eighth.n_reads}x{eighth.n_haps}": host_call_record(native, eighth, dev_index, calls=20, warm=10),
                                      "note": "through gklhip_compute (host threads: the library's own choice, max_threads = 0), back-to-back single calls, "
                                              "median; calls of up to 2048 pairs run fp32 + policy + fp64 of a pair in one wavefront and one launch"}
                # regions with many reads / haplotypes (4k..50k pairs): graded job lengths, two-launch per-pair policy
                for mr, mh in ((400, 40), (250, 128), (1000, 50)):
                    mid = make_batch(a.workload, mr, mh, seed=DEFAULT_SEED)
                    res["small_batch"][f"mid_{mr}x{mh}"] = host_call_record(native, mid, dev_index, calls=30, warm=10)
                try:
                    # the 8-GPU strong-scaling step on one GPU: an eighth of the batch, device-resident, pipelined -- on one
                    # stream, and on two streams through ONE context (the library gives each stream an engine) and through two
                    deighth = native.DeviceBatch.upload(eighth, dev)
                    def shard_steps(ctxs_, streams_, n=200, warm=30):
                        outs_ = [torch.empty(eighth.n_pairs, dtype=torch.float64, device=dev) for _ in streams_]
                        def go(k):
                            with torch.cuda.stream(streams_[k % len(streams_)]):
                                ctxs_[k % len(ctxs_)].compute_device(deighth, outs_[k % len(outs_)], streams_[k % len(streams_)])
                        for k in range(warm):
                            go(k)
                        torch.cuda.synchronize(dev)
                        t1 = time.perf_counter()
                        for k in range(n):
                            go(k)
                        torch.cuda.synchronize(dev)
                        return round((time.perf_counter() - t1) / n * 1e3, 4)
                    ss = [torch.cuda.Stream(dev) for _ in range(2)]
                    with native.PairHmmContext(device=dev_index) as ca, native.PairHmmContext(device=dev_index) as cb:
                        res["small_batch"]["eighth_device_resident"] = {
                            "one_stream_ms_per_step": shard_steps([ca], ss[:1]),
                            "two_streams_one_context_ms_per_step": shard_steps([ca], ss),
                            "two_contexts_ms_per_step": shard_steps([ca, cb], ss),
                            "note": "what one rank of an 8-GPU strong-scaling run does per step (before the gather); bench.py --gpus N "
                                    "alternates two contexts per rank"}
                except Exception as e:
                    res["small_batch"]["eighth_device_resident"] = {"error": repr(e)}
                if os.environ.get("GKL_BENCH_INNER") == "1":
                    res["jni_path"] = {"error": "measured by the parent run in a process of its own (see main)"}
                else:
                    try:
                        jrec, conc = jni_records(batch, c1, res["host_path"]["max_threads_1"]["ms_per_call"],
                                                 res["host_path"]["max_threads_4"]["ms_per_call"])
                        res["jni_path"] = jrec
                        res["jni_path"]["measured_in"] = "the process that measured everything else (torch's and earlier contexts' streams share its hardware queues)"
                        res["small_batch"]["concurrent"] = conc
                    except Exception as e:
                        res["jni_path"] = {"error": repr(e)}
                res["small_batch"]["processes"] = {"error": "measured by the parent run (see main)"}
                # BASELINE config 5 (PDHMM) and SURVEY 8 f4 (Smith-Waterman) in the same driver-run line
                try:
                    res["pdhmm"] = pdhmm_records(dev_index)
                except Exception as e:
                    res["pdhmm"] = {"error": repr(e)}
                try:
                    res["sw"] = sw_records(dev_index)
                except Exception as e:
                    res["sw"] = {"error": repr(e)}
                # reads longer than one wavefront's rows (fp32 > 511 bases, fp64 > 639): workgroups of 2-4 wavefronts per read
                try:
                    lb = make_batch(a.workload, 1000, 32, seed=DEFAULT_SEED, read_len=(600, 1000), hap_len=(900, 1100))
                    dlb = native.DeviceBatch.upload(lb, dev)
                    lout = torch.empty(lb.n_pairs, dtype=torch.float64, device=dev)
                    with native.PairHmmContext(device=dev_index, record_events=1) as lc:
                        def long_call():
                            lc.compute_device(dlb, lout)
                            torch.cuda.synchronize(dev)
                        lms = _median_ms(long_call, 5, 2)
                        lst = lc.stats()
                    res["long_reads"] = {"workload": "1000 reads of 600-1000 bases x 32 haplotypes of 900-1100", "ms_per_call": round(lms, 3),
                                         "gcups": round(lb.cells / lms / 1e6, 1), "fp32_kernel_ms": round(lst["ms_fwd_main"], 3),
                                         "fp32_kernel_gcups": round(lb.cells / lst["ms_fwd_main"] / 1e6, 1),
                                         "fp64_pass_ms": round(lst["ms_fwd_fallback"], 3),
                                         "fallback_fraction": round(lst["n_fallback"] / lb.n_pairs, 4),
                                         "note": "long reads mostly underflow in fp32 (2^120 scaling), so the fp64 pass dominates the call"}
                except Exception as e:
                    res["long_reads"] = {"error": repr(e)}
                # reads of more rows than a workgroup's wavefronts hold (> 2047 bases): super-stripes, the carry row through HBM
                try:
                    lb = make_batch(a.workload, 200, 32, seed=DEFAULT_SEED, read_len=(4000, 6000), hap_len=(5000, 7000))
                    dlb = native.DeviceBatch.upload(lb, dev)
                    lout = torch.empty(lb.n_pairs, dtype=torch.float64, device=dev)
                    with native.PairHmmContext(device=dev_index, record_events=1) as lc:
                        def long5_call():
                            lc.compute_device(dlb, lout)
                            torch.cuda.synchronize(dev)
                        lms = _median_ms(long5_call, 3, 1)
                        lst = lc.stats()
                    res["long_reads_5k"] = {"workload": "200 reads of 4000-6000 bases x 32 haplotypes of 5000-7000", "ms_per_call": round(lms, 3),
                                            "gcups": round(lb.cells / lms / 1e6, 1), "fp32_kernel_ms": round(lst["ms_fwd_main"], 3),
                                            "fp32_kernel_gcups": round(lb.cells / lst["ms_fwd_main"] / 1e6, 1),
                                            "fp64_pass_ms": round(lst["ms_fwd_fallback"], 3),
                                            "fallback_fraction": round(lst["n_fallback"] / lb.n_pairs, 4),
                                            "note": "pairhmm_fwd_super_kernel: super-stripes of 7 compute wavefronts x 512 rows + a helper wavefront that carries the boundary row through HBM"}
                except Exception as e:
                    res["long_reads_5k"] = {"error": repr(e)}
                # the same shape without fallback pairs (one real active region)
                reg = make_batch("region", a.reads, a.haps, seed=DEFAULT_SEED)
                dreg = native.DeviceBatch.upload(reg, dev)
                rout = torch.empty(reg.n_pairs, dtype=torch.float64, device=dev)
                with native.PairHmmContext(device=dev_index, record_events=1) as rc:
                    def reg_call():
                        rc.compute_device(dreg, rout)
                        torch.cuda.synchronize(dev)
                    rms = _median_ms(reg_call, 5, 2)
                    rst = rc.stats()
                res["no_fallback"] = {"workload": f"region: {a.reads} x {a.haps}", "ms_per_call": round(rms, 3),
                                      "gcups": round(reg.cells / rms / 1e6, 1), "kernel_ms": round(rst["ms_fwd_main"], 3),
                                      "kernel_gcups": round(reg.cells / rst["ms_fwd_main"] / 1e6, 1),
                                      "fallback_fraction": round(rst["n_fallback"] / reg.n_pairs, 5)}
            except Exception as e:  # never lose the headline line to a sub-record
                res["extras_error"] = repr(e)
        if world == 1 and not a.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(batch)
            except Exception as e:  # never lose the GPU line to a baseline problem
                res["cpu_baseline"] = {"value": None, "unit": "GCUPS", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e}"}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world > 1 and not a.no_extras:
            # The same batch through the LIBRARY's own multi-device path (one process, GKL_HIP_DEVICES-style context,
            # RCCL gather inside the C ABI) -- what a JVM gets.  After the process group is gone (the other ranks have
            # left their GPUs), in a child process with a time limit: a problem there must not cost the line.
            try:
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--in-library-probe", str(world), "--reads", str(a.reads),
                                    "--haps", str(a.haps), "--workload", a.workload, "--steps", str(min(a.steps, 10)), "--warmup", str(min(a.warmup, 3))],
                                   capture_output=True, text=True, timeout=60,   # (a stuck child must not cost the line: 60 s, then it is killed)
                                   env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
                line = next((ln for ln in reversed(p.stdout.splitlines()) if ln.startswith("{")), None)
                res["in_library"] = json.loads(line) if line else {"error": (p.stderr or p.stdout)[-300:]}
            except Exception as e:
                res["in_library"] = {"error": repr(e)}
            # gklhip_gather_backend of the one-process multi-device context after its calls: "rccl", "peer" or
            # "peer-after-rccl-failure" (+ the devices it drove)
            res["comm"]["in_library_gather"] = {"backend": res["in_library"].get("gather"), "devices": res["in_library"].get("devices"),
                                                "note": res["in_library"].get("gather_note") or res["in_library"].get("error")}
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
