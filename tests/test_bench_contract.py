"""bench.py's output contract, on a small batch: the one JSON line with every field the driver and the judge read,
at N=1 (single stream, roofline from the timed region's HIP events) and for a 2-rank launch (both ranks on the one
test GPU, gloo standing in for RCCL: the strong-scaling shard, the two-stream step overlap, the in-library probe)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "kernels_ms", "fixed_cost_ms", "single_call")


def _line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_single_gpu():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "600", "--haps", "24", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    for k in REQUIRED + ("cpu_baseline", "host_path", "small_batch", "no_fallback", "two_callers"):
        assert k in d, k
    assert d["metric"] == "pairhmm_gcups" and d["unit"] == "GCUPS" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert d["config"]["step_overlap"] == "none" and "model" not in d["config"] and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "TFLOP/s" and r["traffic"] is None and "traffic_from_profile" in r
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    # the dominant kernel's HIP-event time is part of the step it was measured in
    assert d["kernels_ms"]["from"].startswith("HIP events") and d["kernels_ms"]["fwd_main"] <= d["ms_per_step"] * 1.05
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert "error" not in d["two_callers"] and d["two_callers"]["ms_per_step"] > 0
    assert "extras_error" not in d
    # the ceiling in numbers: issue rate of the recurrence's bare instruction mix, measured in the same run
    assert r["issue_ceiling_tflops"] > r["achieved"] and abs(r["frac_of_issue_ceiling"] - r["achieved"] / r["issue_ceiling_tflops"]) < 2e-3
    assert 1.9 < r["issue_ceiling_cycles_per_instr_at_2p4ghz"] < 4.0
    fp = d["fallback_pass"]
    assert fp and 0 < fp["frac"] < 1 and fp["cells"] > 0 and fp["issue_ceiling_tflops"] > fp["achieved"]
    # computeLikelihoodsNative itself (mock JNIEnv): per-call time and its split; concurrent small callers
    j = d["jni_path"]
    assert "error" not in j, j
    for k in ("c2", "c2_max_threads_4", "c1"):
        assert j[k]["ms_per_call"] > 0 and j[k]["marshal_ms"] >= 0 and j[k]["compute_wait_ms"] >= 0 and j[k]["writeback_ms"] >= 0
    for k in ("c2", "c2_max_threads_4"):
        # median of >= 30 calls with its spread, the JNI-call budget per read, nothing -Xcheck:jni would flag
        assert j[k]["calls"] >= 30 and j[k]["p10_ms"] <= j[k]["ms_per_call"] <= j[k]["p90_ms"]
        assert j[k]["jni_calls_per_read"] <= 14 and j[k]["xcheck_violations"] == 0
    assert 0 < j["mock_ns_per_jni_call"] < 100
    # ... and what a JVM's slower JNI functions (+25 ns each) would do to the same call: still pipelined behind the kernels
    slow = j["c2_jni_calls_25ns_slower"]
    assert slow["mock_ns_per_jni_call"] >= 25 and slow["max_threads_1"]["ms_per_call"] > 0 and slow["max_threads_4"]["ms_per_call"] > 0
    assert slow["max_threads_1"]["marshal_ms"] > j["c2"]["marshal_ms"]
    # BASELINE config 5 (PDHMM) and SURVEY 8 f4 (Smith-Waterman) in the same line, each with its kernel time, a roofline
    # fraction that follows from it, and the reference's own kernel on the host beside it
    pd = d["pdhmm"]
    assert "error" not in pd, pd
    for k in ("cross", "paired", "region_276x48_single_call"):
        r = pd[k]["roofline"]
        assert pd[k]["kernel_ms"] > 0 and pd[k]["cells"] > 0
        assert abs(r["achieved"] - 12 * pd[k]["cells"] / pd[k]["kernel_ms"] / 1e9) < 0.02 * r["achieved"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert pd["cross"]["cells"] == pd["paired"]["cells"] == 32 * pd["region_276x48_single_call"]["cells"]
    assert pd["cpu_baseline"]["kind"] == "reference" and pd["cpu_baseline"]["value"] > 0 and pd["cpu_baseline"]["cores"] >= 1
    # the region call through IntelPDHMM.computeLikelihoodsNative itself: 13 JNI calls per read + 7 per haplotype + a few
    jr = pd["jni_region_276x48"]
    assert "error" not in jr, jr
    assert jr["ms_per_call"] > 0 and jr["jni_calls_per_call"] < 276 * 13 + 48 * 7 + 40 and jr["ms_per_call_with_25ns_more_per_jni_call"] > jr["ms_per_call"]
    # GATK calls PDHMM per region from many JVMs, like PairHMM: P processes x one caller of fixture-sized regions
    rp = pd["region_processes"]
    assert "error" not in rp, rp
    assert all(rp[f"processes_{n}"]["aggregate_gcups"] > 0 and rp[f"processes_{n}"]["p50_ms"] > 0 for n in (4, 8))
    sw = d["sw"]
    assert "error" not in sw, sw
    r = sw["batch"]["roofline"]
    assert sw["batch"]["kernel_ms"] > 0 and abs(r["achieved"] - 20 * sw["batch"]["cells"] / sw["batch"]["kernel_ms"] / 1e9) < 0.02 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and sw["batch"]["cpu_baseline"]["kind"] == "reference" and sw["batch"]["cpu_baseline"]["value"] > 0
    # maxNumberOfThreads is a cap on the host log10 threads: the line says what each setting costs
    assert all(d["host_path"][k]["ms_per_call"] > 0 for k in ("max_threads_1", "max_threads_4", "max_threads_auto"))
    assert d["comm"]["ranks_seen"] == 1 and d["comm"]["backend"] == "none" and d["comm"]["gather_bytes_per_rank"] == []
    assert len(d["comm"]["per_rank"]) == 1 and d["comm"]["imbalance"] == 1.0 and d["comm"]["per_rank"][0]["cells"] == d["config"]["cells_per_gpu"]
    conc = d["small_batch"]["concurrent"]
    assert all(conc[f"callers_{n}"]["aggregate_gcups"] > 0 for n in (1, 4, 16))
    # the deployment GATK produces: P processes x one caller on the one GPU
    pr = d["small_batch"]["processes"]
    assert "error" not in pr, pr
    for n in (4, 8, 16):
        assert pr[f"processes_{n}"]["aggregate_gcups"] > 0 and 0 < pr[f"processes_{n}"]["p50_ms"] <= pr[f"processes_{n}"]["p99_ms"]
        assert "first_call_after_idle_ms" in pr[f"processes_{n}"] and "calls_over_5ms" in pr[f"processes_{n}"]
    er = d["small_batch"]["eighth_device_resident"]
    assert "error" not in er and er["two_streams_one_context_ms_per_step"] > 0


@pytest.mark.gpu
def test_bench_line_two_ranks_on_one_gpu():
    env = dict(os.environ, GKL_BENCH_SAME_DEVICE="1", GKL_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", "600", "--haps", "24",
                        "--steps", "4", "--warmup", "2"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["step_overlap"].startswith("2 contexts on 2 streams")
    assert d["kernels_ms"]["from"].startswith("the single-call probe")
    assert 0 < d["roofline"]["frac"] < 1
    # the same batch through the library's own multi-device context (both "devices" = the one GPU: peer copies)
    lib = d["in_library"]
    assert "error" not in lib, lib
    assert lib["devices"] == 2 and lib["bit_identical_to_single_device"] is True
    # the line says who took part in the exchange step; a short-handed group is a non-zero exit, not a smaller number
    c = d["comm"]
    assert c["backend"] == "gloo" and c["ranks_seen"] == 2 and c["world_size"] == 2
    assert len(c["gather_bytes_per_rank"]) == 2 and sum(c["gather_bytes_per_rank"]) == 600 * 24 * 8
    assert c["in_library_gather"]["backend"] == "peer" and c["in_library_gather"]["devices"] == 2
    # every rank's own numbers: the first thing to read when N GPUs land under the expected scaling
    assert [r["rank"] for r in c["per_rank"]] == [0, 1] and c["imbalance"] < 1.1 and c["time_imbalance"] >= 1.0
    for r in c["per_rank"]:
        assert r["cells"] > 0 and r["ms_per_step"] > 0 and r["fwd_main_ms"] > 0 and r["reads"] > 0
        assert r["ms_per_step"] <= d["ms_per_step"] * 1.001          # the line's time is the slowest rank's
    assert sum(r["reads"] for r in c["per_rank"]) == 600


@pytest.mark.gpu
def test_bench_c2_split_is_balanced_and_dry_comm_builds_the_group_only():
    """The C2 batch cut in two (the library's partition rule): cells within 10 % of each other; `--dry-comm` only builds
    the group, runs one gather of the real sizes and prints `comm` -- the first thing to run on a new multi-GPU node."""
    env = dict(os.environ, GKL_BENCH_SAME_DEVICE="1", GKL_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29535", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-comm"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    c = _line(p.stdout)["comm"]
    assert c["dry"] is True and c["ranks_seen"] == 2 and len(c["per_rank"]) == 2 and c["imbalance"] < 1.1
    assert sum(r["reads"] for r in c["per_rank"]) == 10000 and sum(c["gather_bytes_per_rank"]) == 10000 * 128 * 8
    assert all(r["first_gather_ms"] > 0 for r in c["per_rank"])


def test_bench_without_a_launcher_starts_the_launcher_cpu():
    """No GPU here: `python bench.py --gpus 2` must still get as far as starting its two ranks under torch.distributed.run
    (each then refuses to run without an MI355X) and hand their failure back as a non-zero exit with no JSON line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("the GPU twin of this test runs the real thing")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", "50", "--haps", "4", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "starting 2 ranks" in p.stderr and "torch.distributed.run" in p.stderr
    assert p.stderr.count("needs an MI355X") >= 1, p.stderr[-1500:]


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` as the driver types it (no torch.distributed.run around it, WORLD_SIZE unset): bench.py
    starts the two ranks itself on a free port and rank 0 prints the one line, both ranks in the exchange step."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(GKL_BENCH_SAME_DEVICE="1", GKL_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", "600", "--haps", "24", "--steps", "4",
                        "--warmup", "2", "--no-extras"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["steps"] == 4 and d["warmup"] == 2
    c = d["comm"]
    assert c["ranks_seen"] == 2 and c["world_size"] == 2 and len(c["gather_bytes_per_rank"]) == 2
    # a node with fewer GPUs than ranks is a non-zero exit with no line, not a smaller number (no SAME_DEVICE here)
    env.pop("GKL_BENCH_SAME_DEVICE")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", "200", "--haps", "8", "--steps", "1",
                        "--warmup", "0", "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    import torch
    if torch.cuda.device_count() < 2:
        assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_bench_refuses_a_world_size_that_is_not_gpus():
    env = dict(os.environ, GKL_BENCH_SAME_DEVICE="1", GKL_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29534", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--reads", "200", "--haps", "8",
                        "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
