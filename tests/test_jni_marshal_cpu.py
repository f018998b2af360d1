"""The JNI layer of libgkl_pairhmm.so WITHOUT a device: the product's jni_shim.cpp linked against a stub of the seven
C-ABI entry points it calls (tests/native/stub_gklhip.cpp: checksums instead of likelihoods), driven by the mock JVM.
What is pinned here: which bytes reach the C ABI and where the results go (one shot, pipelined ranges, helper threads),
the JNI-call budget per read, what -Xcheck:jni would flag, the JavaVM attach/detach discipline of the helper threads,
the retry after a HIP failure, and every exception path -- the reference's counterpart is JavaData.h:65-154 +
IntelPairHmm.cc:125-181."""
import numpy as np
import pytest

from gkl_amd.synth import make_batch
from tests import mockjni
from tests.mockjni import (ATTACHES, DETACHES, FRAMES_PUSHED, GLOBALS_CREATED, GLOBALS_DELETED, HELPER_JNI_CALLS, JNI_CALLS,
                           MAX_LIVE_REFS, REFS_CREATED, REFS_RELEASED, VIOLATIONS)

IAE = "java/lang/IllegalArgumentException"


@pytest.fixture(scope="module")
def stub():
    lib = mockjni.build_stub()
    lib.stub_reset()
    yield lib
    lib.stub_reset()


def run(b, **kw):
    return mockjni.run(b, lib_path=mockjni.STUB_LIB, **kw)


def test_one_shot_call_marshals_every_byte_within_the_call_budget(stub):
    b = make_batch("hc", 100, 10, seed=5)
    rc, out, cls, msg, k = run(b)
    assert rc == 0, (cls, msg)
    assert out.tobytes() == mockjni.stub_expected(b).tobytes()
    assert k[VIOLATIONS] == 0, msg
    assert k[REFS_CREATED] == k[REFS_RELEASED] > 0          # frames popped, nothing left for the return to Java
    # 13 JNI calls per read (holder, 5 fields, 1 length, 5 region copies, 1 exception check), 4 per haplotype, a frame
    # pair per 32 reads / 96 haplotypes, and a handful per call (array lengths, GetFieldID x 6, GetJavaVM, write-back)
    per_read = (k[JNI_CALLS] - 4 * b.n_haps - 2 * k[FRAMES_PUSHED] - 6 - 1 - 4) / b.n_reads
    assert per_read == 13.0, (per_read, k)
    assert k[FRAMES_PUSHED] == -(-b.n_reads // 32) + 1
    assert k[MAX_LIVE_REFS] <= 6 * 32 and k[ATTACHES] == 0 and k[GLOBALS_CREATED] == 0


@pytest.mark.parametrize("max_threads", [1, 4])
def test_pipelined_call_with_and_without_helper_threads(stub, monkeypatch, max_threads):
    b = make_batch("hc", 600, 12, seed=6)
    exp = mockjni.stub_expected(b)
    monkeypatch.setenv("GKL_HIP_JNI_PIPELINE_PAIRS", "1")
    for rp in ("300", "1200", "2400", "7200"):    # 24, 6, 3 ranges, one range
        monkeypatch.setenv("GKL_HIP_JNI_RANGE_PAIRS", rp)
        stub.stub_delay_us(200)                   # a range computes for a while: marshalling and write-back interleave with it
        rc, out, cls, msg, k = run(b, max_threads=max_threads)
        stub.stub_delay_us(0)
        assert rc == 0, (cls, msg)
        assert out.tobytes() == exp.tobytes(), rp
        assert k[VIOLATIONS] == 0, msg
        assert k[REFS_CREATED] == k[REFS_RELEASED]
        if max_threads == 1 or rp == "7200":
            assert k[ATTACHES] == 0 and k[GLOBALS_CREATED] == 0 and k[HELPER_JNI_CALLS] == 0
        else:
            # helpers attach once, work on a global reference to readDataArray, detach when doneNative retires the slot
            assert 1 <= k[ATTACHES] <= 3 and k[ATTACHES] == k[DETACHES], k
            assert k[GLOBALS_CREATED] == k[GLOBALS_DELETED] == 1
    if max_threads > 1:
        # enough reads and calls that the helpers (started by the first call) get to their share of the ranges
        big = make_batch("hc", 6000, 4, seed=7)
        monkeypatch.setenv("GKL_HIP_JNI_RANGE_PAIRS", "1200")
        k = []
        rc, out, cls, msg, _ = mockjni.run_concurrent(big, 1, iters=6, warm=2, max_threads=max_threads, lib_path=mockjni.STUB_LIB, counters=k)
        assert rc == 0 and out.tobytes() == mockjni.stub_expected(big).tobytes() and k[VIOLATIONS] == 0, (cls, msg)
        assert k[HELPER_JNI_CALLS] > 13 * 300, k
        assert k[ATTACHES] == k[DETACHES] == 3 and k[GLOBALS_CREATED] == k[GLOBALS_DELETED] == 8
    # the default schedule (4/12/28/36/14/6; small ranges merged) and explicit shares
    monkeypatch.delenv("GKL_HIP_JNI_RANGE_PAIRS")
    for sh in (None, "6,47,47", "1,1,1,1,96", "100"):
        if sh:
            monkeypatch.setenv("GKL_HIP_JNI_RANGE_SHARES", sh)
        rc, out, cls, msg, k = run(b, max_threads=max_threads)
        assert rc == 0 and out.tobytes() == exp.tobytes() and k[VIOLATIONS] == 0, (sh, cls, msg)


def test_default_range_schedule(stub, monkeypatch):
    """plan_ranges (jni_shim.cpp): 10 000 x 128 -> six ranges (4 / 12 / 28 / 36 / 14 / 6 per cent) with or without helper
    threads; a 2000 x 100 call (200k pairs) -> two ranges (ranges below 40k / 60k pairs are merged with their neighbour)."""
    monkeypatch.setenv("GKL_HIP_JNI_PIPELINE_PAIRS", "1")
    stub.stub_skip_arithmetic(1)
    try:
        n = (mockjni.C.c_long * 4)()
        for reads, haps, mt, ranges in ((10000, 128, 4, 6), (10000, 128, 1, 6), (2000, 100, 4, 2), (2000, 100, 1, 2), (300, 100, 1, 1)):
            b = make_batch("hc", reads, haps, seed=11)
            stub.stub_reset()
            stub.stub_skip_arithmetic(1)
            rc, _, cls, msg, k = run(b, max_threads=mt)
            assert rc == 0 and k[VIOLATIONS] == 0, (cls, msg)
            stub.stub_counts(n)
            assert n[2] == ranges, (reads, haps, mt, n[2])
    finally:
        stub.stub_reset()


def test_helper_threads_serve_consecutive_and_concurrent_calls(stub, monkeypatch):
    monkeypatch.setenv("GKL_HIP_JNI_PIPELINE_PAIRS", "1")
    monkeypatch.setenv("GKL_HIP_JNI_RANGE_PAIRS", "400")
    b = make_batch("hc", 480, 10, seed=8)
    k = []
    rc, out, cls, msg, _ = mockjni.run_concurrent(b, 3, iters=5, max_threads=3, lib_path=mockjni.STUB_LIB, counters=k)
    assert rc == 0, (cls, msg)
    assert out.tobytes() == mockjni.stub_expected(b).tobytes()
    assert k[VIOLATIONS] == 0, msg
    assert 0 < k[ATTACHES] == k[DETACHES] <= 3 * 2          # (three slots, two helpers each: attached once, not per call)
    assert k[GLOBALS_CREATED] == k[GLOBALS_DELETED] == 15


@pytest.mark.parametrize("pipelined", [False, True])
@pytest.mark.parametrize("max_threads", [1, 3])
def test_marshalling_errors_become_illegal_argument_exceptions(stub, monkeypatch, pipelined, max_threads):
    b = make_batch("hc", 200, 6, seed=9)
    if pipelined:
        monkeypatch.setenv("GKL_HIP_JNI_PIPELINE_PAIRS", "1")
        monkeypatch.setenv("GKL_HIP_JNI_RANGE_PAIRS", "150")
    for where in (0, mockjni.LAST_READ_BAD):      # (the last read: in a pipelined call a helper thread may find it)
        for flags, text in ((mockjni.NULL_READQUALS, "null byte[] field"), (mockjni.NULL_READ_ELEMENT, "null element"),
                            (mockjni.SHORT_QUALS, "shorter than readBases")):
            rc, out, cls, msg, k = run(b, flags=flags | where, max_threads=max_threads)
            assert rc == 2 and cls == IAE and text in msg, (flags, where, cls, msg)
            # the exception is raised on the CALLING thread's JNIEnv whoever found the problem; nothing is written
            assert np.all(out == -12345.0) or pipelined
            if flags == mockjni.SHORT_QUALS:
                # the only calls ever made with an exception pending: the region copies of the same read that follow
                # the too-short array (at most three; the reference reads out of bounds instead)
                assert k[VIOLATIONS] <= 3, k
            else:
                assert k[VIOLATIONS] == 0, k
            assert k[REFS_CREATED] == k[REFS_RELEASED] and k[GLOBALS_CREATED] == k[GLOBALS_DELETED] and k[ATTACHES] == k[DETACHES]
    rc, out, cls, msg, _ = run(b, out_len=b.n_pairs - 1)
    assert rc == 2 and cls == IAE and np.all(out == -12345.0)
    rc, out, cls, msg, _ = run(b)
    assert rc == 0 and out.tobytes() == mockjni.stub_expected(b).tobytes()


@pytest.mark.parametrize("pipelined", [False, True])
def test_a_hip_failure_is_retried_once_on_fresh_contexts(stub, monkeypatch, capfd, pipelined):
    """IntelPairHmm.cc:99-113: the reference always has a CPU kernel, so it never fails mid-run; here one failed
    gklhip_compute makes the slot drop its contexts, take a fresh one and run the call again; a second failure in a row
    is the RuntimeException of IntelPairHmm.cc:141-145's convention."""
    b = make_batch("hc", 300, 8, seed=10)
    exp = mockjni.stub_expected(b)
    if pipelined:
        monkeypatch.setenv("GKL_HIP_JNI_PIPELINE_PAIRS", "1")
        monkeypatch.setenv("GKL_HIP_JNI_RANGE_PAIRS", "400")     # six ranges
    stub.stub_reset()
    stub.stub_fail(3 if pipelined else 1, 1)
    rc, out, cls, msg, k = run(b, max_threads=2)
    assert rc == 0, (cls, msg)
    assert out.tobytes() == exp.tobytes() and k[VIOLATIONS] == 0
    err = capfd.readouterr().err
    assert err.count("retrying the call once on a fresh device context") == 1 and "injected fault" in err
    n = (mockjni.C.c_long * 4)()
    stub.stub_counts(n)
    assert n[3] == 0 and n[0] == n[1] == (3 if pipelined else 2)     # contexts: made, dropped, made again; none left
    stub.stub_reset()
    stub.stub_fail(1, 1000)
    rc, out, cls, msg, k = run(b, max_threads=2)
    assert rc == 2 and cls == "java/lang/RuntimeException" and "injected fault" in msg
    assert k[VIOLATIONS] == 0 and k[GLOBALS_CREATED] == k[GLOBALS_DELETED]
    stub.stub_reset()
    rc, out, cls, msg, _ = run(b)
    assert rc == 0 and out.tobytes() == exp.tobytes()            # the library is usable afterwards


def test_an_idle_slot_gives_back_its_second_engine_and_streams(stub, monkeypatch):
    """An idle JVM that once sent a big batch must not keep the device queues of its pipelined engines (the device's
    scheduler rotates every process's queues: docs/NOTES.md 49, 55): after GKL_HIP_IDLE_RELEASE_MS without a call the
    slot's second context and compute threads go and the first is asked to release what it holds for speed; the next
    call simply makes them again."""
    monkeypatch.setenv("GKL_HIP_JNI_PIPELINE_PAIRS", "1")
    monkeypatch.setenv("GKL_HIP_IDLE_RELEASE_MS", "150")
    monkeypatch.setenv("MOCKJNI_PAUSE_MS", "900")
    b = make_batch("hc", 400, 10, seed=12)
    stub.stub_reset()
    # (the janitor thread reads GKL_HIP_IDLE_RELEASE_MS when the process's first initNative starts it: a fresh process)
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from tests import mockjni\nfrom gkl_amd.synth import make_batch\n"
            "lib = mockjni.build_stub(); b = make_batch('hc', 400, 10, seed=12)\n"
            "rc, out, cls, msg, k = mockjni.run(b, lib_path=mockjni.STUB_LIB, flags=mockjni.PAUSE_AND_AGAIN, max_threads=2)\n"
            "n = (mockjni.C.c_long * 4)(); lib.stub_counts(n); lib.stub_releases.restype = mockjni.C.c_long\n"
            "print(rc, (out == mockjni.stub_expected(b)).all(), k[mockjni.VIOLATIONS], n[0], n[1], n[3], lib.stub_releases())\n") % mockjni.ROOT
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-1500:]
    rc, ok, viol, inits, dones, live, releases = p.stdout.split()[-7:]
    assert (rc, ok, viol) == ("0", "True", "0")
    # two contexts for the first call, the second one given back while idle and made again by the second call: 3 made
    assert (int(inits), int(dones), int(live)) == (3, 3, 0) and int(releases) >= 1


def test_mock_flags_what_xcheck_jni_would():
    """The checker itself: a local reference used on another thread, a JNIEnv on a foreign thread and a reference past
    its frame are violations (mock_jni.cpp) -- otherwise `violations == 0` above would mean nothing."""
    import ctypes as C
    mockjni.build()
    lib = C.CDLL(mockjni.SO)
    lib.mockjni_selfcheck.restype = C.c_int
    assert lib.mockjni_selfcheck() == 0b1111


def test_mock_jni_functions_cost_nanoseconds():
    import ctypes as C
    mockjni.build()
    lib = C.CDLL(mockjni.SO)
    lib.mockjni_selfbench.restype = C.c_double
    ns = min(lib.mockjni_selfbench(4000, 150) for _ in range(3))
    assert 0 < ns < 60, ns
