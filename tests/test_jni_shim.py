"""The JNI drop-in (libgkl_pairhmm.so) driven through a mock JNIEnv -- BASELINE config 1
("JNI plumbing").  CPU tests pin the symbol surface and the exception mapping; the GPU tests
run the reference's own test vectors through the full JNI path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gkl_amd.batch import FlatBatch, HaplotypeDataHolder, ReadDataHolder
from gkl_amd.synth import make_batch
from tests import mockjni

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def small_batch():
    return make_batch("hc", 6, 3, seed=21)


def test_jni_library_exports_the_three_natives():
    hdr = open(os.path.join(ROOT, "include", "gkl_pairhmm_jni.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(Java_com_intel_gkl_pairhmm_IntelPairHmm_\w+)\s*\(", hdr)))
    assert declared == ["Java_com_intel_gkl_pairhmm_IntelPairHmm_computeLikelihoodsNative",
                        "Java_com_intel_gkl_pairhmm_IntelPairHmm_doneNative",
                        "Java_com_intel_gkl_pairhmm_IntelPairHmm_initNative"]
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(mockjni.JNI_LIB)
    for s in declared:
        assert hasattr(lib, s)


def test_jni_slot_indices_follow_the_spec():
    """The clean-room table must use the JNI specification's indices."""
    txt = open(os.path.join(ROOT, "gkl_amd", "csrc", "jni_min.h")).read()
    got = dict(re.findall(r"kJniSlot(\w+) = (\d+)", txt))
    spec = dict(FindClass=6, ThrowNew=14, ExceptionClear=17, PushLocalFrame=19, PopLocalFrame=20, NewGlobalRef=21,
                DeleteGlobalRef=22, DeleteLocalRef=23, GetFieldID=94,
                GetObjectField=95, GetArrayLength=171, GetObjectArrayElement=173,
                GetByteArrayRegion=200, SetDoubleArrayRegion=214, GetJavaVM=219, ExceptionCheck=228)
    for k, v in spec.items():
        assert int(got[k]) == v, k
    # ... and the invocation interface's (JNIInvokeInterface: three reserved slots, DestroyJavaVM, AttachCurrentThread, ...)
    jvm = dict(re.findall(r"kJvmSlot(\w+) = (\d+)", txt))
    assert (int(jvm["DetachCurrentThread"]), int(jvm["GetEnv"]), int(jvm["AttachCurrentThreadAsDaemon"]), int(jvm["Count"])) == (5, 6, 7, 8)


def test_missing_field_is_illegal_argument():
    # JavaData.h:127-133: GetFieldID failure -> IllegalArgumentException("Unable to get field ID")
    rc, _, cls, msg, _ = mockjni.run(small_batch(), flags=mockjni.DROP_GCP_FIELD)
    assert rc == 1 and cls == "java/lang/IllegalArgumentException" and msg == "Unable to get field ID"


def test_compute_before_init_raises():
    rc, _, cls, msg, _ = mockjni.run(small_batch(), flags=mockjni.SKIP_INIT)
    assert rc == 2 and cls == "java/lang/RuntimeException" and "initNative" in msg


def test_no_gpu_means_runtime_exception_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    rc, out, cls, msg, _ = mockjni.run(small_batch())
    assert rc == 1 and cls == "java/lang/RuntimeException" and "device" in msg.lower()
    assert np.all(out == -12345.0)  # nothing was computed behind our back


@pytest.mark.gpu
def test_jni_full_path_matches_oracle_bit_for_bit(oracle):
    b = make_batch("hc", 40, 8, seed=33)
    for use_double in (False, True):
        rc, out, cls, msg, refs = mockjni.run(b, use_double=use_double)
        assert rc == 0, (cls, msg)
        exp = oracle.batch(b, use_double=use_double, n_threads=8)
        assert out.tobytes() == exp.tobytes()
        assert refs[0] == refs[1] > 0  # every local ref handed out was released (frames popped) before the return to Java
        # what the reference's test JVMs check with -Xcheck:jni (build.gradle:101-104): no JNI call with an exception
        # pending, no use of a dead reference, never more local references alive than the frame was pushed for
        assert refs[2] == 0, msg
        assert refs[3] <= 6 * 32


def test_shims_compile_against_a_specification_shaped_jni_header():
    # the real-JDK path (GKL_USE_SYSTEM_JNI) of all four shims: distinct jobject/jclass/jbyteArray... classes and the
    # struct-of-function-pointers table, at least type-checked here (no JDK in the image)
    mockjni.typecheck_sysjni()


@pytest.mark.gpu
def test_system_jni_build_of_the_shim_runs_bit_exact(oracle):
    mockjni.build_sysjni()
    b = make_batch("hc", 40, 8, seed=35)
    rc, out, cls, msg, refs = mockjni.run(b, lib_path=mockjni.SYSJNI_LIB)
    assert rc == 0, (cls, msg)
    assert out.tobytes() == oracle.batch(b, n_threads=8).tobytes()
    assert refs[0] == refs[1] > 0 and refs[2] == 0 and refs[3] <= 6 * 32


@pytest.mark.gpu
def test_jni_lifecycle_follows_the_reference(oracle):
    """initNative only re-sets globals and doneNative is empty in the reference (IntelPairHmm.cc:70-116,189-192):
    calling initNative again (same or other arguments) or computing after doneNative must keep working."""
    b = make_batch("hc", 30, 6, seed=34)
    exp = oracle.batch(b, n_threads=4)
    for flags in (mockjni.REINIT_TWICE, mockjni.COMPUTE_AFTER_DONE, mockjni.REINIT_TWICE | mockjni.COMPUTE_AFTER_DONE):
        rc, out, cls, msg, refs = mockjni.run(b, flags=flags)
        assert rc == 0, (cls, msg)
        assert out.tobytes() == exp.tobytes()
        assert refs[2] == 0, msg


@pytest.mark.gpu
def test_jni_golden_file(golden_cases):
    # dataFileTest (PairHmmUnitTest.java:171-234): 1 read x 1 hap per call, abs tol 1e-5
    for c in golden_cases:
        b = FlatBatch.from_holders([ReadDataHolder(c["read"], c["q"], c["i"], c["d"], c["c"])],
                                   [HaplotypeDataHolder(c["hap"])])
        for use_double in (False, True):
            rc, out, cls, msg, _ = mockjni.run(b, use_double=use_double)
            assert rc == 0, (cls, msg)
            assert abs(out[0] - c["expected"]) <= 1e-5


@pytest.mark.gpu
def test_jni_argument_errors():
    b = small_batch()
    rc, _, cls, msg, _ = mockjni.run(b, flags=mockjni.NULL_READQUALS)
    assert rc == 2 and cls == "java/lang/IllegalArgumentException"
    rc, _, cls, msg, _ = mockjni.run(b, flags=mockjni.SHORT_QUALS)
    assert rc == 2 and cls == "java/lang/IllegalArgumentException"
    rc, _, cls, msg, refs = mockjni.run(b, flags=mockjni.NULL_READ_ELEMENT)
    assert rc == 2 and cls == "java/lang/IllegalArgumentException"
    assert refs[2] == 0 and refs[3] <= 6 * 32   # the error paths, too, make no JNI call with the exception pending
    rc, out, cls, msg, _ = mockjni.run(b, out_len=b.n_pairs - 1)
    assert rc == 2 and cls == "java/lang/IllegalArgumentException" and np.all(out == -12345.0)


@pytest.mark.gpu
@pytest.mark.parametrize("n_threads,slots", [(6, None), (3, "1")])
def test_jni_concurrent_callers_get_their_own_slot(oracle, n_threads, slots, monkeypatch):
    # computeLikelihoodsNative is re-entrant in the reference (SURVEY 8b "Threading"); here every concurrent
    # caller leases a slot (context + stream + pinned arenas; GKL_HIP_SLOTS bounds them, callers queue beyond it)
    if slots:
        monkeypatch.setenv("GKL_HIP_SLOTS", slots)
    b = make_batch("hc", 240, 12, seed=77)
    rc, out, cls, msg, _ = mockjni.run_concurrent(b, n_threads=n_threads, iters=4, max_threads=2)
    assert rc == 0, (cls, msg)
    assert out.tobytes() == oracle.batch(b, n_threads=8).tobytes()


@pytest.mark.gpu
def test_jni_init_and_done_of_another_instance_do_not_disturb_running_calls(oracle, monkeypatch):
    # closing one IntelPairHmm instance (doneNative) or initialising another must not break the calls other
    # threads have in flight -- the reference's doneNative is empty and its initNative only re-sets globals
    monkeypatch.setenv("MOCKJNI_CHURN", "1")
    b = make_batch("hc", 360, 12, seed=79)
    rc, out, cls, msg, _ = mockjni.run_concurrent(b, n_threads=5, iters=8, max_threads=2)
    assert rc == 0, (cls, msg)
    assert out.tobytes() == oracle.batch(b, n_threads=8).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("pipelined", [False, True])
def test_jni_retries_a_failed_call_once_on_a_fresh_context(oracle, monkeypatch, capfd, pipelined):
    """The reference never fails mid-run (it always has a CPU kernel, IntelPairHmm.cc:99-113); a HIP failure here used to
    be a RuntimeException that ends a GATK job of hours.  Now the slot drops its contexts, takes a fresh one and runs the
    call again: one injected failure in a 300 x 24 call -> the oracle's bits and one line on stderr; two in a row ->
    java/lang/RuntimeException (the convention of IntelPairHmm.cc:141-145,171-178), and the library works afterwards."""
    lib = C.CDLL(mockjni.JNI_LIB)
    lib.gklhip_fault_inject.argtypes = [C.c_char_p]
    b = make_batch("hc", 300, 24, seed=91)
    exp = oracle.batch(b, n_threads=8)
    if pipelined:
        monkeypatch.setenv("GKL_HIP_JNI_PIPELINE_PAIRS", "1")
        monkeypatch.setenv("GKL_HIP_JNI_RANGE_PAIRS", "1200")    # six ranges
    try:
        assert lib.gklhip_fault_inject(b"compute:3" if pipelined else b"compute:1") == 0
        rc, out, cls, msg, refs = mockjni.run(b, max_threads=2)
        assert rc == 0, (cls, msg)
        assert out.tobytes() == exp.tobytes() and refs[2] == 0
        err = capfd.readouterr().err
        assert err.count("retrying the call once on a fresh device context") == 1 and "injected fault" in err
        assert lib.gklhip_fault_inject(b"compute:1x1000") == 0
        rc, out, cls, msg, refs = mockjni.run(b, max_threads=2)
        assert rc == 2 and cls == "java/lang/RuntimeException" and "injected fault" in msg, (rc, cls, msg)
    finally:
        lib.gklhip_fault_inject(None)
    rc, out, cls, msg, _ = mockjni.run(b)
    assert rc == 0 and out.tobytes() == exp.tobytes()


@pytest.mark.gpu
def test_jni_idle_slot_releases_its_second_engine_and_computes_again(oracle):
    """The JNI library's janitor (GKL_HIP_IDLE_RELEASE_MS): a pipelined call, 0.9 s of nothing -- the slot's second context,
    compute threads and extra streams go -- and the same call again: the oracle's bits both times (a fresh process: the
    janitor reads its setting when the first initNative starts it)."""
    import subprocess
    import sys
    code = ("import sys, ctypes as C; sys.path.insert(0, %r)\n"
            "import numpy as np\nfrom tests import mockjni\nfrom gkl_amd.synth import make_batch\nfrom oracle.oracle import Oracle\n"
            "b = make_batch('hc', 400, 24, seed=77)\n"
            "rc, out, cls, msg, k = mockjni.run(b, flags=mockjni.PAUSE_AND_AGAIN, max_threads=2)\n"
            "h = (C.c_int64 * 5)(); C.CDLL(mockjni.JNI_LIB).gkl_pairhmm_jni_helpers(h, 0)\n"
            "print(rc, out.tobytes() == Oracle().batch(b, n_threads=8).tobytes(), k[mockjni.VIOLATIONS], h[4])\n") % ROOT
    env = dict(os.environ, GKL_HIP_JNI_PIPELINE_PAIRS="1", GKL_HIP_JNI_RANGE_PAIRS="2400", GKL_HIP_IDLE_RELEASE_MS="150", MOCKJNI_PAUSE_MS="900")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-1500:]
    rc, same, viol, released = p.stdout.split()[-4:]
    assert (rc, same, viol) == ("0", "True", "0") and int(released) >= 1, p.stdout


def _check_jni_onload(monkeypatch, have_gpu):
    # JNI_OnLoad (not in the reference): JNI_ERR without a usable gfx950 device, so that System.load fails and
    # NativeLibraryLoader.load() returns false (GATK then falls back); JNI_VERSION_1_8 with one, or when forced
    for name in ("libgkl_pairhmm.so", "libgkl_pdhmm.so", "libgkl_smithwaterman_hip.so"):
        lib = C.CDLL(os.path.join(os.path.dirname(mockjni.JNI_LIB), name))
        lib.JNI_OnLoad.restype = C.c_int
        lib.JNI_OnLoad.argtypes = [C.c_void_p, C.c_void_p]
        monkeypatch.delenv("GKL_HIP_LOAD_WITHOUT_DEVICE", raising=False)
        assert lib.JNI_OnLoad(None, None) == (0x00010008 if have_gpu else -1), name
        monkeypatch.setenv("GKL_HIP_LOAD_WITHOUT_DEVICE", "1")
        assert lib.JNI_OnLoad(None, None) == 0x00010008


def _check_utils_library(have_gpu):
    """libgkl_utils.so replacement: IntelPairHmm.load() asks isAvxSupported() first
    (IntelPairHmm.java:66-75); here that answers "is a gfx950 device usable"."""
    mockjni.build()
    lib = C.CDLL(mockjni.SO)
    out = (C.c_int * 6)()
    path = os.path.join(ROOT, "gkl_amd", "lib", "libgkl_utils.so")
    assert lib.mockjni_run_utils(path.encode(), out) == 0
    assert bool(out[1]) == have_gpu and bool(out[2]) == have_gpu and out[3] == 0
    assert out[4] >= 1 and out[5] == 1  # FTZ can be switched on, like utils.cc:44-55


def test_jni_onload_probes_for_a_device(monkeypatch):
    import torch
    _check_jni_onload(monkeypatch, torch.cuda.is_available())


def test_utils_library_gates_on_the_gpu():
    import torch
    _check_utils_library(torch.cuda.is_available())


@pytest.mark.gpu
def test_jni_onload_with_a_device(monkeypatch):
    # the device-present branch, in a driver-observed -m gpu process: JNI_OnLoad -> JNI_VERSION_1_8
    _check_jni_onload(monkeypatch, True)


@pytest.mark.gpu
def test_utils_library_with_a_device():
    # isAvxSupported / isAvx2Supported -> true on the GPU box (what IntelPairHmm.load() gates on)
    _check_utils_library(True)


@pytest.mark.gpu
def test_jni_pipelined_big_call_path_bit_exact_and_error_safe(oracle, monkeypatch):
    """Big calls are pipelined: read ranges are marshalled on the calling thread while earlier ranges compute on the
    slot's two engines (jni_shim.cpp).  Forced here on a small batch (GKL_HIP_JNI_PIPELINE_PAIRS / _RANGE_PAIRS): results
    bit-identical to the oracle with 2..16 ranges, both precisions; a holder error in the first range and a too-short
    likelihood array raise the same exceptions as the one-shot path, with ranges in flight or not, and leave the
    library usable."""
    b = make_batch("hc", 400, 24, seed=77)
    exp = oracle.batch(b, n_threads=8)
    expd = oracle.batch(b, use_double=True, n_threads=8)
    monkeypatch.setenv("GKL_HIP_JNI_PIPELINE_PAIRS", "1")
    for rp in ("600", "2400", "4800"):   # 16, 4 and 2 ranges
        monkeypatch.setenv("GKL_HIP_JNI_RANGE_PAIRS", rp)
        rc, out, cls, msg, refs = mockjni.run(b)
        assert rc == 0, (cls, msg)
        assert out.tobytes() == exp.tobytes(), rp
        assert refs[2] == 0 and refs[3] <= 6 * 32, (msg, refs)     # -Xcheck:jni rules hold in the pipelined loop too
    rc, out, cls, msg, _ = mockjni.run(b, use_double=True)
    assert rc == 0 and out.tobytes() == expd.tobytes()
    for flags in (mockjni.NULL_READQUALS, mockjni.SHORT_QUALS, mockjni.NULL_READ_ELEMENT):
        rc, _, cls, msg, refs = mockjni.run(b, flags=flags)
        assert rc == 2 and cls == "java/lang/IllegalArgumentException", (flags, cls, msg)
        # (a too-short quality array shows as the region copy's exception: the up to three copies of the same read
        # behind it are the only JNI calls ever made with an exception pending -- tests/test_jni_marshal_cpu.py)
        assert refs[2] <= (3 if flags == mockjni.SHORT_QUALS else 0), msg
    rc, _, cls, msg, _ = mockjni.run(b, out_len=b.n_pairs - 1)
    assert rc == 2 and cls == "java/lang/IllegalArgumentException"
    rc, out, cls, msg, _ = mockjni.run(b)
    assert rc == 0 and out.tobytes() == exp.tobytes()
    # several Java threads, every one pipelining through its own slot
    rc, out, cls, msg, _ = mockjni.run_concurrent(b, 3, iters=3)
    assert rc == 0, (cls, msg)
    assert out.tobytes() == exp.tobytes()
