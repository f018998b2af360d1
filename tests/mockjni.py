"""ctypes wrapper around tests/native/libmock_jni.so (the mock JVM side)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "native", "libmock_jni.so")
JNI_LIB = os.path.join(ROOT, "gkl_amd", "lib", "libgkl_pairhmm.so")

DROP_GCP_FIELD, NULL_READQUALS, SKIP_INIT, SHORT_QUALS, NULL_READ_ELEMENT = 1, 2, 4, 8, 16


def build():
    src = os.path.join(HERE, "native", "mock_jni.cpp")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", src, "-o", SO, "-ldl"], check=True)


def run(batch, use_double=False, max_threads=1, flags=0, out_len=None, lib_path=JNI_LIB):
    """Drive initNative -> computeLikelihoodsNative -> doneNative through the mock JNIEnv.
    Returns (rc, out, exception_class, exception_message, (refs_created, refs_deleted))."""
    try:
        import torch  # noqa: F401  (same HIP-runtime load order as gkl_amd.native)
    except ImportError:
        pass
    build()
    lib = C.CDLL(SO)
    n = batch.n_pairs if out_len is None else out_len
    out = np.zeros(max(n, 1), np.float64)
    ec, em = C.create_string_buffer(256), C.create_string_buffer(512)
    counters = (C.c_long * 2)()
    ro = np.ascontiguousarray(batch.read_off, np.int64)
    ho = np.ascontiguousarray(batch.hap_off, np.int64)
    arrs = [np.ascontiguousarray(a, np.uint8) for a in (batch.read_bases, batch.read_quals, batch.ins_gop,
                                                        batch.del_gop, batch.gcp, batch.hap_bases)]
    lib.mockjni_run.restype = C.c_int
    rc = lib.mockjni_run(lib_path.encode(), int(use_double), int(max_threads), batch.n_reads, batch.n_haps,
                         ro.ctypes.data_as(C.c_void_p), ho.ctypes.data_as(C.c_void_p),
                         *[a.ctypes.data_as(C.c_void_p) for a in arrs], out.ctypes.data_as(C.c_void_p),
                         int(n), int(flags), ec, em, counters)
    return rc, out[:n], ec.value.decode(), em.value.decode(), (counters[0], counters[1])
