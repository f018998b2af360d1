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
        subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", src, "-o", SO, "-ldl", "-lpthread"], check=True)


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


def run_concurrent(batch, n_threads, iters=1, use_double=False, max_threads=1, lib_path=JNI_LIB):
    """One initNative, n_threads concurrent callers (own JNIEnv each) over read slices, one doneNative.
    Returns (rc, out, exception_class, exception_message, wall_ms)."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    build()
    lib = C.CDLL(SO)
    out = np.zeros(max(batch.n_pairs, 1), np.float64)
    ec, em = C.create_string_buffer(256), C.create_string_buffer(512)
    wall = C.c_double(0.0)
    ro = np.ascontiguousarray(batch.read_off, np.int64)
    ho = np.ascontiguousarray(batch.hap_off, np.int64)
    arrs = [np.ascontiguousarray(a, np.uint8) for a in (batch.read_bases, batch.read_quals, batch.ins_gop,
                                                        batch.del_gop, batch.gcp, batch.hap_bases)]
    lib.mockjni_run_concurrent.restype = C.c_int
    rc = lib.mockjni_run_concurrent(lib_path.encode(), int(use_double), int(max_threads), int(n_threads), int(iters),
                                    batch.n_reads, batch.n_haps, ro.ctypes.data_as(C.c_void_p),
                                    ho.ctypes.data_as(C.c_void_p), *[a.ctypes.data_as(C.c_void_p) for a in arrs],
                                    out.ctypes.data_as(C.c_void_p), ec, em, C.byref(wall))
    return rc, out[:batch.n_pairs], ec.value.decode(), em.value.decode(), wall.value


# ---------------------------------------------------------------- PDHMM
PD_JNI_LIB = os.path.join(ROOT, "gkl_amd", "lib", "libgkl_pdhmm.so")
PD_DROP_PDBASES_FIELD, PD_SKIP_INIT, PD_HOLDERS = 1, 2, 4


def run_pdhmm(b, flags=0, max_memory_mb=512, holders=None):
    """Drive IntelPDHMM's natives through the mock JNIEnv.

    b: PdhmmBatch (padded 1:1) -> computePDHMMNative.  holders=(reads_batch, haps_batch): two PdhmmBatch-like
    objects giving distinct reads / haplotypes -> computeLikelihoodsNative over their cross product.
    Returns (rc, out, exception_class, exception_message)."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    build()
    lib = C.CDLL(SO)
    ec, em = C.create_string_buffer(256), C.create_string_buffer(512)
    if holders is None:
        n_a, n_b, out_len = b.batch, 0, b.batch
        hap_src, read_src = b, b
    else:
        read_src, hap_src = holders
        n_a, n_b, out_len = read_src.batch, hap_src.batch, read_src.batch * hap_src.batch
        flags |= PD_HOLDERS
    out = np.zeros(max(out_len, 1), np.float64)
    arrs = [np.ascontiguousarray(a, np.int8) for a in (hap_src.hap_bases, hap_src.hap_pdbases, read_src.read_bases,
                                                       read_src.read_qual, read_src.read_ins_qual,
                                                       read_src.read_del_qual, read_src.gcp)]
    hl = np.ascontiguousarray(hap_src.hap_lengths, np.int64)
    rl = np.ascontiguousarray(read_src.read_lengths, np.int64)
    lib.mockjni_run_pdhmm.restype = C.c_int
    rc = lib.mockjni_run_pdhmm(PD_JNI_LIB.encode(), int(n_a), int(n_b), int(hap_src.max_hap_len),
                               int(read_src.max_read_len), *[a.ctypes.data_as(C.c_void_p) for a in arrs],
                               hl.ctypes.data_as(C.c_void_p), rl.ctypes.data_as(C.c_void_p),
                               out.ctypes.data_as(C.c_void_p), int(out_len), int(flags), int(max_memory_mb), ec, em)
    return rc, out[:out_len], ec.value.decode(), em.value.decode()


# ---------------------------------------------------------------- Smith-Waterman
SW_JNI_LIB = os.path.join(ROOT, "gkl_amd", "lib", "libgkl_smithwaterman.so")
SW_SKIP_INIT, SW_NULL_REF = 1, 2


def run_sw(ref, alt, params, strategy, cigar_len=None, flags=0, iters=1):
    """initNative -> alignNative x iters -> doneNative through the mock JNIEnv.
    Returns (rc, cigar bytes (trimmed like IntelSmithWaterman.java:150), offset, exception_class, message, wall_ms)."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    build()
    lib = C.CDLL(SO)
    ref, alt = bytes(ref), bytes(alt)
    if cigar_len is None:
        cigar_len = 2 * max(len(ref), len(alt))
    out = C.create_string_buffer(max(cigar_len, 1))
    ec, em = C.create_string_buffer(256), C.create_string_buffer(512)
    off, wall = C.c_int(0), C.c_double(0.0)
    lib.mockjni_run_sw.restype = C.c_int
    rc = lib.mockjni_run_sw(SW_JNI_LIB.encode(), ref, len(ref), alt, len(alt), int(cigar_len), int(params[0]),
                            int(params[1]), int(params[2]), int(params[3]), int(strategy), int(flags), int(iters), out,
                            C.byref(off), ec, em, C.byref(wall))
    return rc, out.raw[:cigar_len].rstrip(b"\0"), off.value, ec.value.decode(), em.value.decode(), wall.value


def run_sw_batch(refs, alts, params, strategy, stride=None):
    """initNative -> alignBatchNative -> doneNative.  Returns (rc, returned, [cigar bytes], offsets, class, message)."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    build()
    lib = C.CDLL(SO)
    n = len(refs)
    refs, alts = [bytes(r) for r in refs], [bytes(a) for a in alts]
    if stride is None:
        stride = 2 * max(max(len(r), len(a)) for r, a in zip(refs, alts))
    ro = np.zeros(n + 1, np.int64)
    ao = np.zeros(n + 1, np.int64)
    np.cumsum([len(r) for r in refs], out=ro[1:])
    np.cumsum([len(a) for a in alts], out=ao[1:])
    rb, ab = b"".join(refs), b"".join(alts)
    cig = np.zeros(n * stride, np.uint8)
    off = np.zeros(n, np.int32)
    ret = C.c_int(0)
    ec, em = C.create_string_buffer(256), C.create_string_buffer(512)
    lib.mockjni_run_sw_batch.restype = C.c_int
    rc = lib.mockjni_run_sw_batch(SW_JNI_LIB.encode(), n, rb, ro.ctypes.data_as(C.c_void_p), ab, ao.ctypes.data_as(C.c_void_p),
                                  int(stride), int(params[0]), int(params[1]), int(params[2]), int(params[3]), int(strategy),
                                  cig.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), C.byref(ret), ec, em)
    rows = cig.reshape(n, stride)
    return rc, ret.value, [rows[k].tobytes().rstrip(b"\0") for k in range(n)], off, ec.value.decode(), em.value.decode()
