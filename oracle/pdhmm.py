"""ctypes front-end of the PDHMM checkers (test infrastructure only, like oracle/oracle.py)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_i8p = C.POINTER(C.c_int8)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)

STATUS = {0: "PDHMM_SUCCESS", 1: "PDHMM_MEMORY_ALLOCATION_FAILED", 2: "PDHMM_INPUT_DATA_ERROR",
          3: "PDHMM_FAILURE", 4: "PDHMM_MEMORY_ACCESS_ERROR"}


def _args(b):
    arrs = [np.ascontiguousarray(a, np.int8) for a in (b.hap_bases, b.hap_pdbases, b.read_bases, b.read_qual,
                                                        b.read_ins_qual, b.read_del_qual, b.gcp)]
    hl = np.ascontiguousarray(b.hap_lengths, np.int64)
    rl = np.ascontiguousarray(b.read_lengths, np.int64)
    res = np.zeros(b.batch, np.float64)
    return arrs, hl, rl, res


class PdhmmOracle:
    """Our plain-C restatement (oracle/pdhmm_oracle.c)."""

    def __init__(self):
        path = os.path.join(_HERE, "liboracle_pdhmm.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.run(["make", "-s", "-C", _HERE, "oracle"], check=True)
        self.lib = lib = C.CDLL(path)
        lib.pdhmm_oracle_compute.restype = C.c_int
        lib.pdhmm_oracle_table.restype = C.c_long
        lib.pdhmm_oracle_init()

    def compute(self, b, semantics=0, threads=8):
        arrs, hl, rl, res = _args(b)
        st = self.lib.pdhmm_oracle_compute(*[a.ctypes.data_as(_i8p) for a in arrs], res.ctypes.data_as(_f64p),
                                           C.c_int64(b.batch), hl.ctypes.data_as(_i64p), rl.ctypes.data_as(_i64p),
                                           C.c_int32(b.max_read_len), C.c_int32(b.max_hap_len), int(semantics), int(threads))
        return st, res

    def compute_reference(self, b, fma_mode=1, ref_batch=0, threads=8):
        """What GKL returns for every POSITION of a paired batch of pairs (pdhmm.h:1133-1290): the vector arithmetic of
        the engine (semantics 2 = AVX-512 object, fma_mode 1; 0 = AVX2 object) for the full groups of SIMD width, the
        scalar engine's (semantics 1) for the last `batch mod width` pairs -- per reference batch of `ref_batch` pairs
        when the pairs are the expansion of a cross product (pdhmm/JavaData.h:83-101,177-242); 0 = one batch."""
        width, sem = (8, 2) if fma_mode else (4, 0)
        st, out = self.compute(b, semantics=sem, threads=threads)
        per = min(ref_batch, b.batch) if ref_batch > 0 else b.batch
        tail = []
        for start in range(0, b.batch, max(per, 1)):
            nb = min(per, b.batch - start)
            tail += list(range(start + nb // width * width, start + nb))
        if tail:
            st1, ser = self.compute(b.subset(tail), semantics=1, threads=threads)   # (the scalar engine's input checks apply to the tail only)
            out[tail] = ser
            st = max(st, st1)
        return st, out

    def table(self, which):
        n = self.lib.pdhmm_oracle_table(which, None, C.c_long(0))
        a = np.empty(n, np.float64)
        self.lib.pdhmm_oracle_table(which, a.ctypes.data_as(_f64p), C.c_long(n))
        return a


class PdhmmReference:
    """The reference's own PDHMM objects (oracle/_ref/libgkl_ref_pdhmm.so); engine 0 scalar, 1 AVX2, 2 AVX-512."""
    path = os.path.join(_HERE, "_ref", "libgkl_ref_pdhmm.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.path)

    def __init__(self):
        self.lib = lib = C.CDLL(self.path)
        lib.ref_pdhmm_compute.restype = C.c_int
        lib.ref_pdhmm_table.restype = C.c_long
        lib.ref_pdhmm_has_avx512.restype = C.c_int
        lib.ref_pdhmm_simd_width.restype = C.c_int

    def has_avx512(self):
        return bool(self.lib.ref_pdhmm_has_avx512())

    def simd_width(self, engine):
        return int(self.lib.ref_pdhmm_simd_width(engine))

    def compute(self, b, engine=1, threads=1):
        arrs, hl, rl, res = _args(b)
        st = self.lib.ref_pdhmm_compute(int(engine), *[a.ctypes.data_as(_i8p) for a in arrs],
                                        res.ctypes.data_as(_f64p), C.c_int64(b.batch), hl.ctypes.data_as(_i64p),
                                        rl.ctypes.data_as(_i64p), C.c_int32(b.max_read_len), C.c_int32(b.max_hap_len),
                                        int(threads))
        return st, res

    def table(self, which):
        n = self.lib.ref_pdhmm_table(which, None, C.c_long(0))
        a = np.empty(n, np.float64)
        self.lib.ref_pdhmm_table(which, a.ctypes.data_as(_f64p), C.c_long(n))
        return a
