"""ctypes front-end of the CHECKERS under oracle/ -- test infrastructure only.

Two libraries, both built by ``oracle/Makefile`` (see ``__graft_entry__.build``):

* ``liboracle_pairhmm.so``  -- our plain-C restatement (``pairhmm_oracle.c``);
* ``_ref/libgkl_ref_pairhmm.so`` -- the reference's own kernel objects compiled
  in place from /root/reference plus ``ref_driver.cpp``.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product package ``gkl_amd`` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_u8p = C.POINTER(C.c_uint8)
_i64p = C.POINTER(C.c_int64)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)


def build(ref: bool = True) -> None:
    """(Re)build the checker libraries. Building the checker is not using it."""
    subprocess.run(["make", "-s", "-C", _HERE, "oracle"], check=True)
    if ref:
        subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)


def _ptr(a, ty):
    return a.ctypes.data_as(ty) if a is not None else None


def _u8(a) -> np.ndarray:
    if isinstance(a, (bytes, bytearray)):
        a = np.frombuffer(bytes(a), dtype=np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


class _BatchRunner:
    """Shared plumbing for the two libraries' batch entry points."""

    def _run_batch(self, fn, batch, use_double, extra, n_threads, want_raw):
        n = batch.n_reads * batch.n_haps
        out = np.empty(n, dtype=np.float64)
        raw32 = np.zeros(n, dtype=np.float32) if want_raw else None
        raw64 = np.zeros(n, dtype=np.float64) if want_raw else None
        used64 = np.zeros(n, dtype=np.uint8) if want_raw else None
        ro = np.ascontiguousarray(batch.read_off, dtype=np.int64)
        ho = np.ascontiguousarray(batch.hap_off, dtype=np.int64)
        args = [batch.n_reads, batch.n_haps, _ptr(ro, _i64p), _ptr(ho, _i64p),
                _ptr(batch.read_bases, _u8p), _ptr(batch.read_quals, _u8p),
                _ptr(batch.ins_gop, _u8p), _ptr(batch.del_gop, _u8p),
                _ptr(batch.gcp, _u8p), _ptr(batch.hap_bases, _u8p), int(use_double)]
        args += extra + [int(n_threads), _ptr(out, _f64p), _ptr(raw32, _f32p),
                         _ptr(raw64, _f64p), _ptr(used64, _u8p)]
        fn(*args)
        if want_raw:
            return out, raw32, raw64, used64
        return out


class Oracle(_BatchRunner):
    """Our own C restatement (kind == "port")."""

    kind = "port"

    def __init__(self, path: str | None = None):
        path = path or os.path.join(_HERE, "liboracle_pairhmm.so")
        if not os.path.exists(path):
            build(ref=False)
        self.lib = lib = C.CDLL(path)
        pair_args = [_u8p, _u8p, _u8p, _u8p, _u8p, C.c_int, _u8p, C.c_int, C.c_int]
        lib.oracle_fwd_f32.argtypes = pair_args
        lib.oracle_fwd_f32.restype = C.c_float
        lib.oracle_fwd_f64.argtypes = pair_args
        lib.oracle_fwd_f64.restype = C.c_double
        lib.oracle_finalize_f32.argtypes = [C.c_float]
        lib.oracle_finalize_f32.restype = C.c_double
        lib.oracle_finalize_f64.argtypes = [C.c_double]
        lib.oracle_finalize_f64.restype = C.c_double
        lib.oracle_batch.argtypes = [C.c_int, C.c_int, _i64p, _i64p, _u8p, _u8p, _u8p, _u8p, _u8p,
                                     _u8p, C.c_int, C.c_int, C.c_int, _f64p, _f32p, _f64p, _u8p]
        lib.oracle_batch.restype = None
        lib.oracle_table_f32.argtypes = [C.c_int, _f32p, C.c_long]
        lib.oracle_table_f32.restype = C.c_long
        lib.oracle_table_f64.argtypes = [C.c_int, _f64p, C.c_long]
        lib.oracle_table_f64.restype = C.c_long
        lib.oracle_max_threads.restype = C.c_int
        lib.oracle_tables_init()

    def max_threads(self) -> int:
        return int(self.lib.oracle_max_threads())

    def pair_raw(self, read, quals, ins, dele, gcp, hap, fma_mode=1):
        rs, q, i, d, c, h = map(_u8, (read, quals, ins, dele, gcp, hap))
        a = [_ptr(rs, _u8p), _ptr(q, _u8p), _ptr(i, _u8p), _ptr(d, _u8p), _ptr(c, _u8p),
             len(rs), _ptr(h, _u8p), len(h), int(fma_mode)]
        return (np.float32(self.lib.oracle_fwd_f32(*a)), np.float64(self.lib.oracle_fwd_f64(*a)))

    def finalize_f32(self, raw) -> float:
        return float(self.lib.oracle_finalize_f32(C.c_float(float(raw))))

    def finalize_f64(self, raw) -> float:
        return float(self.lib.oracle_finalize_f64(C.c_double(float(raw))))

    def finalize(self, raw32, raw64, used64) -> np.ndarray:
        """Vectorised PH/IntelPairHmm.cc:159-165 on raw sums (host libm)."""
        raw32 = np.asarray(raw32, dtype=np.float32)
        raw64 = np.asarray(raw64, dtype=np.float64)
        out = np.empty(raw32.shape, dtype=np.float64)
        for k in range(raw32.size):
            out.flat[k] = (self.finalize_f64(raw64.flat[k]) if used64.flat[k]
                           else self.finalize_f32(raw32.flat[k]))
        return out

    def batch(self, batch, use_double=False, fma_mode=1, n_threads=1, want_raw=False):
        return self._run_batch(self.lib.oracle_batch, batch, use_double, [int(fma_mode)],
                               n_threads, want_raw)

    def table(self, which: int, dtype):
        if np.dtype(dtype) == np.float32:
            n = self.lib.oracle_table_f32(which, None, 0)
            a = np.empty(n, dtype=np.float32)
            self.lib.oracle_table_f32(which, _ptr(a, _f32p), n)
        else:
            n = self.lib.oracle_table_f64(which, None, 0)
            a = np.empty(n, dtype=np.float64)
            self.lib.oracle_table_f64(which, _ptr(a, _f64p), n)
        return a


class Reference(_BatchRunner):
    """The reference's own kernel objects (kind == "reference"); x86 AVX only."""

    kind = "reference"
    path = os.path.join(_HERE, "_ref", "libgkl_ref_pairhmm.so")

    @classmethod
    def available(cls) -> bool:
        return os.path.exists(cls.path)

    def __init__(self, engine: int = 0):
        if not self.available():
            raise FileNotFoundError(self.path + " (run `make -C oracle ref` where /root/reference exists)")
        self.lib = lib = C.CDLL(self.path)
        lib.ref_init.argtypes = [C.c_int]
        lib.ref_init.restype = C.c_int
        lib.ref_has_avx512.restype = C.c_int
        lib.ref_has_avx.restype = C.c_int
        lib.ref_max_threads.restype = C.c_int
        lib.ref_pair_raw.argtypes = [_u8p, _u8p, _u8p, _u8p, _u8p, C.c_int, _u8p, C.c_int, _f32p, _f64p]
        lib.ref_pair_raw.restype = None
        lib.ref_batch.argtypes = [C.c_int, C.c_int, _i64p, _i64p, _u8p, _u8p, _u8p, _u8p, _u8p, _u8p,
                                  C.c_int, C.c_int, _f64p, _f32p, _f64p, _u8p]
        lib.ref_batch.restype = None
        lib.ref_table_f32.argtypes = [C.c_int, _f32p, C.c_long]
        lib.ref_table_f32.restype = C.c_long
        lib.ref_table_f64.argtypes = [C.c_int, _f64p, C.c_long]
        lib.ref_table_f64.restype = C.c_long
        if not lib.ref_has_avx():
            raise RuntimeError("host CPU lacks AVX; the reference kernels cannot run")
        if engine == 2 and not lib.ref_has_avx512():
            raise RuntimeError("host CPU lacks AVX-512")
        self.engine = int(lib.ref_init(engine))  # 1 = AVX (unfused), 2 = AVX-512 (FMA)

    @property
    def fma_mode(self) -> int:
        return 1 if self.engine == 2 else 0

    def has_avx512(self) -> bool:
        return bool(self.lib.ref_has_avx512())

    def max_threads(self) -> int:
        return int(self.lib.ref_max_threads())

    def set_engine(self, engine: int) -> int:
        self.engine = int(self.lib.ref_init(engine))
        return self.engine

    def pair_raw(self, read, quals, ins, dele, gcp, hap):
        rs, q, i, d, c, h = map(_u8, (read, quals, ins, dele, gcp, hap))
        r32 = C.c_float()
        r64 = C.c_double()
        self.lib.ref_pair_raw(_ptr(rs, _u8p), _ptr(q, _u8p), _ptr(i, _u8p), _ptr(d, _u8p),
                              _ptr(c, _u8p), len(rs), _ptr(h, _u8p), len(h),
                              C.byref(r32), C.byref(r64))
        return np.float32(r32.value), np.float64(r64.value)

    def batch(self, batch, use_double=False, n_threads=1, want_raw=False):
        return self._run_batch(self.lib.ref_batch, batch, use_double, [], n_threads, want_raw)

    def table(self, which: int, dtype):
        if np.dtype(dtype) == np.float32:
            n = self.lib.ref_table_f32(which, None, 0)
            a = np.empty(n, dtype=np.float32)
            self.lib.ref_table_f32(which, _ptr(a, _f32p), n)
        else:
            n = self.lib.ref_table_f64(which, None, 0)
            a = np.empty(n, dtype=np.float64)
            self.lib.ref_table_f64(which, _ptr(a, _f64p), n)
        return a
