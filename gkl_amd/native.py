"""ctypes binding of the C ABI (include/gkl_hip_pairhmm.h) -- the only way Python reaches
the HIP kernels.  There is no fallback: if ``libgklhip_pairhmm.so`` is missing or no
gfx950 device is visible, everything here raises.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .batch import FlatBatch
from .errors import IllegalArgumentException, OutOfMemoryError, RuntimeException

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libgklhip_pairhmm.so")

ABI_VERSION = 2
OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_OOM, ERR_HIP, ERR_UNSUPPORTED = range(6)
FINALIZE_REFERENCE_HOST, FINALIZE_DEVICE_F64, FINALIZE_DEVICE_REF32 = 0, 1, 2

_u8p = C.POINTER(C.c_uint8)
_i64p = C.POINTER(C.c_int64)


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("use_double", C.c_int32),
                ("max_threads", C.c_int32), ("fma_mode", C.c_int32), ("finalize", C.c_int32),
                ("record_events", C.c_int32), ("rows_per_lane", C.c_int32)]


class CBatch(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("n_haps", C.c_int32), ("read_off", _i64p), ("hap_off", _i64p),
                ("read_bases", C.c_void_p), ("read_quals", C.c_void_p), ("ins_gop", C.c_void_p),
                ("del_gop", C.c_void_p), ("gcp", C.c_void_p), ("hap_bases", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("n_pairs", C.c_int64), ("n_fallback", C.c_int64), ("cells", C.c_int64),
                ("cells_fp64", C.c_int64), ("n_chunks", C.c_int32), ("n_hap_groups", C.c_int32),
                ("rows_per_lane", C.c_int32), ("n_long_pairs", C.c_int32), ("ms_fwd_main", C.c_float),
                ("ms_fwd_fallback", C.c_float), ("ms_total_device", C.c_float), ("lane_fill", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def load_library(path: Optional[str] = None):
    """dlopen the product library; raises (never falls back) when it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    try:
        # PyTorch-ROCm wheels bundle their own libamdhip64; whichever HIP runtime is loaded
        # first owns the device, so let torch's load first and share it (torch is only used
        # for device memory / streams / torch.distributed, never for compute).
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise RuntimeException(f"{p} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(p)
    lib.gklhip_abi_version.restype = C.c_int
    lib.gklhip_device_count.restype = C.c_int
    lib.gklhip_strerror.restype = C.c_char_p
    lib.gklhip_strerror.argtypes = [C.c_int]
    lib.gklhip_last_error.restype = C.c_char_p
    lib.gklhip_init.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    lib.gklhip_init.restype = C.c_int
    lib.gklhip_init_devices.argtypes = [C.POINTER(Config), C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_void_p)]
    lib.gklhip_init_devices.restype = C.c_int
    lib.gklhip_num_devices.argtypes = [C.c_void_p]
    lib.gklhip_num_devices.restype = C.c_int
    lib.gklhip_gather_backend.argtypes = [C.c_void_p]
    lib.gklhip_gather_backend.restype = C.c_int
    lib.gklhip_measure_issue_ceiling.argtypes = [C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.gklhip_measure_issue_ceiling.restype = C.c_int
    lib.gklhip_small_call_counts.argtypes = [C.c_int, C.POINTER(C.c_int64), C.c_int]
    lib.gklhip_release_idle.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.gklhip_release_idle.restype = C.c_int
    lib.gklhip_fault_inject.argtypes = [C.c_char_p]
    lib.gklhip_fault_inject.restype = C.c_int
    lib.gklhip_small_call_counts.restype = C.c_int
    lib.gklhip_gather_note.argtypes = [C.c_void_p]
    lib.gklhip_gather_note.restype = C.c_char_p
    lib.gklhip_partition_reads.argtypes = [C.c_int32, _i64p, C.c_int32, C.POINTER(C.c_int32)]
    lib.gklhip_partition_reads.restype = C.c_int
    lib.gklhip_rccl_selftest.argtypes = [C.c_int32]
    lib.gklhip_rccl_selftest.restype = C.c_int
    lib.gklhip_done.argtypes = [C.c_void_p]
    lib.gklhip_done.restype = C.c_int
    lib.gklhip_compute.argtypes = [C.c_void_p, C.POINTER(CBatch), C.c_void_p]
    lib.gklhip_compute.restype = C.c_int
    lib.gklhip_compute_device.argtypes = [C.c_void_p, C.POINTER(CBatch), C.c_void_p, C.c_void_p]
    lib.gklhip_compute_device.restype = C.c_int
    lib.gklhip_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    lib.gklhip_get_step_times.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                          C.POINTER(C.c_float)]
    lib.gklhip_get_step_times.restype = C.c_int
    lib.gklhip_get_stats.restype = C.c_int
    lib.gklhip_get_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gklhip_get_raw.restype = C.c_int
    lib.gklhip_get_table_f32.argtypes = [C.c_int, C.c_void_p, C.c_int64]
    lib.gklhip_get_table_f32.restype = C.c_int64
    lib.gklhip_get_table_f64.argtypes = [C.c_int, C.c_void_p, C.c_int64]
    lib.gklhip_get_table_f64.restype = C.c_int64
    if lib.gklhip_abi_version() != ABI_VERSION:
        raise RuntimeException("libgklhip_pairhmm.so ABI mismatch")
    if path is None:
        _lib = lib
    return lib


def _raise(lib, status: int):
    msg = (lib.gklhip_last_error() or b"").decode() or lib.gklhip_strerror(status).decode()
    if status == ERR_INVALID_ARG:
        raise IllegalArgumentException(msg)
    if status == ERR_OOM:
        raise OutOfMemoryError(msg)
    raise RuntimeException(f"{lib.gklhip_strerror(status).decode()}: {msg}")


def host_table(which: int, dtype) -> np.ndarray:
    """Lookup tables exactly as the library uploads them (0 ph2pr, 1 matchToMatch, 2 ph2pr/3)."""
    lib = load_library()
    if np.dtype(dtype) == np.float32:
        n = lib.gklhip_get_table_f32(which, None, 0)
        a = np.empty(n, np.float32)
        lib.gklhip_get_table_f32(which, a.ctypes.data, n)
    else:
        n = lib.gklhip_get_table_f64(which, None, 0)
        a = np.empty(n, np.float64)
        lib.gklhip_get_table_f64(which, a.ctypes.data, n)
    return a


@dataclass
class DeviceBatch:
    """A FlatBatch whose byte arrays live in HBM (torch uint8 tensors own the memory)."""
    host: FlatBatch
    tensors: tuple  # read_bases, read_quals, ins, del, gcp, hap_bases
    read_off: np.ndarray
    hap_off: np.ndarray

    @staticmethod
    def upload(batch: FlatBatch, device="cuda:0") -> "DeviceBatch":
        import torch
        ts = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in
                   (batch.read_bases, batch.read_quals, batch.ins_gop, batch.del_gop, batch.gcp,
                    batch.hap_bases))
        return DeviceBatch(batch, ts, np.ascontiguousarray(batch.read_off, np.int64),
                           np.ascontiguousarray(batch.hap_off, np.int64))

    def c_batch(self) -> CBatch:
        t = self.tensors
        return CBatch(self.host.n_reads, self.host.n_haps, self.read_off.ctypes.data_as(_i64p),
                      self.hap_off.ctypes.data_as(_i64p), *[x.data_ptr() for x in t])


class PinnedBatch:
    """A FlatBatch whose six byte arrays live in page-locked memory from gklhip_host_alloc -- what the JNI shim's
    marshalling arenas are (jni_shim.cpp), so that gklhip_compute's host-to-device copies are plain DMA."""

    def __init__(self, batch: FlatBatch):
        import dataclasses
        lib = load_library()
        lib.gklhip_host_alloc.restype = C.c_void_p
        lib.gklhip_host_alloc.argtypes = [C.c_size_t]
        lib.gklhip_host_free.argtypes = [C.c_void_p]
        self._lib, self._ptrs, arrays = lib, [], {}
        for name in ("read_bases", "read_quals", "ins_gop", "del_gop", "gcp", "hap_bases"):
            src = np.ascontiguousarray(getattr(batch, name), np.uint8)
            p = lib.gklhip_host_alloc(max(1, src.size))
            if not p:
                raise OutOfMemoryError("gklhip_host_alloc failed")
            self._ptrs.append(p)
            dst = np.ctypeslib.as_array((C.c_uint8 * max(1, src.size)).from_address(p))[:src.size]
            dst[:] = src
            arrays[name] = dst
        self.batch = dataclasses.replace(batch, **arrays)

    def close(self):
        for p in self._ptrs:
            self._lib.gklhip_host_free(p)
        self._ptrs = []

    def __enter__(self):
        return self.batch

    def __exit__(self, *a):
        self.close()


def partition_reads(read_off, n_parts: int):
    """The library's sharding rule (gklhip_partition_reads): boundaries of contiguous read ranges balanced by cells."""
    lib = load_library()
    ro = np.ascontiguousarray(read_off, np.int64)
    bounds = (C.c_int32 * (n_parts + 1))()
    st = lib.gklhip_partition_reads(ro.size - 1, ro.ctypes.data_as(_i64p), n_parts, bounds)
    if st != OK:
        _raise(lib, st)
    return list(bounds)


def small_call_counts(device: int = 0, reset: bool = False):
    """(small host-buffer calls, those launched together with other threads' calls, sets of launches) on `device`."""
    lib = load_library()
    out = (C.c_int64 * 3)()
    st = lib.gklhip_small_call_counts(int(device), out, 1 if reset else 0)
    if st != OK:
        _raise(lib, st)
    return int(out[0]), int(out[1]), int(out[2])


def rccl_selftest(device: int = 0) -> None:
    lib = load_library()
    st = lib.gklhip_rccl_selftest(device)
    if st != OK:
        _raise(lib, st)


class PairHmmContext:
    """One gklhip context (= one initNative).  `devices` = a list of device ordinals: every call is sharded over
    them inside the library (a device may appear twice)."""

    def __init__(self, use_double: bool = False, max_threads: int = 0, device: int = -1,
                 fma_mode: int = 1, finalize: int = -1, record_events: bool = False,
                 rows_per_lane: int = 0, lib_path: Optional[str] = None, devices=None):
        self.lib = load_library(lib_path)
        cfg = Config(ABI_VERSION, device, int(use_double), int(max_threads), int(fma_mode),
                     int(finalize), int(record_events), int(rows_per_lane))
        h = C.c_void_p()
        if devices:
            arr = (C.c_int32 * len(devices))(*devices)
            st = self.lib.gklhip_init_devices(C.byref(cfg), arr, len(devices), C.byref(h))
        else:
            st = self.lib.gklhip_init(C.byref(cfg), C.byref(h))
        if st != OK:
            _raise(self.lib, st)
        self.handle = h
        self.use_double = bool(use_double)

    @property
    def n_devices(self) -> int:
        return self.lib.gklhip_num_devices(self.handle)

    @property
    def gather_backend(self) -> str:
        """none | peer | rccl (lazily created: before the first device-resident call, what it will try) |
        peer-after-rccl-failure (gather_note says why)"""
        return ("none", "peer", "rccl", "peer-after-rccl-failure")[self.lib.gklhip_gather_backend(self.handle)]

    def release_idle(self) -> int:
        """Give back the streams / twin engines the context holds only for speed, if it is idle (gklhip_release_idle);
        returns how many streams went."""
        n = C.c_int32(0)
        st = self.lib.gklhip_release_idle(self.handle, C.byref(n))
        if st != OK:
            _raise(self.lib, st)
        return int(n.value)

    def issue_ceiling(self, use_double: bool = False, ms_budget: float = 50.0):
        """(cells per second of the recurrence's bare instruction mix, sustained clock in GHz) -- diagnostics."""
        cells, clk = C.c_double(0.0), C.c_double(0.0)
        st = self.lib.gklhip_measure_issue_ceiling(self.handle, 1 if use_double else 0, float(ms_budget), C.byref(cells), C.byref(clk))
        if st != OK:
            _raise(self.lib, st)
        return cells.value, clk.value

    @property
    def gather_note(self) -> str:
        return (self.lib.gklhip_gather_note(self.handle) or b"").decode()

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gklhip_done(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- host buffers in / out: what the JNI shim does --
    def compute(self, batch: FlatBatch, out: Optional[np.ndarray] = None) -> np.ndarray:
        n = batch.n_pairs
        if out is None:
            out = np.empty(n, np.float64)
        if out.dtype != np.float64 or out.size < n or not out.flags.c_contiguous:
            raise IllegalArgumentException("likelihood array must be contiguous float64 of n_reads*n_haps")
        keep = [np.ascontiguousarray(a, np.uint8) for a in
                (batch.read_bases, batch.read_quals, batch.ins_gop, batch.del_gop, batch.gcp,
                 batch.hap_bases)]
        ro = np.ascontiguousarray(batch.read_off, np.int64)
        ho = np.ascontiguousarray(batch.hap_off, np.int64)
        cb = CBatch(batch.n_reads, batch.n_haps, ro.ctypes.data_as(_i64p), ho.ctypes.data_as(_i64p),
                    *[a.ctypes.data for a in keep])
        st = self.lib.gklhip_compute(self.handle, C.byref(cb), out.ctypes.data)
        if st != OK:
            _raise(self.lib, st)
        return out

    # -- everything resident in HBM; `out` is a torch float64 CUDA tensor --
    def compute_device(self, dbatch: DeviceBatch, out=None, stream=None):
        import torch
        n = dbatch.host.n_pairs
        if out is None:
            out = torch.empty(n, dtype=torch.float64, device=dbatch.tensors[0].device)
        if stream is None:
            stream = torch.cuda.current_stream(out.device)
        cb = dbatch.c_batch()
        st = self.lib.gklhip_compute_device(self.handle, C.byref(cb), out.data_ptr(), stream.cuda_stream)
        if st != OK:
            _raise(self.lib, st)
        return out

    def step_times(self, steps_back: int = 0):
        """record_events=2 contexts: (ms_main, ms_fallback, ms_total_device) of the call `steps_back` calls ago."""
        a, b, t = C.c_float(0), C.c_float(0), C.c_float(0)
        st = self.lib.gklhip_get_step_times(self.handle, int(steps_back), C.byref(a), C.byref(b), C.byref(t))
        if st != OK:
            _raise(self.lib, st)
        return a.value, b.value, t.value

    def stats(self) -> dict:
        s = Stats()
        st = self.lib.gklhip_get_stats(self.handle, C.byref(s))
        if st != OK:
            _raise(self.lib, st)
        return s.as_dict()

    def raw(self, n_pairs: int):
        """Raw scaled sums of the last call: (raw32, raw64, used64)."""
        r32 = np.zeros(n_pairs, np.float32)
        r64 = np.zeros(n_pairs, np.float64)
        u = np.zeros(n_pairs, np.uint8)
        st = self.lib.gklhip_get_raw(self.handle, r32.ctypes.data, r64.ctypes.data, u.ctypes.data)
        if st != OK:
            _raise(self.lib, st)
        return r32, r64, u


# ---------------------------------------------------------------- PDHMM (include/gkl_hip_pdhmm.h)
PDHMM_LIB_PATH = os.environ.get("GKL_AMD_PDHMM_LIB") or os.path.join(LIB_DIR, "libgklhip_pdhmm.so")   # the variable: A/B of library builds (dev tool)


class CPdhmmBatch(C.Structure):
    _fields_ = [("batch", C.c_int32), ("max_hap_len", C.c_int32), ("max_read_len", C.c_int32),
                ("hap_bases", C.c_void_p), ("hap_pdbases", C.c_void_p), ("read_bases", C.c_void_p),
                ("read_qual", C.c_void_p), ("read_ins_qual", C.c_void_p), ("read_del_qual", C.c_void_p),
                ("gcp", C.c_void_p), ("hap_lengths", C.c_void_p), ("read_lengths", C.c_void_p)]


class CPdhmmCross(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("n_haps", C.c_int32), ("max_hap_len", C.c_int32), ("max_read_len", C.c_int32),
                ("hap_bases", C.c_void_p), ("hap_pdbases", C.c_void_p), ("read_bases", C.c_void_p),
                ("read_qual", C.c_void_p), ("read_ins_qual", C.c_void_p), ("read_del_qual", C.c_void_p),
                ("gcp", C.c_void_p), ("hap_lengths", C.c_void_p), ("read_lengths", C.c_void_p)]


_pd_lib = None


def load_pdhmm_library(path: Optional[str] = None):
    global _pd_lib
    if _pd_lib is not None and path is None:
        return _pd_lib
    p = path or PDHMM_LIB_PATH
    try:
        import torch  # noqa: F401  (HIP runtime load order, see load_library)
    except ImportError:
        pass
    if not os.path.exists(p):
        raise RuntimeException(f"{p} is not built (hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(p)
    lib.gklhip_pdhmm_init.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.gklhip_pdhmm_init.restype = C.c_int
    lib.gklhip_pdhmm_set_fma_mode.argtypes = [C.c_void_p, C.c_int]
    lib.gklhip_pdhmm_set_fma_mode.restype = C.c_int
    lib.gklhip_pdhmm_set_tail_mode.argtypes = [C.c_void_p, C.c_int]
    lib.gklhip_pdhmm_set_tail_mode.restype = C.c_int
    lib.gklhip_pdhmm_compute.argtypes = [C.c_void_p, C.POINTER(CPdhmmBatch), C.c_void_p]
    lib.gklhip_pdhmm_compute.restype = C.c_int
    lib.gklhip_pdhmm_compute_cross.argtypes = [C.c_void_p, C.POINTER(CPdhmmCross), C.c_void_p]
    lib.gklhip_pdhmm_compute_cross.restype = C.c_int
    lib.gklhip_pdhmm_compute_cross_batched.argtypes = [C.c_void_p, C.POINTER(CPdhmmCross), C.c_int64, C.c_void_p]
    lib.gklhip_pdhmm_compute_cross_batched.restype = C.c_int
    lib.gklhip_pdhmm_reference_batch_pairs.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64]
    lib.gklhip_pdhmm_reference_batch_pairs.restype = C.c_int64
    lib.gklhip_pdhmm_available_memory_mb.argtypes = [C.c_int32]
    lib.gklhip_pdhmm_available_memory_mb.restype = C.c_int32
    lib.gklhip_pdhmm_done.argtypes = [C.c_void_p]
    lib.gklhip_pdhmm_done.restype = C.c_int
    lib.gklhip_pdhmm_last_kernel_ms.argtypes = [C.c_void_p]
    lib.gklhip_pdhmm_last_kernel_ms.restype = C.c_float
    lib.gklhip_pdhmm_buffer_bytes.argtypes = [C.c_void_p]
    lib.gklhip_pdhmm_buffer_bytes.restype = C.c_int64
    lib.gklhip_pdhmm_last_routing.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.gklhip_pdhmm_last_routing.restype = C.c_int
    lib.gklhip_pdhmm_get_table.argtypes = [C.c_int, C.c_void_p, C.c_int64]
    lib.gklhip_pdhmm_get_table.restype = C.c_int64
    lib.gklhip_pdhmm_last_error.restype = C.c_char_p
    if path is None:
        _pd_lib = lib
    return lib


def pdhmm_host_table(which: int) -> np.ndarray:
    lib = load_pdhmm_library()
    n = lib.gklhip_pdhmm_get_table(which, None, 0)
    a = np.empty(n, np.float64)
    lib.gklhip_pdhmm_get_table(which, a.ctypes.data, n)
    return a


def pdhmm_reference_batch_pairs(max_memory_mb: int, max_read_len: int, max_hap_len: int, total_pairs: int) -> int:
    """Pairs per batch of the reference's computeLikelihoodsNative (pdhmm/JavaData.h:83-101)."""
    return int(load_pdhmm_library().gklhip_pdhmm_reference_batch_pairs(max_memory_mb, max_read_len, max_hap_len, total_pairs))


def pdhmm_available_memory_mb(max_memory_mb: int) -> int:
    """min(maxMemoryInMB, free RAM of the host): what the reference's initNative keeps (pdhmm-implementation.h:204-235)."""
    return int(load_pdhmm_library().gklhip_pdhmm_available_memory_mb(max_memory_mb))


class PdhmmContext:
    """One gklhip_pdhmm context (= IntelPDHMM.initNative)."""

    def __init__(self, device: int = -1, fma_mode: int = 1, reference_tail: Optional[bool] = None, lib_path: Optional[str] = None):
        """fma_mode 1: bit-identical to GKL's AVX-512 PDHMM object, 0: to its AVX2 object.  reference_tail: the last
        `batch mod SIMD width` pairs of every reference batch take the scalar engine's arithmetic, as in GKL -- None:
        the library's setting (default: on; GKL_HIP_PDHMM_TAIL=vector turns it off), True / False: set it.
        lib_path: another build of the library (the all-C++ cross-check build of the tests)."""
        self.lib = load_pdhmm_library(lib_path)
        h = C.c_void_p()
        st = self.lib.gklhip_pdhmm_init(device, C.byref(h))
        if st != OK:
            self._raise(st)
        self.handle = h
        st = self.lib.gklhip_pdhmm_set_fma_mode(self.handle, int(fma_mode))
        if st != OK:
            self._raise(st)
        if reference_tail is not None:
            st = self.lib.gklhip_pdhmm_set_tail_mode(self.handle, 1 if reference_tail else 0)
            if st != OK:
                self._raise(st)

    def _raise(self, status):
        msg = (self.lib.gklhip_pdhmm_last_error() or b"").decode()
        if status == ERR_INVALID_ARG:
            raise IllegalArgumentException(msg)
        if status == ERR_OOM:
            raise OutOfMemoryError(msg)
        raise RuntimeException(msg)

    def compute(self, b) -> np.ndarray:
        keep = [np.ascontiguousarray(a, np.int8) for a in (b.hap_bases, b.hap_pdbases, b.read_bases, b.read_qual,
                                                           b.read_ins_qual, b.read_del_qual, b.gcp)]
        hl = np.ascontiguousarray(b.hap_lengths, np.int64)
        rl = np.ascontiguousarray(b.read_lengths, np.int64)
        cb = CPdhmmBatch(b.batch, b.max_hap_len, b.max_read_len, *[a.ctypes.data for a in keep],
                         hl.ctypes.data, rl.ctypes.data)
        out = np.empty(max(b.batch, 0), np.float64)
        st = self.lib.gklhip_pdhmm_compute(self.handle, C.byref(cb), out.ctypes.data)
        if st != OK:
            self._raise(st)
        return out

    def compute_cross(self, reads, haps, ref_batch_pairs: int = 0) -> np.ndarray:
        """Every read against every haplotype (IntelPDHMM.computeLikelihoods), out[r * n_haps + h].
        ref_batch_pairs: in reference-tail mode, the size of the batches the reference cuts the pair list into
        (reference_batch_pairs()); 0 = one batch.
        reads: PdhmmBatch-like with the five read arrays [n][max_read_len] + read_lengths (its haplotype side is
        ignored); haps: PdhmmBatch-like with hap_bases / hap_pdbases [n][max_hap_len] + hap_lengths."""
        keep = [np.ascontiguousarray(a, np.int8) for a in (haps.hap_bases, haps.hap_pdbases, reads.read_bases,
                                                           reads.read_qual, reads.read_ins_qual, reads.read_del_qual,
                                                           reads.gcp)]
        hl = np.ascontiguousarray(haps.hap_lengths, np.int64)
        rl = np.ascontiguousarray(reads.read_lengths, np.int64)
        cb = CPdhmmCross(reads.batch, haps.batch, haps.max_hap_len, reads.max_read_len, *[a.ctypes.data for a in keep],
                         hl.ctypes.data, rl.ctypes.data)
        out = np.empty(max(reads.batch * haps.batch, 0), np.float64)
        st = self.lib.gklhip_pdhmm_compute_cross_batched(self.handle, C.byref(cb), C.c_int64(ref_batch_pairs), out.ctypes.data)
        if st != OK:
            self._raise(st)
        return out

    def last_kernel_ms(self) -> float:
        return float(self.lib.gklhip_pdhmm_last_kernel_ms(self.handle))

    def buffer_bytes(self) -> int:
        """Device + pinned host bytes the context holds right now (they shrink again after 16 small calls)."""
        return int(self.lib.gklhip_pdhmm_buffer_bytes(self.handle))

    def last_routing(self):
        """Last cross call: its haplotypes by kernel (LDS prior table, predicate, byte-comparing); last paired call: its
        packed jobs (wavefront-loads of whole pairs) by kernel."""
        out = (C.c_int32 * 3)()
        st = self.lib.gklhip_pdhmm_last_routing(self.handle, out)
        if st != OK:
            self._raise(st)
        return int(out[0]), int(out[1]), int(out[2])

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gklhip_pdhmm_done(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------- Smith-Waterman (include/gkl_hip_sw.h)
SW_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libgklhip_sw.so")
SW_SOFTCLIP, SW_INDEL, SW_LEADING_INDEL, SW_IGNORE = 9, 10, 11, 12
_sw_lib = None


class CSwParams(C.Structure):
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("open", C.c_int32), ("extend", C.c_int32)]


def load_sw_library(path: Optional[str] = None):
    global _sw_lib
    if _sw_lib is not None and path is None:
        return _sw_lib
    p = path or SW_LIB_PATH
    try:
        import torch  # noqa: F401  (HIP runtime load order, see load_library)
    except ImportError:
        pass
    if not os.path.exists(p):
        raise RuntimeException(f"{p} is not built (hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(p)
    lib.gklhip_sw_init.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.gklhip_sw_init.restype = C.c_int
    lib.gklhip_sw_done.argtypes = [C.c_void_p]
    lib.gklhip_sw_done.restype = C.c_int
    lib.gklhip_sw_align.argtypes = [C.c_void_p, C.POINTER(CSwParams), C.c_int32, C.c_char_p, C.c_int32, C.c_char_p,
                                    C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
    lib.gklhip_sw_align.restype = C.c_int
    lib.gklhip_sw_align_batch.argtypes = [C.c_void_p, C.POINTER(CSwParams), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.gklhip_sw_align_batch.restype = C.c_int
    lib.gklhip_sw_last_kernel_ms.argtypes = [C.c_void_p]
    lib.gklhip_sw_last_kernel_ms.restype = C.c_float
    lib.gklhip_sw_last_error.restype = C.c_char_p
    if path is None:
        _sw_lib = lib
    return lib


class SwContext:
    """One gklhip_sw context (= IntelSmithWaterman.initNative)."""

    def __init__(self, device: int = -1):
        self.lib = load_sw_library()
        h = C.c_void_p()
        st = self.lib.gklhip_sw_init(device, C.byref(h))
        if st != OK:
            self._raise(st)
        self.handle = h

    def _raise(self, status):
        msg = (self.lib.gklhip_sw_last_error() or b"").decode()
        if status == ERR_INVALID_ARG:
            raise IllegalArgumentException(msg)
        if status == ERR_OOM:
            raise OutOfMemoryError(msg)
        raise RuntimeException(msg)

    def align(self, ref: bytes, alt: bytes, params, strategy: int, cigar_len: Optional[int] = None):
        """(cigar bytes, cigar_count, offset) of one pair; cigar_len defaults to the Java side's 2*max(len)."""
        ref, alt = bytes(ref), bytes(alt)
        if cigar_len is None:
            cigar_len = 2 * max(len(ref), len(alt))
        buf = C.create_string_buffer(max(cigar_len, 1))
        cnt, off = C.c_uint32(0), C.c_int32(0)
        p = CSwParams(*[int(v) for v in params])
        st = self.lib.gklhip_sw_align(self.handle, C.byref(p), int(strategy), ref, len(ref), alt, len(alt), buf,
                                      int(cigar_len), C.byref(cnt), C.byref(off))
        if st != OK:
            self._raise(st)
        return buf.raw[:cigar_len].rstrip(b"\0"), cnt.value, off.value

    @staticmethod
    def pack(refs, alts, cigar_stride: Optional[int] = None):
        """Flat arrays of gklhip_sw_align_batch for lists of byte strings (reusable across calls)."""
        n = len(refs)
        if n != len(alts):
            raise IllegalArgumentException("refs and alts differ in length")
        refs, alts = [bytes(r) for r in refs], [bytes(a) for a in alts]
        if cigar_stride is None:
            cigar_stride = 2 * max([1] + [max(len(r), len(a)) for r, a in zip(refs, alts)])
        ro = np.zeros(n + 1, np.int64)
        ao = np.zeros(n + 1, np.int64)
        np.cumsum([len(r) for r in refs], out=ro[1:])
        np.cumsum([len(a) for a in alts], out=ao[1:])
        rb = np.frombuffer(b"".join(refs) or b"\0", dtype=np.uint8)
        ab = np.frombuffer(b"".join(alts) or b"\0", dtype=np.uint8)
        return {"n": n, "ref_off": ro, "alt_off": ao, "refs": rb, "alts": ab, "cigar_stride": int(cigar_stride),
                "cells": int(sum(len(r) * len(a) for r, a in zip(refs, alts)))}

    def align_packed(self, pk, params, strategy: int):
        """One gklhip_sw_align_batch call on pack()'s arrays: (cigar byte matrix [n][stride], counts, offsets)."""
        n, stride = pk["n"], pk["cigar_stride"]
        cig = np.zeros(max(n, 1) * stride, np.uint8)
        cnt = np.zeros(max(n, 1), np.uint32)
        off = np.zeros(max(n, 1), np.int32)
        p = CSwParams(*[int(v) for v in params])
        st = self.lib.gklhip_sw_align_batch(self.handle, C.byref(p), int(strategy), n, pk["refs"].ctypes.data,
                                            pk["ref_off"].ctypes.data, pk["alts"].ctypes.data, pk["alt_off"].ctypes.data,
                                            cig.ctypes.data, stride, cnt.ctypes.data, off.ctypes.data)
        if st != OK:
            self._raise(st)
        return cig.reshape(max(n, 1), stride)[:n], cnt[:n], off[:n]

    def align_batch(self, refs, alts, params, strategy: int, cigar_stride: Optional[int] = None):
        """Lists of byte strings in, (list of cigar bytes, counts, offsets) out -- one launch for all pairs."""
        rows, cnt, off = self.align_packed(self.pack(refs, alts, cigar_stride), params, strategy)
        return [rows[k].tobytes().rstrip(b"\0") for k in range(len(refs))], cnt.copy(), off.copy()

    def last_kernel_ms(self) -> float:
        return float(self.lib.gklhip_sw_last_kernel_ms(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gklhip_sw_done(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
