"""ctypes wrappers for the Smith-Waterman checkers -- TEST INFRASTRUCTURE, not product code.

`SwOracle`     oracle/liboracle_sw.so      (our scalar restatement, oracle/sw_oracle.c)
`SwReference`  oracle/_ref/libgkl_ref_sw.so (the reference's own AVX2 / AVX-512 objects behind oracle/ref_sw_driver.cpp)
Both: align(ref, alt, (match, mismatch, open, extend), strategy, cigar_len=None) -> (status, cigar bytes (NUL-stripped),
cigar_count, offset)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SOFTCLIP, INDEL, LEADING_INDEL, IGNORE = 9, 10, 11, 12
STRATEGIES = (SOFTCLIP, INDEL, LEADING_INDEL, IGNORE)


def _call(fn, pre, ref, alt, params, strategy, cigar_len):
    ref, alt = bytes(ref), bytes(alt)
    if cigar_len is None:
        cigar_len = 2 * max(len(ref), len(alt))  # IntelSmithWaterman.java:131
    buf = C.create_string_buffer(max(cigar_len, 1))
    count, off = C.c_uint32(0), C.c_int32(0)
    st = fn(*pre, C.c_int32(params[0]), C.c_int32(params[1]), C.c_int32(params[2]), C.c_int32(params[3]),
            ref, C.c_int32(len(ref)), alt, C.c_int32(len(alt)), C.c_int32(strategy), buf, C.c_int32(cigar_len),
            C.byref(count), C.byref(off))
    return st, buf.raw[:cigar_len].rstrip(b"\0"), count.value, off.value


class SwOracle:
    def __init__(self):
        path = os.path.join(HERE, "liboracle_sw.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle_sw.so missing: run `make -C oracle oracle`")
        self.lib = C.CDLL(path)
        self.lib.sw_oracle_align.restype = C.c_int

    def align(self, ref, alt, params, strategy, cigar_len=None):
        return _call(self.lib.sw_oracle_align, (), ref, alt, params, strategy, cigar_len)


class SwReference:
    def __init__(self):
        path = os.path.join(HERE, "_ref", "libgkl_ref_sw.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/_ref/libgkl_ref_sw.so missing: run `make -C oracle ref` where /root/reference exists")
        self.lib = C.CDLL(path)
        self.lib.ref_sw_align.restype = C.c_int
        self.lib.ref_sw_has_avx512.restype = C.c_int

    def has_avx512(self):
        return bool(self.lib.ref_sw_has_avx512())

    def align(self, ref, alt, params, strategy, cigar_len=None, engine=1):
        return _call(self.lib.ref_sw_align, (C.c_int(engine),), ref, alt, params, strategy, cigar_len)
