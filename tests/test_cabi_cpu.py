"""CPU-side checks of the product library: it loads, exports every symbol the headers
declare, builds the same lookup tables as the oracle, and FAILS LOUDLY (no fallback)
when no gfx950 device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:gklhip|Java_com_intel_gkl)_\w+)\s*\(", txt)))


def test_cabi_exports_every_declared_symbol():
    from gkl_amd import native
    lib = native.load_library()
    syms = declared_symbols("gkl_hip_pairhmm.h")
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/gkl_hip_pairhmm.h but not exported"


def test_tables_bit_identical_to_oracle(oracle):
    from gkl_amd import native
    for dt in (np.float32, np.float64):
        ph = oracle.table(0, dt)
        assert native.host_table(0, dt).tobytes() == ph.tobytes()
        mm = oracle.table(1, dt)[:8256]  # quals are &127: only the first 128 triangle rows are reachable
        assert native.host_table(1, dt).tobytes() == mm.tobytes()
        assert native.host_table(2, dt).tobytes() == (ph / dt(3.0)).astype(dt).tobytes()


def test_no_silent_cpu_fallback():
    """Without a GPU, initNative's C-ABI counterpart must fail, not degrade."""
    import torch
    from gkl_amd import native
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the failure path is exercised on CPU-only hosts")
    with pytest.raises(native.RuntimeException) as e:
        native.PairHmmContext()
    assert "no" in str(e.value).lower() and "device" in str(e.value).lower()


def test_missing_library_raises(tmp_path):
    from gkl_amd import native
    with pytest.raises(native.RuntimeException):
        native.load_library(str(tmp_path / "libgklhip_pairhmm.so"))


def test_null_context_is_invalid_argument():
    from gkl_amd import native
    lib = native.load_library()
    st = lib.gklhip_compute(None, None, None)
    assert st == native.ERR_INVALID_ARG
    assert b"NULL" in lib.gklhip_last_error()
