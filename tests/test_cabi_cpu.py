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


HEADER_TO_LIBS = {
    "gkl_hip_pairhmm.h": ["libgklhip_pairhmm.so", "libgkl_pairhmm.so"],
    "gkl_pairhmm_jni.h": ["libgkl_pairhmm.so"],
    "gkl_hip_pdhmm.h": ["libgklhip_pdhmm.so", "libgkl_pdhmm.so"],
    "gkl_pdhmm_jni.h": ["libgkl_pdhmm.so"],
    "gkl_utils_jni.h": ["libgkl_utils.so"],
    "gkl_hip_sw.h": ["libgklhip_sw.so", "libgkl_smithwaterman_hip.so"],
    "gkl_sw_jni.h": ["libgkl_smithwaterman_hip.so"],
}


def test_every_header_symbol_is_exported():
    """Every function include/*.h declares is exported by the library that header describes."""
    try:
        import torch  # noqa: F401  (HIP runtime load order, see gkl_amd.native.load_library)
    except ImportError:
        pass
    assert sorted(HEADER_TO_LIBS) == sorted(f for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h"))
    for header, libs in HEADER_TO_LIBS.items():
        syms = declared_symbols(header)
        assert syms, header
        for lib_name in libs:
            lib = C.CDLL(os.path.join(ROOT, "gkl_amd", "lib", lib_name))
            for s in syms:
                assert hasattr(lib, s), f"{s} declared in include/{header} but not exported by {lib_name}"


def test_planner_packs_every_read_once_and_routes_long_reads():
    from gkl_amd import native
    from gkl_amd.synth import make_batch
    lib = native.load_library()
    b = make_batch("hc", 3000, 40, seed=4)
    lens = b.read_lens.copy()
    lens[7] = 5000  # one read too long for any chunk
    off = np.zeros(lens.size + 1, np.int64)
    off[1:] = np.cumsum(lens)
    for rpl in (4, 8):
        cap = 3000 * 64
        lanes = np.full(cap * 2, -7, np.int32)
        ng, nl = C.c_int32(), C.c_int32()
        lib.gklhip_plan_describe.restype = C.c_int
        n_chunks = lib.gklhip_plan_describe(3000, 40, off.ctypes.data_as(C.c_void_p),
                                            np.ascontiguousarray(b.hap_off).ctypes.data_as(C.c_void_p), rpl,
                                            lanes.ctypes.data_as(C.c_void_p), C.c_int64(cap), C.byref(ng), C.byref(nl))
        assert n_chunks > 0 and nl.value == 1 and ng.value >= 1
        lanes = lanes[: n_chunks * 128].reshape(n_chunks, 64, 2)
        seen = np.zeros(3000, int)
        rows = 0
        for ch in lanes:
            for lane in range(64):
                r, blk = ch[lane]
                if r < 0:
                    continue
                need = (lens[r] + rpl) // rpl  # ceil((R+1)/rpl): R rows + at least one pad row
                assert need <= 64 and 0 <= blk < need
                if blk == 0:
                    seen[r] += 1
                    assert lane + need <= 64 and np.all(ch[lane:lane + need, 0] == r)
                    assert np.array_equal(ch[lane:lane + need, 1], np.arange(need))
                    rows += lens[r]
        assert seen[7] == 0 and np.all(np.delete(seen, 7) == 1)
        assert rows / (n_chunks * 64 * rpl) > 0.88  # best-fit packing keeps the lanes full


def test_planner_job_counts_avoid_a_sparse_second_set_of_wavefront_slots():
    # jobs of the fp32 pass = chunks x haplotype groups on 4096 wavefront slots (1024 SIMDs x 4).  Mid-size and tall
    # batches get about three sets of graded jobs; a count just above one set is cut back to one set (a handful of
    # equal-length jobs in a second set doubled the kernel time: docs/NOTES.md 23, 24); big batches keep full-size groups.
    from gkl_amd import native
    from gkl_amd.synth import make_batch
    lib = native.load_library()
    lib.gklhip_plan_describe.restype = C.c_int

    def jobs(n_reads, n_haps, rpl=8):
        b = make_batch("hc", n_reads, n_haps, seed=11)
        ng = C.c_int32()
        n_chunks = lib.gklhip_plan_describe(n_reads, n_haps, np.ascontiguousarray(b.read_off).ctypes.data_as(C.c_void_p),
                                            np.ascontiguousarray(b.hap_off).ctypes.data_as(C.c_void_p), rpl, None, C.c_int64(0),
                                            C.byref(ng), None)
        assert n_chunks > 0
        return n_chunks, ng.value

    for shape in ((100, 10), (150, 30), (400, 40), (250, 128), (450, 100), (500, 128), (1000, 50), (2000, 30), (5000, 16), (5000, 8)):
        chunks, groups = jobs(*shape)
        n = chunks * groups
        assert groups <= shape[1]
        assert not (4096 < n <= 4096 * 1.3), (shape, chunks, groups)       # never a set and a bit
        if shape[0] * shape[1] >= 30000:
            assert n > 1.5 * 4096, (shape, chunks, groups)                  # two to three sets of graded jobs
    # a GATK-sized region: one haplotype per group (as many short jobs as there are)
    assert jobs(100, 10, rpl=4)[1] >= 8
    # the bench batch keeps its ~2048-column groups (+ the halving tail)
    chunks, groups = jobs(10000, 128)
    assert 18 <= groups <= 26 and chunks > 2500


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


def test_pdhmm_reference_batch_size_formula():
    """gklhip_pdhmm_reference_batch_pairs = the reference's batchSize of computeLikelihoodsNative
    (pdhmm/JavaData.h:86-101): min(totalPairs, maxMemory / ((maxRead * 5 + maxHap * 2) + 8 + 16)), maxMemory in MB capped
    by the host's free RAM at initNative (pdhmm-implementation.h:204-235) -- no device needed."""
    from gkl_amd import native
    f = native.pdhmm_reference_batch_pairs
    per = 151 * 5 + 300 * 2 + 8 + 16
    assert f(1, 151, 300, 10**9) == 1024 * 1024 // per
    assert f(1, 151, 300, 17) == 17                      # fewer pairs than a batch holds
    assert f(0, 151, 300, 17) == 0 and f(1, 0, 300, 17) == 0 and f(1, 151, 300, 0) == 0
    # the free-RAM cap is applied once, by initNative (gklhip_pdhmm_available_memory_mb), not per call
    assert native.pdhmm_available_memory_mb(1) == 1 and native.pdhmm_available_memory_mb(0) == 0
    assert 0 < native.pdhmm_available_memory_mb(10**9) < 10**9   # more than any host has free
    assert f(10**6, 151, 300, 10**12) == 10**6 * 1024 * 1024 // per


def test_generated_asm_header_is_what_its_generator_writes(tmp_path):
    """gkl_amd/csrc/pairhmm_fwd_asm.h and pdhmm_plain_asm.h (the whole-job asm programs of the forward kernels) are generator
    output: git-ignored, written by `make -C gkl_amd/csrc` (python3 is a build dependency; __graft_entry__.build() runs the
    make).  What lies in the build tree must be exactly what the generators write today -- a stale header would mean the
    libraries under test were not built from this tree's generators."""
    import subprocess
    import sys
    for h in ("pairhmm_fwd_asm.h", "pdhmm_plain_asm.h"):
        if not os.path.exists(os.path.join(ROOT, "gkl_amd", "csrc", h)):
            pytest.skip(f"{h} not generated yet: run `make -C gkl_amd/csrc` (or __graft_entry__.build()) first")
    out = tmp_path / "pairhmm_fwd_asm.h"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_fwd_asm.py"), str(out)], check=True, capture_output=True)
    assert out.read_bytes() == open(os.path.join(ROOT, "gkl_amd", "csrc", "pairhmm_fwd_asm.h"), "rb").read(), \
        "regenerate: python tools/gen_fwd_asm.py"
    # ... and the PDHMM table kernel's plain-step run
    out = tmp_path / "pdhmm_plain_asm.h"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_pdhmm_asm.py"), str(out)], check=True, capture_output=True)
    assert out.read_bytes() == open(os.path.join(ROOT, "gkl_amd", "csrc", "pdhmm_plain_asm.h"), "rb").read(), \
        "regenerate: python tools/gen_pdhmm_asm.py"


def test_generated_fp32_programs_keep_the_bank_rule_and_their_registers():
    """Static checks of the generated asm (no GPU): (1) the VGPR file of gfx950 has two banks, even and odd registers, and
    a three-source fp32 op whose sources all sit in one bank issues at half rate (docs/NOTES.md, round 3) -- no v_fmac_f32 /
    v_fma_f32 of the fp32 programs may be monochrome; (2) every register a program names lies inside its configuration's
    map, i.e. below the VGPR budget its kernels are launched with (128 for fp32, 256 for fp64 at 8 / 10 rows, 168 / 128 for
    the narrow fp64 ones)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_fwd_asm", os.path.join(ROOT, "tools", "gen_fwd_asm.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    budgets = {("f32", 8): 128, ("f32", 4): 128, ("f32", 2): 128, ("f64", 10): 256, ("f64", 8): 256, ("f64", 6): 168, ("f64", 4): 128,
               ("f64", 2): 128}
    for ((kind, R), budget), fma in [(kb, f) for kb in budgets.items() for f in (True, False)]:   # both arithmetics (the unfused "...n" programs: round 5)
        c = g.Cfg(f"{kind}r{R}", kind == "f64", R, fma)
        assert c.name.endswith("n") == (not fma)
        for wide in ((False, True) if R >= 8 else (False,)):
            prog = g.program(c, wide)
            regs = set()
            for ins in prog:
                for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", ins):
                    if m.group(3) is not None:
                        regs.add(int(m.group(3)))
                    else:
                        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
                if kind == "f32" and (ins.startswith("v_fmac_f32 ") or ins.startswith("v_fma_f32 ")):
                    ops = [int(x) for x in re.findall(r"\bv(\d+)\b", ins)]
                    srcs = ops if ins.startswith("v_fmac_f32 ") else ops[1:]   # fmac reads its destination as the third source
                    assert len({r % 2 for r in srcs}) == 2, f"monochrome three-source op in {c.name}: {ins}"
            assert max(regs) < budget, (c.name, wide, max(regs), budget)
            assert max(regs) <= (c.last if fma else max(c.last, max(c.TN) + c.w - 1)), (c.name, max(regs), c.last)
            if not fma:   # the unfused arithmetic has no fused operation anywhere
                assert not any(ins.startswith(("v_fma_f", "v_fmac_f")) for ins in prog), c.name
            # labels are unique within the one asm statement
            labels = [ins[:-1] for ins in prog if re.fullmatch(r"\d+:", ins)]
            assert len(labels) == len(set(labels)), (c.name, "duplicate label")
