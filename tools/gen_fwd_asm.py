#!/usr/bin/env python3
"""Generates gkl_amd/csrc/pairhmm_fwd_asm.h: whole-job asm drivers of the forward recurrence,
  fwd_asm_run_f32r8   fp32, 8 rows per lane   (pairhmm_fwd_stream_kernel<float, 8, true> and the other float/8/FMA kernels)
  fwd_asm_run_f64r10  fp64, 10 rows per lane  (pairhmm_fwd_jobs_kernel<double, 10, true>, ..stream_kernel<double, 10, true>)
  fwd_asm_run_f64r8   fp64, 8 rows per lane   (the wide long-read kernel: 16 KB of prior planes per wavefront instead of 20)
  fwd_asm_run_f32r4 / f32r2 / f64r6 / f64r4 / f64r2   the narrow kernels of small and mid-size calls
Arithmetic (operation order, FMA pattern) = WaveJob::advance, i.e. the reference's compute_full_prob with the gcc-11
contraction of the AVX-512 object (reference avx-pairhmm-template.h:208-223); tests/test_gpu_parity.py pins the asm drivers
to the C++ steps and to the oracle bit for bit.

Structure (round 4).  ONE asm statement per haplotype, with its loops inside:
    general steps (the window in which the previous haplotype's separator travels down the array; the fill)
 -> 8-column fast blocks (every lane inside a haplotype)
 -> general steps (the < 8 leftover columns and the separator itself).
The C++ around it is scalar bookkeeping once per haplotype.  (Round 3 had the fast block only, as an asm statement per 8
columns inside a C++ loop, and everything else as compiler-allocated C++ steps; several asm statements sharing pinned
registers make the register allocator copy the whole state between them -- measured: +20 % -- hence one statement.)

What round 3 measured on this chip and this file keeps (tools/gen_ubench_banks2.py, docs/NOTES.md):
  * two VGPR banks (even / odd registers): a three-source fp32 op with all sources in one bank issues at half rate --
    M, X, Y and accumulators in even registers, pGAPM / pXX / pMM / pMX in odd ones (fp32 map);
  * DPP ops grouped back to back behind one `s_nop 1`;  VOP2 encodings where they exist;  no SGPR sources in the hot ops.

A lane's stream entry is a haplotype base code or a SEPARATOR-TYPE word (sign bit): the separator of a haplotype, or the
pre-roll word a lane holds before its first column and behind the last separator (fill and drain).  The general step is
the fast step's arithmetic plus, for lanes on a separator-type entry, M, Y and the running sums ANDed with a per-lane mask
(0 there, ~0 elsewhere: bitwise, so Inf / NaN of an overflowed pair die too -- no EXEC games, no selects) and, behind a
scalar branch taken only when a lane holding the END of a read is on the separator in flight, the store of the pair's sum.
The step that FEEDS a separator also moves the next haplotype's Y0 into the pad row of every read's first lane.

A job whose haplotypes are all longer than the array is deep (skew_max + 1 columns; stream order is by ascending length,
so the first one decides) has one separator in flight at a time and its output column in a scalar register; otherwise
(`multi`) a storing lane reads hap_orig[k] for the k its separator carries -- two more scalar instructions in the store
branch of the first kind, a dependent load per storing step in the second.
Preconditions (checked by the caller, WaveJob::run): fp64: no haplotype of the job contains an 'N' (four prior planes), no
packed output.
"""
import sys

DPP = "wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
U = 8  # columns per fast block

# scalar registers owned by the asm program (clobbered): entries of a block, loop state
S_E0 = 72          # s[72:79]: eight stream entries (s_load_dwordx8: 4-aligned)
S_LIM, S_ORIG, S_CNT, S_PH, S_BT, S_TMP = "s80", "s81", "s82", "s83", "s84", "s85"
S_SAVE = "s[86:87]"
S_SRC = 88         # s[88:89]: the stream pointer (operand src, pinned there)
S_T, S_NEED = "s90", "s91"   # wide variant: steps this wavefront has completed (operand st, pinned there); scratch
S_CLOB = [f"s{i}" for i in range(72, 88)]
RING = 64          # wide variant: slots of a hand-off ring
OOB_PLANE = 512    # fp32 general step: a prior "plane" beyond the CU's 160 KB of LDS whatever the plane size (512 x 512 B = 256 KB)


class Cfg:
    def __init__(self, name, f64, R, fma=True):
        # fma = False: the UNFUSED arithmetic of the reference's AVX translation unit (what GKL computes on a host without
        # AVX-512, reference IntelPairHmm.cc:106-113; gklhip_config.fma_mode 0): every multiply and add rounds on its own --
        # 12 VALU operations per cell instead of 8.  Programs of that kind carry an "n" behind their name.
        self.name, self.f64, self.R, self.fma = name + ("" if fma else "n"), f64, R, fma
        self.w = 2 if f64 else 1
        if not f64:
            # round-3 map: E = even, O = odd registers
            self.M = lambda s: 16 + 2 * s
            self.GAPM = lambda s: 17 + 2 * s
            self.X = lambda s: 32 + 2 * s
            self.PXX = lambda s: 33 + 2 * s
            self.YA = lambda s: 48 + 2 * s
            self.PMM = lambda s: 49 + 2 * s
            self.YB = lambda s: 64 + 2 * s      # Y ping-pong partner (fast block), product temporaries (general step)
            self.PMX = lambda s: 65 + 2 * s
            self.PMY = lambda s: 80 + s
            self.PR = lambda s: 88 + s
            self.RS = [(96, 97, 98), (100, 101, 102)]
            (self.SM, self.SX, self.ENT, self.EAB, self.LMASK, self.DIRECT, self.NDIRECT, self.LOFF, self.ADDR, self.NSEP, self.VAL,
             self.OUTIDX, self.PADSLOT, self.Y0N, self.A, self.P) = 104, 105, 106, 107, 108, 109, 110, 111, 112, 113, 114, 115, 116, 117, 118, 120
            self.KREG = None
            self.WADDR, self.FV = 122, 123      # wide variant: ring address, flag value
            self.last = 123
            self.TN = [124, 125, 126, 127]      # unfused arithmetic: product temporaries (two per row, two rows in flight)
            # prior table [code][plane][lane][min(R, 4) floats] (WaveJob::kVecBytes, kPlanes, kRowBytes): 2 rows per lane read
            # one ds_read_b64 per step, 4 rows one ds_read_b128, 8 rows two
            assert R in (2, 4, 8)
            self.vec_bytes = 8 if R == 2 else 16
            self.planes = 2 if R == 8 else 1
            self.plane_stride = 64 * self.vec_bytes
            self.code_shift = {2: 9, 4: 10, 8: 11}[R]   # log2(kRowBytes)
            self.codes = 5
        else:
            b, W2 = (8 if R <= 6 else 16), 2 * R              # ten-row map: v16..v240; eight rows: v16..v204; six: v8..v160 (three
                                                             # wavefronts per SIMD in the one-pair-per-wavefront kernels), four:
                                                             # v8..v124 (four per SIMD)
            self.M = lambda s: b + 2 * s
            self.X = lambda s: b + W2 + 2 * s
            self.YA = lambda s: b + 2 * W2 + 2 * s
            self.PMM = lambda s: b + 3 * W2 + 2 * s
            self.GAPM = lambda s: b + 4 * W2 + 2 * s
            self.PMX = lambda s: b + 5 * W2 + 2 * s
            self.PXX = lambda s: b + 6 * W2 + 2 * s
            self.PMY = lambda s: b + 7 * W2 + 2 * s
            self.PR = lambda s: b + 8 * W2 + 2 * s         # R / 2 ds_read_b128 (v176..v195 at ten rows: 4-aligned)
            t0 = b + 9 * W2
            self.YB = lambda s: t0 + 2 * (s % 4)             # four rotating product temporaries
            r0 = t0 + 8
            self.RS = [(r0, r0 + 2, r0 + 4), (r0 + 6, r0 + 8, r0 + 10)]
            self.SM, self.SX = r0 + 12, r0 + 14
            m0 = r0 + 16
            (self.ENT, self.EAB, self.LMASK, self.DIRECT, self.NDIRECT, self.LOFF, self.ADDR, self.NSEP) = [m0 + i for i in range(8)]
            self.VAL, self.OUTIDX, self.PADSLOT, self.Y0N, self.A, self.P, self.KREG = m0 + 8, m0 + 10, m0 + 11, m0 + 12, m0 + 14, m0 + 16, m0 + 18
            self.WADDR, self.FV = m0 + 19, m0 + 20
            self.last = m0 + 20
            self.TN = [t0, t0 + 2, t0 + 4, t0 + 6]   # unfused arithmetic: the four rotating product temporaries, two per row
            assert self.PR(0) % 4 == 0 and self.last < 256
            self.codes, self.planes, self.plane_stride, self.code_shift = 4, R // 2, 1024, None   # code * kRowBytes via KREG
            self.vec_bytes = 16

    # operand text
    def v(self, n):
        return f"v[{n}:{n + 1}]" if self.f64 else f"v{n}"

    def v32(self, n):
        return f"v{n}"


def fp(c, op):
    return f"v_{op}_f64" if c.f64 else f"v_{op}_f32"


def dpp_recv(c, dst, src):
    """dst = (bottom row value of the lane above) & lmask: one v_and_b32_dpp per 32-bit half"""
    return [f"v_and_b32_dpp v{dst + h}, v{src + h}, v{c.LMASK} {DPP}" for h in range(c.w)]


def and_mask(c, reg, mask):
    """value & per-lane mask (0 or ~0).  fp64: the HIGH half only -- sign, exponent and the top of the mantissa go, what is
    left is a denormal, and this library's kernels run with fp64 denormals flushed (Makefile: -fdenormal-fp-math=
    preserve-sign, like MXCSR.FTZ on the host), so every consumer (all are fp64 arithmetic) reads it as zero."""
    h = c.w - 1
    return [f"v_and_b32 v{reg + h}, v{reg + h}, v{mask}"]


def mov(c, dst, src):
    return [f"v_mov_b32 v{dst + h}, v{src + h}" for h in range(c.w)]


def prior_loads(c, code_reg):
    """LDS address of the lane's prior rows for base code `code_reg`, and the reads"""
    o = []
    if c.f64:
        o.append(f"v_mad_u32_u24 v{c.ADDR}, v{code_reg}, v{c.KREG}, v{c.LOFF}")
    else:
        o.append(f"v_lshl_add_u32 v{c.ADDR}, v{code_reg}, {c.code_shift}, v{c.LOFF}")
    for pl in range(c.planes):
        d = c.PR(0) + 4 * pl
        off = f" offset:{pl * c.plane_stride}" if pl else ""
        if c.vec_bytes == 8:
            o.append(f"ds_read_b64 v[{d}:{d + 1}], v{c.ADDR}{off}")
        else:
            o.append(f"ds_read_b128 v[{d}:{d + 3}], v{c.ADDR}{off}")
    return o


def recurrence_unfused(c, yo, yn, d, r, general):
    """The same step in the reference's UNFUSED arithmetic (WaveJob::advance with FMA = false; avx-pairhmm-template.h:208-223
    as the AVX translation unit computes it): M-inner = ((Md * pMM + Xd * pGAPM) + Yd * pGAPM), Y = M * pMY + Y * pXX,
    X = M' * pMX + X' * pXX -- every product and every sum its own instruction (12 per cell), two product temporaries per
    row, alternating between two sets so that consecutive rows do not wait for each other's."""
    R, o = c.R, []
    dM, dX, dY = d
    rM, rX, rY = r
    for s in range(R - 1, -1, -1):
        md, xd, yd = (c.M(s - 1), c.X(s - 1), yo(s - 1)) if s > 0 else (dM, dX, dY)
        t1, t2 = c.TN[2 * (s % 2)], c.TN[2 * (s % 2) + 1]
        o.append(f"{fp(c, 'mul')} {c.v(t1)}, {c.v(xd)}, {c.v(c.GAPM(s))}")     # (before X(s) is overwritten: xd is X(s - 1), not X(s))
        o.append(f"{fp(c, 'mul')} {c.v(c.X(s))}, {c.v(md)}, {c.v(c.PMM(s))}")
        o.append(f"{fp(c, 'mul')} {c.v(t2)}, {c.v(c.M(s))}, {c.v(c.PMY(s))}")
        o.append(f"{fp(c, 'add')} {c.v(c.X(s))}, {c.v(c.X(s))}, {c.v(t1)}")
        o.append(f"{fp(c, 'mul')} {c.v(t1)}, {c.v(yd)}, {c.v(c.GAPM(s))}")
        o.append(f"{fp(c, 'mul')} {c.v(yn(s))}, {c.v(yo(s))}, {c.v(c.PXX(s))}")
        o.append(f"{fp(c, 'add')} {c.v(c.X(s))}, {c.v(c.X(s))}, {c.v(t1)}")
        o.append(f"{fp(c, 'add')} {c.v(yn(s))}, {c.v(t2)}, {c.v(yn(s))}")
    o.append("s_waitcnt lgkmcnt(0)")
    if general and not c.f64:
        for s in range(R):
            o.append(f"v_mul_legacy_f32 {c.v(c.M(s))}, {c.v(c.X(s))}, {c.v(c.PR(s))}")
    else:
        for s in range(R):
            o.append(f"{fp(c, 'mul')} {c.v(c.M(s))}, {c.v(c.X(s))}, {c.v(c.PR(s))}")
        if general:
            for s in range(R):
                o += and_mask(c, c.M(s), c.NSEP)
    for s in range(R):
        ms, xs = (c.M(s - 1), c.X(s - 1)) if s else (rM, rX)
        t1 = c.TN[2 * (s % 2)]
        o.append(f"{fp(c, 'mul')} {c.v(t1)}, {c.v(xs)}, {c.v(c.PXX(s))}")
        o.append(f"{fp(c, 'mul')} {c.v(c.X(s))}, {c.v(ms)}, {c.v(c.PMX(s))}")
        o.append(f"{fp(c, 'add')} {c.v(c.X(s))}, {c.v(c.X(s))}, {c.v(t1)}")
    o.append(f"{fp(c, 'add')} {c.v(c.SM)}, {c.v(c.SM)}, {c.v(c.M(R - 1))}")
    return o


def recurrence(c, yo, yn, d, r, general):
    """M-inner + Y bottom-up, prior multiply, X column top-down, running sum of M.  yo/yn: Y source / destination register
    functions (the same one: in place through v_fma with a product temporary)."""
    if not c.fma:
        return recurrence_unfused(c, yo, yn, d, r, general)
    R, o = c.R, []
    dM, dX, dY = d
    rM, rX, rY = r
    inplace = yo is yn
    for s in range(R - 1, -1, -1):
        md, xd, yd = (c.M(s - 1), c.X(s - 1), yo(s - 1)) if s > 0 else (dM, dX, dY)
        t = c.YB(s) if inplace else yn(s)
        o.append(f"{fp(c, 'mul')} {c.v(c.X(s))}, {c.v(md)}, {c.v(c.PMM(s))}")
        o.append(f"{fp(c, 'mul')} {c.v(t)}, {c.v(c.M(s))}, {c.v(c.PMY(s))}")
        o.append(f"{fp(c, 'fmac')} {c.v(c.X(s))}, {c.v(xd)}, {c.v(c.GAPM(s))}")
        if inplace:
            o.append(f"{fp(c, 'fma')} {c.v(yn(s))}, {c.v(yo(s))}, {c.v(c.PXX(s))}, {c.v(t)}")   # = fmac into the product: one rounding
        else:
            o.append(f"{fp(c, 'fmac')} {c.v(yn(s))}, {c.v(yo(s))}, {c.v(c.PXX(s))}")
        o.append(f"{fp(c, 'fmac')} {c.v(c.X(s))}, {c.v(yd)}, {c.v(c.GAPM(s))}")
    o.append("s_waitcnt lgkmcnt(0)")
    if general and not c.f64:
        # no haplotype base in the column: the lane's prior rows were read from beyond the LDS allocation, i.e. are 0
        # (prior_loads), and v_mul_legacy_f32 makes 0 * anything = 0, also for the Inf / NaN of an overflowed pair
        for s in range(R):
            o.append(f"v_mul_legacy_f32 {c.v(c.M(s))}, {c.v(c.X(s))}, {c.v(c.PR(s))}")
    else:
        for s in range(R):
            o.append(f"{fp(c, 'mul')} {c.v(c.M(s))}, {c.v(c.X(s))}, {c.v(c.PR(s))}")
        if general:
            for s in range(R):
                o += and_mask(c, c.M(s), c.NSEP)     # no haplotype base in the column: M = 0 (bitwise: NaN / Inf too)
    o.append(f"{fp(c, 'mul')} {c.v(c.X(0))}, {c.v(rM)}, {c.v(c.PMX(0))}")
    o.append(f"{fp(c, 'fmac')} {c.v(c.X(0))}, {c.v(rX)}, {c.v(c.PXX(0))}")
    for s in range(1, R):
        o.append(f"{fp(c, 'mul')} {c.v(c.X(s))}, {c.v(c.M(s - 1))}, {c.v(c.PMX(s))}")
        o.append(f"{fp(c, 'fmac')} {c.v(c.X(s))}, {c.v(c.X(s - 1))}, {c.v(c.PXX(s))}")
    o.append(f"{fp(c, 'add')} {c.v(c.SM)}, {c.v(c.SM)}, {c.v(c.M(R - 1))}")
    return o


def wide_handoff(c, target, y_reg, lab):
    """wide variant, end of a step: the bottom row of lane 63 goes to the next wavefront's ring slot of this step, the
    previous wavefront's slot of this step lands in lane 0's row-above registers `target` (= the DPP fetch's target)."""
    R = c.R
    sz = 8 if c.f64 else 4
    slot_shift = 5 if c.f64 else 4
    o = []
    o.append("s_cmp_eq_u32 %[has_out], 0")
    o.append(f"s_cbranch_scc1 {lab}f")
    o.append(f"s_and_b32 {S_NEED}, {S_T}, {RING - 1}")
    o.append(f"s_lshl_b32 {S_NEED}, {S_NEED}, {slot_shift}")
    o.append(f"s_add_u32 {S_NEED}, {S_NEED}, %[ring_out]")
    o.append(f"v_mov_b32 v{c.WADDR}, {S_NEED}")
    o.append("s_mov_b32 exec_lo, 0")
    o.append("s_brev_b32 exec_hi, 1")                          # lane 63 alone
    w = "ds_write_b64" if c.f64 else "ds_write_b32"
    for k, reg in enumerate((c.M(R - 1), c.X(R - 1), y_reg)):
        off = f" offset:{k * sz}" if k else ""
        o.append(f"{w} v{c.WADDR}, {c.v(reg)}{off}")
    o.append("s_mov_b64 exec, -1")
    o.append(f"{lab}:")
    o.append("s_cmp_eq_u32 %[has_in], 0")
    o.append(f"s_cbranch_scc1 {lab + 1}f")
    o.append(f"s_and_b32 {S_NEED}, {S_T}, {RING - 1}")
    o.append(f"s_lshl_b32 {S_NEED}, {S_NEED}, {slot_shift}")
    o.append(f"s_add_u32 {S_NEED}, {S_NEED}, %[ring_in]")
    o.append(f"v_mov_b32 v{c.WADDR}, {S_NEED}")
    o.append("s_mov_b64 exec, 1")                              # lane 0 alone
    if c.f64:
        for k in range(3):
            off = f" offset:{k * 8}" if k else ""
            o.append(f"ds_read_b64 {c.v(target[k])}, v{c.WADDR}{off}")
    else:
        o.append(f"ds_read_b96 v[{target[0]}:{target[0] + 2}], v{c.WADDR}")
    o.append("s_mov_b64 exec, -1")
    o.append(f"{lab + 1}:")
    o.append(f"s_add_u32 {S_T}, {S_T}, 1")
    return o


def wide_wait(c, lab):
    """wide variant, before a group of up to 8 steps: the previous wavefront must have finished the steps whose rows this
    group fetches; the next wavefront must have fetched the ring slots this group overwrites."""
    o = []
    o.append("s_cmp_eq_u32 %[has_in], 0")
    o.append(f"s_cbranch_scc1 {lab + 1}f")
    o.append(f"s_add_u32 {S_NEED}, {S_T}, 8")
    o.append(f"s_min_u32 {S_NEED}, {S_NEED}, %[t_end]")
    o.append(f"{lab}:")
    o.append(f"v_mov_b32 v{c.FV}, %[flag_prod]")
    o.append(f"ds_read_b32 v{c.FV}, v{c.FV}")
    o.append("s_waitcnt lgkmcnt(0)")
    o.append(f"v_readfirstlane_b32 {S_TMP}, v{c.FV}")
    o.append(f"s_cmp_ge_u32 {S_TMP}, {S_NEED}")
    o.append(f"s_cbranch_scc1 {lab + 1}f")
    o.append("s_sleep 1")
    o.append(f"s_branch {lab}b")
    o.append(f"{lab + 1}:")
    o.append("s_cmp_eq_u32 %[has_out], 0")
    o.append(f"s_cbranch_scc1 {lab + 3}f")
    o.append(f"s_add_u32 {S_NEED}, {S_T}, 8")
    o.append(f"{lab + 2}:")
    o.append(f"v_mov_b32 v{c.FV}, %[flag_cons]")
    o.append(f"ds_read_b32 v{c.FV}, v{c.FV}")
    o.append("s_waitcnt lgkmcnt(0)")
    o.append(f"v_readfirstlane_b32 {S_TMP}, v{c.FV}")
    o.append(f"s_add_u32 {S_TMP}, {S_TMP}, {RING}")
    o.append(f"s_cmp_ge_u32 {S_TMP}, {S_NEED}")
    o.append(f"s_cbranch_scc1 {lab + 3}f")
    o.append("s_sleep 1")
    o.append(f"s_branch {lab + 2}b")
    o.append(f"{lab + 3}:")
    return o


def wide_publish(c):
    """wide variant, after a group: tell the neighbours how many steps this wavefront has completed"""
    return [f"v_mov_b32 v{c.FV}, {S_T}", f"v_mov_b32 v{c.WADDR}, %[flag_own]", "s_mov_b64 exec, 1",
            f"ds_write_b32 v{c.WADDR}, v{c.FV}", "s_mov_b64 exec, -1"]


def fast_step(c, u, last, e, wide=False, lab=0):
    """one stream column, every lane inside a haplotype.  fp32: Y ping-pongs between Ya and Yb (VOP2 v_fmac); fp64: in place."""
    R = c.R
    if c.f64:
        yo = yn = c.YA
    else:
        yo, yn = (c.YA, c.YB) if u % 2 == 0 else (c.YB, c.YA)
    r, d = c.RS[u % 2], c.RS[(u + 1) % 2]   # r: the row above at THIS column; d: at the previous one, overwritten by this step's fetch
    o = [f"v_and_or_b32 v{c.ENT}, {e}, v{c.DIRECT}, v{c.EAB}"]
    o += prior_loads(c, c.ENT)
    o += recurrence(c, yo, yn, d, r, False)
    o.append("s_nop 1")
    if not last:
        o.append(f"v_and_b32_dpp v{c.EAB}, v{c.ENT}, v{c.NDIRECT} {DPP}")
    o += dpp_recv(c, d[0], c.M(R - 1))
    o += dpp_recv(c, d[2], yn(R - 1))
    o += dpp_recv(c, d[1], c.X(R - 1))
    o.append(f"{fp(c, 'add')} {c.v(c.SX)}, {c.v(c.SX)}, {c.v(c.X(R - 1))}")
    if wide:
        o += wide_handoff(c, d, yn(R - 1), lab)
    return o


def fast_block(c, ents, wide=False):
    o = ["s_nop 1", f"v_and_b32_dpp v{c.EAB}, v{c.ENT}, v{c.NDIRECT} {DPP}"]
    for u in range(U):
        o += fast_step(c, u, u == U - 1, ents[u], wide, 300 + 4 * u)
    return o


def general_step(c, e, lab, wide=False):
    """one stream column, any entry kind; parity-neutral (Y in Ya in place, row-above set in RS[0], diagonal set in RS[1]).
    `lab`: base of this copy's local labels.  Scalar state: S_CNT steps left in this run (counted here), S_BT = the value of
    S_CNT at which the step feeds the haplotype's separator (0xffffffff: never), S_ORIG the output column of the separator in flight."""
    R = c.R
    r, d = c.RS[0], c.RS[1]
    L = lambda k: str(lab + k)
    o = []
    # the step that feeds the separator: from here on lanes meet THIS haplotype's separator (the previous one has left the array)
    o.append(f"s_cmp_eq_u32 {S_CNT}, {S_BT}")
    o.append(f"s_cselect_b32 {S_ORIG}, %[orig], {S_ORIG}")
    o += ["s_nop 1", f"v_and_b32_dpp v{c.EAB}, v{c.ENT}, v{c.NDIRECT} {DPP}"]
    o.append(f"v_and_or_b32 v{c.ENT}, {e}, v{c.DIRECT}, v{c.EAB}")
    o.append(f"v_not_b32 v{c.NSEP}, v{c.ENT}")
    o.append(f"v_ashrrev_i32 v{c.NSEP}, 31, v{c.NSEP}")      # ~0: haplotype base in this column, 0: separator-type entry
    if c.f64:
        o.append(f"v_and_b32 v{c.VAL}, v{c.ENT}, v{c.NSEP}") # base code (plane 0 for separator-type entries: any valid address)
    else:
        # base code; OOB_PLANE for separator-type entries: that plane starts beyond the CU's whole LDS, and a DS read beyond
        # the workgroup's allocation returns 0 (ISA manuals since GCN3; tools/ubench_lds_oob.hip checks it on this chip) --
        # the zero prior the column needs, without touching the eight products afterwards
        o.append(f"v_min_u32 v{c.VAL}, {OOB_PLANE}, v{c.ENT}")
    o += prior_loads(c, c.VAL)
    o += recurrence(c, c.YA, c.YA, d, r, True)
    o.append(f"{fp(c, 'add')} {c.v(c.SX)}, {c.v(c.SX)}, {c.v(c.X(R - 1))}")
    for k in range(3):
        o += mov(c, d[k], r[k])
    # the pair's result: lanes that hold the LAST row of a read (%[outmask]) and are on the separator in flight
    o.append(f"{fp(c, 'add')} {c.v(c.VAL)}, {c.v(c.SM)}, {c.v(c.SX)}")
    # a haplotype's separator: any entry below the pre-roll words (signed; S_LIM = kEntNoEmit).  With one separator in the
    # array at a time its output column is S_ORIG; with several (%[multi]: a haplotype no longer than the array is deep)
    # every lane looks its own up by the stream-order index its separator carries
    o.append(f"v_cmp_gt_i32 vcc, {S_LIM}, v{c.ENT}")
    o.append("s_and_b64 vcc, vcc, %[outmask]")                # (SCC = some lane stores)
    o.append(f"s_cbranch_scc0 {L(1)}f")
    o.append(f"s_and_saveexec_b64 {S_SAVE}, vcc")
    o.append(f"v_add_u32 v{c.Y0N}, {S_ORIG}, v{c.OUTIDX}")   # pair index (Y0N doubles as the index temporary)
    o.append("s_cmp_eq_u32 %[multi], 0")
    o.append(f"s_cbranch_scc1 {L(4)}f")
    o.append(f"v_and_b32 v{c.Y0N}, 0x3fffffff, v{c.ENT}")
    o.append(f"v_mad_u64_u32 v[{c.A}:{c.A + 1}], vcc, v{c.Y0N}, 4, %[haporig]")
    o.append(f"global_load_dword v{c.Y0N}, v[{c.A}:{c.A + 1}], off")
    o.append("s_waitcnt vmcnt(0)")
    o.append(f"v_add_u32 v{c.Y0N}, v{c.Y0N}, v{c.OUTIDX}")
    o.append(f"{L(4)}:")
    o.append(f"v_mad_u64_u32 v[{c.A}:{c.A + 1}], vcc, v{c.Y0N}, {8 if c.f64 else 4}, %[raw]")
    o.append(f"global_store_dword{'x2' if c.f64 else ''} v[{c.A}:{c.A + 1}], {c.v(c.VAL)}, off")
    if not c.f64:
        o.append("s_cmp_eq_u64 %[packed], 0")
        o.append(f"s_cbranch_scc1 {L(2)}f")
        o.append(f"v_cmp_gt_f32 vcc, %[minacc], v{c.VAL}")   # IntelPairHmm.cc:159: sum < 1e-28f -> fp64 pass (NaN stays fp32)
        o.append(f"v_cndmask_b32_e64 v{c.P}, v{c.VAL}, 0, vcc")
        o.append(f"v_cndmask_b32_e64 v{c.P + 1}, -1, 0, vcc")
        o.append(f"v_mad_u64_u32 v[{c.A}:{c.A + 1}], vcc, v{c.Y0N}, 8, %[packed]")
        o.append(f"global_store_dwordx2 v[{c.A}:{c.A + 1}], v[{c.P}:{c.P + 1}], off")
        o.append(f"{L(2)}:")
    o.append(f"s_mov_b64 exec, {S_SAVE}")
    o.append(f"{L(1)}:")
    for s in range(R):
        o += and_mask(c, c.YA(s), c.NSEP)                     # column-0 state of the next haplotype: Y = 0 ...
    o += and_mask(c, c.SM, c.NSEP)
    o += and_mask(c, c.SX, c.NSEP)
    # ... and, in the step that feeds the separator, Y0 of the next haplotype in the pad row of every read's first lane
    # (those lanes take the entry directly, so they are on the separator right now; PADSLOT is -1 in every other lane)
    o.append(f"s_cmp_lg_u32 {S_CNT}, {S_BT}")
    o.append(f"s_cbranch_scc1 {L(3)}f")
    if c.f64:
        o.append(f"v_mov_b32 v{c.Y0N}, %[y0n_lo]")
        o.append(f"v_mov_b32 v{c.Y0N + 1}, %[y0n_hi]")
    else:
        o.append(f"v_mov_b32 v{c.Y0N}, %[y0n]")
    for s in range(R):
        o.append(f"v_cmp_eq_u32 vcc, {s}, v{c.PADSLOT}")
        for h in range(c.w):
            o.append(f"v_cndmask_b32_e32 v{c.YA(s) + h}, v{c.YA(s) + h}, v{c.Y0N + h}, vcc")
    o.append(f"{L(3)}:")
    o.append("s_nop 1")
    o += dpp_recv(c, r[0], c.M(R - 1))
    o += dpp_recv(c, r[2], c.YA(R - 1))
    o += dpp_recv(c, r[1], c.X(R - 1))
    if wide:
        o += wide_handoff(c, r, c.YA(R - 1), lab + 5)
    return o


def program(c, wide=False):
    """the asm statement: general(%[n_pre]) -> fast(%[n_blk] blocks) -> general(%[n_post], its last step feeding the
    separator when %[has_sep])."""
    ents = [f"s{S_E0 + u}" for u in range(U)]
    load = [f"s_load_dwordx8 s[{S_E0}:{S_E0 + 7}], s[{S_SRC}:{S_SRC + 1}], 0x0",
            f"s_add_u32 s{S_SRC}, s{S_SRC}, 32", f"s_addc_u32 s{S_SRC + 1}, s{S_SRC + 1}, 0",
            "s_waitcnt lgkmcnt(0)"]
    o = [f"s_mov_b32 {S_LIM}, 0xbffffffe", f"s_mov_b32 {S_ORIG}, %[orig_old]",
         f"s_mov_b32 {S_CNT}, %[n_pre]", f"s_mov_b32 {S_PH}, 0", f"s_mov_b32 {S_BT}, -1"]
    o.append("60:")                                            # ---- a run of S_CNT general steps
    o.append(f"s_cmp_eq_u32 {S_CNT}, 0")
    o.append("s_cbranch_scc1 70f")
    o.append("61:")
    o += load
    if wide:
        o += wide_wait(c, 200)
    for u in range(U):
        o += general_step(c, ents[u], 100 + 10 * u, wide)
        if wide and u == U - 1:
            o += wide_publish(c)
        o.append(f"s_sub_u32 {S_CNT}, {S_CNT}, 1")
        o.append(f"s_cmp_eq_u32 {S_CNT}, 0")
        if u < U - 1:
            o.append(f"s_cbranch_scc1 {62 + u}f")
        else:
            o.append("s_cbranch_scc0 61b")
    o.append("s_branch 70f")
    for u in range(U - 1):                                     # the run ended inside a block: give the unread entries back
        o.append(f"{62 + u}:")
        o.append(f"s_sub_u32 s{S_SRC}, s{S_SRC}, {4 * (U - 1 - u)}")
        o.append(f"s_subb_u32 s{S_SRC + 1}, s{S_SRC + 1}, 0")
        if u < U - 2:
            o.append("s_branch 70f")
    o.append("70:")
    if wide:
        o += wide_publish(c)
    o.append(f"s_cmp_lg_u32 {S_PH}, 0")
    o.append("s_cbranch_scc1 99f")
    o.append(f"s_mov_b32 {S_CNT}, %[n_blk]")                   # ---- fast blocks
    o.append("80:")
    o.append(f"s_cmp_eq_u32 {S_CNT}, 0")
    o.append("s_cbranch_scc1 81f")
    o += load
    if wide:
        o += wide_wait(c, 210)
    o += fast_block(c, ents, wide)
    if wide:
        o += wide_publish(c)
    o.append(f"s_sub_u32 {S_CNT}, {S_CNT}, 1")
    o.append("s_branch 80b")
    o.append("81:")
    o.append(f"s_mov_b32 {S_PH}, 1")                           # ---- leftover columns + the separator
    o.append(f"s_mov_b32 {S_CNT}, %[n_post]")
    o.append(f"s_cmp_lg_u32 %[has_sep], 0")
    o.append(f"s_cselect_b32 {S_BT}, 1, -1")
    o.append("s_branch 60b")
    o.append("99:")
    return o


def decls(c, o):
    R = c.R
    T = "double" if c.f64 else "float"
    inout, consts = [], []
    for s in range(R):
        inout.append((f"j.M[{s}]", c.M(s), f"m{s}"))
    for s in range(R):
        inout.append((f"j.X[{s}]", c.X(s), f"x{s}"))
    for s in range(R):
        inout.append((f"j.Y[{s}]", c.YA(s), f"y{s}"))
    inout += [("j.rM", c.RS[0][0], "rm"), ("j.rX", c.RS[0][1], "rx"), ("j.rY", c.RS[0][2], "ry"),
              ("j.dM", c.RS[1][0], "dm"), ("j.dX", c.RS[1][1], "dx"), ("j.dY", c.RS[1][2], "dy"),
              ("j.sM", c.SM, "sm"), ("j.sX", c.SX, "sx")]
    for s in range(R):
        consts += [(f"j.pMM[{s}]", c.PMM(s), f"pmm{s}"), (f"j.pGAPM[{s}]", c.GAPM(s), f"pgapm{s}"), (f"j.pMX[{s}]", c.PMX(s), f"pmx{s}"),
                   (f"j.pXX[{s}]", c.PXX(s), f"pxx{s}"), (f"j.pMY[{s}]", c.PMY(s), f"pmy{s}")]
    for expr, reg, name in inout + consts:
        o.append(f"  {T} {name} = {expr};")
    return inout, consts


def hard(c, reg, wide=None):
    """hard-register constraint of a value living in VGPR `reg` (a pair for fp64 values)"""
    w = c.w if wide is None else wide
    return "{v[%d:%d]}" % (reg, reg + 1) if w == 2 else "{v%d}" % reg


def emit_asm(o, ins, outs, inp, clob, indent="    "):
    o.append(indent + "asm volatile(")
    for i in ins:
        o.append(f"{indent}    \"{i}\\n\\t\"")
    o.append(f"{indent}    : {outs}")
    o.append(f"{indent}    : {inp}")
    o.append(f"{indent}    : " + ", ".join(f"\"{x}\"" for x in clob) + ");")


def clobbers(c):
    R = c.R
    cl = set()
    for s in range(R):
        for h in range(c.w):
            cl.add(c.YB(s) + h)
            cl.add(c.PR(s) + h)
    for reg, n in ((c.EAB, 1), (c.ADDR, 1), (c.NSEP, 1), (c.VAL, c.w), (c.Y0N, c.w), (c.A, 2), (c.P, 2)):
        for h in range(n):
            cl.add(reg + h)
    if not c.fma:
        for t in c.TN:
            for h in range(c.w):
                cl.add(t + h)
    return [f"v{x}" for x in sorted(cl)] + S_CLOB + ["vcc", "scc", "memory"]


def driver(c, o, wide=False):
    T = "double" if c.f64 else "float"
    CT = "ConstF64" if c.f64 else "ConstF32"
    o.append("")
    o.append(f"// One whole job (haplotypes [hap_begin, hap_end) streamed through the loaded rows) of WaveJob<{T}, {c.R}, true> in pinned")
    o.append("// registers: see tools/gen_fwd_asm.py for the structure, the register map and the preconditions the caller checks.")
    o.append("template <class Job, class Args>")
    if wide:
        o.append("// WIDE variant: the read spans the n_waves wavefronts of a workgroup (lanes 64 * wave .. 64 * wave + 63 of one systolic")
        o.append("// array); the bottom row of a wavefront's lane 63 reaches the next wavefront's lane 0 through a ring in LDS, a step")
        o.append("// counter per wavefront (flags) keeps producer and consumer within the ring of each other.  Every wavefront runs the")
        o.append("// same number of steps: 64 * wave steps of pre-roll first (its lanes' skew), pre-roll words after its last separator.")
        o.append(f"__device__ __forceinline__ void fwd_asm_run_wide_{c.name}(Job& j, const Args& a, int lane, int hap_begin, int hap_end, int wave, int n_waves,")
        o.append("                                                       uint32_t ring_in, uint32_t ring_out, uint32_t flag_own, uint32_t flag_prod, uint32_t flag_cons) {")
        o.append("  wave = __builtin_amdgcn_readfirstlane(wave); n_waves = __builtin_amdgcn_readfirstlane(n_waves);")
    else:
        o.append(f"__device__ __forceinline__ void fwd_asm_run_{c.name}(Job& j, const Args& a, int lane, int hap_begin, int hap_end) {{")
    o.append("  hap_begin = __builtin_amdgcn_readfirstlane(hap_begin); hap_end = __builtin_amdgcn_readfirstlane(hap_end);")
    o.append("  ConstI32* hap_pos = (ConstI32*)a.hap_pos;")
    o.append("  ConstI32* hap_len = (ConstI32*)a.hap_len;")
    o.append("  ConstI32* hap_orig = (ConstI32*)a.hap_orig;")
    o.append(f"  {CT}* y0s = ({CT}*)a.y0;")
    o.append("  const int sb = hap_pos[hap_begin];")
    o.append(f"  j.reset_state(y0s[hap_begin]);")
    inout, consts = decls(c, o)
    o.append("  uint32_t ent = kEntPreroll;")
    o.append("  const uint32_t lmask = j.lmask, direct = j.direct, ndirect = ~j.direct;")
    o.append("  const uint32_t loff = (uint32_t)(uintptr_t)j.lds + (uint32_t)lane * (uint32_t)Job::kVecBytes;")
    o.append("  const uint32_t outidx = j.out_read >= 0 ? (uint32_t)j.out_read * (uint32_t)a.b.n_haps : 0u;")
    o.append("  const int32_t padslot = j.padb_slot;")
    extra_v = ", ".join(f"\"{hard(c, r, 1)}\"({n})" for r, n in ((c.LMASK, "lmask"), (c.DIRECT, "direct"), (c.NDIRECT, "ndirect"),
                                                              (c.LOFF, "loff"), (c.OUTIDX, "outidx"), (c.PADSLOT, "padslot")))
    if c.f64:
        o.append("  const uint32_t kreg = (uint32_t)Job::kRowBytes;")
        extra_v += f", \"{hard(c, c.KREG, 1)}\"(kreg)"
    o.append("  const uint64_t outmask = __ballot(j.out_read >= 0);")
    o.append("  const uint64_t raw = (uint64_t)(uintptr_t)a.raw;")
    if not c.f64:
        o.append("  const uint64_t packed = (uint64_t)(uintptr_t)a.packed_out;")
        o.append("  const float minacc = 1e-28f;")
    o.append("  const int skew = j.skew_max;")
    o.append("  uint64_t src = (uint64_t)(uintptr_t)(a.stream + sb);")
    o.append("  uint32_t orig_old = 0;")
    o.append("  // several separators in the array at once (stream order is by ascending length: the first haplotype decides)?")
    o.append("  const uint32_t multi = hap_len[hap_begin] <= skew ? 1u : 0u;")
    o.append("  const uint64_t haporig = (uint64_t)(uintptr_t)a.hap_orig;")
    o.append("  int t = 0, fast_from = skew;   // the fill: the most skewed lane meets its first column at t = skew")
    if wide:
        o.append("  const int delay = 64 * wave;")
        o.append("  const int t_end = 64 * (n_waves - 1) + (hap_pos[hap_end - 1] - sb + hap_len[hap_end - 1]) + 64;   // steps of every wavefront")
        o.append("  const uint32_t has_in = wave > 0 ? 1u : 0u, has_out = wave + 1 < n_waves ? 1u : 0u;")
        o.append("  uint32_t st = 0;")
        o.append("  const uint64_t stream0 = src;")
        o.append("  for (int k = hap_begin - 1; k <= hap_end; k++) {")
    else:
        o.append("  for (int k = hap_begin; k <= hap_end; k++) {")
    o.append("    int n_pre, n_blk, n_post, has_sep;")
    o.append("    uint32_t orig = 0;")
    o.append(f"    {T} y0n = 0;")
    if wide:
        o.append("    if (k < hap_begin) {")
        o.append("      // this wavefront's lanes start 64 * wave steps after the first wavefront's: pre-roll until then")
        o.append("      if (delay == 0) continue;")
        o.append("      n_pre = delay; n_blk = 0; n_post = 0; has_sep = 0;")
        o.append("      src = (uint64_t)(uintptr_t)kPrerollWords;")
        o.append("    } else if (k < hap_end) {")
        o.append("      if (k == hap_begin) src = stream0;")
    else:
        o.append("    if (k < hap_end) {")
    o.append("      const int sep_at = hap_pos[k] - sb + hap_len[k];   // stream-relative position of this haplotype's separator")
    o.append("      n_pre = fast_from - t;                              // the window of the previous separator (the fill)")
    o.append("      if (n_pre < 0) n_pre = 0;")
    o.append("      if (n_pre > sep_at - t) n_pre = sep_at - t;")
    o.append("      const int rem = sep_at - t - n_pre;")
    o.append("      n_blk = rem >> 3;")
    o.append("      n_post = (rem & 7) + 1;                             // leftover columns + the separator itself")
    o.append("      has_sep = 1;")
    o.append("      orig = (uint32_t)hap_orig[k];")
    o.append("      if (k + 1 < hap_end) y0n = y0s[k + 1];")
    o.append("      t = sep_at + 1;")
    o.append("      fast_from = sep_at + skew + 1;")
    o.append("    } else {")
    o.append("      // drain: the last separator travels down the array, nothing new enters")
    if wide:
        o.append("      n_pre = t_end - (delay + t); n_blk = 0; n_post = 0; has_sep = 0;")
    else:
        o.append("      n_pre = skew; n_blk = 0; n_post = 0; has_sep = 0;")
    o.append("      src = (uint64_t)(uintptr_t)kPrerollWords;")
    o.append("    }")
    o.append("    // (wave-uniform by construction; said explicitly so that they are SGPR operands in every kernel this is inlined into)")
    for nm in ("n_pre", "n_blk", "n_post", "has_sep"):
        o.append(f"    {nm} = __builtin_amdgcn_readfirstlane({nm});")
    o.append("    const uint32_t multi_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)multi);")
    o.append("    const uint64_t haporig_s = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(haporig >> 32)) << 32) |")
    o.append("                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)haporig);")
    for nm in ("orig", "orig_old"):
        o.append(f"    {nm} = (uint32_t)__builtin_amdgcn_readfirstlane((int){nm});")
    o.append("    src = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(src >> 32)) << 32) |")
    o.append("          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)src);")
    if c.f64:
        o.append("    const uint64_t y0bits = (uint64_t)__double_as_longlong(y0n);")
        o.append("    const uint32_t y0n_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)y0bits);")
        o.append("    const uint32_t y0n_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(y0bits >> 32));")
        y0in = "[y0n_lo] \"s\"(y0n_lo), [y0n_hi] \"s\"(y0n_hi)"
    else:
        o.append("    const uint32_t y0n_b = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(y0n));")
        y0in = "[y0n] \"s\"(y0n_b)"
    outs = ", ".join(f"\"+{hard(c, reg)}\"({name})" for _, reg, name in inout) + f", \"+{hard(c, c.ENT, 1)}\"(ent), " \
        f"\"+&{{s[{S_SRC}:{S_SRC + 1}]}}\"(src)"   # (early clobber: the program advances it while it still reads its scalar inputs)
    inp = ", ".join(f"\"{hard(c, reg)}\"({name})" for _, reg, name in consts) + ", " + extra_v + \
        ", [orig] \"s\"(orig), [orig_old] \"s\"(orig_old), [multi] \"s\"(multi_s), [haporig] \"s\"(haporig_s), [outmask] \"s\"(outmask), " \
        "[raw] \"s\"(raw), " + ("" if c.f64 else "[packed] \"s\"(packed), [minacc] \"s\"(minacc), ") + y0in + \
        ", [n_pre] \"s\"(n_pre), [n_blk] \"s\"(n_blk), [n_post] \"s\"(n_post), [has_sep] \"s\"(has_sep)"
    clob = clobbers(c)
    if wide:
        # early clobber: without it the compiler may keep an input of the same VALUE (orig = 0, n_blk = 0 in the pre-roll
        # call, where st is 0 too) in the very register the program counts its steps in
        outs += f", \"+&{{{S_T}}}\"(st)"
        for nm in ("has_in", "has_out", "ring_in", "ring_out", "flag_own", "flag_prod", "flag_cons", "t_end"):
            o.append(f"    const uint32_t {nm}_s = (uint32_t)__builtin_amdgcn_readfirstlane((int){nm});")
            inp += f", [{nm}] \"s\"({nm}_s)"
        clob = clob + [f"v{c.WADDR}", f"v{c.FV}", S_NEED]
    prog = program(c, wide)
    emit_asm(o, prog, outs, inp, clob, "    ")
    o.append("    orig_old = orig;")
    o.append("  }")
    o.append("}")
    return prog


def legacy_fast(o):
    """fwd_fast_asm_f32r8: the fast blocks alone, state in and out through the job's members -- for jobs that fail the
    whole-job driver's preconditions (and GKLHIP_ASM_GENERAL=0): the C++ general step runs around it, as in round 3."""
    c = Cfg("f32r8", False, 8)
    o.append("")
    o.append("// Runs blocks of 8 in-haplotype columns while t + 8 <= end; state in and out through the job's members.")
    o.append("template <class Job>")
    o.append("__device__ __forceinline__ void fwd_fast_asm_f32r8(Job& j, StreamWord* sp, int& t, int end, int lane) {")
    o.append("  if (t + 8 > end) return;")
    inout, consts = decls(c, o)
    o.append("  uint32_t ent = j.ent;")
    o.append("  const uint32_t lmask = j.lmask, direct = j.direct, ndirect = ~j.direct;")
    o.append("  const uint32_t loff = (uint32_t)(uintptr_t)j.lds + (uint32_t)lane * 16u;")
    o.append("  for (; t + 8 <= end; t += 8) {")
    o.append("    uint32_t e[8];")
    o.append("#pragma unroll")
    o.append("    for (int u = 0; u < 8; u++) e[u] = sp[t + u];")
    outs = ", ".join(f"\"+{hard(c, reg)}\"({name})" for _, reg, name in inout) + f", \"+{hard(c, c.ENT, 1)}\"(ent)"
    inp = ", ".join(f"\"{hard(c, reg)}\"({name})" for _, reg, name in consts) + \
        f", \"{hard(c, c.LMASK, 1)}\"(lmask), \"{hard(c, c.DIRECT, 1)}\"(direct), \"{hard(c, c.NDIRECT, 1)}\"(ndirect), \"{hard(c, c.LOFF, 1)}\"(loff), " + \
        ", ".join(f"[e{u}] \"s\"(e[{u}])" for u in range(U))
    clob = [f"v{c.YB(s)}" for s in range(c.R)] + [f"v{c.PR(s)}" for s in range(c.R)] + [f"v{c.EAB}", f"v{c.ADDR}"]
    emit_asm(o, fast_block(c, [f"%[e{u}]" for u in range(U)]), outs, inp, clob, "    ")
    o.append("  }")
    for expr, reg, name in inout:
        o.append(f"  {expr} = {name};")
    o.append("  j.ent = ent;")
    o.append("}")


def main(path):
    o = []
    o.append("// GENERATED by tools/gen_fwd_asm.py -- do not edit; see that file for the why, the structure and the register maps.")
    o.append("// Hand-allocated asm drivers of the PairHMM forward recurrence for gfx950; arithmetic = WaveJob::advance (reference")
    o.append("// avx-pairhmm-template.h:208-223 in the AVX-512 object's FMA pattern).")
    o.append("#pragma once")
    o.append("namespace gklhip {")
    o.append("constexpr uint32_t kEntPreroll = 0xBFFFFFFFu;   // separator-type entry that no haplotype owns (fill and drain)")
    o.append("constexpr uint32_t kEntNoEmit = 0xBFFFFFFEu;    // separator \"in flight\" while there is none: matches no entry")
    o.append("// the drain's entries (and a wide job's pre-roll and tail: up to 64 steps per wavefront of the read's array + 64), read")
    o.append("// eight at a time; kPrerollMaxWaves: the deepest array a wide program may be run for (reads of 64 * 64 * RPL - 1 bases)")
    o.append("constexpr int kPrerollMaxWaves = 64;")
    o.append("__device__ const uint32_t kPrerollWords[64 * kPrerollMaxWaves + 72] = {")
    for _ in range((64 * 64 + 72) // 8):
        o.append("    kEntPreroll, kEntPreroll, kEntPreroll, kEntPreroll, kEntPreroll, kEntPreroll, kEntPreroll, kEntPreroll,")
    o.append("};")
    o.append("typedef const int32_t __attribute__((address_space(4))) ConstI32;   // plan arrays: written by an earlier kernel, s_load here")
    o.append("typedef const float __attribute__((address_space(4))) ConstF32;")
    o.append("typedef const double __attribute__((address_space(4))) ConstF64;")
    legacy_fast(o)
    stats = []
    shapes = (("f32r8", False, 8, True), ("f64r10", True, 10, True), ("f64r8", True, 8, True),
              # the narrow kernels of small and mid-size calls (one pair or a few reads per wavefront)
              # (wide variants of the 2- and 4-row programs were generated and tried in round 5 -- a lone GATK-sized call with a pair of
              #  128+ bases on TWO wavefronts -- and lost: docs/NOTES.md 42; `True` in the last column brings them back)
              ("f32r4", False, 4, False), ("f32r2", False, 2, False), ("f64r6", True, 6, False), ("f64r4", True, 4, False), ("f64r2", True, 2, False))
    # every program twice: the AVX-512 object's contraction (fma_mode 1) and the AVX object's unfused arithmetic (fma_mode 0, "...n")
    for c, wide in [(Cfg(nm, f64, R, fma), w) for fma in (True, False) for nm, f64, R, w in shapes]:
        driver(c, o)
        if wide:
            driver(c, o, wide=True)
        fb = fast_block(c, [f"s{S_E0 + u}" for u in range(U)])
        gs = general_step(c, "s72", 100)
        nv = lambda ins: sum(1 for i in ins if i.startswith("v_"))
        stats.append(f"{c.name}: fast block {nv(fb)} VALU = {nv(fb) / (U * c.R):.3f} per cell; general step {nv(gs)} VALU "
                     f"({nv(gs) - (6 + (2 if c.f64 else 1) + 2 * c.R * c.w // c.w)} without the rare store / Y0 sections: see the listing)")
    o.append("}  // namespace gklhip")
    open(path, "w").write("\n".join(o) + "\n")
    print("\n".join(stats))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gkl_amd/csrc/pairhmm_fwd_asm.h")
