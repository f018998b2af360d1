#!/usr/bin/env python3
"""Generates gkl_amd/csrc/pairhmm_fwd_fast_asm.h: the unrolled fast loop of the fp32 / 8-rows-per-lane / FMA forward
kernel as ONE hand-allocated inline-asm block per 8 stream columns.

Why asm (measured on MI355X, tools/gen_ubench_banks2.py -> gpurun_out/regbank{2,3}.txt, four wavefronts per SIMD):
  * the VGPR file has two banks, even and odd register numbers: a three-source VALU op (v_fmac_f32, v_fma_f32) whose three
    sources are all even or all odd issues in 4.45 cycles instead of 2.35.  hipcc's allocation of the C++ loop puts 67 of
    its 256 v_fmac_f32 there.  Here: M, X, Y and the fmac accumulators live in even registers, pGAPM and pXX in odd ones --
    no three-source op is ever monochrome.
  * a DPP op that follows a non-DPP VALU op costs the SIMD ~2.3 cycles on top of its own 2.3 (26 without an s_nop in front);
    four DPP ops back to back behind one `s_nop 1` cost what four v_mov cost.  Here the step's four hand-off DPPs (the three
    values of the bottom row and the stream entry) form one group at the end of the step.
  * VOP3-encoded v_mul_legacy_f32 2.8 cycles against v_mul_f32 2.33; in the fast loop every column has a haplotype base, so
    the plain multiply is exact-equivalent (prior 0 only meets the all-zero pad rows).
  * any SGPR source makes a VALU op a 4.4-cycle op; v_and_or_b32 / v_lshl_add_u32 are 4.4 anyway (VOP3 integer).

The arithmetic (operation order, FMA pattern) is WaveJob::advance's, i.e. the reference's compute_full_prob with the
gcc-11 contraction of the AVX-512 object (reference avx-pairhmm-template.h:208-223); results are bit-identical to the
C++ step (tests/test_gpu_parity.py runs both builds).

Register map (v16..v122), E = even, O = odd:
  M[s]  E v16+2s      pGAPM[s] O v17+2s
  X[s]  E v32+2s      pXX[s]   O v33+2s
  Ya[s] E v48+2s      pMM[s]   O v49+2s
  Yb[s] E v64+2s      pMX[s]   O v65+2s      (Ya/Yb ping-pong so that the Y update is a VOP2 v_fmac)
  pMY[s]   v80+s
  prior[s] v88+s (two ds_read_b128)
  R0 = v96,v97,v98 (M,X,Y of the row above), R1 = v100,v101,v102
  sM v104  sX v105  ent v106  eabove v107  lmask v108  direct v109  ndirect v110  lane_off v111  addr v112
"""
import sys

U = 8
R = 8

M = lambda s: f"v{16 + 2 * s}"
GAPM = lambda s: f"v{17 + 2 * s}"
X = lambda s: f"v{32 + 2 * s}"
PXX = lambda s: f"v{33 + 2 * s}"
YA = lambda s: f"v{48 + 2 * s}"
PMM = lambda s: f"v{49 + 2 * s}"
YB = lambda s: f"v{64 + 2 * s}"
PMX = lambda s: f"v{65 + 2 * s}"
PMY = lambda s: f"v{80 + s}"
PR = lambda s: f"v{88 + s}"
RSET = [("v96", "v97", "v98"), ("v100", "v101", "v102")]
SM, SX, ENT, EAB, LMASK, DIRECT, NDIRECT, LOFF, ADDR = "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112"
DPP = "wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"


def step(u, last):
    """one stream column; Y set: even steps read Ya write Yb, odd steps the reverse"""
    yo, yn = (YA, YB) if u % 2 == 0 else (YB, YA)
    rM, rX, rY = RSET[u % 2]          # the row above at THIS column (fetched at the end of the previous step)
    dM, dX, dY = RSET[(u + 1) % 2]    # ... at the previous column (diagonal inputs); overwritten by this step's fetch
    o = []
    # the lane's entry for this column, its prior rows
    o.append(f"v_and_or_b32 {ENT}, %[e{u}], {DIRECT}, {EAB}")
    o.append(f"v_lshl_add_u32 {ADDR}, {ENT}, 11, {LOFF}")
    o.append(f"ds_read_b128 v[88:91], {ADDR}")
    o.append(f"ds_read_b128 v[92:95], {ADDR} offset:1024")
    # bottom-up: M-inner of row s into X[s]'s register (old X[s] is dead once row s+1 has read it), Y update of row s
    for s in range(R - 1, -1, -1):
        md, xd, yd = (M(s - 1), X(s - 1), yo(s - 1)) if s > 0 else (dM, dX, dY)
        o.append(f"v_mul_f32 {X(s)}, {md}, {PMM(s)}")
        o.append(f"v_mul_f32 {yn(s)}, {M(s)}, {PMY(s)}")
        o.append(f"v_fmac_f32 {X(s)}, {xd}, {GAPM(s)}")
        o.append(f"v_fmac_f32 {yn(s)}, {yo(s)}, {PXX(s)}")
        o.append(f"v_fmac_f32 {X(s)}, {yd}, {GAPM(s)}")
    o.append("s_waitcnt lgkmcnt(0)")
    for s in range(R):
        o.append(f"v_mul_f32 {M(s)}, {X(s)}, {PR(s)}")
    # X column, top-down
    o.append(f"v_mul_f32 {X(0)}, {rM}, {PMX(0)}")
    o.append(f"v_fmac_f32 {X(0)}, {rX}, {PXX(0)}")
    for s in range(1, R):
        o.append(f"v_mul_f32 {X(s)}, {M(s - 1)}, {PMX(s)}")
        o.append(f"v_fmac_f32 {X(s)}, {X(s - 1)}, {PXX(s)}")
    o.append(f"v_add_f32 {SM}, {SM}, {M(R - 1)}")
    # hand-off group: four DPP ops back to back behind one s_nop (X last: written two instructions earlier)
    o.append("s_nop 1")
    if not last:
        o.append(f"v_and_b32_dpp {EAB}, {ENT}, {NDIRECT} {DPP}")
    o.append(f"v_and_b32_dpp {dM}, {M(R - 1)}, {LMASK} {DPP}")
    o.append(f"v_and_b32_dpp {dY}, {yn(R - 1)}, {LMASK} {DPP}")
    o.append(f"v_and_b32_dpp {dX}, {X(R - 1)}, {LMASK} {DPP}")
    o.append(f"v_add_f32 {SX}, {SX}, {X(R - 1)}")
    return o


def block():
    o = [f"s_nop 1", f"v_and_b32_dpp {EAB}, {ENT}, {NDIRECT} {DPP}"]
    for u in range(U):
        o += step(u, u == U - 1)
    return o


def main(path):
    ins = block()
    inout = []   # (c++ expr, register, name)
    for s in range(R):
        inout.append((f"j.M[{s}]", M(s), f"m{s}"))
    for s in range(R):
        inout.append((f"j.X[{s}]", X(s), f"x{s}"))
    for s in range(R):
        inout.append((f"j.Y[{s}]", YA(s), f"y{s}"))
    # at block entry r* = the row above at the block's first column = R0 of step 0, d* = R1
    inout += [("j.rM", RSET[0][0], "rm"), ("j.rX", RSET[0][1], "rx"), ("j.rY", RSET[0][2], "ry"),
              ("j.dM", RSET[1][0], "dm"), ("j.dX", RSET[1][1], "dx"), ("j.dY", RSET[1][2], "dy"),
              ("j.sM", SM, "sm"), ("j.sX", SX, "sx")]
    consts = []
    for s in range(R):
        consts += [(f"j.pMM[{s}]", PMM(s), f"pmm{s}"), (f"j.pGAPM[{s}]", GAPM(s), f"pgapm{s}"), (f"j.pMX[{s}]", PMX(s), f"pmx{s}"),
                   (f"j.pXX[{s}]", PXX(s), f"pxx{s}"), (f"j.pMY[{s}]", PMY(s), f"pmy{s}")]
    clob = [YB(s) for s in range(R)] + [PR(s) for s in range(R)] + [EAB, ADDR]
    o = []
    o.append("// GENERATED by tools/gen_fwd_fast_asm.py -- do not edit; see that file for the why and the register map.")
    o.append("// The unrolled fast loop of pairhmm_fwd_stream_kernel<float, 8, true> (and the other float/8/FMA kernels) as one")
    o.append("// hand-allocated asm block per 8 stream columns; arithmetic = WaveJob::advance (reference")
    o.append("// avx-pairhmm-template.h:208-223 in the AVX-512 object's FMA pattern).")
    o.append("#pragma once")
    o.append("namespace gklhip {")
    o.append("// Runs blocks of 8 in-haplotype columns while t + 8 <= end; state in and out through the job's members.")
    o.append("template <class Job>")
    o.append("__device__ __forceinline__ void fwd_fast_asm_f32r8(Job& j, StreamWord* sp, int& t, int end, int lane) {")
    o.append("  if (t + 8 > end) return;")
    for expr, reg, name in inout:
        o.append(f"  register float {name} asm(\"{reg}\") = {expr};")
    for expr, reg, name in consts:
        o.append(f"  register float {name} asm(\"{reg}\") = {expr};")
    o.append(f"  register uint32_t ent asm(\"{ENT}\") = j.ent;")
    o.append(f"  register uint32_t lmask asm(\"{LMASK}\") = j.lmask;")
    o.append(f"  register uint32_t direct asm(\"{DIRECT}\") = j.direct;")
    o.append(f"  register uint32_t ndirect asm(\"{NDIRECT}\") = ~j.direct;")
    o.append(f"  register uint32_t loff asm(\"{LOFF}\") = (uint32_t)(uintptr_t)j.lds + (uint32_t)lane * 16u;")
    o.append("  for (; t + 8 <= end; t += 8) {")
    o.append("    uint32_t e[8];")
    o.append("#pragma unroll")
    o.append("    for (int u = 0; u < 8; u++) e[u] = sp[t + u];")
    o.append("    asm volatile(")
    for i in ins:
        o.append(f"        \"{i}\\n\\t\"")
    outs = ", ".join(f"\"+v\"({name})" for _, _, name in inout) + ", \"+v\"(ent)"
    inp = ", ".join(f"\"v\"({name})" for _, _, name in consts) + ", \"v\"(lmask), \"v\"(direct), \"v\"(ndirect), \"v\"(loff), " + \
        ", ".join(f"[e{u}] \"s\"(e[{u}])" for u in range(U))
    o.append(f"        : {outs}")
    o.append(f"        : {inp}")
    o.append("        : " + ", ".join(f"\"{c}\"" for c in clob) + ");")
    o.append("  }")
    for expr, reg, name in inout:
        o.append(f"  {expr} = {name};")
    o.append("  j.ent = ent;")
    o.append("}")
    o.append("}  // namespace gklhip")
    open(path, "w").write("\n".join(o) + "\n")
    n_valu = sum(1 for i in ins if i.startswith("v_"))
    print(f"{len(ins)} instructions per block, {n_valu} VALU = {n_valu / 64:.3f} per cell")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gkl_amd/csrc/pairhmm_fwd_fast_asm.h")
