#!/usr/bin/env python3
"""Dev tool (not product): generates tools/ubench_regbank.hip -- issue-rate microbenchmarks with EXPLICIT VGPR numbers,
to find out which operand placements let a gfx950 SIMD issue a wave64 v_fma_f32 / v_fmac_f32 / v_mul_f32 at its 2-cycle
rate (round 2 measured 2.5 cycles for two-operand ops but 3.2-4.5 for FMAs with compiler-chosen registers).

Every variant is one loop of 64 independent instructions over 16 accumulator chains; whole loop in one asm statement."""
import sys

VARIANTS = []


def variant(name, instrs):
    VARIANTS.append((name, instrs))


def acc(bank, i, base=24):
    """i-th accumulator register in a given bank (reg % 4 == bank)"""
    return base + 4 * i + bank


def rep(fn):
    return [fn(i % 8) for i in range(64)]


def set_a():
    # --- three-operand FMA: placement of the three sources (bank = reg % 4) ---
    variant("fma  A(seq) * M(v0) + C(v1)", rep(lambda i: f"v_fma_f32 v{24+i}, v{24+i}, v0, v1"))
    variant("fma  A(b0) * M(b1) + C(b2)   all distinct", rep(lambda i: f"v_fma_f32 v{acc(0,i)}, v{acc(0,i)}, v1, v2"))
    variant("fma  A(b0) * M(b0) + C(b0)   all same bank", rep(lambda i: f"v_fma_f32 v{acc(0,i)}, v{acc(0,i)}, v4, v8"))
    variant("fma  A(b0) * M(b1) + C(b1)   M,C same", rep(lambda i: f"v_fma_f32 v{acc(0,i)}, v{acc(0,i)}, v1, v5"))
    variant("fma  A(b0) * M(b0) + C(b1)   A,M same", rep(lambda i: f"v_fma_f32 v{acc(0,i)}, v{acc(0,i)}, v4, v1"))
    variant("fma  D(b3) = A(b0)*M(b1)+C(b2)", rep(lambda i: f"v_fma_f32 v{acc(3,i)}, v{acc(0,i)}, v1, v2"))
    variant("fma  D(b0) = A(b0')*M(b1)+C(b2)", rep(lambda i: f"v_fma_f32 v{acc(0,i)}, v{acc(0,(i+1)%8)}, v1, v2"))
    variant("fma  A*A + C(b2)", rep(lambda i: f"v_fma_f32 v{acc(0,i)}, v{acc(0,i)}, v{acc(0,i)}, v2"))
    variant("fma  A(b0) * s4 + C(b2)", rep(lambda i: f"v_fma_f32 v{acc(0,i)}, v{acc(0,i)}, s4, v2"))
    variant("fma  A(b0) * M(b1) + 1.0", rep(lambda i: f"v_fma_f32 v{acc(0,i)}, v{acc(0,i)}, v1, 1.0"))
    variant("fma  A(b0) * s4 + 1.0", rep(lambda i: f"v_fma_f32 v{acc(0,i)}, v{acc(0,i)}, s4, 1.0"))
    # per-chain distinct multiplicands: a_i = a_i * b_i + c_i with banks 0,1,2
    variant("fma  A(b0)*B_i(b1)+C_i(b2)  distinct per chain", rep(lambda i: f"v_fma_f32 v{acc(0,i)}, v{acc(0,i)}, v{acc(1,i)}, v{acc(2,i)}"))
    variant("fma  A(b0)*B_i(b0)+C_i(b0)  distinct, one bank", rep(lambda i: f"v_fma_f32 v{acc(0,i)}, v{acc(0,i)}, v{acc(0,i,56)}, v{acc(0,i,88)}"))
    # --- VOP2 fmac ---
    variant("fmac A(b0) += B_i(b1) * M(b2)", rep(lambda i: f"v_fmac_f32 v{acc(0,i)}, v{acc(1,i)}, v2"))
    variant("fmac A(b0) += B_i(b0) * M(b0)", rep(lambda i: f"v_fmac_f32 v{acc(0,i)}, v{acc(0,i,56)}, v4"))
    variant("fmac A(b0) += M(b1) * C(b2)", rep(lambda i: f"v_fmac_f32 v{acc(0,i)}, v1, v2"))
    variant("fmac A(b0) += s4 * C(b2)", rep(lambda i: f"v_fmac_f32 v{acc(0,i)}, s4, v2"))
    variant("fmac A(seq) += B(seq) * M(v0)", rep(lambda i: f"v_fmac_f32 v{24+i}, v{56+i}, v0"))
    # --- two-operand ---
    variant("mul  A(b0) * M(b1)", rep(lambda i: f"v_mul_f32 v{acc(0,i)}, v{acc(0,i)}, v1"))
    variant("mul  A(b0) * M(b0)", rep(lambda i: f"v_mul_f32 v{acc(0,i)}, v{acc(0,i)}, v4"))
    variant("mul  D(b2) = A(b0) * B_i(b1)", rep(lambda i: f"v_mul_f32 v{acc(2,i)}, v{acc(0,i)}, v{acc(1,i)}"))
    variant("mul  A(b0) * s4", rep(lambda i: f"v_mul_f32 v{acc(0,i)}, s4, v{acc(0,i)}"))
    variant("mul_legacy A(b0) * M(b1)", rep(lambda i: f"v_mul_legacy_f32 v{acc(0,i)}, v{acc(0,i)}, v1"))
    variant("add  A(b0) + M(b1)", rep(lambda i: f"v_add_f32 v{acc(0,i)}, v{acc(0,i)}, v1"))
    variant("mov  D(b1) = A(b0)", rep(lambda i: f"v_mov_b32 v{acc(1,i)}, v{acc(0,i)}"))
    # --- mixes shaped like the recurrence: mul, fma, fma, mul / mul, fma / mul, fma ---
    def cell_mix(clean):
        out = []
        for r in range(8):
            a = acc(0, r)          # state regs bank 0
            b = acc(1, r) if clean else acc(0, r, 56)
            c = acc(2, r) if clean else acc(0, r, 88)
            t = acc(3, r) if clean else acc(1, r, 56)
            out += [f"v_mul_f32 v{t}, v{a}, v{b}",
                    f"v_fmac_f32 v{t}, v{b}, v{c}",
                    f"v_fmac_f32 v{t}, v{a}, v{c}",
                    f"v_mul_f32 v{t}, v{t}, v{c}",
                    f"v_mul_f32 v{a}, v{a}, v{b}",
                    f"v_fmac_f32 v{a}, v{c}, v{b}",
                    f"v_mul_f32 v{c}, v{c}, v{b}",
                    f"v_fmac_f32 v{c}, v{t}, v{b}"]
        return out
    variant("cell mix 4 mul + 4 fmac, banks spread", cell_mix(True))
    variant("cell mix 4 mul + 4 fmac, one bank", cell_mix(False))
    # --- packed ---
    variant("pk_fma A(b0,1) * M(b2,3) + C(b2,3)'", rep(lambda i: f"v_pk_fma_f32 v[{acc(0,i)}:{acc(0,i)+1}], v[{acc(0,i)}:{acc(0,i)+1}], v[2:3], v[6:7]"))
    variant("pk_fma A(b0,1) * B_i(b2,3) + C_i(b0,1)", rep(lambda i: f"v_pk_fma_f32 v[{acc(0,i)}:{acc(0,i)+1}], v[{acc(0,i)}:{acc(0,i)+1}], v[{acc(2,i)}:{acc(2,i)+1}], v[{acc(0,i,56)}:{acc(0,i,56)+1}]"))
    variant("pk_mul A(b0,1) * M(b2,3)", rep(lambda i: f"v_pk_mul_f32 v[{acc(0,i)}:{acc(0,i)+1}], v[{acc(0,i)}:{acc(0,i)+1}], v[2:3]"))
    variant("pk_fma A * s[4:5] + C(b2,3)", rep(lambda i: f"v_pk_fma_f32 v[{acc(0,i)}:{acc(0,i)+1}], v[{acc(0,i)}:{acc(0,i)+1}], s[4:5], v[2:3]"))
    # --- DPP ---
    variant("and_dpp D(b1) = shr(A(b0)) & M(b2)", rep(lambda i: f"v_and_b32_dpp v{acc(1,i)}, v{acc(0,i)}, v2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"))
    variant("mov_dpp D(b1) = shr(A(b0))", rep(lambda i: f"v_mov_b32_dpp v{acc(1,i)}, v{acc(0,i)} wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"))
    variant("mov_dpp row_shr:1 D(b1) = A(b0)", rep(lambda i: f"v_mov_b32_dpp v{acc(1,i)}, v{acc(0,i)} row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"))
    # 1 DPP among 15 muls (how much does an isolated DPP cost)
    variant("15 mul + 1 and_dpp(wave_shr)", [(f"v_and_b32_dpp v{acc(1,i%8)}, v{acc(0,i%8)}, v2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" if i % 16 == 7 else f"v_mul_f32 v{acc(0,i%8)}, v{acc(0,i%8)}, v1") for i in range(64)])
    variant("15 mul + 1 mov_dpp(row_shr)", [(f"v_mov_b32_dpp v{acc(1,i%8)}, v{acc(0,i%8)} row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" if i % 16 == 7 else f"v_mul_f32 v{acc(0,i%8)}, v{acc(0,i%8)}, v1") for i in range(64)])
    # --- f64 ---
    variant("fma_f64 A(b0,1)*M(b2,3)+C(b2,3)'", rep(lambda i: f"v_fma_f64 v[{acc(0,i)}:{acc(0,i)+1}], v[{acc(0,i)}:{acc(0,i)+1}], v[2:3], v[6:7]"))
    variant("fma_f64 A(b0,1)*B_i(b2,3)+C_i(b0,1)", rep(lambda i: f"v_fma_f64 v[{acc(0,i)}:{acc(0,i)+1}], v[{acc(0,i)}:{acc(0,i)+1}], v[{acc(2,i)}:{acc(2,i)+1}], v[{acc(0,i,56)}:{acc(0,i,56)+1}]"))
    variant("mul_f64 A(b0,1)*M(b2,3)", rep(lambda i: f"v_mul_f64 v[{acc(0,i)}:{acc(0,i)+1}], v[{acc(0,i)}:{acc(0,i)+1}], v[2:3]"))
    variant("add_f64 A(b0,1)+M(b2,3)", rep(lambda i: f"v_add_f64 v[{acc(0,i)}:{acc(0,i)+1}], v[{acc(0,i)}:{acc(0,i)+1}], v[2:3]"))




def set_b():
    """second round: the bank function, SGPR sources, VOP3 encodings, DPP in context, LDS reads among VALU, fp64 moves"""
    M = lambda i: f"v_mul_f32 v{acc(0,i%8)}, v{acc(0,i%8)}, v1"
    # (1) bank function: fmac D_i += A_i * B with D_i = 24+8i (all = 0 mod 8), A_i = D_i + a_off, B = v{b}
    for a_off in (1, 2, 3, 4, 5, 6, 7):
        for b in (0, 4, 8, 16):
            variant(f"fmac D(8k) += A(8k+{a_off}) * v{b}", [f"v_fmac_f32 v{24+8*(i%8)}, v{24+8*(i%8)+a_off}, v{b}" for i in range(64)])
    for b in (1, 2, 3, 5, 12, 20):
        variant(f"fmac D(8k) += A(8k+4) * v{b}", [f"v_fmac_f32 v{24+8*(i%8)}, v{24+8*(i%8)+4}, v{b}" for i in range(64)])
    variant("fmac D(8k) += A(8k) * v4   (A = D)", [f"v_fmac_f32 v{24+8*(i%8)}, v{24+8*(i%8)}, v4" for i in range(64)])
    variant("fmac D(8k) += A(8k+1) * A(8k+1)", [f"v_fmac_f32 v{24+8*(i%8)}, v{24+8*(i%8)+1}, v{24+8*(i%8)+1}" for i in range(64)])
    variant("mul D(8k) = A(8k+1) * A(8k+1)", [f"v_mul_f32 v{24+8*(i%8)}, v{24+8*(i%8)+1}, v{24+8*(i%8)+1}" for i in range(64)])
    # (2) SGPR / constant sources and encodings
    variant("mov v, s4", [f"v_mov_b32 v{acc(0,i%8)}, s4" for i in range(64)])
    variant("mov v, 1.0", [f"v_mov_b32 v{acc(0,i%8)}, 1.0" for i in range(64)])
    variant("mov v, 0x12345 (literal)", [f"v_mov_b32 v{acc(0,i%8)}, 0x12345" for i in range(64)])
    variant("and_or v, s4, v, v (VOP3)", [f"v_and_or_b32 v{acc(0,i%8)}, s4, v1, v{acc(0,i%8)}" for i in range(64)])
    variant("and_or v, v, v, v (VOP3)", [f"v_and_or_b32 v{acc(0,i%8)}, v2, v1, v{acc(0,i%8)}" for i in range(64)])
    variant("lshl_or v, v, 11, v (VOP3)", [f"v_lshl_or_b32 v{acc(0,i%8)}, v{acc(0,i%8)}, 11, v1" for i in range(64)])
    variant("mul_f32_e64 A * M", [f"v_mul_f32_e64 v{acc(0,i%8)}, v{acc(0,i%8)}, v1" for i in range(64)])
    variant("or_b32 e32 v, v, v", [f"v_or_b32 v{acc(0,i%8)}, v{acc(0,i%8)}, v1" for i in range(64)])
    variant("cndmask e32 (vcc)", [f"v_cndmask_b32 v{acc(0,i%8)}, v{acc(0,i%8)}, v1, vcc" for i in range(64)])
    variant("cndmask e64 (s[8:9])", [f"v_cndmask_b32_e64 v{acc(0,i%8)}, v{acc(0,i%8)}, v1, s[8:9]" for i in range(64)])
    variant("readlane s, v, 3", [f"v_readlane_b32 s{10+(i%8)}, v{acc(0,i%8)}, 3" for i in range(64)])
    variant("add_u32 e32", [f"v_add_u32 v{acc(0,i%8)}, v{acc(0,i%8)}, v1" for i in range(64)])
    # (3) SALU among VALU: does it take issue time from the wave's VALU stream?
    variant("64 mul", [M(i) for i in range(64)])
    variant("64 mul + 16 s_add interleaved", sum([[M(4*k), M(4*k+1), M(4*k+2), M(4*k+3), "s_add_u32 s10, s10, 1"] for k in range(16)], []))
    variant("64 mul + 16 s_nop 0 interleaved", sum([[M(4*k), M(4*k+1), M(4*k+2), M(4*k+3), "s_nop 0"] for k in range(16)], []))
    variant("64 mul + 16 s_nop 1 interleaved", sum([[M(4*k), M(4*k+1), M(4*k+2), M(4*k+3), "s_nop 1"] for k in range(16)], []))
    # (4) DPP in context: 60 mul + 4 DPP, placed differently.  DPP reads bank-1 registers nobody writes in the loop
    D = lambda j, pre="": pre + f"v_and_b32_dpp v{acc(2,j)}, v{acc(1,j)}, v2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
    mul60 = [M(i) for i in range(60)]
    variant("60 mul + 4 dpp spread (no nop)", sum([mul60[15*k:15*k+15] + [D(k)] for k in range(4)], []))
    variant("60 mul + 4 dpp spread (s_nop 0 each)", sum([mul60[15*k:15*k+15] + ["s_nop 0", D(k)] for k in range(4)], []))
    variant("60 mul + 4 dpp spread (s_nop 1 each)", sum([mul60[15*k:15*k+15] + ["s_nop 1", D(k)] for k in range(4)], []))
    variant("60 mul + 4 dpp spread (s_nop 3 each)", sum([mul60[15*k:15*k+15] + ["s_nop 3", D(k)] for k in range(4)], []))
    variant("60 mul + 4 dpp grouped (no nop)", mul60 + [D(k) for k in range(4)])
    variant("60 mul + 4 dpp grouped (s_nop 1 first)", mul60 + ["s_nop 1"] + [D(k) for k in range(4)])
    variant("60 mul + 4 dpp grouped (s_nop 1 first, s_nop 1 after)", mul60 + ["s_nop 1"] + [D(k) for k in range(4)] + ["s_nop 1"])
    variant("60 mul + 4 dpp grouped (s_nop 3 first)", mul60 + ["s_nop 3"] + [D(k) for k in range(4)])
    variant("60 mul + 4 dpp pairs (s_nop 1)", mul60[:30] + ["s_nop 1", D(0), D(1)] + mul60[30:] + ["s_nop 1", D(2), D(3)])
    variant("60 mul + 4 mov_dpp row_shr grouped (s_nop 1)", mul60 + ["s_nop 1"] + [f"v_mov_b32_dpp v{acc(2,k)}, v{acc(1,k)} row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" for k in range(4)])
    variant("60 mul + 4 mul_dpp (fused op) grouped (s_nop 1)", mul60 + ["s_nop 1"] + [f"v_mul_f32_dpp v{acc(2,k)}, v{acc(1,k)}, v2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" for k in range(4)])
    variant("60 mul + 4 ds_bpermute", mul60 + [f"ds_bpermute_b32 v{acc(2,k)}, v3, v{acc(1,k)}" for k in range(4)] + ["s_waitcnt lgkmcnt(0)"])
    variant("60 mul + 4 permlane32_swap", mul60 + [f"v_permlane32_swap_b32 v{acc(2,k)}, v{acc(1,k)}" for k in range(4)])
    variant("60 mul + 4 v_mov (baseline for the above)", mul60 + [f"v_mov_b32 v{acc(2,k)}, v{acc(1,k)}" for k in range(4)])
    # fmac instead of mul as the neighbours
    F = lambda i: f"v_fmac_f32 v{acc(0,i%8)}, v{acc(1,i%8)}, v2"
    fm60 = [F(i) for i in range(60)]
    variant("60 fmac + 4 dpp spread (s_nop 1 each)", sum([fm60[15*k:15*k+15] + ["s_nop 1", D(k)] for k in range(4)], []))
    variant("60 fmac + 4 dpp grouped (s_nop 1 first)", fm60 + ["s_nop 1"] + [D(k) for k in range(4)])
    # (5) LDS reads among VALU
    variant("60 mul + 2 ds_read_b128", mul60[:30] + ["ds_read_b128 v[88:91], v3"] + mul60[30:] + ["ds_read_b128 v[92:95], v3 offset:1024", "s_waitcnt lgkmcnt(0)"])
    variant("60 mul + 4 ds_read_b64", sum([mul60[15*k:15*k+15] + [f"ds_read_b64 v[{88+2*k}:{89+2*k}], v3 offset:{512*k}"] for k in range(4)], []) + ["s_waitcnt lgkmcnt(0)"])
    variant("60 mul + 8 ds_read_b32", sum([mul60[7*k:7*k+7] + [f"ds_read_b32 v{88+k}, v3 offset:{256*k}"] for k in range(8)], []) + mul60[56:] + ["s_waitcnt lgkmcnt(0)"])
    # (6) fp64
    P = lambda b, i, base=24: f"v[{acc(b,i,base)}:{acc(b,i,base)+1}]"
    variant("mov_b64 D(b2,3) = A(b0,1)", [f"v_mov_b64 {P(2,i%8)}, {P(0,i%8)}" for i in range(64)])
    variant("2x mov_b32 for a 64-bit copy", sum([[f"v_mov_b32 v{acc(2,i%8)}, v{acc(0,i%8)}", f"v_mov_b32 v{acc(3,i%8)}, v{acc(1,i%8)}"] for i in range(32)], []))
    variant("max_f64", [f"v_max_f64 {P(0,i%8)}, {P(0,i%8)}, v[2:3]" for i in range(64)])
    variant("fma_f64 D=A*B+C all distinct regs", [f"v_fma_f64 {P(0,i%8)}, {P(2,i%8)}, {P(0,i%8,56)}, {P(2,i%8,56)}" for i in range(64)])
    variant("fma_f64 A = A*s[4:5] + C", [f"v_fma_f64 {P(0,i%8)}, {P(0,i%8)}, s[4:5], v[2:3]" for i in range(64)])
    variant("cndmask pair (64-bit select)", sum([[f"v_cndmask_b32 v{acc(0,i%8)}, v{acc(0,i%8)}, v2, vcc", f"v_cndmask_b32 v{acc(1,i%8)}, v{acc(1,i%8)}, v3, vcc"] for i in range(32)], []))
    variant("32 fma_f64 + 32 mul_f32 alternating", sum([[f"v_fma_f64 {P(0,i%8)}, {P(0,i%8)}, v[2:3], v[6:7]", f"v_mul_f32 v{acc(0,i%8,56)}, v{acc(0,i%8,56)}, v1"] for i in range(32)], []))
    variant("32 fma_f64 + 32 v_and_b32 alternating", sum([[f"v_fma_f64 {P(0,i%8)}, {P(0,i%8)}, v[2:3], v[6:7]", f"v_and_b32 v{acc(0,i%8,56)}, v{acc(0,i%8,56)}, v1"] for i in range(32)], []))



def set_c():
    """third round: compare + select costs (the PDHMM match predicate)"""
    A = lambda i: acc(0, i % 8)
    B = lambda i: acc(1, i % 8)
    Cc = lambda i: acc(2, i % 8)
    variant("cmp_e32 -> vcc ; 2 cndmask_e32 (vcc)", sum([[f"v_cmp_lt_u32 vcc, v1, v{A(i)}", f"v_cndmask_b32 v{B(i)}, v{B(i)}, v2, vcc", f"v_cndmask_b32 v{Cc(i)}, v{Cc(i)}, v3, vcc"] for i in range(21)], []))
    variant("cmp_e64 -> s[8:9] ; 2 cndmask_e64", sum([[f"v_cmp_lt_u32_e64 s[8:9], v1, v{A(i)}", f"v_cndmask_b32_e64 v{B(i)}, v{B(i)}, v2, s[8:9]", f"v_cndmask_b32_e64 v{Cc(i)}, v{Cc(i)}, v3, s[8:9]"] for i in range(21)], []))
    variant("cmp_e32 only", [f"v_cmp_lt_u32 vcc, v1, v{A(i)}" for i in range(64)])
    variant("cmp_e64 only", [f"v_cmp_lt_u32_e64 s[8:9], v1, v{A(i)}" for i in range(64)])
    variant("cndmask_e32 only (vcc set once before the loop)", [f"v_cndmask_b32 v{B(i)}, v{B(i)}, v2, vcc" for i in range(64)])
    variant("and + cmp_e32 + 2 cndmask_e32 (the predicate, vcc)", sum([[f"v_and_b32 v{A(i)}, v1, v{Cc(i)}", f"v_cmp_lt_u32 vcc, v2, v{A(i)}", f"v_cndmask_b32 v{B(i)}, v{B(i)}, v2, vcc", f"v_cndmask_b32 v{B(i)+1}, v{B(i)+1}, v3, vcc"] for i in range(16)], []))
    variant("and + cmp_e64 + 2 cndmask_e64 (the predicate, sgpr pair)", sum([[f"v_and_b32 v{A(i)}, v1, v{Cc(i)}", f"v_cmp_lt_u32_e64 s[8:9], v2, v{A(i)}", f"v_cndmask_b32_e64 v{B(i)}, v{B(i)}, v2, s[8:9]", f"v_cndmask_b32_e64 v{B(i)+1}, v{B(i)+1}, v3, s[8:9]"] for i in range(16)], []))
    variant("bfe_i32 + 2 bfi (mask select)", sum([[f"v_bfe_i32 v{A(i)}, v1, v{Cc(i)}, 1", f"v_bfi_b32 v{B(i)}, v{A(i)}, v2, v{B(i)}", f"v_bfi_b32 v{B(i)+1}, v{A(i)}, v3, v{B(i)+1}"] for i in range(21)], []))
    P = lambda b, i, base=24: f"v[{base+8*(i%8)+2*b}:{base+8*(i%8)+2*b+1}]"
    variant("fma_f64 x3 + predicate(vcc) per 'row'", sum([[f"v_fma_f64 {P(0,i)}, {P(0,i)}, v[2:3], v[6:7]", f"v_and_b32 v{88+(i%8)}, v1, v{96+(i%8)}", f"v_fma_f64 {P(1,i)}, {P(1,i)}, v[2:3], v[6:7]", f"v_cmp_lt_u32 vcc, v2, v{88+(i%8)}", f"v_fma_f64 {P(2,i)}, {P(2,i)}, v[2:3], v[6:7]", f"v_cndmask_b32 v{104+(i%8)}, v{104+(i%8)}, v2, vcc", f"v_cndmask_b32 v{112+(i%4)}, v{112+(i%4)}, v3, vcc"] for i in range(9)], []))
    variant("fma_f64 x3 + predicate(sgpr) per 'row'", sum([[f"v_fma_f64 {P(0,i)}, {P(0,i)}, v[2:3], v[6:7]", f"v_and_b32 v{88+(i%8)}, v1, v{96+(i%8)}", f"v_fma_f64 {P(1,i)}, {P(1,i)}, v[2:3], v[6:7]", f"v_cmp_lt_u32_e64 s[8:9], v2, v{88+(i%8)}", f"v_fma_f64 {P(2,i)}, {P(2,i)}, v[2:3], v[6:7]", f"v_cndmask_b32_e64 v{104+(i%8)}, v{104+(i%8)}, v2, s[8:9]", f"v_cndmask_b32_e64 v{112+(i%4)}, v{112+(i%4)}, v3, s[8:9]"] for i in range(9)], []))
    variant("fma_f64 x3 only (27)", sum([[f"v_fma_f64 {P(0,i)}, {P(0,i)}, v[2:3], v[6:7]", f"v_fma_f64 {P(1,i)}, {P(1,i)}, v[2:3], v[6:7]", f"v_fma_f64 {P(2,i)}, {P(2,i)}, v[2:3], v[6:7]"] for i in range(9)], []))


def main(path, which="b"):
    {"a": set_a, "b": set_b, "c": set_c}[which]()
    o = []
    o.append("// GENERATED by tools/gen_ubench_banks2.py -- dev tool (not product).  hipcc --offload-arch=gfx950 -O3 -o /tmp/ub tools/ubench_regbank.hip")
    o.append("#include <hip/hip_runtime.h>\n#include <cstdio>\n")
    o.append("#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf(\"HIP error %s at %d\\n\", hipGetErrorString(e), __LINE__); return 1; } } while (0)")
    clob = ", ".join(f'"v{i}"' for i in range(120))
    for n, (name, instrs) in enumerate(VARIANTS):
        body = "\\n\"\n      \"".join(instrs)
        o.append(f"// {name}")
        o.append(f"__global__ __launch_bounds__(256) void k{n}(float* out, int iters) {{")
        o.append("  __shared__ float lds[4096]; lds[threadIdx.x] = 1.f; __syncthreads();")
        o.append("  asm volatile(")
        # init every VGPR with a finite value that keeps chains bounded (x*1+0 style): 1.0 everywhere, then the loop
        o.append("      \"s_mov_b32 s4, 1.0\\n s_mov_b32 s5, 1.0\\n\"")
        for r in range(0, 120, 8):
            o.append("      \"" + "".join(f"v_mov_b32 v{r+j}, 1.0\\n " for j in range(8)) + "\"")
        o.append("      \"v_mov_b32 v2, 0\\n v_mov_b32 v3, 0\\n v_mov_b32 v6, 0\\n v_mov_b32 v7, 0\\n\"")
        o.append("      \"s_mov_b64 s[8:9], -1\\n v_mbcnt_lo_u32_b32 v3, -1, 0\\n v_mbcnt_hi_u32_b32 v3, -1, v3\\n v_lshlrev_b32 v3, 4, v3\\n\"")
        o.append("      \"s_mov_b32 s6, %0\\n\"")
        o.append("      \"1:\\n\"")
        o.append(f"      \"{body}\\n\"")
        o.append("      \"s_sub_u32 s6, s6, 1\\n s_cmp_lg_u32 s6, 0\\n s_cbranch_scc1 1b\\n\"")
        o.append(f"      :: \"s\"(iters) : \"s4\", \"s5\", \"s6\", \"scc\", {clob});")
        o.append("  if (iters < 0) out[threadIdx.x] = lds[threadIdx.x + 1];")
        o.append("}\n")
    o.append("typedef void (*kern_t)(float*, int);")
    o.append("struct V { const char* name; kern_t k; int n; };")
    o.append("static V variants[] = {")
    for n, (name, _) in enumerate(VARIANTS):
        o.append(f"  {{\"{name}\", k{n}, {len(VARIANTS[n][1])}}},")
    o.append("};")
    o.append(r'''
int main(int argc, char** argv) {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const double clk = p.clockRate * 1e-6;
  printf("device %s  CUs %d  clock %.2f GHz\n", p.gcnArchName, p.multiProcessorCount, clk);
  float* d; CHECK(hipMalloc(&d, 4096));
  const int ITER = 4096;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int wps : {4, 2}) {
    const int blocks = p.multiProcessorCount * wps;
    for (auto& v : variants) {
      hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), 0, 0, d, 56);
      CHECK(hipDeviceSynchronize());
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), 0, 0, d, ITER);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const double wave_iters = (double)blocks * 4 * ITER;
      const double cyc = (double)p.multiProcessorCount * 4 * clk * 1e9 / (wave_iters / (best * 1e-3));
      printf("%-56s w/SIMD=%d %7.3f ms %8.1f cyc/iter  n=%3d  %.2f cyc/instr\n", v.name, wps, best, cyc, v.n, cyc / v.n);
    }
  }
  return 0;
}
''')
    open(path, "w").write("\n".join(o))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "tools/ubench_regbank.hip", sys.argv[2] if len(sys.argv) > 2 else "b")
