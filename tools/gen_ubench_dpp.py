#!/usr/bin/env python3
"""Dev tool: cost of a DPP op as a function of how long ago its source register was written (tools/ubench_dpp.hip)."""
import re, sys
sys.argv = [sys.argv[0], "tools/ubench_dpp.hip"]
src = open("tools/gen_ubench_mix.py").read()
# reuse the generator with another variant table
head, tail = src.split("variants = {}", 1)
body_start = tail.index("src = ['// generated")
new_variants = '''
def fmacs(n, base=0):
    return [f"v_fmac_f32 v{base+i}, v{40+i}, v{80+i}" for i in range(n)]
variants["8 fmac (baseline)"] = fmacs(8)
variants["7 fmac + v_and plain (src v6)"] = fmacs(7) + ["v_and_b32 v20, v6, v31"]
for k in (1, 2, 3, 4, 5, 6, 7):
    variants[f"7 fmac + nop + and_dpp, src written {k} fmac(s) earlier"] = fmacs(7) + ["s_nop 1", f"v_and_b32_dpp v20, v{7-k}, v31 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"]
variants["7 fmac + nop + and_dpp, src never written (cold)"] = fmacs(7) + ["s_nop 1", "v_and_b32_dpp v20, v30, v31 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"]
variants["7 fmac + and_dpp cold, no nop"] = fmacs(7) + ["v_and_b32_dpp v20, v30, v31 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"]
variants["7 fmac + mov v30,v30 + nop + and_dpp v30 (re-warmed)"] = fmacs(7) + ["v_mov_b32 v30, v30", "s_nop 1", "v_and_b32_dpp v20, v30, v31 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"]
variants["15 fmac + nop + and_dpp src 15 earlier"] = [f"v_fmac_f32 v{i}, v{40+i}, v{80+i}" for i in range(15)] + ["s_nop 1", "v_and_b32_dpp v20, v0, v31 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"]
variants["15 fmac + v_and plain src v0"] = [f"v_fmac_f32 v{i}, v{40+i}, v{80+i}" for i in range(15)] + ["v_and_b32 v20, v0, v31"]
variants["7 fmac + nop + 3 and_dpp hot (v4 v5 v6)"] = fmacs(7) + ["s_nop 1"] + [f"v_and_b32_dpp v{20+i}, v{4+i}, v31 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" for i in range(3)]
variants["7 fmac + 3 v_and plain"] = fmacs(7) + [f"v_and_b32 v{20+i}, v{4+i}, v31" for i in range(3)]
variants["7 fmac + 3 x (nop + and_dpp) hot"] = fmacs(7) + sum([["s_nop 1", f"v_and_b32_dpp v{20+i}, v{4+i}, v31 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"] for i in range(3)], [])
variants["7 fmac + 3 x (nop0 + and_dpp) hot"] = fmacs(7) + sum([["s_nop 0", f"v_and_b32_dpp v{20+i}, v{4+i}, v31 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"] for i in range(3)], [])
variants["dpp spread: (2 fmac, nop, dpp) x3 + fmac"] = sum([[f"v_fmac_f32 v{2*i}, v{40+2*i}, v{80+2*i}", f"v_fmac_f32 v{2*i+1}, v{41+2*i}, v{81+2*i}", "s_nop 1", f"v_and_b32_dpp v{20+i}, v{2*i}, v31 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"] for i in range(3)], []) + ["v_fmac_f32 v6, v46, v86"]
variants["dpp spread no nop: (2 fmac, dpp) x3 + fmac"] = sum([[f"v_fmac_f32 v{2*i}, v{40+2*i}, v{80+2*i}", f"v_fmac_f32 v{2*i+1}, v{41+2*i}, v{81+2*i}", f"v_and_b32_dpp v{20+i}, v{2*i}, v31 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"] for i in range(3)], []) + ["v_fmac_f32 v6, v46, v86"]
variants["dpp spread s_nop 0: (2 fmac, nop0, dpp) x3 + fmac"] = sum([[f"v_fmac_f32 v{2*i}, v{40+2*i}, v{80+2*i}", f"v_fmac_f32 v{2*i+1}, v{41+2*i}, v{81+2*i}", "s_nop 0", f"v_and_b32_dpp v{20+i}, v{2*i}, v31 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"] for i in range(3)], []) + ["v_fmac_f32 v6, v46, v86"]
variants["7 fmac + 3 v_and plain (again)"] = fmacs(7) + [f"v_and_b32 v{20+i}, v{4+i}, v31" for i in range(3)]
variants["4 mul + 4 fmac"] = [f"v_mul_f32 v{20+i}, v{40+i}, v{80+i}" for i in range(4)] + [f"v_fmac_f32 v{20+i}, v{44+i}, v{84+i}" for i in range(4)]
variants["3 mul + 1 mul_legacy + 4 fmac"] = [f"v_mul_f32 v{20+i}, v{40+i}, v{80+i}" for i in range(3)] + ["v_mul_legacy_f32 v23, v43, v83"] + [f"v_fmac_f32 v{20+i}, v{44+i}, v{84+i}" for i in range(4)]
variants["8 mul_legacy"] = [f"v_mul_legacy_f32 v{20+i}, v{40+i}, v{80+i}" for i in range(8)]
variants["8 mul"] = [f"v_mul_f32 v{20+i}, v{40+i}, v{80+i}" for i in range(8)]
variants["7 fmac + v_and_or_b32"] = fmacs(7) + ["v_and_or_b32 v20, v30, v31, v32"]
variants["7 fmac + v_lshl_or_b32"] = fmacs(7) + ["v_lshl_or_b32 v20, v30, 11, v31"]
variants["7 fmac + v_add_f32"] = fmacs(7) + ["v_add_f32 v20, v20, v6"]
variants["kernel row: mul fmac fmac mul_legacy mul fmac mul fmac (dep)"] = ["v_mul_f32 v20, v40, v80", "v_fmac_f32 v20, v41, v81", "v_fmac_f32 v20, v42, v81", "v_mul_legacy_f32 v21, v20, v82", "v_mul_f32 v22, v43, v83", "v_fmac_f32 v22, v44, v84", "v_mul_f32 v23, v45, v85", "v_fmac_f32 v23, v46, v84"]
variants["8 fmac e64 (v_fma_f32 d,a,b,d)"] = [f"v_fma_f32 v{i}, v{40+i}, v{80+i}, v{i}" for i in range(8)]
variants["8 fmac, operands in 3 banks (i, 41+i, 82+i)"] = [f"v_fmac_f32 v{i}, v{41+i}, v{82+i}" for i in range(8)]
variants["8 fmac, src0/src1 same bank, dst other (i, 41+i, 81+i)"] = [f"v_fmac_f32 v{i}, v{41+i}, v{81+i}" for i in range(8)]
variants["8 fmac, dst/src1 same bank (i, 41+i, 80+i)"] = [f"v_fmac_f32 v{i}, v{41+i}, v{80+i}" for i in range(8)]
variants["8 mul, operands in 3 banks (20+i, 41+i, 82+i)"] = [f"v_mul_f32 v{20+i}, v{41+i}, v{82+i}" for i in range(8)]
variants["8 mul, src same bank (20+i, 41+i, 81+i)"] = [f"v_mul_f32 v{20+i}, v{41+i}, v{81+i}" for i in range(8)]
variants["8 fmac, src1 = one shared reg (i, 41+i, 90)"] = [f"v_fmac_f32 v{i}, v{41+i}, v90" for i in range(8)]
variants["8 fmac, src0 sgpr (i, s4, 82+i)"] = [f"v_fmac_f32 v{i}, s4, v{82+i}" for i in range(8)]
variants["8 mul_f64 (same banks)"] = [f"v_mul_f64 v[{2*i}:{2*i+1}], v[{40+2*i}:{41+2*i}], v[{80+2*i}:{81+2*i}]" for i in range(8)]
variants["8 mul_f64 (srcs in different banks)"] = [f"v_mul_f64 v[{2*i}:{2*i+1}], v[{40+2*i}:{41+2*i}], v[{82+2*i}:{83+2*i}]" for i in range(8)]
variants["8 fmac_f64 (same banks)"] = [f"v_fmac_f64 v[{2*i}:{2*i+1}], v[{40+2*i}:{41+2*i}], v[{80+2*i}:{81+2*i}]" for i in range(8)]
variants["8 fmac_f64 (srcs other banks: 2i, 42+2i... )"] = [f"v_fmac_f64 v[{2*i}:{2*i+1}], v[{42+2*i}:{43+2*i}], v[{80+2*i}:{81+2*i}]" for i in range(8)]
variants["8 add_f64"] = [f"v_add_f64 v[{2*i}:{2*i+1}], v[{40+2*i}:{41+2*i}], v[{80+2*i}:{81+2*i}]" for i in range(8)]
variants["8 max_f64"] = [f"v_max_f64 v[{2*i}:{2*i+1}], v[{40+2*i}:{41+2*i}], v[{80+2*i}:{81+2*i}]" for i in range(8)]
variants["8 mov_b64"] = [f"v_mov_b64 v[{2*i}:{2*i+1}], v[{40+2*i}:{41+2*i}]" for i in range(8)]
variants["4 x (and, cmp_lt vcc, cndmask, cndmask)"] = sum([[f"v_and_b32 v{20+i}, v30, v{40+i}", f"v_cmp_lt_u32 vcc, s4, v{20+i}", f"v_cndmask_b32 v{2*i}, v{50+i}, v{60+i}, vcc", f"v_cndmask_b32 v{2*i+1}, v{70+i}, v{80+i}, vcc"] for i in range(2)], [])
variants["PD row x2: mul fmac fmac mul mul fmac (f64) + and cmp cnd cnd"] = sum([[f"v_mul_f64 v[{a}:{a+1}], v[40:41], v[80:81]", f"v_fmac_f64 v[{a}:{a+1}], v[42:43], v[82:83]", f"v_fmac_f64 v[{a}:{a+1}], v[44:45], v[82:83]", f"v_and_b32 v30, v31, v{60+a}", "v_cmp_lt_u32 vcc, s4, v30", f"v_cndmask_b32 v32, v50, v52, vcc", f"v_cndmask_b32 v33, v51, v53, vcc", f"v_mul_f64 v[{a+2}:{a+3}], v[{a}:{a+1}], v[32:33]", f"v_mul_f64 v[{a+4}:{a+5}], v[46:47], v[84:85]", f"v_fmac_f64 v[{a+4}:{a+5}], v[48:49], v[86:87]"] for a in (0, 8)], [])
def dppx(ctrl, d, s_, nop=True, op="v_and_b32_dpp", extra="row_mask:0xf bank_mask:0xf bound_ctrl:1"):
    third = ", v31" if op.startswith("v_and") else ""
    return (["s_nop 1"] if nop else []) + [f"{op} v{d}, v{s_}{third} {ctrl} {extra}"]
for name, ctrl in (("wave_shr:1", "wave_shr:1"), ("row_shr:1", "row_shr:1"), ("row_bcast:15", "row_bcast:15"), ("quad_perm", "quad_perm:[0,0,1,2]"), ("row_ror:1", "row_ror:1")):
    # 3 producers (fmac), 3 dpp of them, 3 consumers of the dpp results, + 16 filler fmacs: the kernel's hand-off in miniature
    variants[f"handoff {name}: 16 fmac + 3 prod + 3 dpp(nop) + 3 cons"] = fmacs(16, 4) + [f"v_fmac_f32 v{i}, v{60+i}, v{90+i}" for i in range(3)] + sum([dppx(ctrl, 20 + i, i) for i in range(3)], []) + [f"v_fmac_f32 v{24+i}, v{20+i}, v{70+i}" for i in range(3)]
variants["handoff none: 16 fmac + 3 prod + 3 v_and + 3 cons"] = fmacs(16, 4) + [f"v_fmac_f32 v{i}, v{60+i}, v{90+i}" for i in range(3)] + [f"v_and_b32 v{20+i}, v{i}, v31" for i in range(3)] + [f"v_fmac_f32 v{24+i}, v{20+i}, v{70+i}" for i in range(3)]
variants["handoff bpermute: 16 fmac + 3 prod + 3 ds_bpermute + 3 cons"] = fmacs(16, 4) + [f"v_fmac_f32 v{i}, v{60+i}, v{90+i}" for i in range(3)] + [f"ds_bpermute_b32 v{20+i}, v34, v{i}" for i in range(3)] + ["s_waitcnt lgkmcnt(0)"] + [f"v_fmac_f32 v{24+i}, v{20+i}, v{70+i}" for i in range(3)]
variants["handoff lds: 16 fmac + 3 prod + ds_write_b96 + ds_read_b96 + 3 cons"] = fmacs(16, 4) + [f"v_fmac_f32 v{i}, v{60+i}, v{90+i}" for i in range(3)] + ["ds_write_b96 v35, v[0:2]", "ds_read_b96 v[20:22], v38", "s_waitcnt lgkmcnt(0)"] + [f"v_fmac_f32 v{24+i}, v{20+i}, v{70+i}" for i in range(3)]
variants["handoff wave_shr far: 3 prod + 16 fmac + 3 dpp(nop) + 3 cons"] = [f"v_fmac_f32 v{i}, v{60+i}, v{90+i}" for i in range(3)] + fmacs(16, 4) + sum([dppx("wave_shr:1", 20 + i, i) for i in range(3)], []) + [f"v_fmac_f32 v{24+i}, v{20+i}, v{70+i}" for i in range(3)]
variants["handoff wave_shr, consumers far: 3 prod + 3 dpp(nop) + 16 fmac + 3 cons"] = [f"v_fmac_f32 v{i}, v{60+i}, v{90+i}" for i in range(3)] + sum([dppx("wave_shr:1", 20 + i, i) for i in range(3)], []) + fmacs(16, 4) + [f"v_fmac_f32 v{24+i}, v{20+i}, v{70+i}" for i in range(3)]
def rows8():
    out = []
    for r in range(8):
        a, b, c, d = 4 * r % 20, (4 * r + 1) % 20, (4 * r + 2) % 20, (4 * r + 3) % 20
        out += [f"v_mul_f32 v{a}, v40, v80", f"v_fmac_f32 v{a}, v41, v81", f"v_fmac_f32 v{a}, v42, v81", f"v_mul_legacy_f32 v{b}, v{a}, v82", f"v_mul_f32 v{c}, v43, v83", f"v_fmac_f32 v{c}, v44, v84", f"v_mul_f32 v{d}, v45, v85", f"v_fmac_f32 v{d}, v46, v84"]
    return out
dpp = lambda d, s_: ["s_nop 1", f"v_and_b32_dpp v{d}, v{s_}, v31 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"]
variants["STEP now: 64 FP + 4 dpp(nop) + and_or + lshl_or + 2 add + 2 ds_read_b128"] = rows8() + dpp(20, 1) + dpp(21, 2) + dpp(22, 3) + dpp(23, 30) + ["v_and_or_b32 v30, s4, v31, v23", "v_lshl_or_b32 v33, v30, 11, v34", "ds_read_b128 v[24:27], v33", "ds_read_b128 v[50:53], v33 offset:1024", "v_add_f32 v36, v36, v1", "v_add_f32 v37, v37, v2"]
variants["STEP lds: 64 FP + ds_write_b128 + ds_read_b128 + cndmask + or + 2 add + 2 ds_read_b128"] = rows8() + ["ds_write_b128 v35, v[0:3]", "ds_read_b128 v[20:23], v38", "v_and_b32 v39, s4, v31", "v_or_b32 v30, v39, v23", "v_lshl_or_b32 v33, v30, 11, v34", "ds_read_b128 v[24:27], v33", "ds_read_b128 v[50:53], v33 offset:1024", "v_add_f32 v36, v36, v1", "v_add_f32 v37, v37, v2"]
variants["STEP FP only: 64 FP"] = rows8()
variants["7 fmac + cndmask_dpp vcc (cold)"] = fmacs(7) + ["v_cndmask_b32_dpp v20, v30, v31, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"]
'''
exec(head + "variants = {}" + new_variants + tail[body_start:])
