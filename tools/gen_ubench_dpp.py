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
variants["7 fmac + cndmask_dpp vcc (cold)"] = fmacs(7) + ["v_cndmask_b32_dpp v20, v30, v31, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"]
'''
exec(head + "variants = {}" + new_variants + tail[body_start:])
