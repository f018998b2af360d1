#!/usr/bin/env python3
"""Generator of gkl_amd/csrc/pdhmm_plain_asm.h: the plain-step run of the PDHMM table kernel (pdhmm_kernel.h,
PdJob<FMA = true, kTab = true>) as one asm statement with its loop inside and every value in a pinned register.

What a plain step is: pdhmm_kernel.h step_plain -- the in-place update of the three live matrices of a lane's six rows
when no lane of the wavefront is in or next to a deletion (reference pdhmm.h:427-431, the AVX-512 object's contraction):
    D[s] = fma(D[s], tii[s], M[s] * tmd[s])
    M[s] = prior[s] * fma(M[s-1], tmm[s], fma(D[s-1], tim[s], I[s-1] * tim[s]))       (old values of row s-1)
    I[s] = fma(I'[s-1], tii[s], M'[s-1] * tmi[s])                                       (new values of row s-1)
    sum += M'[5] + I'[5]
The same operations on the same values in the same order per value as the C++ step; what the program changes is the
ORDER ACROSS values and where they live:
  * the three dependent operations of a row's M are issued row-interleaved (six independent operations between an
    operation and its consumer) instead of back to back;
  * D runs one step ahead in a second register set (D of step t+1 needs only M of step t and D of step t): its twelve
    operations fill the gaps of the insertion chain, the one truly serial part of a step (I[s] needs I'[s-1]);
  * the roles of the two hand-off sets (row above at the previous column / at this column) alternate, the loop is
    unrolled by four steps (two entry pairs in flight), nothing is copied;
  * no idle test: the caller runs this only where every lane is inside its haplotype (steps top .. H-1 of a job);
  * no ballot: the caller knows the length of the run from the haplotype's next-special-column table.
Per step: 50 fp64 operations, 6 DPP moves (hand-off), 1 address operation = 57 vector instructions (the C++ loop: 62.5).
"""
import os
import sys

KNOB = set(os.environ.get('PD_ASM_KNOBS', '').split(','))   # timing experiments only (wrong results): novmem, nolds, nodpp

R = 6
MM = lambda s: 2 + 2 * s          # noqa: E731
IM = lambda s: 14 + 2 * s         # noqa: E731
DA = lambda s: 26 + 2 * s         # noqa: E731
DB = lambda s: 38 + 2 * s         # noqa: E731
TMM = lambda s: 50 + 2 * s        # noqa: E731
TIM = lambda s: 62 + 2 * s        # noqa: E731
TMI = lambda s: 74 + 2 * s        # noqa: E731
TII = lambda s: 86 + 2 * s        # noqa: E731
TMD = lambda s: 98 + 2 * s        # noqa: E731
U = lambda s: 110 + 2 * s         # noqa: E731
PR = lambda s: 122 + 2 * s        # noqa: E731
DD = lambda k: 134 + 2 * k        # noqa: E731
RR = lambda k: 140 + 2 * k        # noqa: E731
SUM, TMP, LMASK, TABL, PADDR, VOFF, EA, EB = 146, 148, 150, 151, 152, 153, 154, 156
LAST = 157


def v2(r):
    return f"v[{r}:{r + 1}]"


def mul(d, a, b):
    return f"v_mul_f64 {v2(d)}, {v2(a)}, {v2(b)}"


def fma(d, a, b, c):
    return f"v_fma_f64 {v2(d)}, {v2(a)}, {v2(b)}, {v2(c)}"


def add(d, a, b):
    return f"v_add_f64 {v2(d)}, {v2(a)}, {v2(b)}"


def prior_reads():
    return [f"ds_read_b128 v[{PR(0)}:{PR(0) + 3}], v{PADDR}",
            f"ds_read_b128 v[{PR(2)}:{PR(2) + 3}], v{PADDR} offset:1024",
            f"ds_read_b128 v[{PR(4)}:{PR(4) + 3}], v{PADDR} offset:2048"]


def b_ops(dcur, dnext):
    """D one step ahead: dnext[s] = fma(dcur[s], tii[s], M[s] * tmd[s]); multiplies first, then the fmas"""
    return [mul(dnext(s), MM(s), TMD(s)) for s in range(R)] + [fma(dnext(s), dcur(s), TII(s), dnext(s)) for s in range(R)]


def step(dprev, dthis, diag, top, next_ent, pre=None, post=None):
    """one plain step.  dprev: D of the previous step (the diagonal inputs), dthis: D of this step (computed a step ago),
    this step writes D of the next step over dprev.  diag / top: the hand-off sets; the new hand-off lands in diag."""
    o = []
    for s in range(R):
        o.append(mul(U(s), IM(s - 1) if s else diag(1), TIM(s)))
    for s in range(R):
        o.append(fma(U(s), dprev(s - 1) if s else diag(2), TIM(s), U(s)))
    for s in range(R):
        o.append(fma(U(s), MM(s - 1) if s else diag(0), TMM(s), U(s)))
    o.append("s_waitcnt lgkmcnt(0)")
    for s in range(R):
        o.append(mul(MM(s), PR(s), U(s)))
    # the prior registers are free again: the next step's priors are fetched now, a whole insertion chain ahead of their use
    if pre:
        o += pre
    o.append(f"v_and_or_b32 v{PADDR}, {next_ent}, s95, v{TABL}")
    o += prior_reads()
    for s in range(R):
        o.append(mul(U(s), MM(s - 1) if s else top(0), TMI(s)))
    b = b_ops(dthis, dprev)
    for s in range(R):
        o.append(fma(IM(s), IM(s - 1) if s else top(1), TII(s), U(s)))
        o += b[2 * s:2 * s + 2]
    o.append(add(TMP, MM(R - 1), IM(R - 1)))
    dpp = "wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
    for k, src in enumerate((MM(R - 1), IM(R - 1), dthis(R - 1))):
        o.append(f"v_and_b32_dpp v{diag(k)}, v{src}, v{LMASK} {dpp}")
        o.append(f"v_and_b32_dpp v{diag(k) + 1}, v{src + 1}, v{LMASK} {dpp}")
    o.append(add(SUM, SUM, TMP))
    if post:
        o += post
    return o


def program():
    o = []
    o.append(f"global_load_dwordx2 v[{EA}:{EA + 1}], v{VOFF}, s[92:93]")
    o.append(f"global_load_dwordx2 v[{EB}:{EB + 1}], v{VOFF}, s[92:93] offset:8")
    o.append("s_add_u32 s92, s92, 16")
    o.append("s_addc_u32 s93, s93, 0")
    o += b_ops(DA, DB)
    o.append("s_waitcnt vmcnt(1)")
    o.append(f"v_and_or_b32 v{PADDR}, v{EA}, s95, v{TABL}")
    o += prior_reads()
    o.append("1:")
    load_a = [f"global_load_dwordx2 v[{EA}:{EA + 1}], v{VOFF}, s[92:93]"]
    load_b = [f"global_load_dwordx2 v[{EB}:{EB + 1}], v{VOFF}, s[92:93] offset:8",
              "s_add_u32 s92, s92, 16", "s_addc_u32 s93, s93, 0"]
    o += step(DA, DB, DD, RR, f"v{EA + 1}")
    # (EA is dead once the second step's prior address is made: its reload goes out then; EB must have landed by the end
    #  of the second step -- the older of the two loads in flight)
    o += step(DB, DA, RR, DD, f"v{EB}", pre=["s_waitcnt vmcnt(0)"], post=load_a)
    o += step(DA, DB, DD, RR, f"v{EB + 1}")
    o += step(DB, DA, RR, DD, f"v{EA}", pre=["s_waitcnt vmcnt(0)"], post=load_b)
    o.append("s_sub_u32 s94, s94, 1")
    o.append("s_cmp_lg_u32 s94, 0")
    o.append("s_cbranch_scc1 1b")
    o.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return o


def strip(o):
    r = []
    for i in o:
        if 'novmem' in KNOB and (i.startswith('global_load') or 'vmcnt' in i):
            continue
        if 'nolds' in KNOB and (i.startswith('ds_read') or i == 's_waitcnt lgkmcnt(0)'):
            continue
        if 'nodpp' in KNOB and '_dpp' in i:
            continue
        r.append(i)
    return r


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "pdhmm_plain_asm.h"
    o = []
    o.append("// GENERATED by tools/gen_pdhmm_asm.py -- do not edit.  The plain-step run of the PDHMM table kernel in pinned registers.")
    o.append("#pragma once")
    o.append("namespace gklhip {")
    o.append("// `n4` x 4 plain steps of PdJob<true, false, true, true> (see the generator for the structure).  `base`: the")
    o.append("// wavefront's entry pointer for this step (uniform), `voff`: the lane's byte offset from it; every lane's entries")
    o.append("// of these steps are real columns without a special bit (the caller's job).  Ends with nothing in flight;")
    o.append("// j.asm_next[0..3]: the lane's entries of the four steps behind the run (the loop's own look-ahead).")
    o.append("template <class Job>")
    o.append("__device__ __forceinline__ void pd_plain_run_asm(Job& j, const uint32_t* base_ptr, uint32_t voff, int n4) {")
    inout = []
    for s in range(R):
        inout.append((f"j.mm[{s}]", MM(s), f"mm{s}"))
    for s in range(R):
        inout.append((f"j.im[{s}]", IM(s), f"im{s}"))
    for s in range(R):
        inout.append((f"j.dm[{s}]", DA(s), f"dm{s}"))
    for k in range(3):
        inout.append((f"j.d[{k}]", DD(k), f"dd{k}"))
    for k in range(3):
        inout.append((f"j.r[{k}]", RR(k), f"rr{k}"))
    inout.append(("j.sum", SUM, "sum"))
    consts = []
    for s in range(R):
        consts += [(f"j.tmm[{s}]", TMM(s), f"tmm{s}"), (f"j.tim[{s}]", TIM(s), f"tim{s}"), (f"j.tmi[{s}]", TMI(s), f"tmi{s}"),
                   (f"j.tii[{s}]", TII(s), f"tii{s}"), (f"j.tmd[{s}]", TMD(s), f"tmd{s}")]
    for expr, reg, name in inout + consts:
        o.append(f"  double {name} = {expr};")
    o.append("  uint64_t base = (uint64_t)(uintptr_t)base_ptr;")
    o.append("  base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |")
    o.append("         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);")
    o.append("  uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane(n4);")
    o.append("  const uint32_t msk = (uint32_t)__builtin_amdgcn_readfirstlane((int)kPdTabOffsetMask);")
    o.append("  const uint32_t lmask = j.lmask, tabl = j.tab_lane;")
    o.append("  asm volatile(")
    for ins in strip(program()):
        o.append(f"      \"{ins}\\n\\t\"")
    outs = ", ".join(f"\"+{{v[{reg}:{reg + 1}]}}\"({name})" for _, reg, name in inout)
    o.append(f"      : {outs}, \"+&{{s94}}\"(cnt), \"+&{{s[92:93]}}\"(base), \"=&{{v{EA}}}\"(j.asm_next[0]), \"=&{{v{EA + 1}}}\"(j.asm_next[1]), "
             f"\"=&{{v{EB}}}\"(j.asm_next[2]), \"=&{{v{EB + 1}}}\"(j.asm_next[3])")
    ins = ", ".join(f"\"{{v[{reg}:{reg + 1}]}}\"({name})" for _, reg, name in consts)
    o.append(f"      : {ins}, \"{{v{LMASK}}}\"(lmask), \"{{v{TABL}}}\"(tabl), \"{{v{VOFF}}}\"(voff), \"{{s95}}\"(msk)")
    clob = [f"v{r}" for r in range(DB(0), DB(R - 1) + 2)] + [f"v{r}" for r in range(U(0), PR(R - 1) + 2)] + \
           [f"v{TMP}", f"v{TMP + 1}", f"v{PADDR}"] + ["scc", "memory"]
    o.append("      : " + ", ".join(f"\"{c}\"" for c in clob) + ");")
    for expr, reg, name in inout:
        o.append(f"  {expr} = {name};")
    o.append("}")
    o.append("}  // namespace gklhip")
    text = "\n".join(o) + "\n"
    # %L / %H: low / high half of a 64-bit scalar operand -- clang has no such modifier for AMDGPU; the pair is passed as
    # two 32-bit operands instead
    with open(out, "w") as f:
        f.write(text)
    print(f"{out}: {len(program())} instructions")




def ubench(path):
    """dev tool: the step body in isolation (tools/ubench_pdstep.hip), variants with parts removed, 2 wavefronts per SIMD"""
    global KNOB
    variants = [("full step (no entry loads)", {"novmem"}), ("fp64 only", {"novmem", "nolds", "nodpp"}),
                ("fp64 + prior reads", {"novmem", "nodpp"}), ("fp64 + hand-off DPP", {"novmem", "nolds"}),
                ("full step + s_nop 1 before the DPP group", {"novmem", "nop"}),
                ("full step with entry loads", set())]
    o = ["// generated by tools/gen_pdhmm_asm.py --ubench (dev tool): the PDHMM plain step in isolation", "#include <hip/hip_runtime.h>",
         "#include <cstdio>", "#include <cstdint>",
         "#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf(\"HIP %s line %d\\n\", hipGetErrorString(e), __LINE__); return 1;} } while(0)"]
    allv = ", ".join(f"\"v{r}\"" for r in range(2, LAST + 1))
    for i, (name, knobs) in enumerate(variants):
        KNOB = knobs
        body = strip(program())
        if "nop" in knobs:
            nb = []
            for ins in body:
                if "_dpp" in ins and (not nb or "_dpp" not in nb[-1]):
                    nb.append("s_nop 1")
                nb.append(ins)
            body = nb
        o.append(f"__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k{i}(const uint32_t* ent, double* out, int iters) {{")
        o.append("  __shared__ __attribute__((aligned(16))) double l[18 * 1024 / 8];")
        o.append("  for (int k = threadIdx.x; k < 18 * 128; k += 64) l[k] = 0.999;")
        o.append("  uint64_t base = (uint64_t)(uintptr_t)ent; uint32_t cnt = (uint32_t)iters;")
        o.append("  base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);")
        o.append("  cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);")
        o.append("  asm volatile(")
        for r in range(2, 150, 2):
            o.append(f"      \"v_mov_b32 v{r}, 0\\n\\tv_mov_b32 v{r + 1}, 0x3fe00000\\n\\t\"")
        o.append(f"      \"v_mov_b32 v{LMASK}, -1\\n\\tv_lshlrev_b32 v{TABL}, 4, %2\\n\\tv_mov_b32 v{VOFF}, 0\\n\\tv_mov_b32 v{EA}, 0\\n\\tv_mov_b32 v{EA + 1}, 0\\n\\tv_mov_b32 v{EB}, 0\\n\\tv_mov_b32 v{EB + 1}, 0\\n\\t\"")
        for ins in body:
            o.append(f"      \"{ins}\\n\\t\"")
        o.append(f"      : \"+&{{s94}}\"(cnt), \"+&{{s[92:93]}}\"(base) : \"v\"(threadIdx.x), \"{{s95}}\"(0x7fffu) : {allv}, \"scc\", \"memory\");")
        o.append("  if (iters < 0) out[threadIdx.x] = l[threadIdx.x];")
        o.append("}")
    # latency probes: one dependent chain of fp64 operations
    for i, (name, ins) in enumerate((("dependent v_fma_f64 chain", "v_fma_f64 v[2:3], v[2:3], v[4:5], v[6:7]"),
                                     ("dependent v_mul_f64 chain", "v_mul_f64 v[2:3], v[2:3], v[4:5]"),
                                     ("independent v_fma_f64 x8", None))):
        o.append(f"__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void c{i}(const uint32_t* ent, double* out, int iters) {{")
        o.append("  __shared__ double l[18 * 1024 / 8]; l[threadIdx.x] = 0;")
        seq = [ins] * 8 if ins else [f"v_fma_f64 v[{10 + 2 * k}:{11 + 2 * k}], v[{10 + 2 * k}:{11 + 2 * k}], v[4:5], v[6:7]" for k in range(8)]
        o.append("  for (int i = 0; i < iters * 28; i++) asm volatile(\"" + "\\n\\t".join(seq) + "\" ::: " + ", ".join(f"\"v{r}\"" for r in range(2, 30)) + ");")
        o.append("  if (iters < 0) out[threadIdx.x] = l[threadIdx.x];")
        o.append("}")
    o.append("int main() {")
    o.append("  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0)); const double clk = 2.4e9;")
    o.append("  uint32_t* ent; double* out; CHECK(hipMalloc(&ent, 1 << 24)); CHECK(hipMemset(ent, 0, 1 << 24)); CHECK(hipMalloc(&out, 4096));")
    o.append("  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));")
    o.append("  const int iters = 2000;")
    o.append("  for (int waves = 1; waves <= 2; waves++) {")
    o.append("    const int blocks = p.multiProcessorCount * 4 * waves;")
    o.append("    printf(\"--- %d wavefront(s) per SIMD, 4 steps per iteration; cycles at 2.4 GHz\\n\", waves);")
    o.append("    float ms;")
    for i, (name, knobs) in enumerate(variants):
        o.append(f"    k{i}<<<blocks, 64>>>(ent, out, 10); CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(e0)); k{i}<<<blocks, 64>>>(ent, out, iters); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));")
        o.append(f"    printf(\"%-50s %8.3f ms  %7.1f cycles per step per wavefront, %7.1f per step per SIMD\\n\", \"{name}\", ms, ms * 1e-3 * clk / (iters * 4.0), ms * 1e-3 * clk / (iters * 4.0 * waves));")
    for i, name in enumerate(("dependent v_fma_f64 chain", "dependent v_mul_f64 chain", "independent v_fma_f64 x8")):
        o.append(f"    c{i}<<<blocks, 64>>>(ent, out, 10); CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(e0)); c{i}<<<blocks, 64>>>(ent, out, iters); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));")
        o.append(f"    printf(\"%-50s %8.3f ms  %7.2f cycles per operation per wavefront\\n\", \"{name}\", ms, ms * 1e-3 * clk / (iters * 28.0 * 8.0));")
    o.append("  }")
    o.append("  return 0;")
    o.append("}")
    KNOB = set()
    with open(path, "w") as f:
        f.write("\n".join(o) + "\n")


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[1] == "--ubench":
    ubench(sys.argv[2])
elif __name__ == "__main__":
    main()
