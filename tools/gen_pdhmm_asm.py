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
DD = lambda k: 134 + 2 * k if k < 3 else 194 + 2 * (k - 3)        # noqa: E731  (3..5: the row above's branch copies, whole-job program only)
RR = lambda k: 140 + 2 * k if k < 3 else 200 + 2 * (k - 3)        # noqa: E731
SUM, TMP, LMASK, TABL, PADDR, VOFF, EA, EB = 146, 148, 150, 151, 152, 153, 154, 156
LAST = 157
# whole-job program: the branch copies and the lane's state field
BMM = lambda s: 158 + 2 * s       # noqa: E731
BIM = lambda s: 170 + 2 * s       # noqa: E731
BDM = lambda s: 182 + 2 * s       # noqa: E731
VST = 206
JOB_LAST = 206
# scalar registers of the whole-job program
S_T, S_N, S_TOP, S_C, S_NS, S_SV, S_AD, S_X = 80, 81, 82, 83, 84, 86, 88, 90


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
    if 'predicate' in KNOB:   # timing experiment: the match predicate per cell instead of the table (what an asm predicate kernel would issue)
        o.append(f"v_and_b32 v{VST}, 0x40000000, {next_ent}")
        o.append(f"v_cmp_eq_u32_e32 vcc, 0, v{VST}")
        for s in range(R):
            o.append(f"v_and_b32 v{VST}, {next_ent}, v{TMM(s)}")
            o.append(f"v_cmp_lt_u32_e32 vcc, s95, v{VST}")
            o.append(f"v_cndmask_b32 v{PR(s)}, v{TIM(s)}, v{TII(s)}, vcc")
            o.append(f"v_cndmask_b32 v{PR(s) + 1}, v{TIM(s) + 1}, v{TII(s) + 1}, vcc")
    else:
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


# ---------------------------------------------------------------------------------------------------------------
# The whole-job program (GKL_PD_ASM = 2, the default): every step of one haplotype against the wavefront's rows inside
# ONE asm statement -- plain steps and general steps share all their invariants, so the program switches between them
# per pair of steps with one scalar compare and nothing is copied at a switch.
#   * No idle test anywhere: an idle entry (before a lane's first column, behind its last) names a prior class BEYOND
#     the wavefront's LDS table (kPdTabIdleOffset; a DS read beyond the workgroup's allocation returns 0,
#     tools/ubench_lds_oob.hip).  With prior 0 a lane that has not started stays in its initial state exactly
#     (M = 0 * x, I = fma(0, tii, 0 * tmi), D = fma(0, tii, 0) -- the pad row's D = fma(init, 1, 0)) and a lane that
#     is done adds +0 to its sum; what such lanes hand down is only ever consumed by lanes in the same condition.
#   * A general step is the plain step plus, under EXEC masks: AFTER_DEL lanes merge their left / diagonal values
#     with the branch copies first (and redo the D that was computed a step ahead from the unmerged values); every lane
#     but the INSIDE_DEL ones then takes its branch copies; DEL_END lanes redo their insertion chain with merged
#     inputs behind the normal one; twelve hand-off DPPs instead of six (pdhmm_kernel.h step_general is the C++ form).
#   * Mode per pair of steps (t, t+1): general iff a special column can reach a lane by step t + 3 (two steps of
#     lead-in rebuild the branch copies and both generations of the row above's) <=> ns[max(t - top, 0)] <= t + 3;
#     the table value is kept in a scalar register and read again only when its column has left the last lane.
def general_pre(dprev, dthis, diag, cur_ent, lab):
    o = [f"v_bfe_u32 v{VST}, {cur_ent}, 16, 2",
         f"v_cmp_eq_u32_e32 vcc, 2, v{VST}",
         f"s_and_saveexec_b64 s[{S_SV}:{S_SV + 1}], vcc",
         f"s_cbranch_execz {lab}f"]
    for s in range(R):
        o.append(f"v_max_f64 {v2(MM(s))}, {v2(MM(s))}, {v2(BMM(s))}")
        o.append(f"v_max_f64 {v2(IM(s))}, {v2(IM(s))}, {v2(BIM(s))}")
        o.append(f"v_max_f64 {v2(dprev(s))}, {v2(dprev(s))}, {v2(BDM(s))}")
    for k in range(3):
        o.append(f"v_max_f64 {v2(diag(k))}, {v2(diag(k))}, {v2(diag(k + 3))}")
    o += [mul(dthis(s), MM(s), TMD(s)) for s in range(R)] + [fma(dthis(s), dprev(s), TII(s), dthis(s)) for s in range(R)]
    o += [f"{lab}:",
          f"s_mov_b64 exec, s[{S_SV}:{S_SV + 1}]",
          f"v_cmp_ne_u32_e32 vcc, 1, v{VST}",
          f"s_and_saveexec_b64 s[{S_SV}:{S_SV + 1}], vcc"]
    for s in range(R):
        o.append(f"v_mov_b64 {v2(BMM(s))}, {v2(MM(s))}")
        o.append(f"v_mov_b64 {v2(BIM(s))}, {v2(IM(s))}")
        o.append(f"v_mov_b64 {v2(BDM(s))}, {v2(dprev(s))}")
    o.append(f"s_mov_b64 exec, s[{S_SV}:{S_SV + 1}]")
    return o


def general_del_end(top, cur_ent, lab):
    o = [f"v_and_b32 v{VST}, 0x40000, {cur_ent}",
         f"v_cmp_ne_u32_e32 vcc, 0, v{VST}",
         f"s_and_saveexec_b64 s[{S_SV}:{S_SV + 1}], vcc",
         f"s_cbranch_execz {lab}f"]
    for s in range(R):
        bm, lm = (BMM(s - 1), MM(s - 1)) if s else (top(3), top(0))
        bi, li = (BIM(s - 1), IM(s - 1)) if s else (top(4), top(1))
        o.append(f"v_max_f64 {v2(U(s))}, {v2(bm)}, {v2(lm)}")
        o.append(f"v_max_f64 {v2(TMP)}, {v2(bi)}, {v2(li)}")
        o.append(mul(U(s), U(s), TMI(s)))
        o.append(fma(IM(s), TMP, TII(s), U(s)))
    o += [f"{lab}:", f"s_mov_b64 exec, s[{S_SV}:{S_SV + 1}]"]
    return o


def job_step(dprev, dthis, diag, top, next_ent, cur_ent, general, lab, pre=None, post=None):
    o = []
    if general:
        o += general_pre(dprev, dthis, diag, cur_ent, lab)
    for s in range(R):
        o.append(mul(U(s), IM(s - 1) if s else diag(1), TIM(s)))
    for s in range(R):
        o.append(fma(U(s), dprev(s - 1) if s else diag(2), TIM(s), U(s)))
    for s in range(R):
        o.append(fma(U(s), MM(s - 1) if s else diag(0), TMM(s), U(s)))
    o.append("s_waitcnt lgkmcnt(0)")
    for s in range(R):
        o.append(mul(MM(s), PR(s), U(s)))
    if pre:
        o += pre
    if 'predicate' in KNOB:   # timing experiment: the match predicate per cell instead of the table (what an asm predicate kernel would issue)
        o.append(f"v_and_b32 v{VST}, 0x40000000, {next_ent}")
        o.append(f"v_cmp_eq_u32_e32 vcc, 0, v{VST}")
        for s in range(R):
            o.append(f"v_and_b32 v{VST}, {next_ent}, v{TMM(s)}")
            o.append(f"v_cmp_lt_u32_e32 vcc, s95, v{VST}")
            o.append(f"v_cndmask_b32 v{PR(s)}, v{TIM(s)}, v{TII(s)}, vcc")
            o.append(f"v_cndmask_b32 v{PR(s) + 1}, v{TIM(s) + 1}, v{TII(s) + 1}, vcc")
    else:
        o.append(f"v_and_or_b32 v{PADDR}, {next_ent}, s95, v{TABL}")
        o += prior_reads()
    for s in range(R):
        o.append(mul(U(s), MM(s - 1) if s else top(0), TMI(s)))
    b = b_ops(dthis, dprev)
    for s in range(R):
        o.append(fma(IM(s), IM(s - 1) if s else top(1), TII(s), U(s)))
        o += b[2 * s:2 * s + 2]
    if general:
        o += general_del_end(top, cur_ent, lab + 1)
    o.append(add(TMP, MM(R - 1), IM(R - 1)))
    dpp = "wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
    srcs = [MM(R - 1), IM(R - 1), dthis(R - 1)] + ([BMM(R - 1), BIM(R - 1), BDM(R - 1)] if general else [])
    for k, src in enumerate(srcs):
        o.append(f"v_and_b32_dpp v{diag(k)}, v{src}, v{LMASK} {dpp}")
        o.append(f"v_and_b32_dpp v{diag(k) + 1}, v{src + 1}, v{LMASK} {dpp}")
    o.append(add(SUM, SUM, TMP))
    if post:
        o += post
    return o


def job_program():
    o = []
    o.append(f"global_load_dwordx2 v[{EA}:{EA + 1}], v{VOFF}, s[92:93]")
    o.append(f"global_load_dwordx2 v[{EB}:{EB + 1}], v{VOFF}, s[92:93] offset:8")
    o.append("s_add_u32 s92, s92, 16")
    o.append("s_addc_u32 s93, s93, 0")
    # the mode of the first pair: ns[0]
    o.append(f"s_load_dword s{S_C}, s[{S_NS}:{S_NS + 1}], 0x0")
    o.append(f"s_mov_b32 s{S_T}, 0")
    # initial state: M = I = 0, branch copies 0, both hand-off sets 0 but the row above's D at this column (the pad row
    # can be the last row of the lane above), sum 0; D of the rows comes in (0, the pad row's: INITIAL_CONDITION / H)
    for r in [MM(s) for s in range(R)] + [IM(s) for s in range(R)] + [BMM(s) for s in range(R)] + [BIM(s) for s in range(R)] + \
             [BDM(s) for s in range(R)] + [DD(k) for k in range(6)] + [RR(k) for k in (0, 1, 3, 4, 5)] + [SUM]:
        o.append(f"v_mov_b64 {v2(r)}, 0")
    dpp = "wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
    o.append(f"v_and_b32_dpp v{RR(2)}, v{DA(R - 1)}, v{LMASK} {dpp}")
    o.append(f"v_and_b32_dpp v{RR(2) + 1}, v{DA(R - 1) + 1}, v{LMASK} {dpp}")
    o += b_ops(DA, DB)
    o.append("s_waitcnt vmcnt(1)")
    o.append(f"v_and_or_b32 v{PADDR}, v{EA}, s95, v{TABL}")
    o += prior_reads()
    o.append("s_waitcnt lgkmcnt(0)")   # (the scalar load; the first step waits for the priors again anyway)
    load_a = [f"global_load_dwordx2 v[{EA}:{EA + 1}], v{VOFF}, s[92:93]"]
    load_b = [f"global_load_dwordx2 v[{EB}:{EB + 1}], v{VOFF}, s[92:93] offset:8",
              "s_add_u32 s92, s92, 16", "s_addc_u32 s93, s93, 0"]

    def pair(lab0, ents, waits, posts):
        """two steps; `ents`: (cur, next) registers of the two steps"""
        q = []
        # s_c: the first special column at or behind the oldest column in flight (ns[max(t - top, 0)]).  It stays the
        # same until that column has left the last lane (t - top > s_c): only then the table is read again -- a scalar
        # load per special column instead of one per pair of steps (whose latency every step's lgkmcnt wait would pay).
        q.append(f"s_sub_i32 s{S_AD}, s{S_T}, s{S_TOP}")
        q.append(f"s_cmp_gt_i32 s{S_AD}, s{S_C}")
        q.append(f"s_cbranch_scc0 {lab0 + 6}f")
        q.append(f"s_lshl_b32 s{S_AD}, s{S_AD}, 2")          # (t - top > s_c >= 0 here)
        q.append(f"s_add_u32 s{S_AD}, s{S_NS}, s{S_AD}")
        q.append(f"s_addc_u32 s{S_AD + 1}, s{S_NS + 1}, 0")
        if 'nosmem' not in KNOB:
            q.append(f"s_load_dword s{S_C}, s[{S_AD}:{S_AD + 1}], 0x0")
            q.append("s_waitcnt lgkmcnt(0)")
        q.append(f"{lab0 + 6}:")
        # mode of this pair: general iff a special column can reach a lane by step t + 3
        q.append(f"s_add_i32 s{S_X}, s{S_T}, 3")
        q.append(f"s_cmp_le_i32 s{S_C}, s{S_X}")
        if 'allplain' in KNOB:
            q.append("s_cmp_lg_u32 0, 0")
        if 'allgeneral' in KNOB:
            q.append("s_cmp_eq_u32 0, 0")
        q.append(f"s_cbranch_scc1 {lab0}f")
        roles = [(DA, DB, DD, RR), (DB, DA, RR, DD)]
        for k in range(2):
            q += job_step(*roles[k], ents[k][1], ents[k][0], False, 0, pre=waits[k], post=posts[k])
        q.append(f"s_branch {lab0 + 1}f")
        q.append(f"{lab0}:")
        for k in range(2):
            q += job_step(*roles[k], ents[k][1], ents[k][0], True, lab0 + 2 + 2 * k, pre=waits[k], post=posts[k])
        q.append(f"{lab0 + 1}:")
        q.append(f"s_add_i32 s{S_T}, s{S_T}, 2")
        return q

    o.append("1:")
    w = ["s_waitcnt vmcnt(0)"]
    o += pair(10, [(f"v{EA}", f"v{EA + 1}"), (f"v{EA + 1}", f"v{EB}")], [None, w], [None, load_a])
    o += pair(20, [(f"v{EB}", f"v{EB + 1}"), (f"v{EB + 1}", f"v{EA}")], [None, w], [None, load_b])
    o.append(f"s_cmp_lt_i32 s{S_T}, s{S_N}")
    o.append("s_cbranch_scc1 1b")
    o.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return o


def emit_job(o):
    o.append("// One haplotype against the wavefront's rows, start to end (PdJob<true, false, true, true>, GKL_PD_ASM = 2; see the")
    o.append("// generator).  `base_ptr`: the wavefront's entry pointer at step 0 (uniform: column 0 minus `top`), `voff`: the lane's")
    o.append("// byte offset from it, `ns`: the haplotype's next-special-column table; j.dm: the rows' initial D; returns the sum in j.sum.")
    o.append("template <class Job>")
    o.append("__device__ __forceinline__ void pd_job_asm(Job& j, const uint32_t* base_ptr, uint32_t voff, int n_steps, int top, const int32_t* ns_ptr) {")
    o.append("  uint64_t base = (uint64_t)(uintptr_t)base_ptr, nsb = (uint64_t)(uintptr_t)ns_ptr;")
    o.append("  base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |")
    o.append("         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);")
    o.append("  nsb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(nsb >> 32)) << 32) |")
    o.append("        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)nsb);")
    o.append("  const uint32_t nst = (uint32_t)__builtin_amdgcn_readfirstlane(n_steps), tp = (uint32_t)__builtin_amdgcn_readfirstlane(top);")
    o.append("  const uint32_t msk = (uint32_t)__builtin_amdgcn_readfirstlane((int)kPdTabOffsetMask);")
    o.append("  const uint32_t lmask = j.lmask, tabl = j.tab_lane;")
    consts = []
    for s in range(R):
        consts += [(f"j.tmm[{s}]", TMM(s), f"tmm{s}"), (f"j.tim[{s}]", TIM(s), f"tim{s}"), (f"j.tmi[{s}]", TMI(s), f"tmi{s}"),
                   (f"j.tii[{s}]", TII(s), f"tii{s}"), (f"j.tmd[{s}]", TMD(s), f"tmd{s}")]
    dms = [(f"j.dm[{s}]", DA(s), f"dm{s}") for s in range(R)]
    for expr, reg, name in dms + consts:
        o.append(f"  double {name} = {expr};")
    o.append("  double sum;")
    o.append("  asm volatile(")
    for ins in strip(job_program()):
        o.append(f"      \"{ins}\\n\\t\"")
    outs = ", ".join(f"\"+{{v[{reg}:{reg + 1}]}}\"({name})" for _, reg, name in dms)
    o.append(f"      : {outs}, \"=&{{v[{SUM}:{SUM + 1}]}}\"(sum), \"+&{{s[92:93]}}\"(base)")
    ins = ", ".join(f"\"{{v[{reg}:{reg + 1}]}}\"({name})" for _, reg, name in consts)
    o.append(f"      : {ins}, \"{{v{LMASK}}}\"(lmask), \"{{v{TABL}}}\"(tabl), \"{{v{VOFF}}}\"(voff), \"{{s95}}\"(msk), "
             f"\"{{s{S_N}}}\"(nst), \"{{s{S_TOP}}}\"(tp), \"{{s[{S_NS}:{S_NS + 1}]}}\"(nsb)")
    pinned = {reg + h for _, reg, _ in dms + consts for h in (0, 1)} | {SUM, SUM + 1, LMASK, TABL, VOFF}
    clob = [f"v{r}" for r in range(2, JOB_LAST + 1) if r not in pinned] + \
           [f"s{S_T}", f"s{S_C}", f"s{S_SV}", f"s{S_SV + 1}", f"s{S_AD}", f"s{S_AD + 1}", f"s{S_X}", "vcc", "scc", "memory"]
    o.append("      : " + ", ".join(f"\"{c}\"" for c in clob) + ");")
    o.append("  j.sum = sum;")
    o.append("}")


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
    emit_job(o)
    o.append("}  // namespace gklhip")
    text = "\n".join(o) + "\n"
    # %L / %H: low / high half of a 64-bit scalar operand -- clang has no such modifier for AMDGPU; the pair is passed as
    # two 32-bit operands instead
    with open(out, "w") as f:
        f.write(text)
    print(f"{out}: plain run {len(program())} instructions, whole job {len(job_program())}")




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
