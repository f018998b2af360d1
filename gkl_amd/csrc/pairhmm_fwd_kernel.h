// PairHMM forward recurrence for gfx950 (MI355X) -- device code.
//
// What it computes (per (read, haplotype) pair) is the reference's
// compute_full_prob_* (reference src/main/native/pairhmm/avx-pairhmm-template.h:
// 235-372): the scaled likelihood  sum_j M[R][j] + sum_j X[R][j]  of the M/X/Y
// forward recurrence (computeMXY, :208-223) with the per-row transition
// probabilities of initializeVectors (:106-152) and the match/mismatch prior of
// stripeINITIALIZATION (:181-183).  Results are bit-identical to GKL's objects:
// the operation order and FMA pattern are the reference's (FMA=true: the gcc-11
// contraction of the AVX-512 TU; FMA=false: the unfused AVX TU), the tables come
// from the host, and fp32 denormals are flushed like MXCSR.FTZ does.
//
// How it is mapped to the machine is new (the reference's stripe/bit-mask SIMD
// scheme, :26-98,160-202, is not reproduced):
//
//   * A wavefront is a 64-lane systolic array.  Each lane owns RPL consecutive
//     read rows in registers (state M/X/Y + five transition probabilities per
//     row); a 64-lane "chunk" holds several reads packed back to back.
//   * Haplotypes are *streamed* through the array: lane L works on stream
//     position t-L at step t, so the three-term dependency (diagonal, up, left)
//     only ever crosses from lane L-1 to lane L, as one DPP wave_shr:1 per value.
//     Many haplotypes are chained in one stream (separator entries reset the
//     column state and emit results), so the 64-step pipeline fill is paid once
//     per job instead of once per pair.
//   * Row 0 of each read (M=X=0, Y=2^120/H) is a "pad" row with degenerate
//     transition probabilities, so a read boundary inside the array costs no
//     instructions; lanes that start a read zero their incoming values with a
//     per-lane AND mask (isolates pairs from each other, NaN/Inf included).
//   * The match/mismatch prior is a table in LDS (5 base codes x rows in fp32,
//     4 in fp64 where an 'N' column is gathered from the rows' own planes) indexed
//     by the haplotype base code of the lane's current column: one ds_read_b128
//     per four rows replaces a compare+select per cell.
//   * No MFMA: this is a recurrence, not a contraction.  The roofline is the fp32
//     (fp64) vector FMA rate; HBM traffic is ~1 KB per pair.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gklhip {

constexpr uint32_t kEntIdle = 5u;           // stream entry: idle column (prior 0, no LDS row)
constexpr uint32_t kEntSep = 0x80000000u;   // stream entry: separator | stream-order hap index
constexpr int kLanes = 64;
// The haplotype stream is written by an earlier kernel and only read here: reading it through the constant address
// space lets every (wave-uniform) entry load be a scalar s_load, also in the job-list kernel whose atomics make the
// compiler treat plain global memory as clobbered (there the entries came through global_load_dwordx4 + vmcnt waits).
typedef const uint32_t __attribute__((address_space(4))) StreamWord;
}  // namespace gklhip
#include "pairhmm_fwd_asm.h"  // generated: fwd_asm_run_f32r8 / fwd_asm_run_f64r10 (whole jobs, hand-allocated), fwd_fast_asm_f32r8 (fast blocks only)
#ifndef GKLHIP_FAST_ASM
#define GKLHIP_FAST_ASM 1  // 0: the C++ fast step everywhere (A/B and parity cross-check builds)
#endif
namespace gklhip {

template <typename T>
struct DevTables {
  const T* ph2pr;  // [128]
  const T* div3;   // [128]
  const T* mm;     // [8256]
};

struct DevBatch {
  const uint8_t* read_bases;
  const uint8_t* read_quals;
  const uint8_t* ins;
  const uint8_t* del;
  const uint8_t* gcp;
  const int64_t* read_off;  // device copy, [n_reads+1]
  int32_t n_reads, n_haps;
};

// One haplotype stream: haps [hap_begin, hap_end) in stream order, their columns
// and separators starting at stream[stream_begin], followed by >= 64 idle entries.
struct HapGroup {
  int32_t hap_begin, hap_end, stream_begin, pad_;
};

struct LaneSlot {  // one lane of a chunk: which read it holds, and which RPL-row block of it
  int32_t read;    // -1 = idle lane
  int32_t block;   // 0 .. n_blocks-1
};

struct FwdJob {  // one wave job of the job-list pass: stream haps [hap_begin, hap_end) through a chunk
  int32_t chunk, hap_begin, hap_end;
  int32_t klass; // the planner's length class (ordering of the job list)
};

template <typename T>
struct FwdArgs {
  DevBatch b;
  DevTables<T> tab;
  const uint32_t* stream;
  const int32_t* hap_len;   // [n_haps] stream order
  const int32_t* hap_pos;   // [n_haps] stream index of column 1, stream order
  const int32_t* hap_orig;  // [n_haps] stream order -> caller's hap index
  const T* y0;              // [n_haps] stream order: INITIAL_CONSTANT / (T)haplen (host-computed)
  const uint8_t* hap_has_n; // [n_haps] stream order: the haplotype contains an 'N' (set by prep_kernel)
  const HapGroup* groups;
  int32_t n_groups;
  const LaneSlot* chunk_lanes;  // [n_chunks * 64]
  int32_t n_chunks;
  int32_t chunk_stride;  // blocks per haplotype group in the streaming kernels' grid (>= n_chunks, see fwd_stream_block)
  T* raw;  // [n_reads * n_haps], r-major, scaled likelihood sums
  // Host-buffer calls (reference-exact log10 on the host): the packed result word of a pair (packed_word below; fp32
  // pass: the tagged fp32 sum, or 0 = "pending" when the policy will send the pair to the fp64 pass) goes straight to
  // pinned host memory from the lane that stores the sum -- the PCIe writes of a whole batch (10 MB) then trickle out
  // under the forward kernels instead of taking 0.4 ms of the policy kernel behind them.  NULL: no such output.
  uint64_t* packed_out;
  // fp64 recomputation pass: the policy's flags.  Its jobs are (chunk, haplotype run) rectangles, so a few per cent of the
  // pairs it computes were NOT flagged (they ride along): those keep their fp32 word.  NULL: every pair is wanted.
  const uint8_t* packed_only_flagged;
  // job-list mode (packed fp64 fallback): jobs are (chunk, first hap, end hap) in stream order
  const FwdJob* jobs;
  const int32_t* job_count;
  int32_t* job_next;
  int32_t asm_general;  // whole jobs in the generated asm programs (GKLHIP_ASM_GENERAL=0: C++ general steps around the fp32 fast block, C++ fp64 -- A/B and cross-checks)
  // long-read jobs of a call whose reads need super-stripes: pairhmm_fwd_super_kernel takes the jobs that meet its programs'
  // preconditions (super_takes below) and the striped kernel, launched behind it over the same list with long_filter = 2
  // and its own counter, the rest -- the C++ stripes would cost the super kernel 25 registers and a wavefront per SIMD
  int32_t long_filter;
  int64_t super_steps;  // steps a super-stripe carry row holds
};

// Host finalisation (reference-exact log10f / log10 of the host libm) needs, per pair, either the raw
// fp32 sum or the raw fp64 sum: one 8-byte word carries both cases -- the double's bits, or
// 0xFFFFFFFF:float bits (a double whose high word is all ones is a NaN no computation here produces; if
// one ever does it is replaced by the default NaN, which finalises to NaN all the same).
constexpr uint64_t kPackedF32Tag = 0xFFFFFFFF00000000ull;
__device__ __forceinline__ uint64_t packed_word(double raw) {
  const uint64_t bits = (uint64_t)__double_as_longlong(raw);
  return (bits & kPackedF32Tag) == kPackedF32Tag ? 0x7FF8000000000000ull : bits;
}

// the store of a pair's result (all kernels): the raw sum, and the packed word when the call wants one
template <typename T>
__device__ __forceinline__ void emit_result(const FwdArgs<T>& a, int64_t idx, T v) {
  a.raw[idx] = v;
  if (a.packed_out) {
    if (sizeof(T) == 4) {
      const float f = (float)v;
      // the policy's test (IntelPairHmm.cc:159; NaN compares false and stays fp32); plan_policy_kernel applies the same one
      a.packed_out[idx] = f < 1e-28f ? 0ull : (kPackedF32Tag | (uint64_t)__float_as_uint(f));
    } else if (!a.packed_only_flagged || a.packed_only_flagged[idx]) {
      a.packed_out[idx] = packed_word((double)v);
    }
  }
}

// ---- cross-lane helpers -----------------------------------------------------
// wave_shr:1 (DPP ctrl 0x138): lane L reads lane L-1 across the whole wavefront; lane 0 reads 0
// (bound_ctrl on) or keeps `old` (bound_ctrl off).
__device__ __forceinline__ uint32_t dpp_shr1_keep(uint32_t old, uint32_t src) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t dpp_shr1_zero(uint32_t src) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, 0x138, 0xf, 0xf, true);
}
// value of the lane above, ANDed with this lane's mask (0 for lanes that start a read or are idle, ~0 otherwise): one
// v_and_b32_dpp.  Written as asm WITH an s_nop in front: a DPP op issued straight behind another VALU op of the same
// wavefront costs the SIMD ~5 extra cycles (all its wavefronts wait), behind an s_nop ~2 (tools/ubench_dpp.hip: seven
// fmac + three DPP ops spread among them take 42.6 cycles per SIMD without and 34.3-34.7 with the nops, 27.2 with
// plain v_and); the compiler only adds the nops the register hazard rules demand.  Not volatile: it schedules freely.
__device__ __forceinline__ uint32_t dpp_shr1_and_nop(uint32_t src, uint32_t mask) {
  uint32_t d;
  asm("s_nop 1\n\tv_and_b32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(src), "v"(mask));
  return d;
}
__device__ __forceinline__ float recv_above(float v, uint32_t lmask) {
  return __uint_as_float(dpp_shr1_and_nop(__float_as_uint(v), lmask));   // fp32 kernel: -2.2 % (A/B on one box)
}
// (fp64: the two halves through the compiler-scheduled builtin -- with the nops, one per half or one per value, the
//  fp64 kernels measured the same)
__device__ __forceinline__ double recv_above(double v, uint32_t lmask) {
  const uint64_t u = (uint64_t)__double_as_longlong(v);
  const uint32_t lo = dpp_shr1_zero((uint32_t)u) & lmask;
  const uint32_t hi = dpp_shr1_zero((uint32_t)(u >> 32)) & lmask;
  return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// broadcast lane `src` (wave-uniform index) of v
__device__ __forceinline__ float read_lane(float v, int src) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), src));
}
__device__ __forceinline__ double read_lane(double v, int src) {
  const uint64_t u = (uint64_t)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, src);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), src);
  return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// a*b + c*d in the reference's two patterns (c*d is the term rounded first).
template <bool FMA, typename T>
__device__ __forceinline__ T mul_add2(T a, T b, T c, T d) {
  if (FMA) return fma_t(a, b, c * d);  // fma(a, b, c*d)
  return c * d + a * b;
}
// "* prior" of the M update.  fp32 uses v_mul_legacy_f32: identical to the IEEE multiply for the
// finite non-negative values of this recurrence, but 0 * (Inf|NaN) = 0, so a column whose prior
// is 0 by construction (separator / idle / pad rows) clears M -- and through M the X chain --
// even if an earlier pair overflowed.  fp64 has no such opcode: its general step selects.
__device__ __forceinline__ float mul_prior(float a, float b) {
  float r;  // (hipcc 7.2 exposes no builtin for it; plain, non-volatile asm keeps it schedulable)
  asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double mul_prior(double a, double b) { return a * b; }

// M-state inner sum: ((Md*pMM + Xd*pGAPM) + Yd*pGAPM)   (template.h:213)
template <bool FMA, typename T>
__device__ __forceinline__ T m_inner(T md, T xd, T yd, T pmm, T pgapm) {
  if (FMA) return fma_t(yd, pgapm, fma_t(xd, pgapm, md * pmm));
  return (md * pmm + xd * pgapm) + yd * pgapm;
}

// ---- the per-wave job ---------------------------------------------------------
template <typename T, int RPL, bool FMA>
struct WaveJob {
  static constexpr int kVecBytes = RPL * (int)sizeof(T) < 16 ? RPL * (int)sizeof(T) : 16;
  static constexpr int kPerVec = kVecBytes / (int)sizeof(T);        // rows per LDS vector (16 bytes; 8 for two fp32 rows per lane)
  static constexpr int kPlanes = RPL / kPerVec;                      // 16-byte vectors per lane per code
  static constexpr int kRowBytes = kPlanes * kLanes * kVecBytes;     // one base code, all rows
  // Prior table: one plane per haplotype base code.  fp32 keeps five (A C T G N = 10 KB, 4 wavefronts per SIMD
  // either way).  fp64 keeps four: 5 x 3 KB would let only 10 blocks share a CU's LDS (2.5 per SIMD), 4 x 3 KB
  // lets 13 (3 per SIMD, what the 152-160 VGPRs allow) -- measured 4 % on the fp64 pass.  A haplotype 'N'
  // matches every read base, so its prior is the row's OWN plane; such columns are rare and take the general
  // step, which gathers them row by row (haplotypes containing an N never enter the unrolled loop).
  static constexpr int kCodes = sizeof(T) == 8 ? 4 : 5;
  static constexpr int kLdsBytes = kCodes * kRowBytes;  // idle columns are handled in step_any
  // fp32, 8 rows per lane, the AVX-512 object's FMA pattern (the default arithmetic): the unrolled loop is the generated asm block
  static constexpr bool kAsmFast = GKLHIP_FAST_ASM && sizeof(T) == 4 && (RPL == 8 || RPL == 4 || RPL == 2);   // (both arithmetics: FMA = false takes the "...n" programs)
  // fp64, 10 rows per lane (the packed recomputation pass and the all-fp64 mode), same arithmetic: whole jobs in asm
  // (and 8: the wide long-read kernel, whose workgroups hold several wavefronts' prior tables)
  static constexpr bool kAsm64 = GKLHIP_FAST_ASM && sizeof(T) == 8 && (RPL == 10 || RPL == 8 || RPL == 6 || RPL == 4 || RPL == 2);
  static_assert(RPL % kPerVec == 0, "RPL must fill whole 16-byte vectors");
  using Vec = T __attribute__((ext_vector_type(kPerVec)));

  // registers
  T M[RPL], X[RPL], Y[RPL];
  T pMM[RPL], pGAPM[RPL], pMX[RPL], pXX[RPL], pMY[RPL];
  T dM, dX, dY;       // row above at the previous column (diagonal inputs)
  T rM, rX, rY;       // row above at this lane's NEXT column, fetched at the end of the previous step
  T sM, sX;           // running sums of the lane's bottom row
  uint32_t ent;       // this lane's current stream entry
  uint32_t lmask;     // 0 if this lane starts a read / is idle
  int32_t out_read;   // read whose LAST row is this lane's bottom row, else -1
  int32_t padb_slot;  // slot of the Y0-holding pad row in this lane, else -1
  uint32_t own_codes; // kCodes == 4: 2 bits per row, the plane holding the row's match prior (its own base)
  uint32_t direct;    // ~0: this lane takes the step's stream entry itself (first lane of a read, idle lane), 0: from the lane above
  int32_t skew_max;   // wave-uniform: the largest skew of a lane = how many steps an entry travels through the array
  unsigned char* lds; // this wave's prior table

  // Load one lane's rows: transition probabilities in registers, priors in LDS.
  // Read layout inside a chunk: n_blocks = ceil((R+1)/RPL) lanes, p = n_blocks*RPL-R
  // pad rows first (p-1 all-zero rows, then the Y0 row), then the R real rows, so the
  // read's last row is always the bottom row of its last lane.
  //
  // Skew: the lanes of ONE read must see column j one step after the lane above (systolic hand-off), but
  // different reads of a chunk are independent, so every read starts its own skew at 0: its first lane takes
  // the step's stream entry directly, lane b of the read sees it b steps later.  An entry (a separator in
  // particular) therefore leaves the array after `skew_max` = (largest lane count of a read in the chunk) - 1
  // steps instead of 63: shorter fill/drain per job and a shorter general-step window behind each separator.
  // `full_skew` (the striped long-read path, whose carry logic counts on lane L = skew L) keeps 0..63.
  __device__ __forceinline__ void setup(const FwdArgs<T>& a, int lane, LaneSlot slot, bool full_skew = false) {
    int R = 0, first = 0;
    int64_t roff = 0;
    out_read = -1;
    padb_slot = -1;
    lmask = 0u;
    {
      const bool takes = full_skew ? lane == 0 : (lane == 0 || slot.read < 0 || slot.block == 0);
      direct = takes ? ~0u : 0u;
      asm("" : "+v"(direct));  // opaque bit mask: as a bool the compiler makes shift_entry a v_mov + v_cndmask on top of the DPP
      int m = full_skew ? kLanes - 1 : (slot.read >= 0 ? slot.block : 0);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const int o = __shfl_xor(m, off, kLanes);
        m = o > m ? o : m;
      }
      skew_max = __builtin_amdgcn_readfirstlane(m);
    }
    if (slot.read >= 0) {
      roff = a.b.read_off[slot.read];
      R = (int)(a.b.read_off[slot.read + 1] - roff);
      const int n_blocks = (R + RPL) / RPL;  // ceil((R+1)/RPL)
      const int pads = n_blocks * RPL - R;
      first = slot.block * RPL - pads;       // read-row index (0-based) of slot 0; negative = pad
      if (slot.block == n_blocks - 1) out_read = slot.read;
      if (slot.block != 0) lmask = ~0u;
    }
    T match[RPL], mism[RPL];
    int code[RPL];
#pragma unroll
    for (int s = 0; s < RPL; s++) {
      const int v = first + s;
      pMM[s] = pGAPM[s] = pMX[s] = pXX[s] = pMY[s] = T(0);
      match[s] = mism[s] = T(0);
      code[s] = 0;
      if (slot.read >= 0 && v >= 0) {
        const int64_t at = roff + v;
        const int qi = a.b.ins[at] & 127, qd = a.b.del[at] & 127, qc = a.b.gcp[at] & 127;
        const int qq = a.b.read_quals[at] & 127;
        const int mx = qi > qd ? qi : qd, mn = qi > qd ? qd : qi;
        pMM[s] = a.tab.mm[((mx * (mx + 1)) >> 1) + mn];
        const T pc = a.tab.ph2pr[qc];
        pGAPM[s] = T(1) - pc;
        pMX[s] = a.tab.ph2pr[qi];
        pXX[s] = pc;  // == pYY
        pMY[s] = a.tab.ph2pr[qd];
        match[s] = T(1) - a.tab.ph2pr[qq];
        mism[s] = a.tab.div3[qq];
        const uint8_t bb = a.b.read_bases[at];
        code[s] = bb == 'C' ? 1 : bb == 'T' ? 2 : bb == 'G' ? 3 : bb == 'N' ? 4 : 0;
      } else if (slot.read >= 0 && v == -1) {
        pXX[s] = T(1);  // pad row holding Y0: Y stays constant, M = X = 0
        padb_slot = s;
        code[s] = -1;
      } else {
        code[s] = -1;
      }
    }
    own_codes = 0;
#pragma unroll
    for (int s = 0; s < RPL; s++) own_codes |= (uint32_t)((code[s] >= 0 && code[s] < 4) ? code[s] : 0) << (2 * s);
    // prior table: [base code][plane][lane][kPerVec rows]
#pragma unroll
    for (int c = 0; c < kCodes; c++) {
#pragma unroll
      for (int pl = 0; pl < kPlanes; pl++) {
        Vec v;
#pragma unroll
        for (int k = 0; k < kPerVec; k++) {
          const int s = pl * kPerVec + k;
          const bool real = code[s] >= 0;
          const bool hit = (c == code[s]) || (c == 4) || (code[s] == 4);
          v[k] = !real ? T(0) : (hit ? match[s] : mism[s]);
        }
        *reinterpret_cast<Vec*>(lds + c * kRowBytes + pl * (kLanes * kVecBytes) + lane * kVecBytes) = v;
      }
    }
  }

  __device__ __forceinline__ void reset_state(T y0) {
#pragma unroll
    for (int s = 0; s < RPL; s++) {
      M[s] = X[s] = T(0);
      Y[s] = (s == padb_slot) ? y0 : T(0);
    }
    dM = dX = dY = T(0);
    sM = sX = T(0);
    ent = kEntIdle;
    fetch_above();
  }

  // Cross-lane hand-off, issued at the END of a step so its DPP latency hides behind the
  // next step's M/Y updates: the lane above has just finished the column this lane does next.
  __device__ __forceinline__ void fetch_above() {
    rM = recv_above(M[RPL - 1], lmask);
    rX = recv_above(X[RPL - 1], lmask);
    rY = recv_above(Y[RPL - 1], lmask);
  }

  __device__ __forceinline__ void load_priors(uint32_t code, int lane, T* pr) const {
    // 24-bit multiply: kRowBytes is 3072 in fp64 and a plain v_mul_lo_u32 issues at quarter rate
    const unsigned char* p = lds + __umul24(code, (uint32_t)kRowBytes) + (uint32_t)lane * kVecBytes;
#pragma unroll
    for (int pl = 0; pl < kPlanes; pl++) {
      const Vec v = *reinterpret_cast<const Vec*>(p + pl * (kLanes * kVecBytes));
#pragma unroll
      for (int k = 0; k < kPerVec; k++) pr[pl * kPerVec + k] = v[k];
    }
  }
  // kCodes == 4, haplotype base 'N': row s takes its prior from the plane of its own base (a read-base 'N' row
  // and pad rows hold the same value in every plane).
  __device__ __forceinline__ void load_priors_n(int lane, T* pr) const {
#pragma unroll
    for (int s = 0; s < RPL; s++) {
      const uint32_t c = (own_codes >> (2 * s)) & 3u;
      pr[s] = *reinterpret_cast<const T*>(lds + c * (uint32_t)kRowBytes + (s / kPerVec) * (kLanes * kVecBytes) +
                                          (uint32_t)lane * kVecBytes + (s % kPerVec) * sizeof(T));
    }
  }

  // One anti-diagonal step of the recurrence for this lane's RPL rows.
  __device__ __forceinline__ void advance(const T* pr, T* nM, T* nX, T* nY) const {
    nM[0] = mul_prior(m_inner<FMA>(dM, dX, dY, pMM[0], pGAPM[0]), pr[0]);
#pragma unroll
    for (int s = 1; s < RPL; s++)
      nM[s] = mul_prior(m_inner<FMA>(M[s - 1], X[s - 1], Y[s - 1], pMM[s], pGAPM[s]), pr[s]);
#pragma unroll
    for (int s = 0; s < RPL; s++) nY[s] = mul_add2<FMA>(Y[s], pXX[s], M[s], pMY[s]);  // (:222)
    nX[0] = mul_add2<FMA>(rX, pXX[0], rM, pMX[0]);                                      // (:219)
#pragma unroll
    for (int s = 1; s < RPL; s++) nX[s] = mul_add2<FMA>(nX[s - 1], pXX[s], nM[s - 1], pMX[s]);
  }

  // This step's entry of the lane: the scalar `entry` for lanes that start a skew, else the lane above's
  // previous one (v_and_b32_dpp + v_and_or_b32).
  __device__ __forceinline__ void shift_entry(uint32_t entry) {
    uint32_t above;  // (lane above's entry) & ~direct; written as asm because the compiler leaves this AND outside the
                     // DPP move (three instructions); s_nop 1 = the wait states a DPP read needs after a VALU write
    asm("s_nop 1\n\tv_and_b32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=v"(above) : "v"(ent), "v"(~direct));
    ent = (entry & direct) | above;
  }

  // Fast step: every lane is inside a haplotype (entry = base code 0..4).
  __device__ __forceinline__ void step_fast(uint32_t entry, int lane) {
    shift_entry(entry);
    T pr[RPL], nM[RPL], nX[RPL], nY[RPL];
    load_priors(ent, lane, pr);
    advance(pr, nM, nX, nY);
#pragma unroll
    for (int s = 0; s < RPL; s++) { M[s] = nM[s]; X[s] = nX[s]; Y[s] = nY[s]; }
    dM = rM; dX = rX; dY = rY;
    fetch_above();
    sM = sM + nM[RPL - 1];  // ascending-column sums (:354-369)
    sX = sX + nX[RPL - 1];
  }

  // General step: lanes may be idle, inside a haplotype, or on a separator (end of haplotype:
  // emit the result, return to the column-0 state of the next one).  Branch-free except for
  // the result store: M and X clear themselves on a prior-0 column (see mul_prior; fp64
  // selects explicitly), Y and the running sums are selected.
  // `k_cur` (wave-uniform): the haplotype whose separator is travelling through the array during this
  // step, with its output column `orig_cur` and the next haplotype's Y0 `y0_next` already in scalar
  // registers (the caller loads them once per haplotype) -- so the lane on the separator stores its result
  // and restarts without waiting for two dependent global loads.  Any other separator (haplotypes
  // shorter than the array, the striped long-read path) takes the loads.
  __device__ __forceinline__ void step_any(const FwdArgs<T>& a, uint32_t entry, int lane,
                                           int hap_begin, int hap_end, int k_cur, int orig_cur, T y0_next) {
    constexpr bool kSelfClearing = sizeof(T) == 4;
    shift_entry(entry);
    const bool sep = (int32_t)ent < 0;
    const bool off = sep || ent == kEntIdle;  // no haplotype base in this column: prior 0
    const bool is_n = kCodes == 4 && ent == 4u;
    T pr[RPL], nM[RPL], nX[RPL], nY[RPL];
    load_priors((off || is_n) ? 0u : ent, lane, pr);
    if (kCodes == 4 && __ballot(is_n) != 0) {
      T pn[RPL];
      load_priors_n(lane, pn);
#pragma unroll
      for (int s = 0; s < RPL; s++) pr[s] = is_n ? pn[s] : pr[s];
    }
#pragma unroll
    for (int s = 0; s < RPL; s++) pr[s] = off ? T(0) : pr[s];
    advance(pr, nM, nX, nY);
    const T tM = sM + nM[RPL - 1], tX = sX + nX[RPL - 1];  // on a separator both addends are 0
    T y0n = T(0);
    if (sep) {
      const int k = (int)(ent & 0x7fffffffu);
      if (k == k_cur) {  // k_cur is always inside [hap_begin, hap_end)
        if (out_read >= 0) emit_result(a, (int64_t)out_read * a.b.n_haps + orig_cur, sM + sX);
        y0n = y0_next;
      } else {
        const bool mine = (k >= hap_begin) && (k < hap_end);
        if (mine && out_read >= 0) emit_result(a, (int64_t)out_read * a.b.n_haps + a.hap_orig[k], sM + sX);
        if (mine && k + 1 < hap_end) {
          y0n = a.y0[k + 1];
          // consume the load inside this (rare) branch: left pending, its s_waitcnt vmcnt(0) lands behind the join and
          // EVERY general step then also waits for the previous step's result store (loads and stores share vmcnt)
          asm volatile("" :: "v"(y0n));
        }
      }
    }
#pragma unroll
    for (int s = 0; s < RPL; s++) {
      M[s] = (!kSelfClearing && sep) ? T(0) : nM[s];
      X[s] = (!kSelfClearing && sep) ? T(0) : nX[s];
      Y[s] = sep ? ((s == padb_slot) ? y0n : T(0)) : nY[s];
    }
    sM = sep ? T(0) : tM;
    sX = sep ? T(0) : tX;
    dM = rM; dX = rX; dY = rY;
    fetch_above();
  }

  // fp64 general step (the steps behind a separator, the fill and the drain, haplotypes with an 'N'): the FAST step's
  // arithmetic for every lane, then two small fix-ups under the EXEC mask instead of selecting every state value
  // (a 64-bit select is two instructions, a masked 64-bit move is one, and the general step's 36 selects per step made
  // it 1.9x a fast step on a pass where every fifth step is one):
  //   * lanes on a separator or idle entry (no haplotype base in the column): their new M and X are zeroed before
  //     they enter the state and the running sums -- the prior they multiplied with is some plane's, not 0;
  //   * lanes on a separator: store the pair's sum, return to the column-0 state of the next haplotype.
  __device__ __forceinline__ void step_win(const FwdArgs<T>& a, uint32_t entry, int lane,
                                           int hap_begin, int hap_end, int k_cur, int orig_cur, T y0_next) {
    shift_entry(entry);
    const bool sep = (int32_t)ent < 0;
    const bool off = sep || ent == kEntIdle;
    const bool is_n = ent == 4u;  // haplotype 'N': the row's prior is its own plane (four planes in fp64, see kCodes)
    T pr[RPL], nM[RPL], nX[RPL], nY[RPL];
    load_priors((off || is_n) ? 0u : ent, lane, pr);
    if (__ballot(is_n) != 0) {
      asm volatile("" ::: "memory");  // a real branch: speculated, its six LDS reads and twelve selects ran in every step
      T pn[RPL];
      load_priors_n(lane, pn);
#pragma unroll
      for (int s = 0; s < RPL; s++) pr[s] = is_n ? pn[s] : pr[s];
    }
    advance(pr, nM, nX, nY);
    if (off) {
      asm volatile("" ::: "memory");  // keeps this a masked block of moves (if-converted it is two selects per value again)
#pragma unroll
      for (int s = 0; s < RPL; s++) { nM[s] = T(0); nX[s] = T(0); }
    }
#pragma unroll
    for (int s = 0; s < RPL; s++) { M[s] = nM[s]; X[s] = nX[s]; Y[s] = nY[s]; }
    sM = sM + nM[RPL - 1];
    sX = sX + nX[RPL - 1];
    if (sep) {
      asm volatile("" ::: "memory");
      const int k = (int)(ent & 0x7fffffffu);
      T y0n = T(0);
      if (k == k_cur) {  // k_cur is always inside [hap_begin, hap_end)
        if (out_read >= 0) emit_result(a, (int64_t)out_read * a.b.n_haps + orig_cur, sM + sX);
        y0n = y0_next;
      } else {
        const bool mine = (k >= hap_begin) && (k < hap_end);
        if (mine && out_read >= 0) emit_result(a, (int64_t)out_read * a.b.n_haps + a.hap_orig[k], sM + sX);
        if (mine && k + 1 < hap_end) {
          y0n = a.y0[k + 1];
          asm volatile("" :: "v"(y0n));  // consume the load inside this (rare) branch, see step_any
        }
      }
#pragma unroll
      for (int s = 0; s < RPL; s++) Y[s] = (s == padb_slot) ? y0n : T(0);  // Y0 into the lane's pad row, 0 elsewhere
      sM = T(0);
      sX = T(0);
    }
    dM = rM; dX = rX; dY = rY;
    fetch_above();
  }

  // Stream haplotypes [hap_begin, hap_end) (stream order) through the loaded rows.
  __device__ __forceinline__ void run(const FwdArgs<T>& a, int lane, int hap_begin, int hap_end) {
    constexpr int U = 8;
    if constexpr (kAsmFast || kAsm64) {
      // the whole job in the generated asm program (tools/gen_fwd_asm.py; a job with a haplotype no longer than the array
      // is deep has several separators in flight: its lanes look their output column up themselves); fp64: no haplotype
      // with an 'N' (four prior planes: such columns gather their priors row by row, see step_win)
      bool whole = a.asm_general;
      if constexpr (kAsm64) {
        if (whole) {
          if (a.packed_out) whole = false;
          bool any_n = false;
          for (int k = hap_begin + lane; k < hap_end; k += kLanes) any_n |= a.hap_has_n[k] != 0;
          if (__ballot(any_n) != 0) whole = false;
        }
      }
      if (whole) {
        if constexpr (FMA) {
          if constexpr (kAsm64 && RPL == 10)     fwd_asm_run_f64r10(*this, a, lane, hap_begin, hap_end);
          else if constexpr (kAsm64 && RPL == 8) fwd_asm_run_f64r8(*this, a, lane, hap_begin, hap_end);
          else if constexpr (kAsm64 && RPL == 6) fwd_asm_run_f64r6(*this, a, lane, hap_begin, hap_end);
          else if constexpr (kAsm64 && RPL == 4) fwd_asm_run_f64r4(*this, a, lane, hap_begin, hap_end);
          else if constexpr (kAsm64)             fwd_asm_run_f64r2(*this, a, lane, hap_begin, hap_end);
          else if constexpr (RPL == 8)           fwd_asm_run_f32r8(*this, a, lane, hap_begin, hap_end);
          else if constexpr (RPL == 4)           fwd_asm_run_f32r4(*this, a, lane, hap_begin, hap_end);
          else                                   fwd_asm_run_f32r2(*this, a, lane, hap_begin, hap_end);
        } else {   // the AVX translation unit's unfused arithmetic (fma_mode 0)
          if constexpr (kAsm64 && RPL == 10)     fwd_asm_run_f64r10n(*this, a, lane, hap_begin, hap_end);
          else if constexpr (kAsm64 && RPL == 8) fwd_asm_run_f64r8n(*this, a, lane, hap_begin, hap_end);
          else if constexpr (kAsm64 && RPL == 6) fwd_asm_run_f64r6n(*this, a, lane, hap_begin, hap_end);
          else if constexpr (kAsm64 && RPL == 4) fwd_asm_run_f64r4n(*this, a, lane, hap_begin, hap_end);
          else if constexpr (kAsm64)             fwd_asm_run_f64r2n(*this, a, lane, hap_begin, hap_end);
          else if constexpr (RPL == 8)           fwd_asm_run_f32r8n(*this, a, lane, hap_begin, hap_end);
          else if constexpr (RPL == 4)           fwd_asm_run_f32r4n(*this, a, lane, hap_begin, hap_end);
          else                                   fwd_asm_run_f32r2n(*this, a, lane, hap_begin, hap_end);
        }
        return;
      }
    }
    const int sb = a.hap_pos[hap_begin];
    StreamWord* sp = (StreamWord*)(a.stream + sb);
    reset_state(a.y0[hap_begin]);
    int t = 0;
    int fast_from = skew_max;  // the fill: the most skewed lane is idle until t = skew_max
    int k_cur = -1, orig_cur = 0;  // the separator in flight (none during the fill)
    T y0_next = T(0);
    for (int k = hap_begin; k < hap_end; k++) {
      const int sep_at = a.hap_pos[k] - sb + a.hap_len[k];  // stream-relative separator position
      if (kCodes == 4 && a.hap_has_n[k]) fast_from = sep_at;  // an 'N' somewhere in it: general steps throughout
      const int slow_end = fast_from < sep_at ? fast_from : sep_at;
      run_any(a, sp, t, slow_end, lane, hap_begin, hap_end, k_cur, orig_cur, y0_next);
      if constexpr (kAsmFast && RPL == 8 && FMA) {   // (the fast blocks alone exist for the contracted arithmetic only)
        fwd_fast_asm_f32r8(*this, sp, t, sep_at, lane);
      } else {
        for (; t + U <= sep_at; t += U) {
          uint32_t e[U];
#pragma unroll
          for (int u = 0; u < U; u++) e[u] = sp[t + u];
#pragma unroll
          for (int u = 0; u < U; u++) step_fast(e[u], lane);
        }
      }
      // < U leftover columns, still all in-haplotype.  fp32: they join the general steps below (unrolled by four with
      // the entries prefetched: measured 1.3 % faster than single fast steps that each wait for their scalar load);
      // fp64, whose general step is far more expensive: one unrolled block of four fast steps, then single ones.
      if (sizeof(T) == 8 && t >= fast_from) {
        if (t + 4 <= sep_at) {
          uint32_t e[4];
#pragma unroll
          for (int u = 0; u < 4; u++) e[u] = sp[t + u];
#pragma unroll
          for (int u = 0; u < 4; u++) step_fast(e[u], lane);
          t += 4;
        }
        for (; t < sep_at; t++) step_fast(sp[t], lane);
      }
      run_any(a, sp, t, sep_at, lane, hap_begin, hap_end, k_cur, orig_cur, y0_next);
      fast_from = sep_at + skew_max + 1;
      k_cur = k;  // from stream position sep_at on, lanes meet this haplotype's separator
      orig_cur = a.hap_orig[k];
      y0_next = k + 1 < hap_end ? a.y0[k + 1] : T(0);
    }
    run_any(a, sp, t, fast_from, lane, hap_begin, hap_end, k_cur, orig_cur, y0_next);  // drain
  }

  // General steps for stream positions [t, end): four at a time so the stream entries come from
  // one scalar load issued ahead of use and the state needs no loop-carried register copies.
  __device__ __forceinline__ void run_any(const FwdArgs<T>& a, StreamWord* sp, int& t, int end,
                                          int lane, int hap_begin, int hap_end, int k_cur, int orig_cur, T y0_next) {
    constexpr int V = 4;
    if (sizeof(T) == 8) {  // (compile time: the fp64 kernels carry one kind of general step only -- two do not fit 168 VGPRs)
      for (; t + V <= end; t += V) {
        uint32_t e[V];
#pragma unroll
        for (int u = 0; u < V; u++) e[u] = sp[t + u];
#pragma unroll
        for (int u = 0; u < V; u++) step_win(a, e[u], lane, hap_begin, hap_end, k_cur, orig_cur, y0_next);
      }
      for (; t < end; t++) step_win(a, sp[t], lane, hap_begin, hap_end, k_cur, orig_cur, y0_next);
      return;
    }
    for (; t + V <= end; t += V) {
      uint32_t e[V];
#pragma unroll
      for (int u = 0; u < V; u++) e[u] = sp[t + u];
#pragma unroll
      for (int u = 0; u < V; u++) step_any(a, e[u], lane, hap_begin, hap_end, k_cur, orig_cur, y0_next);
    }
    for (; t < end; t++) step_any(a, sp[t], lane, hap_begin, hap_end, k_cur, orig_cur, y0_next);
  }

  // One 64-lane STRIPE of a read that is longer than a chunk (the reference's stripe loop with
  // its shiftOutM/X/Y carry arrays, PH/avx-pairhmm-template.h:249,291-323, at 64*RPL rows per
  // stripe).  The row above lane 0 is the bottom row of the previous stripe: it arrives through
  // `cin` (one (M,X,Y) triple per stream position, read 64 positions at a time, coalesced, and
  // handed to lane 0 with v_readlane); this stripe's own bottom row leaves through `cout` the
  // same way.  Every step runs the general body -- this path is for rare inputs.
  __device__ __forceinline__ void run_stripe(const FwdArgs<T>& a, int lane, int hap_begin, int hap_end,
                                             const T* __restrict__ cin, T* __restrict__ cout, int clen) {
    const int sb = a.hap_pos[hap_begin];
    const uint32_t* __restrict__ sp = a.stream + sb;
    reset_state(a.y0[hap_begin]);
    const int t_end = a.hap_pos[hap_end - 1] - sb + a.hap_len[hap_end - 1] + kLanes;  // last separator + 64
    T ciM = T(0), ciX = T(0), ciY = T(0), coM = T(0), coX = T(0), coY = T(0);
    // column-0 state of the boundary row (its Y is Y0 when that row is the read's pad row):
    // the diagonal input of lane 0's first column.  Slot [3*clen ..] of the carry buffer.
    if (cout && lane == kLanes - 1) { cout[3 * clen] = M[RPL - 1]; cout[3 * clen + 1] = X[RPL - 1]; cout[3 * clen + 2] = Y[RPL - 1]; }
    if (cin && lane == 0) { dM = cin[3 * clen]; dX = cin[3 * clen + 1]; dY = cin[3 * clen + 2]; }
    for (int t = 0; t < t_end; t++) {
      if (cin) {
        if ((t & 63) == 0) {
          ciM = cin[t + lane]; ciX = cin[clen + t + lane]; ciY = cin[2 * clen + t + lane];
        }
        const T vM = read_lane(ciM, t & 63), vX = read_lane(ciX, t & 63), vY = read_lane(ciY, t & 63);
        if (lane == 0) { rM = vM; rX = vX; rY = vY; }
      }
      step_any(a, sp[t], lane, hap_begin, hap_end, -1, 0, T(0));
      if (cout) {
        const int p = t - (kLanes - 1);  // stream position lane 63 has just finished
        if (p >= 0) {
          const T bM = read_lane(M[RPL - 1], kLanes - 1), bX = read_lane(X[RPL - 1], kLanes - 1),
                  bY = read_lane(Y[RPL - 1], kLanes - 1);
          if (lane == (p & 63)) { coM = bM; coX = bX; coY = bY; }
          if ((p & 63) == 63 || t == t_end - 1) {
            const int base = p & ~63;
            cout[base + lane] = coM; cout[clen + base + lane] = coX; cout[2 * clen + base + lane] = coY;
          }
        }
      }
    }
  }
};

// ---- kernels -------------------------------------------------------------------
// Main pass: block (one wavefront) = (chunk of packed reads) x (haplotype group).
template <typename T, int RPL, bool FMA>
__device__ __forceinline__ void fwd_stream_block(const FwdArgs<T>& a, int block, unsigned char* lds) {
  using Job = WaveJob<T, RPL, FMA>;
  const int lane = threadIdx.x;
  // group-major: the groups come in order of decreasing length.  `chunk_stride` = n_chunks rounded up to a multiple of 8
  // in big launches: workgroups go to the chip's 8 XCDs round-robin by index, so a chunk then meets all its haplotype
  // groups on ONE XCD and its read rows (re-read by every job of the chunk) stay in that XCD's L2 instead of being
  // fetched through all eight (the <= 7 blocks per group beyond n_chunks leave at once).
  const int g = block / a.chunk_stride;
  const int chunk = block - g * a.chunk_stride;
  if (chunk >= a.n_chunks) return;
  const HapGroup grp = a.groups[g];
  Job job;
  job.lds = lds;
  job.setup(a, lane, a.chunk_lanes[(int64_t)chunk * kLanes + lane]);
  __syncthreads();
  job.run(a, lane, grp.hap_begin, grp.hap_end);
}
template <typename T, int RPL, bool FMA>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 8 ? 2 : 4))) void pairhmm_fwd_stream_kernel(FwdArgs<T> a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[WaveJob<T, RPL, FMA>::kLdsBytes];
  fwd_stream_block<T, RPL, FMA>(a, (int)blockIdx.x, lds);
}

// Job-list pass (packed fp64 recomputation): persistent wavefronts pull (chunk, haplotype run)
// jobs built on the device from the fallback flags; chunks come from a second read packing
// that groups reads with similar fallback patterns.  fp64 (here and in the streaming kernel): two wavefronts per SIMD
// -- at three (168 VGPRs) the general step behind the separators spills 15-27 registers at 6 rows per lane (3.45 ->
// 3.28 ms), and the 256-VGPR budget then holds 8 rows per lane without a spill (-> 2.91 ms), 10 with a few (-> 2.77).
template <typename T, int RPL, bool FMA>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 8 ? 2 : 4))) void pairhmm_fwd_jobs_kernel(FwdArgs<T> a) {
  using Job = WaveJob<T, RPL, FMA>;
  __shared__ __attribute__((aligned(16))) unsigned char lds[Job::kLdsBytes];
  const int lane = threadIdx.x;
  const int n = *a.job_count;
  Job job;
  job.lds = lds;
  int loaded_chunk = INT32_MIN;
  for (;;) {
    int idx = 0;
    if (lane == 0) idx = atomicAdd(a.job_next, 1);
    idx = __builtin_amdgcn_readfirstlane(idx);
    if (idx >= n) break;
    const FwdJob j = a.jobs[idx];
    if (j.chunk != loaded_chunk) {
      __syncthreads();  // previous job's LDS reads are done
      job.setup(a, lane, a.chunk_lanes[(int64_t)j.chunk * kLanes + lane]);
      __syncthreads();
      loaded_chunk = j.chunk;
    }
    job.run(a, lane, j.hap_begin, j.hap_end);
  }
}

// Small calls (one GATK active region: at most a few thousand pairs): precision policy, fp64 recomputation and
// finalisation of ONE pair per wavefront in one launch -- no job list, no packing, no second and third launch.  The
// wavefront of a pair whose fp32 sum passed the policy writes its result and leaves; the others hold the read alone
// (lanes 0 .. ceil((R+1)/RPL)-1) and stream that one haplotype.  Lane use is poor (a 100-base read fills a quarter of
// the lanes at 6 rows each) and irrelevant: such a call leaves most of the chip idle, what counts is the length of
// the dependent chain.
struct PairPolicyArgs {
  const float* raw32;
  double* out;              // final doubles, or packed words (mode == kModePackedWords)
  uint8_t* used64;
  int32_t* count;           // number of pairs recomputed
  const int32_t* hap_sidx;  // caller's haplotype index -> stream order
  int32_t mode;             // gklhip_finalize device modes, -1: none, kModePackedWords
  float log10_init_f;
  double log10_init32_as_f64, log10_init_d;
};
constexpr int kModePackedWords = -2;

// the policy's test on one pair; true: the pair must be recomputed in double precision
__device__ __forceinline__ bool pair_policy_head(const PairPolicyArgs& q, int64_t p, int lane) {
  const float v = q.raw32[p];
  const bool fails = v < 1e-28f;  // NaN compares false and stays fp32, like the reference (IntelPairHmm.cc:159)
  if (lane == 0) q.used64[p] = fails ? 1 : 0;
  if (!fails) {
    if (lane == 0) {
      if (q.mode == kModePackedWords) reinterpret_cast<uint64_t*>(q.out)[p] = kPackedF32Tag | (uint64_t)__float_as_uint(v);
      else if (q.mode == 1) q.out[p] = log10((double)v) - q.log10_init32_as_f64;                        // GKLHIP_FINALIZE_DEVICE_F64
      else if (q.mode == 2) q.out[p] = (double)((float)log10((double)v) - q.log10_init_f);             // GKLHIP_FINALIZE_DEVICE_REF32
    }
    return false;
  }
  if (lane == 0) atomicAdd(q.count, 1);
  return true;
}
template <int RPL, bool FMA>
__device__ __forceinline__ void pair_policy_recompute(const FwdArgs<double>& a, const PairPolicyArgs& q, int64_t p, int r, int R, int k,
                                                      unsigned char* lds) {
  using Job = WaveJob<double, RPL, FMA>;
  const int lane = threadIdx.x & 63;
  LaneSlot slot;
  slot.read = lane < (R + RPL) / RPL ? r : -1;
  slot.block = lane;
  Job job;
  job.lds = lds;
  job.setup(a, lane, slot);
  __builtin_amdgcn_wave_barrier();  // (the table is this wavefront's own: LDS operations of a wavefront execute in order)
  job.run(a, lane, k, k + 1);
  if (job.out_read >= 0) {  // the lane that stored the pair's sum
    const double sum = a.raw[p];
    if (q.mode == kModePackedWords) reinterpret_cast<uint64_t*>(q.out)[p] = packed_word(sum);
    else if (q.mode >= 0) q.out[p] = log10(sum) - q.log10_init_d;
  }
}
// One pair per wavefront.  MAXR rows per lane hold the longest read of the call; a pair whose read fits fewer rows per
// lane takes the narrower variant (same number of steps, half the instructions per step).
template <int MAXR, bool FMA>
__device__ __forceinline__ void pair_policy_block(const FwdArgs<double>& a, const PairPolicyArgs& q, int64_t p, unsigned char* lds) {
  const int lane = threadIdx.x;
  if (!pair_policy_head(q, p, lane)) return;
  const int r = (int)(p / a.b.n_haps), k = q.hap_sidx[(int)(p - (int64_t)r * a.b.n_haps)];
  const int R = (int)(a.b.read_off[r + 1] - a.b.read_off[r]);
  if (MAXR > 2 && R <= 2 * kLanes - 1)      pair_policy_recompute<2, FMA>(a, q, p, r, R, k, lds);
  else if (MAXR > 4 && R <= 4 * kLanes - 1) pair_policy_recompute<4, FMA>(a, q, p, r, R, k, lds);
  else                                      pair_policy_recompute<MAXR, FMA>(a, q, p, r, R, k, lds);
}
template <int RPL, bool FMA>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void pairhmm_pair_policy_kernel(FwdArgs<double> a, PairPolicyArgs q) {
  constexpr int kLds2 = WaveJob<double, 2, FMA>::kLdsBytes, kLds4 = WaveJob<double, 4, FMA>::kLdsBytes, kLdsR = WaveJob<double, RPL, FMA>::kLdsBytes;
  __shared__ __attribute__((aligned(16))) unsigned char lds[kLdsR > kLds4 ? (kLdsR > kLds2 ? kLdsR : kLds2) : (kLds4 > kLds2 ? kLds4 : kLds2)];
  pair_policy_block<RPL, FMA>(a, q, (int64_t)blockIdx.x, lds);
}

// Tiny calls (one GATK active region, up to a couple of thousand pairs): the WHOLE pair in one wavefront and one launch
// behind the preparation -- the pair's fp32 recurrence with the read alone in the wavefront (lanes 0 .. ceil((R+1)/RPL)-1,
// rows per lane by the read's length), the policy on its own sum and, when that fails, the fp64 recomputation right
// away.  Against the packed fp32 pass + per-pair policy launch this wastes lanes (irrelevant: the chip is mostly idle)
// and saves a launch, a dispatch of one block per pair and the chaining of several haplotypes per fp32 job: the
// dependent chain of a 100 x 10 region is one pair's ~350 steps at two rows per lane instead of a chunk's at four.
template <int RPL, bool FMA>
__device__ __forceinline__ float pair_fp32_alone(const FwdArgs<float>& f, int64_t p, int r, int R, int k, unsigned char* lds) {
  using Job = WaveJob<float, RPL, FMA>;
  const int lane = threadIdx.x & 63;
  LaneSlot slot;
  slot.read = lane < (R + RPL) / RPL ? r : -1;
  slot.block = lane;
  Job job;
  job.lds = lds;
  job.setup(f, lane, slot);
  __builtin_amdgcn_wave_barrier();  // (the table is this wavefront's own: LDS operations of a wavefront execute in order)
  job.run(f, lane, k, k + 1);
  // the lane holding the read's last row stored the sum (emit_result / the asm program): it reads its own store back
  // (past the vector cache: another wavefront of this CU may have pulled the line in before the store) and broadcasts it
  float v = 0.0f;
  if (job.out_read >= 0) {
    __builtin_amdgcn_s_waitcnt(0);
    v = __hip_atomic_load(f.raw + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const uint64_t m = __ballot(job.out_read >= 0);
  return read_lane(v, (int)__builtin_ctzll(m));
}
template <int MAXR64, bool FMA>
__device__ __forceinline__ void pair_fused_block(const FwdArgs<float>& f, const FwdArgs<double>& d, const PairPolicyArgs& q, int64_t p,
                                                 unsigned char* lds) {
  const int lane = threadIdx.x;
  const int r = (int)(p / f.b.n_haps), k = q.hap_sidx[(int)(p - (int64_t)r * f.b.n_haps)];
  const int R = (int)(f.b.read_off[r + 1] - f.b.read_off[r]);
  // (MAXR64 <= 4: the variant for calls whose reads have at most 255 bases -- no 8-row fp32 and no 6-row fp64 code in the
  //  kernel, 128 VGPRs, four wavefronts per SIMD instead of three: these wavefronts wait on their own dependent chains,
  //  so a SIMD's throughput under many concurrent callers is the number of wavefronts it holds)
  float v;
  if (R <= 2 * kLanes - 1)                     v = pair_fp32_alone<2, FMA>(f, p, r, R, k, lds);
  else if (MAXR64 <= 4 || R <= 4 * kLanes - 1) v = pair_fp32_alone<4, FMA>(f, p, r, R, k, lds);
  else                                         v = pair_fp32_alone<8, FMA>(f, p, r, R, k, lds);
  const bool fails = v < 1e-28f;  // NaN compares false and stays fp32, like the reference (IntelPairHmm.cc:159)
  if (lane == 0) {
    q.used64[p] = fails ? 1 : 0;
    if (fails) atomicAdd(q.count, 1);
    else if (q.mode == kModePackedWords) reinterpret_cast<uint64_t*>(q.out)[p] = kPackedF32Tag | (uint64_t)__float_as_uint(v);
    else if (q.mode == 1) q.out[p] = log10((double)v) - q.log10_init32_as_f64;                        // GKLHIP_FINALIZE_DEVICE_F64
    else if (q.mode == 2) q.out[p] = (double)((float)log10((double)v) - q.log10_init_f);             // GKLHIP_FINALIZE_DEVICE_REF32
  }
  if (!fails) return;
  __syncthreads();  // the fp32 table's readers are done: the fp64 table takes its place
  if (MAXR64 > 2 && R <= 2 * kLanes - 1)      pair_policy_recompute<2, FMA>(d, q, p, r, R, k, lds);
  else if (MAXR64 > 4 && R <= 4 * kLanes - 1) pair_policy_recompute<4, FMA>(d, q, p, r, R, k, lds);
  else                                        pair_policy_recompute<MAXR64, FMA>(d, q, p, r, R, k, lds);
}
template <int MAXR64, bool FMA>
struct PairFusedLds {
  static constexpr int a = WaveJob<float, (MAXR64 <= 4 ? 4 : 8), FMA>::kLdsBytes, b = WaveJob<double, 2, FMA>::kLdsBytes,
                       c = WaveJob<double, 4, FMA>::kLdsBytes, d = WaveJob<double, MAXR64, FMA>::kLdsBytes;
  static constexpr int ab = a > b ? a : b, cd = c > d ? c : d, bytes = ab > cd ? ab : cd;
};
template <int MAXR64, bool FMA>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MAXR64 <= 4 ? 4 : 3))) void pairhmm_pair_fused_kernel(FwdArgs<float> f, FwdArgs<double> d,
                                                                                                       PairPolicyArgs q) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[PairFusedLds<MAXR64, FMA>::bytes];
  pair_fused_block<MAXR64, FMA>(f, d, q, (int64_t)blockIdx.x, lds);
}

// The same with the fp64 recomputation SPECULATED: two wavefronts per pair, one runs the fp32 recurrence, the other the
// fp64 one at the same time; the policy then picks.  Five times the fp64 work a call needs (16 % of the pairs fail) --
// chosen only when the call is alone on the device (one HaplotypeCaller thread sending region after region: the usual
// way GATK runs), where nothing else wants those SIMDs and the call's latency is its longest dependent chain:
// max(fp32, fp64) instead of fp32 + fp64.
template <int MAXR64, bool FMA>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(3))) void pairhmm_pair_spec_kernel(FwdArgs<float> f, FwdArgs<double> d,
                                                                                                        PairPolicyArgs q) {
  constexpr int kL2 = WaveJob<double, 2, FMA>::kLdsBytes, kL4 = WaveJob<double, 4, FMA>::kLdsBytes, kLR = WaveJob<double, MAXR64, FMA>::kLdsBytes;
  __shared__ __attribute__((aligned(16))) unsigned char lds32[WaveJob<float, 8, FMA>::kLdsBytes];
  __shared__ __attribute__((aligned(16))) unsigned char lds64[kLR > kL4 ? (kLR > kL2 ? kLR : kL2) : (kL4 > kL2 ? kL4 : kL2)];
  __shared__ float s_v;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t p = blockIdx.x;
  const int r = (int)(p / f.b.n_haps), k = q.hap_sidx[(int)(p - (int64_t)r * f.b.n_haps)];
  const int R = (int)(f.b.read_off[r + 1] - f.b.read_off[r]);
  if (wave == 0) {
    float v;
    if (R <= 2 * kLanes - 1)      v = pair_fp32_alone<2, FMA>(f, p, r, R, k, lds32);
    else if (R <= 4 * kLanes - 1) v = pair_fp32_alone<4, FMA>(f, p, r, R, k, lds32);
    else                          v = pair_fp32_alone<8, FMA>(f, p, r, R, k, lds32);
    if (lane == 0) s_v = v;
  } else {
    PairPolicyArgs none = q;
    none.mode = -1;   // the raw fp64 sum only: whether it is wanted is decided below
    if (MAXR64 > 2 && R <= 2 * kLanes - 1)      pair_policy_recompute<2, FMA>(d, none, p, r, R, k, lds64);
    else if (MAXR64 > 4 && R <= 4 * kLanes - 1) pair_policy_recompute<4, FMA>(d, none, p, r, R, k, lds64);
    else                                        pair_policy_recompute<MAXR64, FMA>(d, none, p, r, R, k, lds64);
    __threadfence();  // the sum (a global store of this wavefront's last lane) before the barrier
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = s_v;
    const bool fails = v < 1e-28f;  // NaN compares false and stays fp32, like the reference (IntelPairHmm.cc:159)
    q.used64[p] = fails ? 1 : 0;
    if (fails) {
      atomicAdd(q.count, 1);
      const double sum = __hip_atomic_load(d.raw + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (q.mode == kModePackedWords) reinterpret_cast<uint64_t*>(q.out)[p] = packed_word(sum);
      else if (q.mode >= 0) q.out[p] = log10(sum) - q.log10_init_d;
    } else if (q.mode == kModePackedWords) reinterpret_cast<uint64_t*>(q.out)[p] = kPackedF32Tag | (uint64_t)__float_as_uint(v);
    else if (q.mode == 1) q.out[p] = log10((double)v) - q.log10_init32_as_f64;                        // GKLHIP_FINALIZE_DEVICE_F64
    else if (q.mode == 2) q.out[p] = (double)((float)log10((double)v) - q.log10_init_f);             // GKLHIP_FINALIZE_DEVICE_REF32
  }
}

// Mid-size calls (thousands of pairs): the same per-pair policy in two launches.  One block per PAIR leaves the
// dispatcher at ~150-200 blocks per microsecond, and the recomputing blocks scattered through a grid of 16 000 start up
// to 100 us late (tools/small_scaling.py: 66 us for 1000 pairs, 165 us for 16 000 with the same share recomputing).
// Here a thread per pair applies the policy and compacts the failing pairs into a list; the recomputing wavefronts
// then sit at the FRONT of a grid half the size and start at once.
__global__ __launch_bounds__(256) void pairhmm_pair_flag_kernel(PairPolicyArgs q, int32_t n_pairs, int32_t* list) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool fails = false;
  if (p < n_pairs) {
    const float v = q.raw32[p];
    fails = v < 1e-28f;  // NaN compares false and stays fp32, like the reference (IntelPairHmm.cc:159)
    q.used64[p] = fails ? 1 : 0;
    if (!fails) {
      if (q.mode == kModePackedWords) reinterpret_cast<uint64_t*>(q.out)[p] = kPackedF32Tag | (uint64_t)__float_as_uint(v);
      else if (q.mode == 1) q.out[p] = log10((double)v) - q.log10_init32_as_f64;                        // GKLHIP_FINALIZE_DEVICE_F64
      else if (q.mode == 2) q.out[p] = (double)((float)log10((double)v) - q.log10_init_f);             // GKLHIP_FINALIZE_DEVICE_REF32
    }
  }
  const uint64_t m = __ballot(fails);
  if (m) {
    int base = 0;
    if (lane == 0) base = atomicAdd(q.count, __popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    if (fails) list[base + __popcll(m & ((1ull << lane) - 1ull))] = p;
  }
}
template <int MAXR, bool FMA>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void pairhmm_pair_recompute_kernel(FwdArgs<double> a, PairPolicyArgs q,
                                                                                                           const int32_t* list) {
  constexpr int kLds2 = WaveJob<double, 2, FMA>::kLdsBytes, kLds4 = WaveJob<double, 4, FMA>::kLdsBytes, kLdsR = WaveJob<double, MAXR, FMA>::kLdsBytes;
  __shared__ __attribute__((aligned(16))) unsigned char lds[kLdsR > kLds4 ? (kLdsR > kLds2 ? kLdsR : kLds2) : (kLds4 > kLds2 ? kLds4 : kLds2)];
  const int n = *reinterpret_cast<const volatile int32_t*>(q.count);
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int64_t p = list[i];
    const int r = (int)(p / a.b.n_haps), k = q.hap_sidx[(int)(p - (int64_t)r * a.b.n_haps)];
    const int R = (int)(a.b.read_off[r + 1] - a.b.read_off[r]);
    if (MAXR > 2 && R <= 2 * kLanes - 1)      pair_policy_recompute<2, FMA>(a, q, p, r, R, k, lds);
    else if (MAXR > 4 && R <= 4 * kLanes - 1) pair_policy_recompute<4, FMA>(a, q, p, r, R, k, lds);
    else                                      pair_policy_recompute<MAXR, FMA>(a, q, p, r, R, k, lds);
    __syncthreads();  // (a block that takes a second pair reuses the prior table)
  }
}

// Long-read pass: a read with more rows than one chunk holds is processed stripe by stripe by
// one persistent wavefront per (read, haplotype run) job; `carry` is per-wavefront scratch for
// the two ping-pong carry rows (2 x 3 x carry_len values).
template <typename T, int RPL, bool FMA>
__device__ __forceinline__ void long_job_striped(const FwdArgs<T>& a, const FwdJob& j, T* my, int carry_len, unsigned char* lds) {
  using Job = WaveJob<T, RPL, FMA>;
  const int lane = threadIdx.x & 63;
  const int64_t cstride = 3 * (int64_t)carry_len + 64;  // (M,X,Y) rows + the column-0 triple
  Job job;
  job.lds = lds;
  const int r = a.chunk_lanes[(int64_t)j.chunk * kLanes].read;  // pseudo-chunk: lane 0 names the read
  const int R = (int)(a.b.read_off[r + 1] - a.b.read_off[r]);
  const int n_blocks = (R + RPL) / RPL;
  const int n_stripes = (n_blocks + kLanes - 1) / kLanes;
  const int first_cnt = n_blocks - kLanes * (n_stripes - 1);  // lanes of the first (partial) stripe
  for (int st = 0; st < n_stripes; st++) {
    LaneSlot slot;
    if (st == 0) {
      slot.read = lane >= kLanes - first_cnt ? r : -1;
      slot.block = lane - (kLanes - first_cnt);
    } else {
      slot.read = r;
      slot.block = first_cnt + (st - 1) * kLanes + lane;
    }
    __builtin_amdgcn_wave_barrier();  // (one wavefront owns this table: LDS operations of a wavefront execute in order)
    job.setup(a, lane, slot, /*full_skew=*/true);
    __builtin_amdgcn_wave_barrier();
    const T* cin = st > 0 ? my + (int64_t)((st + 1) & 1) * cstride : nullptr;
    T* cout = st + 1 < n_stripes ? my + (int64_t)(st & 1) * cstride : nullptr;
    job.run_stripe(a, lane, j.hap_begin, j.hap_end, cin, cout, carry_len);
    __threadfence_block();  // this stripe's carry stores before the next stripe's carry loads
  }
}
// Does the super-stripe kernel (kRplSuper rows per lane) run this long-read job, or does it fall to the one-wavefront stripes?
// (Wave-uniform; evaluated the same way by both kernels.)
constexpr int kRplSuper = 8;
template <typename T>
__device__ __forceinline__ bool super_takes(const FwdArgs<T>& a, const FwdJob& j, int lane) {
  using Job = WaveJob<T, kRplSuper, true>;
  if (!(Job::kAsmFast || Job::kAsm64) || !a.asm_general || a.super_steps <= 0) return false;
  const int r = a.chunk_lanes[(int64_t)j.chunk * kLanes].read;
  const int R = (int)(a.b.read_off[r + 1] - a.b.read_off[r]);
  const int G = ((R + kRplSuper) / kRplSuper + kLanes - 1) / kLanes;
  const int64_t t_end = 64 * (int64_t)(G - 1) + (a.hap_pos[j.hap_end - 1] - a.hap_pos[j.hap_begin] + a.hap_len[j.hap_end - 1]) + 64;
  // (a job that starts with a haplotype shorter than a wavefront is deep stays with the one-wavefront stripes: its streams are
  //  short against an array of G x 64 lanes -- 300 reads of 5 kb x 128 haplotypes of 30-63 bases: 37 ms striped, 144 ms here)
  if (G > kPrerollMaxWaves || t_end > a.super_steps || a.hap_len[j.hap_begin] <= kLanes - 1) return false;
  if (Job::kAsm64) {
    if (a.packed_out) return false;
    bool any_n = false;
    for (int k = j.hap_begin + lane; k < j.hap_end; k += kLanes) any_n |= a.hap_has_n[k] != 0;
    if (__ballot(any_n) != 0) return false;
  }
  return true;
}

template <typename T, int RPL, bool FMA>
__global__ __launch_bounds__(64) void pairhmm_fwd_long_kernel(FwdArgs<T> a, T* carry, int carry_len) {
  using Job = WaveJob<T, RPL, FMA>;
  __shared__ __attribute__((aligned(16))) unsigned char lds[Job::kLdsBytes];
  const int lane = threadIdx.x;
  const int n = *a.job_count;
  const int64_t cstride = 3 * (int64_t)carry_len + 64;
  T* my = carry + (int64_t)blockIdx.x * 2 * cstride;
  for (;;) {
    int idx = 0;
    if (lane == 0) idx = atomicAdd(a.job_next, 1);
    idx = __builtin_amdgcn_readfirstlane(idx);
    if (idx >= n) break;
    const FwdJob j = a.jobs[idx];
    if (a.long_filter == 2 && super_takes(a, j, lane)) continue;   // the super-stripe kernel's
    long_job_striped<T, RPL, FMA>(a, j, my, carry_len, lds);
  }
}

// Long reads at speed (round 4): a read of more rows than one wavefront holds spans the kWideWaves wavefronts of ONE
// workgroup -- lanes 64 w .. 64 w + 63 of one systolic array, the bottom row of a wavefront's lane 63 reaching the next
// wavefront's lane 0 through a ring in LDS (the reference's stripes with their carry row, avx-pairhmm-template.h:249,
// 291-323, side by side instead of one after the other) -- and every wavefront runs the generated whole-job asm program
// (fwd_asm_run_wide_*).  Reads of up to 4 x 64 x RPL - 1 bases; a job that fails the program's preconditions (longer
// reads, a haplotype no longer than a wavefront is deep, fp64: a haplotype with an 'N', the unfused arithmetic) is
// striped through memory by the workgroup's first wavefront as before.
// W = wavefronts per workgroup (2..4), chosen per call from its longest read: the workgroup's LDS is W prior tables, and
// a call whose reads need two wavefronts should not pay for four (fp64: 20 KB each -- one workgroup per CU at W = 4,
// three at W = 2).
constexpr int kWideWavesMax = 4;
template <typename T, int RPL, bool FMA, int kWideWaves>
__global__ __launch_bounds__(64 * kWideWaves) void pairhmm_fwd_wide_kernel(FwdArgs<T> a, T* carry, int carry_len) {
  using Job = WaveJob<T, RPL, FMA>;
  constexpr int kSlot = sizeof(T) == 8 ? 32 : 16;   // one (M, X, Y) triple
  constexpr int kRingSlots = 64;                    // = RING of tools/gen_fwd_asm.py
  __shared__ __attribute__((aligned(16))) unsigned char tables[kWideWaves][Job::kLdsBytes];
  __shared__ __attribute__((aligned(16))) unsigned char rings[kWideWaves - 1][kRingSlots * kSlot];
  __shared__ uint32_t flags[kWideWaves];
  __shared__ int32_t s_job;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = *a.job_count;
  const int64_t cstride = 3 * (int64_t)carry_len + 64;
  T* my = carry + (int64_t)blockIdx.x * 2 * cstride;
  for (;;) {
    __syncthreads();  // the previous job is done in every wavefront: rings and flags are free
    if (threadIdx.x == 0) s_job = atomicAdd(a.job_next, 1);
    if (threadIdx.x < kWideWaves) flags[threadIdx.x] = 0;
    __syncthreads();
    const int idx = __builtin_amdgcn_readfirstlane(s_job);
    if (idx >= n) break;
    const FwdJob j = a.jobs[idx];
    const int r = a.chunk_lanes[(int64_t)j.chunk * kLanes].read;
    const int R = (int)(a.b.read_off[r + 1] - a.b.read_off[r]);
    const int n_blocks = (R + RPL) / RPL;
    const int n_waves = (n_blocks + kLanes - 1) / kLanes;
    bool wide = (Job::kAsmFast || Job::kAsm64) && RPL >= 8 && a.asm_general && n_waves <= kWideWaves;
    if (Job::kAsm64 && wide) {
      if (a.packed_out) wide = false;
      bool any_n = false;
      for (int k = j.hap_begin + lane; k < j.hap_end; k += kLanes) any_n |= a.hap_has_n[k] != 0;
      if (__ballot(any_n) != 0) wide = false;
    }
    if (!wide) {
      if (wave == 0) long_job_striped<T, RPL, FMA>(a, j, my, carry_len, tables[0]);
      continue;
    }
    if (wave >= n_waves) continue;
    if constexpr ((Job::kAsmFast || Job::kAsm64) && RPL >= 8) {
      LaneSlot slot;
      slot.block = wave * kLanes + lane;
      slot.read = slot.block < n_blocks ? r : -1;
      Job job;
      job.lds = tables[wave];
      job.setup(a, lane, slot, /*full_skew=*/true);
      __builtin_amdgcn_wave_barrier();
      const uint32_t ring_in = wave > 0 ? (uint32_t)(uintptr_t)rings[wave - 1] : 0u;
      const uint32_t ring_out = wave + 1 < kWideWaves ? (uint32_t)(uintptr_t)rings[wave < kWideWaves - 1 ? wave : 0] : 0u;
      const uint32_t f_own = (uint32_t)(uintptr_t)&flags[wave], f_prod = (uint32_t)(uintptr_t)&flags[wave > 0 ? wave - 1 : 0],
                     f_cons = (uint32_t)(uintptr_t)&flags[wave + 1 < kWideWaves ? wave + 1 : wave];
      if constexpr (FMA) {
        if constexpr (Job::kAsm64 && RPL == 10) fwd_asm_run_wide_f64r10(job, a, lane, j.hap_begin, j.hap_end, wave, n_waves, ring_in, ring_out, f_own, f_prod, f_cons);
        else if constexpr (Job::kAsm64)         fwd_asm_run_wide_f64r8(job, a, lane, j.hap_begin, j.hap_end, wave, n_waves, ring_in, ring_out, f_own, f_prod, f_cons);
        else                                    fwd_asm_run_wide_f32r8(job, a, lane, j.hap_begin, j.hap_end, wave, n_waves, ring_in, ring_out, f_own, f_prod, f_cons);
      } else {
        if constexpr (Job::kAsm64 && RPL == 10) fwd_asm_run_wide_f64r10n(job, a, lane, j.hap_begin, j.hap_end, wave, n_waves, ring_in, ring_out, f_own, f_prod, f_cons);
        else if constexpr (Job::kAsm64)         fwd_asm_run_wide_f64r8n(job, a, lane, j.hap_begin, j.hap_end, wave, n_waves, ring_in, ring_out, f_own, f_prod, f_cons);
        else                                    fwd_asm_run_wide_f32r8n(job, a, lane, j.hap_begin, j.hap_end, wave, n_waves, ring_in, ring_out, f_own, f_prod, f_cons);
      }
    }
  }
}


// The helper wavefront of a super-stripe (pairhmm_fwd_super_kernel): producer of the ring into the first compute wavefront
// (slots filled from the previous super-stripe's carry row `cin`), consumer of the ring out of the last one (slots copied into
// this super-stripe's carry row `cout`); lane l moves the slot of step 64 k + l.
template <int kSlot>
__device__ __attribute__((noinline)) void super_helper(const uint64_t* cin, uint64_t* cout, uint32_t* flags, unsigned char* ring_first, unsigned char* ring_last,
                                                       int f_first, int f_last, int f_drain, bool feed, bool drain, int t_end, int lane) {
  constexpr int kRingSlots = 64, kWords = kSlot / 8;
  typedef volatile uint32_t __attribute__((address_space(3))) LdsU32;
  typedef volatile uint64_t __attribute__((address_space(3))) LdsU64;
  LdsU32* vflags = (LdsU32*)(uintptr_t)(uint32_t)(uintptr_t)flags;   // (LDS addresses: the low 32 bits of the generic pointers)
  LdsU64* ring_f = (LdsU64*)(uintptr_t)(uint32_t)(uintptr_t)ring_first;
  LdsU64* ring_d = (LdsU64*)(uintptr_t)(uint32_t)(uintptr_t)ring_last;
  int tf = feed ? 0 : t_end, td = drain ? 0 : t_end;
  // lane l carries the slot of step 64 k + l (= ring slot l): the 64 steps in hand and the 64 behind them, fetched a whole
  // ring ahead (an HBM round trip is longer than a group of eight steps).  Every pass moves as many steps as the
  // neighbours allow -- up to a ring's worth -- not one group: at 8 fp32 rows per lane a group of eight steps is ~2 us of a
  // compute wavefront, no more than one pass of this loop.
  uint64_t cur[kWords], nxt[kWords];
  auto fetch = [&](uint64_t (&dst)[kWords], int first) {
#pragma unroll
    for (int w = 0; w < kWords; w++)
      dst[w] = (feed && first + lane < t_end) ? __hip_atomic_load(const_cast<uint64_t*>(cin) + (int64_t)(first + lane) * kWords + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  };
  fetch(cur, 0);
  fetch(nxt, kRingSlots);
  while (tf < t_end || td < t_end) {
    bool moved = false;
    if (tf < t_end) {
      // slot T may be overwritten once the first compute wavefront has fetched step T - 64; steps in hand end at the ring's end
      int upto = (int)vflags[f_first] + kRingSlots;
      const int hand = (tf & ~(kRingSlots - 1)) + kRingSlots;
      upto = upto < hand ? upto : hand;
      upto = upto < t_end ? upto : t_end;
      if (upto < t_end) upto &= ~7;   // whole groups of eight (the consumer waits for group ends)
      if (upto > tf) {
        const int first = tf & (kRingSlots - 1), cnt = upto - tf;
        if (lane >= first && lane < first + cnt) {
#pragma unroll
          for (int w = 0; w < kWords; w++) ring_f[lane * kWords + w] = cur[w];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        tf = upto;
        if (lane == 0) vflags[0] = (uint32_t)tf;
        if ((tf & (kRingSlots - 1)) == 0) {
#pragma unroll
          for (int w = 0; w < kWords; w++) cur[w] = nxt[w];
          fetch(nxt, tf + kRingSlots);
        }
        moved = true;
      }
    }
    if (td < t_end) {
      int done = (int)vflags[f_last];          // steps the last compute wavefront has finished (published per group of eight)
      done = done < t_end ? done : t_end;
      done = done < td + kRingSlots ? done : td + kRingSlots;
      if (done > td) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int cnt = done - td;
        if (lane < cnt) {
          const int at = ((td + lane) & (kRingSlots - 1)) * kWords;
#pragma unroll
          for (int w = 0; w < kWords; w++) cout[(int64_t)(td + lane) * kWords + w] = ring_d[at + w];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        td = done;
        if (lane == 0) vflags[f_drain] = (uint32_t)td;
        moved = true;
      }
    }
    if (!moved) __builtin_amdgcn_s_sleep(1);
  }
}

// Reads of ANY length at the wide kernel's speed (round 5): a read whose rows need more wavefronts than one workgroup
// holds is processed in SUPER-STRIPES of kW wavefronts -- the reference's stripes with their carry row
// (avx-pairhmm-template.h:249, 291-323), here 64 * kW * RPL rows at a time.  Inside a super-stripe the kW compute wavefronts
// are the wide kernel's (same generated programs, the LDS rings between them); what crosses from one super-stripe to the
// next -- the bottom row of its last lane, one (M, X, Y) triple per step -- goes through HBM, carried by ONE helper
// wavefront per workgroup that plays the neighbour on both open ends: towards the first compute wavefront it is the
// producer of a ring (slots filled from the previous super-stripe's carry row), towards the last one the consumer of a
// ring (slots copied out into this super-stripe's carry row), with the same step-counter protocol the compute
// wavefronts use among themselves.  The programs run on GLOBAL wavefront indices (wavefront g of G: 64 g steps of
// pre-roll, t_end of the whole array), so step T of every super-stripe is step T of the one long array and the carry
// row is simply indexed by T; the later super-stripes pay 64 * kW * s extra pre-roll steps for it (~5 % at 15 kb).
// A job that fails the programs' preconditions is left to the striped kernel, launched behind this one (FwdArgs::long_filter).
template <typename T, int RPL, int kW, bool FMA>
__global__ __launch_bounds__(64 * (kW + 1)) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 8 ? 2 : 4))) void pairhmm_fwd_super_kernel(FwdArgs<T> a, unsigned char* xcarry) {
  const int64_t xsteps = a.super_steps;
  using Job = WaveJob<T, RPL, FMA>;
  constexpr int kSlot = sizeof(T) == 8 ? 32 : 16;   // one (M, X, Y) triple (the wide programs' ring slot)
  constexpr int kRingSlots = 64;
  constexpr int kWords = kSlot / 8;                 // 64-bit words per slot
  __shared__ __attribute__((aligned(16))) unsigned char tables[kW][Job::kLdsBytes];
  __shared__ __attribute__((aligned(16))) unsigned char rings[kW + 1][kRingSlots * kSlot];   // ring w: into compute wavefront w (0: from the helper); ring kW: into the helper
  __shared__ uint32_t flags[kW + 2];   // [0] helper as producer (steps fed), [1 + w] compute wavefront w, [kW + 1] helper as consumer (steps drained)
  __shared__ int32_t s_job;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;   // wv < kW: compute, kW: the helper
  const int n = *a.job_count;
  uint64_t* xbuf = reinterpret_cast<uint64_t*>(xcarry + (int64_t)blockIdx.x * 2 * xsteps * kSlot);
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) s_job = atomicAdd(a.job_next, 1);
    __syncthreads();
    const int idx = __builtin_amdgcn_readfirstlane(s_job);
    if (idx >= n) break;
    const FwdJob j = a.jobs[idx];
    const int r = a.chunk_lanes[(int64_t)j.chunk * kLanes].read;
    const int R = (int)(a.b.read_off[r + 1] - a.b.read_off[r]);
    const int n_blocks = (R + RPL) / RPL;
    const int G = (n_blocks + kLanes - 1) / kLanes;          // wavefronts of the whole array
    const int t_end = 64 * (G - 1) + (a.hap_pos[j.hap_end - 1] - a.hap_pos[j.hap_begin] + a.hap_len[j.hap_end - 1]) + 64;   // = the programs' own
    static_assert(RPL == kRplSuper, "super_takes speaks for this kernel");
    if (!super_takes(a, j, lane)) continue;   // (the striped kernel's, launched behind this one)
    if constexpr ((Job::kAsmFast || Job::kAsm64) && RPL >= 8) {
      const int n_stripes = (G + kW - 1) / kW;
      for (int st = 0; st < n_stripes; st++) {
        __syncthreads();   // the previous super-stripe is done in every wavefront: tables, rings and flags are free, its carry row is complete
        if (threadIdx.x < kW + 2) flags[threadIdx.x] = 0;
        __syncthreads();
        const int nw = G - st * kW < kW ? G - st * kW : kW;   // compute wavefronts of this super-stripe
        const bool feed = st > 0, drain = st + 1 < n_stripes;
        if (wv < nw) {
          const int g = st * kW + wv;
          LaneSlot slot;
          slot.block = g * kLanes + lane;
          slot.read = slot.block < n_blocks ? r : -1;
          Job job;
          job.lds = tables[wv];
          job.setup(a, lane, slot, /*full_skew=*/true);
          __builtin_amdgcn_wave_barrier();
          const uint32_t ring_in = (uint32_t)(uintptr_t)rings[wv];
          const uint32_t ring_out = (uint32_t)(uintptr_t)rings[wv + 1 < nw ? wv + 1 : kW];
          const uint32_t f_own = (uint32_t)(uintptr_t)&flags[1 + wv], f_prod = (uint32_t)(uintptr_t)&flags[wv],
                         f_cons = (uint32_t)(uintptr_t)&flags[wv + 1 < nw ? wv + 2 : kW + 1];
          if constexpr (FMA) {
            if constexpr (Job::kAsm64 && RPL == 10) fwd_asm_run_wide_f64r10(job, a, lane, j.hap_begin, j.hap_end, g, G, ring_in, ring_out, f_own, f_prod, f_cons);
            else if constexpr (Job::kAsm64)         fwd_asm_run_wide_f64r8(job, a, lane, j.hap_begin, j.hap_end, g, G, ring_in, ring_out, f_own, f_prod, f_cons);
            else                                    fwd_asm_run_wide_f32r8(job, a, lane, j.hap_begin, j.hap_end, g, G, ring_in, ring_out, f_own, f_prod, f_cons);
          } else {
            if constexpr (Job::kAsm64 && RPL == 10) fwd_asm_run_wide_f64r10n(job, a, lane, j.hap_begin, j.hap_end, g, G, ring_in, ring_out, f_own, f_prod, f_cons);
            else if constexpr (Job::kAsm64)         fwd_asm_run_wide_f64r8n(job, a, lane, j.hap_begin, j.hap_end, g, G, ring_in, ring_out, f_own, f_prod, f_cons);
            else                                    fwd_asm_run_wide_f32r8n(job, a, lane, j.hap_begin, j.hap_end, g, G, ring_in, ring_out, f_own, f_prod, f_cons);
          }
        } else if (wv == kW && (feed || drain)) {
          // the helper (its own function: its registers must not weigh on the compute wavefronts' allocation)
          super_helper<kSlot>(reinterpret_cast<const uint64_t*>(xbuf + (int64_t)((st + 1) & 1) * xsteps * kWords), xbuf + (int64_t)(st & 1) * xsteps * kWords,
                              flags, rings[0], rings[kW], /*flag of the first compute wavefront*/ 1, /*of the last*/ nw, /*own, as consumer*/ kW + 1,
                              feed, drain, t_end, lane);
          __threadfence();   // this super-stripe's carry row before the next one's loads
        }
      }
    }
  }
}

}  // namespace gklhip
