// Pairwise Smith-Waterman with affine gaps, back-track and CIGAR text for gfx950 -- device code
// (SURVEY.md 8 f4).
//
// What it computes: exactly what the reference's runSWOnePairBT_<engine> does (reference
// src/main/native/smithwaterman/PairWiseSW.h:65-263 smithWatermanBackTrack, :265-452 getCIGAR,
// cell rule MAIN_CODE :27-62) -- int32 scores, strict-">" tie breaks, the four overhang strategies,
// the last-row / last-column maximum with its order-dependent ties, the CIGAR text with the reference's
// "skip what does not fit / has length 0" rule.  Integer work: results are bit-exact (oracle/sw_oracle.c).
//
// Mapping: the reference sweeps anti-diagonals with one SIMD vector; here a 64-lane wavefront is a systolic
// array like the PairHMM kernels' -- lane L owns 1..8 consecutive rows (reference bases; chosen per pair so that
// all 64 lanes are in use) in registers, the
// alternate sequence streams through the lanes one column per step, the row above arrives by DPP
// wave_shr:1 (H and F, 2 values per step).  Sequences longer than 64 lanes' worth of rows run as stripes, the
// boundary row (H, F per column) carried through HBM.  One wavefront owns one pair from fill to text;
// persistent wavefronts pull pairs (longest first) from a counter.
//   * back-track: 4 bits per cell (two "who won" bits + the two gap-extension bits), shifted into one 32-bit
//     word per (row block, column) by compare + add-with-carry and stored step-major ([step][lane], the
//     anti-diagonal order the wavefront produces them in): one fully coalesced 256-byte store per step, about
//     nrow*ncol/2 bytes per pair with 8 rows per lane instead of the reference's 2 bytes per cell of a
//     1024-stride matrix;
//   * most steps have every lane inside its column range: that steady phase runs without a single select
//     (in ramp-up and drain the lanes outside their column range sit the step out under the EXEC mask);
//   * maximum: H of the last row / last column goes to two small arrays; the order-dependent tie rule only
//     ever matters among candidates equal to the global maximum, so a wave-parallel max is followed by an
//     in-order pass over those candidates (ballot + scalar loop);
//   * trace: the walker is wave-uniform (scalar registers) and advances a RUN per iteration: one gather of
//     back-track words looks 32 cells up the diagonal, 16 up the column and 16 along the row, a ballot tells
//     how far the current state (matches / deletion extension / insertion extension) carries.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pairhmm_fwd_kernel.h"  // dpp_shr1_keep, dpp_shr1_zero, kLanes

namespace gklhip {

constexpr int32_t kSwLow = INT32_MIN / 2;      // LOW_INIT_VALUE, smithwaterman_common.h:85
constexpr int32_t kSwCutoff = -100000000;      // MATRIX_MIN_CUTOFF, :84
enum { kSwMatch = 0, kSwInsert = 1, kSwDelete = 2, kSwInsertExt = 4, kSwDeleteExt = 8 };
enum { kSwSoftclip = 9, kSwIndel = 10, kSwLeadingIndel = 11, kSwIgnore = 12 };
// back-track word of one (row block, column): 4 bits per row, row 0 of the block in the highest used nibble;
// nibble = insert-extension << 3 | delete-extension << 2 | "insert beats match" << 1 | "delete beats both"
enum { kBtDel = 1, kBtIns = 2, kBtDelExt = 4, kBtInsExt = 8 };

struct SwPair {
  int64_t ref_off, alt_off;  // into SwArgs::seq
  int32_t nrow, ncol;        // len1 (reference, rows), len2 (alternate, columns)
  int64_t text_off;          // bytes: CIGAR text, [cigar_len], zero-filled by the host API
  int32_t cigar_len;
  int32_t rpl;               // rows per lane of this pair's fill, 1..8: the smallest that holds the reference in as few
                             // 64-lane stripes as 8 rows per lane would need (chosen by the host, sw_api.hip)
};

struct SwArgs {
  const uint8_t* seq;
  const SwPair* pairs;
  const int32_t* order;      // pair indices, longest first
  int32_t n_pairs;
  int32_t match, mismatch, open, extend, strategy;
  // scratch, one slab per persistent wavefront (sized by the host for the largest pair of the batch):
  uint32_t* bt;              // back-track words [stripe][step t < ncol + 64][lane]: what lane L stored at step t
                             // belongs to row block stripe*64 + L, column t - L + 1
  int32_t* aux;              // last_row[ncol + 1], last_col[nrow + 1], 2 x (carryH, carryF)[ncol + 65]
  int32_t* ops;              // run-length ops of the walk, [nrow + ncol + 4]
  int64_t bt_stride, aux_stride, ops_stride;   // slab sizes in elements
  char* text;
  int32_t* result;           // per pair: [0] alignment offset, [1] text bytes written, [2] max_i, [3] max_j
  int32_t* next;
};

__device__ __forceinline__ int32_t sw_readlane(int32_t v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ int32_t sw_max(int32_t a, int32_t b) { return a > b ? a : b; }
__device__ __forceinline__ int32_t sw_abs(int32_t a) { return a < 0 ? -a : a; }

// lane 0 of v = a scalar (no select, no exec juggling)
__device__ __forceinline__ void sw_writelane0(int32_t& v, int32_t scalar) {
  asm("v_writelane_b32 %0, %1, 0" : "+v"(v) : "s"(scalar));
}

// acc = 2 * acc + (a > b) / (a >= b): compare into VCC, then add-with-carry shifts the flag in -- two VALU
// instructions per back-track bit and no SGPR-pair traffic (the compiler's cmp + cndmask + or takes three
// plus wait states).
__device__ __forceinline__ void sw_flag_gt(uint32_t& acc, int32_t x, int32_t y) {
  asm("v_cmp_gt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
}
__device__ __forceinline__ void sw_flag_ge(uint32_t& acc, int32_t x, int32_t y) {
  asm("v_cmp_ge_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
}

// One lane's rows of one stripe.
template <int RPL>
struct SwLane {
  int32_t hl[RPL], e[RPL];   // H[i][j-1], E[i][j-1]
  uint32_t x[RPL];           // reference base of the row (0x100: no row)
  int32_t hd;                // H[i0-1][j-1]: diagonal input of the first row
  int32_t in_h, in_f;        // H, F of the row above at this step's column
  int32_t out_h, out_f;      // H, F of this lane's last row at the column just finished
  uint32_t ent;              // alternate base of this lane's column, 0x100 = none

  // MAIN_CODE (PairWiseSW.h:27-62) for the lane's RPL cells of one column.  kAll: every lane is inside its
  // column range (steady phase), so nothing needs protecting; otherwise `act` guards the state.
  template <bool kAll>
  __device__ __forceinline__ uint32_t step(int32_t open, int32_t extend, int32_t match, int32_t mismatch, bool act) {
    int32_t top_h = in_h, top_f = in_f, diag = hd;
    uint32_t word = 0;
#pragma unroll
    for (int s = 0; s < RPL; s++) {
      const int32_t open_h = hl[s] + open, ext_h = e[s] + extend;        // :29-33
      const int32_t e11 = sw_max(open_h, ext_h);
      sw_flag_ge(word, ext_h, open_h);                                   // INSERT_EXT unless open > ext, :34-35
      const int32_t ext_v = top_f + extend, open_v = top_h + open;       // :39-42
      const int32_t f11 = sw_max(ext_v, open_v);
      sw_flag_ge(word, ext_v, open_v);                                   // DELETE_EXT unless open > ext, :43-44
      const int32_t m11 = diag + (x[s] == ent ? match : mismatch);       // :47-51
      const int32_t h0 = sw_max(kSwCutoff, m11);                          // :52
      sw_flag_gt(word, e11, h0);                                         // insertion beats the diagonal, :53,55
      const int32_t h1 = sw_max(h0, e11);
      sw_flag_gt(word, f11, h1);                                         // deletion beats both, :56,58
      const int32_t h11 = sw_max(h1, f11);
      diag = hl[s];
      hl[s] = (kAll || act) ? h11 : hl[s];
      e[s] = (kAll || act) ? e11 : e[s];
      top_h = h11;
      top_f = f11;
    }
    hd = (kAll || act) ? in_h : hd;
    out_h = top_h;
    out_f = top_f;
    return word;
  }
};

// ---- phase 1: fill.  Writes back-track words, last_row[1..ncol], last_col[1..nrow].
template <int RPL>
__device__ __forceinline__ void sw_fill(const SwArgs& a, const SwPair& p, int lane) {
  constexpr int kStripeRows = kLanes * RPL;
  const int nrow = p.nrow, ncol = p.ncol;
  const uint8_t* ref = a.seq + p.ref_off;
  const uint8_t* alt = a.seq + p.alt_off;
  const bool indel = a.strategy == kSwIndel || a.strategy == kSwLeadingIndel;
  const int32_t open = a.open, extend = a.extend, match = a.match, mismatch = a.mismatch;
  int32_t* last_row = a.aux + (int64_t)blockIdx.x * a.aux_stride;
  int32_t* last_col = last_row + (ncol + 1);
  int32_t* carry = last_col + (nrow + 1);
  const int cstride = ncol + 65;
  uint32_t* bt = a.bt + (int64_t)blockIdx.x * a.bt_stride;
  const int n_stripes = (nrow + kStripeRows - 1) / kStripeRows;
  for (int st = 0; st < n_stripes; st++) {
    const int row0 = st * kStripeRows + lane * RPL;       // rows row0+1 .. row0+RPL
    const int rows_here = nrow - st * kStripeRows;        // rows from this stripe on
    const int lanes_used = rows_here >= kStripeRows ? kLanes : (rows_here + RPL - 1) / RPL;
    const int n_steps = ncol + lanes_used - 1;
    const int32_t* cin_h = carry + ((st + 1) & 1) * 2 * cstride;
    const int32_t* cin_f = cin_h + cstride;
    int32_t* cout_h = carry + (st & 1) * 2 * cstride;
    int32_t* cout_f = cout_h + cstride;
    const bool has_next = st + 1 < n_stripes;
    const bool mine = lane < lanes_used;                  // lanes past the last row stay out of memory
    SwLane<RPL> L;
#pragma unroll
    for (int s = 0; s < RPL; s++) {
      const int i = row0 + 1 + s;
      L.x[s] = i <= nrow ? (uint32_t)ref[i - 1] : 0x100u;  // never equals a byte
      L.hl[s] = indel ? open + (i - 1) * extend : 0;       // H[i][0], PairWiseSW.h:194-203
      L.e[s] = kSwLow;                                     // :205
    }
#pragma unroll
    for (int s = 0; s < RPL; s++) asm volatile("" :: "v"(L.x[s]));  // wait for the base loads before the step loops, not in them
    L.hd = (row0 == 0 || !indel) ? 0 : open + (row0 - 1) * extend;  // H[row0][0]
    L.in_h = 0; L.in_f = kSwLow; L.out_h = 0; L.out_f = kSwLow;
    L.ent = 0x100u;
    int32_t ci_h = 0, ci_f = kSwLow;
    const int last_s = nrow - 1 - row0;                    // row slot that holds row nrow in this lane (if 0..RPL-1)
    const bool holds_last = last_s >= 0 && last_s < RPL;
    uint32_t* bt_st = bt + (int64_t)st * (ncol + kLanes) * kLanes + lane;  // [t * 64]: one coalesced 256-byte store per step

    // the alternate bases arrive 64 at a time (one coalesced load per 64 steps), then one v_readlane per step
    uint32_t achunk = 0x100u;
    auto next_entry = [&](int t) -> uint32_t {
      if ((t & 63) == 0) {
        achunk = t + lane < ncol ? (uint32_t)alt[t + lane] : 0x100u;
        // consume the load HERE: otherwise its s_waitcnt vmcnt(0) lands in front of the v_readlane of every step and
        // each step also waits for the previous step's back-track store (loads and stores share vmcnt on gfx9)
        asm volatile("" :: "v"(achunk));
      }
      return (uint32_t)sw_readlane((int32_t)achunk, t & 63);
    };
    auto general_step = [&](int t) {
      const uint32_t entry = next_entry(t);
      L.ent = dpp_shr1_keep(entry, L.ent);
      const int j = t - lane + 1;
      const bool act = j >= 1 && j <= ncol && mine;
      if (st > 0) {                      // lane 0's row above comes from the previous stripe
        if ((t & 63) == 0) {
          const int c = t + 1 + lane;
          ci_h = c <= ncol ? cin_h[c] : 0;
          ci_f = c <= ncol ? cin_f[c] : kSwLow;
          asm volatile("" :: "v"(ci_h), "v"(ci_f));  // same: wait for the carry loads once per 64 steps, not per step
        }
        const int32_t vh = sw_readlane(ci_h, t & 63), vf = sw_readlane(ci_f, t & 63);
        if (lane == 0) { L.in_h = vh; L.in_f = vf; }
      } else if (lane == 0) {
        L.in_h = indel ? open + (j - 1) * extend : 0;       // H[0][j]
        L.in_f = kSwLow;                                    // F[0][j], :204
      }
      // lanes outside their column range sit the step out (EXEC mask) instead of computing it and selecting the
      // old state back: the guarded step costs the wavefront what a steady one does
      uint32_t word = 0;
      if (act) word = L.template step<true>(open, extend, match, mismatch, true);
      bt_st[(int64_t)t * kLanes] = word;   // unconditionally: slots outside the matrix are never read
      if (act) {
        if (holds_last) {
          int32_t h_last = L.hl[0];
#pragma unroll
          for (int s = 1; s < RPL; s++) h_last = s == last_s ? L.hl[s] : h_last;
          last_row[j] = h_last;
        }
        if (j == ncol) {
#pragma unroll
          for (int s = 0; s < RPL; s++)
            if (row0 + 1 + s <= nrow) last_col[row0 + 1 + s] = L.hl[s];
        }
        if (has_next && lane == kLanes - 1) { cout_h[j] = L.out_h; cout_f[j] = L.out_f; }
      }
      // the row above for the next step: lane L-1 has just finished the column lane L takes next
      const int32_t nh = (int32_t)dpp_shr1_zero((uint32_t)L.out_h), nf = (int32_t)dpp_shr1_zero((uint32_t)L.out_f);
      if (lane != 0) { L.in_h = nh; L.in_f = nf; }
    };

    int t = 0;
    if (st == 0 && !has_next) {
      // single stripe (every pair up to 64*RPL rows): ramp-up with guards, then a steady phase in which every
      // lane is inside its column range -- no selects, lane 0's boundary value from a scalar register -- then
      // the drain with guards again.
      const int steady_end = ncol - 1;                   // steps [lanes_used - 1, ncol) have all lanes active; the last of
                                                         // them (lane 0 on column ncol: last_col) is left to the guarded step
      for (; t < lanes_used - 1 && t < n_steps; t++) general_step(t);
      int32_t bnd = indel ? open + t * extend : 0;        // H[0][t + 1]
      const int32_t bnd_step = indel ? extend : 0;
      for (; t < steady_end; t++) {
        L.ent = dpp_shr1_keep(next_entry(t), L.ent);
        sw_writelane0(L.in_h, bnd);
        sw_writelane0(L.in_f, kSwLow);
        bnd += bnd_step;
        const uint32_t word = L.template step<true>(open, extend, match, mismatch, true);
        bt_st[(int64_t)t * kLanes] = word;
        if (mine) {
          if (holds_last) {
            int32_t h_last = L.hl[0];
#pragma unroll
            for (int s = 1; s < RPL; s++) h_last = s == last_s ? L.hl[s] : h_last;
            last_row[t - lane + 1] = h_last;
          }
        }
        L.in_h = (int32_t)dpp_shr1_zero((uint32_t)L.out_h);
        L.in_f = (int32_t)dpp_shr1_zero((uint32_t)L.out_f);
      }
    }
    for (; t < n_steps; t++) general_step(t);
    // carry rows, last_row/last_col and the back-track are read back by THIS wavefront only: workgroup scope (wait
    // for the stores) is all it takes.  A device-scope __threadfence() is `buffer_wbl2 sc1` + `buffer_inv sc1` on
    // gfx950 -- a write-back and invalidate of the XCD's whole L2, twice per pair and wavefront.
    __threadfence_block();
  }
}

// ---- phase 2: (max_i, max_j) of PairWiseSW.h:207-232.
__device__ __forceinline__ void sw_find_max(const SwArgs& a, const SwPair& p, int lane, int32_t* out_i, int32_t* out_j) {
  const int nrow = p.nrow, ncol = p.ncol;
  const int32_t* last_row = a.aux + (int64_t)blockIdx.x * a.aux_stride;
  const int32_t* last_col = last_row + (ncol + 1);
  const bool use_row = a.strategy == kSwSoftclip || a.strategy == kSwIgnore;
  // global maximum over the candidates
  int32_t m = INT32_MIN;
  if (use_row)
    for (int j = 1 + lane; j <= ncol; j += kLanes) m = sw_max(m, last_row[j]);
  for (int i = 1 + lane; i <= nrow; i += kLanes) m = sw_max(m, last_col[i]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = sw_max(m, __shfl_xor(m, off, kLanes));
  m = __builtin_amdgcn_readfirstlane(m);  // every lane holds the maximum: keep the rest of the pass scalar
  // candidates equal to the maximum, in the reference's order: anti-diagonal d ascending; within one d the
  // last-row cell (nrow, d - nrow) first, then the last-column cell (d - ncol, ncol)
  int32_t max_score = INT32_MIN, max_i = 0, max_j = 0;
  for (int d0 = 1; d0 <= nrow + ncol; d0 += kLanes) {
    const int d = d0 + lane;
    const int jr = d - nrow, ic = d - ncol;
    const bool rc = use_row && jr >= 1 && jr <= ncol && d <= nrow + ncol && last_row[jr] == m;
    const bool cc = ic >= 1 && ic <= nrow && d <= nrow + ncol && last_col[ic] == m;
    uint64_t rmask = __ballot(rc), cmask = __ballot(cc);
    uint64_t any = rmask | cmask;
    while (any) {
      const int l = __builtin_ctzll(any);
      const int dd = d0 + l;
      if ((rmask >> l) & 1) {
        const int j = dd - nrow;
        if (max_score < m || (max_score == m && sw_abs(nrow - j) < sw_abs(max_i - max_j))) {
          max_score = m; max_i = nrow; max_j = j;
        }
      }
      if ((cmask >> l) & 1) {
        const int i = dd - ncol;
        if (max_score < m || (max_score == m && (max_j == ncol || sw_abs(i - ncol) <= sw_abs(max_i - max_j)))) {
          max_score = m; max_i = i; max_j = ncol;
        }
      }
      any &= any - 1;
    }
  }
  *out_i = __builtin_amdgcn_readfirstlane(max_i);
  *out_j = __builtin_amdgcn_readfirstlane(max_j);
}

// fast_itoa of smithwaterman_common.cc:26-58 (0 has no digits; negatives get a '-').  Wave-uniform: every lane
// runs it, `store` (lane 0) alone writes.
__device__ __forceinline__ int sw_itoa(char* ptr, int32_t number, bool store) {
  const bool neg = number < 0;
  if (neg) number = -number;
  int digits = 0;
  for (int32_t c = number; c > 0; c /= 10) digits++;
  if (!ptr) return digits + (neg ? 1 : 0);
  if (neg) { if (store) *ptr = '-'; ptr++; }
  for (int k = digits - 1; k >= 0; k--) { if (store) ptr[k] = (char)('0' + number % 10); number /= 10; }
  return digits + (neg ? 1 : 0);
}

// ---- phase 3: back-track walk (getCIGAR, PairWiseSW.h:265-452) and text.  Everything here is
// wave-uniform (max_i / max_j arrive through v_readfirstlane, so the walker lives in scalar registers and the
// loops are scalar branches); lane 0 does the few stores.
__device__ __forceinline__ void sw_trace(const SwArgs& a, const SwPair& p, int pair_index, int lane, int32_t max_i,
                                         int32_t max_j) {
  const int nrow = p.nrow, ncol = p.ncol;
  const uint32_t* bt = a.bt + (int64_t)blockIdx.x * a.bt_stride;
  const int rpl = p.rpl;
  const int64_t stripe_words = (int64_t)(ncol + kLanes) * kLanes;
  int32_t* ops = a.ops + (int64_t)blockIdx.x * a.ops_stride;  // run-length ops in walk order: op << 28 | length
  int n_ops = 0;
  int cur_op = -1;
  int32_t cur_len = 0;
  auto push = [&](int op, int32_t len) {  // adjacent equal operations merge (:397-415)
    if (op == cur_op) { cur_len += len; return; }
    if (cur_op >= 0) { if (lane == 0) ops[n_ops] = (int32_t)(((uint32_t)cur_op << 28) | ((uint32_t)cur_len & 0xffffu)); n_ops++; }
    cur_op = op; cur_len = len;
  };
  int i, j;
  if (a.strategy == kSwIndel) { i = nrow; j = ncol; }
  else if (a.strategy == kSwLeadingIndel) { i = max_i; j = ncol; }
  else { i = max_i; j = max_j; }
  if (j < ncol) push(kSwSoftclip, ncol - j);
  // The walk, one RUN per iteration instead of one cell.  The wavefront looks ahead along the three directions the
  // walker can take from (i, j) -- 32 cells up the diagonal, 16 up the column, 16 along the row -- with ONE gather
  // of back-track words (every lane fetches the word of its own cell), then a ballot tells how far the current
  // state carries: a run of plain matches in state 0, a run of extension flags inside an insertion / deletion.
  // Cell by cell this is the reference's loop (:295-395): a cell is consumed only if the one before it left the
  // walker in the state the run assumes, and i, j > 0 is part of a cell being there at all.
  const uint64_t magic = (1ull << 32) / (uint64_t)rpl + 1u;  // (x * magic) >> 32 == x / rpl for x < 2^27 (rpl <= 8)
  int state = 0;
  int budget = nrow + ncol + 2;  // every cell consumes a row or a column: a hard bound, whatever the memory holds
  while (i > 0 && j > 0 && budget > 0) {
    const int k = lane < 32 ? lane : (lane - 32) & 15;
    const int ci = lane < 48 ? i - k : i;              // lanes 0-31 diagonal, 32-47 up (k = 0: the current cell), 48-63 left
    const int cj = (lane < 32 || lane >= 48) ? j - k : j;
    const bool there = ci > 0 && cj > 0;
    uint32_t nib = 0;
    if (there) {
      const uint32_t x = (uint32_t)(ci - 1);
      const uint32_t tb = (uint32_t)(((uint64_t)x * magic) >> 32), slot = x - tb * (uint32_t)rpl, tc = (uint32_t)(cj - 1);
      // row block tb = stripe tb / 64, lane tb % 64, stored at step tc + lane
      const uint32_t w = bt[(int64_t)(tb >> 6) * stripe_words + (int64_t)(tc + (tb & 63u)) * kLanes + (tb & 63u)];
      nib = (w >> (4u * ((uint32_t)rpl - 1u - slot))) & 0xfu;
    }
    int n;  // cells consumed by this iteration (>= 1)
    if (state == 0) {
      const uint32_t plain = (uint32_t)__ballot(there && (nib & (kBtDel | kBtIns)) == 0);  // low half: the diagonal
      n = plain == 0xffffffffu ? 32 : __builtin_ctz(~plain);   // leading plain matches
      if (n > 0) {
        n = n < budget ? n : budget;
        push(kSwMatch, n);
        i -= n; j -= n;
      } else {  // the current cell opens a deletion or an insertion (deletion wins, smithwaterman_common.h:44-48)
        const uint32_t nib0 = (uint32_t)sw_readlane((int32_t)nib, 0);
        n = 1;
        if (nib0 & kBtDel) { i--; push(kSwDelete, 1); state = (nib0 & kBtDelExt) ? kSwDeleteExt : 0; }
        else { j--; push(kSwInsert, 1); state = (nib0 & kBtInsExt) ? kSwInsertExt : 0; }
      }
    } else {
      // inside a deletion (walking up, lanes 32-47) or an insertion (walking left, lanes 48-63): every cell is
      // consumed; the first one without the extension flag ends the run and returns to state 0
      const bool del = state == kSwDeleteExt;
      const uint64_t bal = __ballot(there && (nib & (del ? kBtDelExt : kBtInsExt)) != 0);
      const uint64_t thr = __ballot(there);
      const uint32_t ext = (uint32_t)(bal >> (del ? 32 : 48)) & 0xffffu, have = (uint32_t)(thr >> (del ? 32 : 48)) & 0xffffu;
      const int lead = __builtin_ctz(~ext | 0x10000u);          // leading cells that keep the state (0..16)
      const int cells = __builtin_ctz(~have | 0x10000u);        // cells that exist (>= 1)
      n = lead + 1 <= cells ? lead + 1 : cells;                 // + the cell that clears the flag, if it is there
      if (n > 16) n = 16;
      if (n > lead) state = 0;
      n = n < budget ? n : budget;
      cur_len += n;
      if (del) i -= n; else j -= n;
    }
    budget -= n;
  }
  int32_t offset;
  if (a.strategy == kSwSoftclip) {
    if (j > 0) push(kSwSoftclip, j);
    offset = (int16_t)i;
  } else if (a.strategy == kSwIgnore) {
    if (j > 0) push(cur_op, j);           // repeats the previous operation (:372-377)
    offset = (int16_t)(i - j);
  } else {
    if (i > 0) push(kSwDelete, i);
    else if (j > 0) push(kSwInsert, j);
    offset = 0;
  }
  if (cur_op >= 0) { if (lane == 0) ops[n_ops] = (int32_t)(((uint32_t)cur_op << 28) | ((uint32_t)cur_len & 0xffffu)); n_ops++; }
  __threadfence_block();  // ops[] is read back by this wavefront
  // text, last operation first (:417-449); lengths are int16 in the reference.  Executed by the whole
  // wavefront with lane 0 storing: a lane-0-only region at the end of the persistent loop invites the compiler
  // to thread it into the next iteration's lane-0 atomic, which tears the wavefront apart (observed: hang).
  const bool store = lane == 0;
  int cur_size = 0;
  char* text = a.text + p.text_off;
  for (int k = n_ops - 1; k >= 0; k--) {
    const int32_t v = __builtin_amdgcn_readfirstlane(ops[k]);
    const int op = (int)((uint32_t)v >> 28);
    const int32_t len = (int16_t)(v & 0xffff);
    const char c = op == kSwMatch ? 'M' : op == kSwInsert ? 'I' : op == kSwDelete ? 'D' : op == kSwSoftclip ? 'S' : 'R';
    const int need = sw_itoa(nullptr, len, false) + 1;
    if (need > 1 && cur_size + need <= p.cigar_len) {
      cur_size += sw_itoa(text + cur_size, len, store);
      if (store) text[cur_size] = c;
      cur_size++;
    }
  }
  if (store) {
    int32_t* r = a.result + (int64_t)pair_index * 4;
    r[0] = offset; r[1] = cur_size; r[2] = max_i; r[3] = budget > 0 ? max_j : -1;
  }
}

#ifndef GKL_SW_WAVES
#define GKL_SW_WAVES 4
#endif
constexpr int kSwWavesPerSimd = GKL_SW_WAVES;
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GKL_SW_WAVES))) void sw_align_kernel(SwArgs a) {
  const int lane = threadIdx.x;
  for (;;) {
    int k = 0;
    if (lane == 0) k = atomicAdd(a.next, 1);
    k = __builtin_amdgcn_readfirstlane(k);
    if (k >= a.n_pairs) break;
    const int pi = a.order[k];
    const SwPair p = a.pairs[pi];
    switch (p.rpl) {
      case 1: sw_fill<1>(a, p, lane); break;
      case 2: sw_fill<2>(a, p, lane); break;
      case 3: sw_fill<3>(a, p, lane); break;
      case 4: sw_fill<4>(a, p, lane); break;
      case 5: sw_fill<5>(a, p, lane); break;
      case 6: sw_fill<6>(a, p, lane); break;
      case 7: sw_fill<7>(a, p, lane); break;
      default: sw_fill<8>(a, p, lane); break;
    }
    int32_t max_i = 0, max_j = 0;
#ifndef GKL_SW_ABL   // timing ablations (tools): 1 = fill only, 2 = fill + maximum; results are WRONG when defined
    sw_find_max(a, p, lane, &max_i, &max_j);
    sw_trace(a, p, pi, lane, max_i, max_j);
#elif GKL_SW_ABL == 2
    sw_find_max(a, p, lane, &max_i, &max_j);
    if (lane == 0) a.result[(int64_t)pi * 4 + 2] = max_i + max_j;
#endif
    __syncthreads();  // one wavefront per block: free, and it keeps the iterations of the persistent loop apart
  }
}

}  // namespace gklhip
