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
// array like the PairHMM kernels' -- lane L owns 4 consecutive rows (reference bases) in registers, the
// alternate sequence streams through the lanes one column per step, the row above arrives by DPP
// wave_shr:1 (H and F, 2 values per step).  Sequences longer than 256 rows run as consecutive stripes, the
// boundary row (H, F per column) carried through HBM.  One wavefront owns one pair from fill to text;
// persistent wavefronts pull pairs (longest first) from a counter.
//   * back-track: 4 bits per cell (2 direction bits + the two gap-extension bits), 16 bits per (4-row
//     block, column), written as one dword per two columns: nrow*ncol/2 bytes per pair instead of the
//     reference's 2 bytes per cell of a 1024-stride matrix;
//   * maximum: H of the last row / last column goes to two small arrays; the order-dependent tie rule only
//     ever matters among candidates equal to the global maximum, so a wave-parallel max is followed by an
//     in-order pass over those candidates (ballot + scalar loop);
//   * trace: the walker is wave-uniform (scalar registers); the wavefront prefetches a 16-row x 32-column
//     tile of back-track nibbles per global load and walks inside it with v_readlane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pairhmm_fwd_kernel.h"  // dpp_shr1_keep, dpp_shr1_zero, kLanes

namespace gklhip {

constexpr int kSwRpl = 4;
constexpr int kSwStripeRows = kLanes * kSwRpl;
constexpr int32_t kSwLow = INT32_MIN / 2;      // LOW_INIT_VALUE, smithwaterman_common.h:85
constexpr int32_t kSwCutoff = -100000000;      // MATRIX_MIN_CUTOFF, :84
enum { kSwMatch = 0, kSwInsert = 1, kSwDelete = 2, kSwInsertExt = 4, kSwDeleteExt = 8 };
enum { kSwSoftclip = 9, kSwIndel = 10, kSwLeadingIndel = 11, kSwIgnore = 12 };

struct SwPair {
  int64_t ref_off, alt_off;  // into SwArgs::seq
  int32_t nrow, ncol;        // len1 (reference, rows), len2 (alternate, columns)
  int64_t bt_off;            // 16-bit units: [(nrow + 3) / 4][ncolp], ncolp = ncol rounded up to even
  int64_t aux_off;           // int32 units: last_row[ncol + 1], last_col[nrow + 1], 2 x (carryH, carryF)[ncol + 65]
  int64_t ops_off;           // int32 units: run-length ops of the walk, [nrow + ncol + 4]
  int64_t text_off;          // bytes: CIGAR text, [cigar_len], zero-filled by the host API
  int32_t cigar_len, pad_;
};

struct SwArgs {
  const uint8_t* seq;
  const SwPair* pairs;
  const int32_t* order;      // pair indices, longest first
  int32_t n_pairs;
  int32_t match, mismatch, open, extend, strategy;
  uint16_t* bt;
  int32_t* aux;
  int32_t* ops;
  char* text;
  int32_t* result;           // per pair: [0] alignment offset, [1] text bytes written, [2] max_i, [3] max_j
  int32_t* next;
};

__device__ __forceinline__ int32_t sw_readlane(int32_t v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ int32_t sw_max(int32_t a, int32_t b) { return a > b ? a : b; }
__device__ __forceinline__ int32_t sw_abs(int32_t a) { return a < 0 ? -a : a; }

// ---- phase 1: fill.  Writes back-track nibbles, last_row[1..ncol], last_col[1..nrow].
__device__ __forceinline__ void sw_fill(const SwArgs& a, const SwPair& p, int lane) {
  const int nrow = p.nrow, ncol = p.ncol;
  const int ncolp = (ncol + 1) & ~1;
  const uint8_t* ref = a.seq + p.ref_off;
  const uint8_t* alt = a.seq + p.alt_off;
  const bool indel = a.strategy == kSwIndel || a.strategy == kSwLeadingIndel;
  const int32_t open = a.open, extend = a.extend, match = a.match, mismatch = a.mismatch;
  int32_t* last_row = a.aux + p.aux_off;
  int32_t* last_col = last_row + (ncol + 1);
  int32_t* carry = last_col + (nrow + 1);
  const int cstride = ncol + 65;
  uint16_t* bt = a.bt + p.bt_off;
  const int n_stripes = (nrow + kSwStripeRows - 1) / kSwStripeRows;
  for (int st = 0; st < n_stripes; st++) {
    const int row0 = st * kSwStripeRows + lane * kSwRpl;  // rows row0+1 .. row0+4
    const int rows_here = nrow - st * kSwStripeRows;      // rows of this stripe (may exceed 256)
    const int lanes_used = rows_here >= kSwStripeRows ? kLanes : (rows_here + kSwRpl - 1) / kSwRpl;
    const int n_steps = ncol + lanes_used - 1;
    const int32_t* cin_h = carry + ((st + 1) & 1) * 2 * cstride;
    const int32_t* cin_f = cin_h + cstride;
    int32_t* cout_h = carry + (st & 1) * 2 * cstride;
    int32_t* cout_f = cout_h + cstride;
    const bool has_next = st + 1 < n_stripes;
    int32_t hl[kSwRpl], e[kSwRpl];
    uint32_t x[kSwRpl];
    bool valid[kSwRpl];
#pragma unroll
    for (int s = 0; s < kSwRpl; s++) {
      const int i = row0 + 1 + s;
      valid[s] = i <= nrow;
      x[s] = valid[s] ? ref[i - 1] : 0x100u;               // never equals a byte
      hl[s] = indel ? open + (i - 1) * extend : 0;          // H[i][0], PairWiseSW.h:194-203
      e[s] = kSwLow;                                        // :205
    }
    // H[row0][0]: the diagonal input of this lane's first row at column 1
    int32_t hd = (row0 == 0 || !indel) ? 0 : open + (row0 - 1) * extend;
    int32_t in_h = 0, in_f = kSwLow;     // row above at this step's column (lane > 0: by DPP)
    int32_t out_h = 0, out_f = kSwLow;
    uint32_t ent = 0x100u;               // alternate base of this lane's column, 0x100 = none
    uint32_t acc = 0;
    int32_t ci_h = 0, ci_f = kSwLow;
    const int blk = row0 / kSwRpl;
    const int last_s = nrow - 1 - row0;  // row slot that holds row nrow in this lane (if 0..3)
    for (int t = 0; t < n_steps; t++) {
      const uint32_t entry = t < ncol ? (uint32_t)alt[t] : 0x100u;
      ent = dpp_shr1_keep(entry, ent);
      const int j = t - lane + 1;
      const bool act = j >= 1 && j <= ncol && valid[0];  // lanes past the last row stay out of memory
      if (st > 0) {                      // lane 0's row above comes from the previous stripe
        if ((t & 63) == 0) {
          const int c = t + 1 + lane;
          ci_h = c <= ncol ? cin_h[c] : 0;
          ci_f = c <= ncol ? cin_f[c] : kSwLow;
        }
        const int32_t vh = sw_readlane(ci_h, t & 63), vf = sw_readlane(ci_f, t & 63);
        if (lane == 0) { in_h = vh; in_f = vf; }
      } else if (lane == 0) {
        in_h = indel ? open + (j - 1) * extend : 0;         // H[0][j]
        in_f = kSwLow;                                      // F[0][j], :204
      }
      int32_t top_h = in_h, top_f = in_f, diag = hd;
      uint32_t nib = 0;
      int32_t h_last = 0;
#pragma unroll
      for (int s = 0; s < kSwRpl; s++) {
        const int32_t open_h = hl[s] + open, ext_h = e[s] + extend;        // MAIN_CODE :29-33
        const int32_t e11 = sw_max(open_h, ext_h);
        uint32_t code = open_h > ext_h ? 0u : (uint32_t)kSwInsertExt;       // :34-35
        const int32_t ext_v = top_f + extend, open_v = top_h + open;       // :39-42
        const int32_t f11 = sw_max(ext_v, open_v);
        code |= open_v > ext_v ? 0u : (uint32_t)kSwDeleteExt;               // :43-44
        const int32_t m11 = diag + (x[s] == ent ? match : mismatch);       // :47-51
        int32_t h11 = sw_max(kSwCutoff, m11);                               // :52
        uint32_t dir = e11 > h11 ? (uint32_t)kSwInsert : (uint32_t)kSwMatch;  // :53,55
        h11 = sw_max(h11, e11);
        dir = f11 > h11 ? (uint32_t)kSwDelete : dir;                        // :56,58
        h11 = sw_max(h11, f11);
        nib |= (code | dir) << (4 * s);
        diag = hl[s];
        hl[s] = act ? h11 : hl[s];
        e[s] = act ? e11 : e[s];
        top_h = h11;
        top_f = f11;
        if (s == last_s) h_last = h11;
      }
      hd = act ? in_h : hd;
      out_h = top_h;
      out_f = top_f;
      if (act) {
        // two columns of 4 nibbles per dword; an odd last column goes out alone
        if (((j - 1) & 1) == 0) {
          acc = nib;
          if (j == ncol) *reinterpret_cast<uint32_t*>(bt + (int64_t)blk * ncolp + (j - 1)) = acc;
        } else {
          acc |= nib << 16;
          *reinterpret_cast<uint32_t*>(bt + (int64_t)blk * ncolp + (j - 2)) = acc;
        }
        if (last_s >= 0 && last_s < kSwRpl) last_row[j] = h_last;
        if (j == ncol) {
#pragma unroll
          for (int s = 0; s < kSwRpl; s++)
            if (valid[s]) last_col[row0 + 1 + s] = hl[s];
        }
        if (has_next && lane == kLanes - 1) { cout_h[j] = out_h; cout_f[j] = out_f; }
      }
      // the row above for the next step: lane L-1 has just finished the column lane L takes next
      const int32_t nh = (int32_t)dpp_shr1_zero((uint32_t)out_h), nf = (int32_t)dpp_shr1_zero((uint32_t)out_f);
      if (lane != 0) { in_h = nh; in_f = nf; }
    }
    __threadfence();  // carry rows, last_row/last_col and the back-track are read back by this wavefront
  }
}

// ---- phase 2: (max_i, max_j) of PairWiseSW.h:207-232.
__device__ __forceinline__ void sw_find_max(const SwArgs& a, const SwPair& p, int lane, int32_t* out_i, int32_t* out_j) {
  const int nrow = p.nrow, ncol = p.ncol;
  const int32_t* last_row = a.aux + p.aux_off;
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
  const int ncolp = (ncol + 1) & ~1;
  const uint32_t* bt32 = reinterpret_cast<const uint32_t*>(a.bt + p.bt_off);
  const int row_dwords = ncolp >> 1;
  int32_t* ops = a.ops + p.ops_off;  // run-length ops in walk order: op << 28 | length
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
  int state = 0;
  int budget = nrow + ncol + 2;  // every step consumes a row or a column: a hard bound, whatever the memory holds
  while (i > 0 && j > 0 && budget > 0) {
    // tile: 4 row blocks (16 rows) x 16 dwords (32 columns) ending at the walker's block / dword
    const int b0 = (i - 1) >> 2, c0 = (j - 1) >> 1;
    const int tb = b0 - (lane >> 4), tc = c0 - (lane & 15);
    uint32_t tile = 0;
    if (tb >= 0 && tc >= 0) tile = bt32[(int64_t)tb * row_dwords + tc];
    while (i > 0 && j > 0 && budget > 0) {
      const int b = (i - 1) >> 2, c = (j - 1) >> 1;
      if (b0 - b > 3 || c0 - c > 15) break;
      budget--;
      const uint32_t w = (uint32_t)sw_readlane((int32_t)tile, ((b0 - b) << 4) | (c0 - c));
      const int btr = (int)((w >> (16 * ((j - 1) & 1) + 4 * ((i - 1) & 3))) & 0xfu);
      if (state == kSwInsertExt) { j--; cur_len++; state = btr & kSwInsertExt; }
      else if (state == kSwDeleteExt) { i--; cur_len++; state = btr & kSwDeleteExt; }
      else {
        const int dir = btr & 3;
        if (dir == kSwMatch) { i--; j--; push(kSwMatch, 1); state = 0; }
        else if (dir == kSwInsert) { j--; push(kSwInsert, 1); state = btr & kSwInsertExt; }
        else { i--; push(kSwDelete, 1); state = btr & kSwDeleteExt; }
      }
    }
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
  __threadfence();
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

__global__ __launch_bounds__(64) void sw_align_kernel(SwArgs a) {
  const int lane = threadIdx.x;
  for (;;) {
    int k = 0;
    if (lane == 0) k = atomicAdd(a.next, 1);
    k = __builtin_amdgcn_readfirstlane(k);
    if (k >= a.n_pairs) break;
    const int pi = a.order[k];
    const SwPair p = a.pairs[pi];
    sw_fill(a, p, lane);
    int32_t max_i = 0, max_j = 0;
    sw_find_max(a, p, lane, &max_i, &max_j);
    sw_trace(a, p, pi, lane, max_i, max_j);
    __syncthreads();  // one wavefront per block: free, and it keeps the iterations of the persistent loop apart
  }
}

}  // namespace gklhip
