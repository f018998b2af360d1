/*
 * oracle/sw_oracle.c -- TEST INFRASTRUCTURE (the checker), NOT PRODUCT CODE.
 *
 * Plain scalar C restatement of the reference's pairwise Smith-Waterman with affine gaps, back-track and
 * CIGAR text.  Citations: /root/reference/src/main/native/smithwaterman (SW/).
 * The reference sweeps anti-diagonals with AVX2 / AVX-512 vectors (SW/PairWiseSW.h:65-263); the cell
 * update is a pure function of the left, top and diagonal neighbours, so this file sweeps rows and keeps
 * only what the reference's results depend on:
 *   - the cell rule of MAIN_CODE (SW/PairWiseSW.h:27-62) incl. its strict-">" tie breaks,
 *   - the boundary values written after every anti-diagonal (:194-205),
 *   - the order in which the last-row / last-column candidates update the maximum (:207-232),
 *   - getCIGAR (:265-452) statement by statement, incl. fast_itoa's "0 has no digits" (SW/smithwaterman_common.cc:26-58).
 * Pinned bit-exact (CIGAR bytes, count, offset) against oracle/_ref/libgkl_ref_sw.so, i.e. the reference's own
 * AVX2 and AVX-512 objects, on random and adversarial inputs for all four overhang strategies
 * (tests/test_sw.py); the reference's own tests hold only "1M" and "1M1I" (SmithWatermanUnitTest.java:171-205).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { SW_MATCH = 0, SW_INSERT = 1, SW_DELETE = 2, SW_INSERT_EXT = 4, SW_DELETE_EXT = 8 };   /* SW/smithwaterman_common.h:44-48 */
enum { SW_SOFTCLIP = 9, SW_INDEL = 10, SW_LEADING_INDEL = 11, SW_IGNORE = 12 };            /* :49-52 */
#define SW_MIN_CUTOFF (-100000000)          /* MATRIX_MIN_CUTOFF :84 */
#define SW_LOW_INIT (INT32_MIN / 2)         /* LOW_INIT_VALUE :85 */

static int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }
static int32_t iabs(int32_t a) { return a < 0 ? -a : a; }

/* SW/smithwaterman_common.cc:26-58 */
static int32_t fast_itoa(char* ptr, int32_t number) {
  int neg = 0;
  if (number < 0) { number = -number; neg = 1; }
  int32_t cp = number, digits = 0;
  while (cp > 0) { cp /= 10; digits++; }
  if (!ptr) return digits + neg;
  if (neg) *(ptr++) = '-';
  for (int i = digits - 1; i >= 0; i--) { ptr[i] = (char)('0' + number % 10); number /= 10; }
  return digits + neg;
}

/* Fills bt[(i-1)*ncol + (j-1)] for 1<=i<=nrow, 1<=j<=ncol and returns max_i/max_j. 0 ok, 1 allocation failure. */
static int sw_fill(int32_t match, int32_t mismatch, int32_t open, int32_t extend, const uint8_t* s1, int32_t nrow,
                   const uint8_t* s2, int32_t ncol, int strategy, uint8_t* bt, int32_t* out_max_i, int32_t* out_max_j) {
  int32_t* Hprev = (int32_t*)malloc(sizeof(int32_t) * ((size_t)ncol + 1));
  int32_t* Hcur = (int32_t*)malloc(sizeof(int32_t) * ((size_t)ncol + 1));
  int32_t* F = (int32_t*)malloc(sizeof(int32_t) * ((size_t)ncol + 1));
  int32_t* last_row = (int32_t*)malloc(sizeof(int32_t) * ((size_t)ncol + 1));
  int32_t* last_col = (int32_t*)malloc(sizeof(int32_t) * ((size_t)nrow + 1));
  if (!Hprev || !Hcur || !F || !last_row || !last_col) { free(Hprev); free(Hcur); free(F); free(last_row); free(last_col); return 1; }
  const int indel = strategy == SW_INDEL || strategy == SW_LEADING_INDEL;
  /* row 0 and column 0: 0, or the cost of a leading gap (SW/PairWiseSW.h:194-203); H[0][0] = 0 (:106) */
  Hprev[0] = 0;
  for (int32_t j = 1; j <= ncol; j++) Hprev[j] = indel ? open + (j - 1) * extend : 0;
  for (int32_t j = 0; j <= ncol; j++) F[j] = SW_LOW_INIT;                     /* :96-99, :204 */
  for (int32_t i = 1; i <= nrow; i++) {
    Hcur[0] = indel ? open + (i - 1) * extend : 0;
    int32_t E = SW_LOW_INIT;                                                   /* :100-103, :205 */
    for (int32_t j = 1; j <= ncol; j++) {
      const int32_t ext_h = E + extend, open_h = Hcur[j - 1] + open;          /* :29-33 */
      const int32_t e11 = imax(open_h, ext_h);
      int ext = (open_h > ext_h) ? 0 : SW_INSERT_EXT;                          /* :34-35 */
      E = e11;
      const int32_t ext_v = F[j] + extend, open_v = Hprev[j] + open;          /* :39-42 */
      const int32_t f11 = imax(ext_v, open_v);
      if (!(open_v > ext_v)) ext |= SW_DELETE_EXT;                             /* :43-44 */
      F[j] = f11;
      const int32_t m11 = Hprev[j - 1] + (s1[i - 1] == s2[j - 1] ? match : mismatch);  /* :47-51 */
      int32_t h11 = imax(SW_MIN_CUTOFF, m11);                                  /* :52 */
      int b = SW_MATCH;
      if (e11 > h11) b = SW_INSERT;                                            /* :53,55 */
      h11 = imax(h11, e11);
      if (f11 > h11) b = SW_DELETE;                                            /* :56,58 */
      h11 = imax(h11, f11);
      bt[(size_t)(i - 1) * ncol + (j - 1)] = (uint8_t)(b | ext);              /* :59 */
      Hcur[j] = h11;
    }
    last_col[i] = Hcur[ncol];
    if (i == nrow) memcpy(last_row, Hcur, sizeof(int32_t) * ((size_t)ncol + 1));
    int32_t* t = Hprev; Hprev = Hcur; Hcur = t;
  }
  /* maximum over the last row (SOFTCLIP / IGNORE only) and the last column, in anti-diagonal order (:207-232) */
  int32_t max_score = INT32_MIN, max_i = 0, max_j = 0;
  for (int32_t d = 1; d <= nrow + ncol; d++) {
    if (d >= nrow + 1) {            /* ilo == nrow + 1: cell (nrow, d - nrow) */
      const int32_t j = d - nrow, score = last_row[j];
      if (strategy == SW_SOFTCLIP || strategy == SW_IGNORE) {
        if (max_score < score || (max_score == score && iabs(nrow - j) < iabs(max_i - max_j))) {
          max_score = score; max_i = nrow; max_j = j;
        }
      }
    }
    if (d >= ncol + 1) {            /* jhi == ncol + 1: cell (d - ncol, ncol) */
      const int32_t i = d - ncol, score = last_col[i];
      if (max_score < score || (max_score == score && (max_j == ncol || iabs(i - ncol) <= iabs(max_i - max_j)))) {
        max_score = score; max_i = i; max_j = ncol;
      }
    }
  }
  *out_max_i = max_i; *out_max_j = max_j;
  free(Hprev); free(Hcur); free(F); free(last_row); free(last_col);
  return 0;
}

/* SW/PairWiseSW.h:265-452.  ops[2k] = operation, ops[2k+1] = length, in back-track order. */
static void sw_cigar(const uint8_t* bt, int32_t nrow, int32_t ncol, int strategy, int32_t max_i, int32_t max_j,
                     int16_t* ops, char* cigar, int32_t cigar_len, uint32_t* count, int32_t* offset) {
  int16_t i, j;
  int32_t n = 0;
  if (strategy == SW_INDEL) { i = (int16_t)nrow; j = (int16_t)ncol; }
  else if (strategy == SW_LEADING_INDEL) { i = (int16_t)max_i; j = (int16_t)ncol; }
  else { i = (int16_t)max_i; j = (int16_t)max_j; }
  if (j < ncol) { ops[2 * n] = SW_SOFTCLIP; ops[2 * n + 1] = (int16_t)(ncol - j); n++; }
  int state = 0;
  while (i > 0 && j > 0) {
    const int btr = bt[(size_t)(i - 1) * ncol + (j - 1)];
    if (state == SW_INSERT_EXT) { j--; ops[2 * n - 1]++; state = btr & SW_INSERT_EXT; }
    else if (state == SW_DELETE_EXT) { i--; ops[2 * n - 1]++; state = btr & SW_DELETE_EXT; }
    else {
      switch (btr & 3) {
        case SW_MATCH: i--; j--; ops[2 * n] = SW_MATCH; ops[2 * n + 1] = 1; state = 0; n++; break;
        case SW_INSERT: j--; ops[2 * n] = SW_INSERT; ops[2 * n + 1] = 1; state = btr & SW_INSERT_EXT; n++; break;
        case SW_DELETE: i--; ops[2 * n] = SW_DELETE; ops[2 * n + 1] = 1; state = btr & SW_DELETE_EXT; n++; break;
      }
    }
  }
  int16_t off;
  if (strategy == SW_SOFTCLIP) {
    if (j > 0) { ops[2 * n] = SW_SOFTCLIP; ops[2 * n + 1] = j; n++; }
    off = i;
  } else if (strategy == SW_IGNORE) {
    if (j > 0) { ops[2 * n] = ops[2 * (n - 1)]; ops[2 * n + 1] = j; n++; }
    off = (int16_t)(i - j);
  } else {
    if (i > 0) { ops[2 * n] = SW_DELETE; ops[2 * n + 1] = i; n++; }
    else if (j > 0) { ops[2 * n] = SW_INSERT; ops[2 * n + 1] = j; n++; }
    off = 0;
  }
  int32_t last = 0;
  int16_t prev = ops[0];
  for (int32_t k = 1; k < n; k++) {
    const int16_t cur = ops[2 * k];
    if (cur == prev) ops[2 * last + 1] = (int16_t)(ops[2 * last + 1] + ops[2 * k + 1]);
    else { last++; ops[2 * last] = cur; ops[2 * last + 1] = ops[2 * k + 1]; prev = cur; }
  }
  int32_t cur_size = 0;
  for (int32_t k = last; k >= 0; k--) {
    char c;
    switch (ops[2 * k]) {
      case SW_MATCH: c = 'M'; break;
      case SW_INSERT: c = 'I'; break;
      case SW_DELETE: c = 'D'; break;
      case SW_SOFTCLIP: c = 'S'; break;
      default: c = 'R'; break;
    }
    const int32_t need = fast_itoa(NULL, ops[2 * k + 1]) + 1;
    if (cur_size >= 0 && need > 1 && cur_size + need <= cigar_len) {
      cur_size += fast_itoa(cigar + cur_size, ops[2 * k + 1]);
      cigar[cur_size++] = c;
    }
  }
  *count = (uint32_t)strnlen(cigar, (size_t)cur_size);
  *offset = off;
}

/* Same contract as runSWOnePairBT_<engine> (SW/PairWiseSW.h:454-501); cigar is zero-filled first like a fresh
 * Java byte[].  Returns 0, or 1 on allocation failure. */
int sw_oracle_align(int32_t match, int32_t mismatch, int32_t open, int32_t extend, const uint8_t* seq1, int32_t len1,
                    const uint8_t* seq2, int32_t len2, int32_t strategy, char* cigar, int32_t cigar_len,
                    uint32_t* cigar_count, int32_t* offset) {
  memset(cigar, 0, (size_t)cigar_len);
  *cigar_count = 0;
  *offset = 0;
  uint8_t* bt = (uint8_t*)malloc((size_t)len1 * (size_t)len2 + 1);
  int16_t* ops = (int16_t*)malloc(sizeof(int16_t) * 2 * ((size_t)len1 + (size_t)len2 + 4));
  if (!bt || !ops) { free(bt); free(ops); return 1; }
  int32_t max_i = 0, max_j = 0;
  int st = sw_fill(match, mismatch, open, extend, seq1, len1, seq2, len2, strategy, bt, &max_i, &max_j);
  if (st == 0) sw_cigar(bt, len1, len2, strategy, max_i, max_j, ops, cigar, cigar_len, cigar_count, offset);
  free(bt);
  free(ops);
  return st;
}
