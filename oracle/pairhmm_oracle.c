/*
 * oracle/pairhmm_oracle.c -- TEST INFRASTRUCTURE (the checker), NOT PRODUCT CODE.
 *
 * A plain scalar C restatement of the reference's PairHMM forward path, used
 * only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg to
 * check the HIP kernels.  The product library (gkl_amd/csrc) never links,
 * loads or calls anything in this directory.
 *
 * Parity pinning (tests/test_oracle.py): this file is bit-identical, on the raw
 * kernel sums and on the final log10 values, to the reference's own objects
 * built by oracle/Makefile (`make ref`) on the 104 golden cases of
 * src/test/resources/pairhmm-testdata.txt plus seeded synthetic cases, in both
 * precisions and both FMA patterns, and within 1e-5 of the expected values the
 * reference's tests store (T/pairhmm/PairHmmUnitTest.java:88,221).
 *
 * Citations are path:line under /root/reference/src/main/native/pairhmm (PH/).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#if defined(__x86_64__)
#include <xmmintrin.h>
#endif
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- constants: PH/Context.h:30-34, PH/pairhmm_common.h:39 ---- */
#define O_MAX_QUAL 254
#define O_JAC_TOL 8.0
#define O_JAC_STEP 0.0001
#define O_JAC_INV_STEP (1.0 / O_JAC_STEP)
#define O_JAC_SIZE 80001 /* (int)(8.0 / 0.0001) + 1 */
#define O_MM_SIZE (((O_MAX_QUAL + 1) * (O_MAX_QUAL + 2)) >> 1)
#define O_MIN_ACCEPTED 1e-28f

static float  f_ph2pr[128], f_jac[O_JAC_SIZE], f_mm[O_MM_SIZE];
static double d_ph2pr[128], d_jac[O_JAC_SIZE], d_mm[O_MM_SIZE];
static float  f_init_const, f_log10_init;
static double d_init_const, d_log10_init;
static int tables_ready = 0;
static uint8_t conv[256];

/* PH/Context.h:91-94 (fastRound), evaluated in T */
static int fround_f(float d) { return (d > 0.0f) ? (int)(d + 0.5f) : (int)(d - 0.5f); }
static int fround_d(double d) { return (d > 0.0) ? (int)(d + 0.5) : (int)(d - 0.5); }

/* PH/Context.h:96-122 (approximateLog10SumLog10), evaluated in T */
static float approx_f(float small, float big) {
  if (small > big) { float t = big; big = small; small = t; }
  if (isinf(small) || isinf(big)) return big;
  float diff = big - small;
  if (diff >= (float)O_JAC_TOL) return big;
  int ind = fround_f((float)(diff * ((float)O_JAC_INV_STEP)));
  return big + f_jac[ind];
}
static double approx_d(double small, double big) {
  if (small > big) { double t = big; big = small; small = t; }
  if (isinf(small) || isinf(big)) return big;
  double diff = big - small;
  if (diff >= O_JAC_TOL) return big;
  int ind = fround_d(diff * O_JAC_INV_STEP);
  return big + d_jac[ind];
}

/* Tables: PH/Context.h:65-89 (jacobian, matchToMatch), :133-148 / :174-189 */
void oracle_tables_init(void) {
  if (tables_ready) return;
  const double INV_LN10 = 0.434294; /* truncated on purpose: PH/Context.h:78 */
  for (int k = 0; k < O_JAC_SIZE; k++) {
    double v = log10(1.0 + pow(10.0, -((double)k) * O_JAC_STEP));
    f_jac[k] = (float)v;
    d_jac[k] = v;
  }
  for (int i = 0, offset = 0; i <= O_MAX_QUAL; offset += ++i) {
    for (int j = 0; j <= i; j++) {
      /* "(NUMBER)-0.1*(NUMBER)i": the double literal -0.1 is cast to T first */
      double ls_f = approx_f((float)-0.1 * (float)i, (float)-0.1 * (float)j);
      double ls_d = approx_d((double)-0.1 * (double)i, (double)-0.1 * (double)j);
      double m_f = log1p(-fmin(1.0, pow(10, ls_f))) * INV_LN10;
      double m_d = log1p(-fmin(1.0, pow(10, ls_d))) * INV_LN10;
      f_mm[offset + j] = (float)pow(10, m_f);
      d_mm[offset + j] = pow(10, m_d);
    }
  }
  for (int x = 0; x < 128; x++) {
    d_ph2pr[x] = pow(10.0, -((double)x) / 10.0);   /* :139 */
    f_ph2pr[x] = powf(10.f, -((float)x) / 10.f);   /* :180 */
  }
  d_init_const = ldexp(1.0, 1020);  d_log10_init = log10(d_init_const);   /* :142-143 */
  f_init_const = ldexpf(1.f, 120);  f_log10_init = log10f(f_init_const);  /* :183-184 */
  /* PH/pairhmm_common.h:53-62: zero-initialised table, five entries set */
  memset(conv, 0, sizeof conv);
  conv['A'] = 0; conv['C'] = 1; conv['T'] = 2; conv['G'] = 3; conv['N'] = 4;
  tables_ready = 1;
}

/* set_mm_prob: PH/Context.h:156-167,197-209 (quals are &127 so the
 * MAX_QUAL<maxQual branch is dead) */
static int mm_index(int ins, int del) {
  int mn = del, mx = ins;
  if (ins <= del) { mn = ins; mx = del; }
  return ((mx * (mx + 1)) >> 1) + mn;
}

long oracle_table_f32(int which, float* dst, long cap) {
  oracle_tables_init();
  const float* src; long n; float tmp[2];
  switch (which) {
    case 0: src = f_ph2pr; n = 128; break;
    case 1: src = f_mm; n = O_MM_SIZE; break;
    case 2: src = f_jac; n = O_JAC_SIZE; break;
    case 3: tmp[0] = f_init_const; tmp[1] = f_log10_init; src = tmp; n = 2; break;
    default: return -1;
  }
  if (dst) memcpy(dst, src, sizeof(float) * (size_t)(n < cap ? n : cap));
  return n;
}
long oracle_table_f64(int which, double* dst, long cap) {
  oracle_tables_init();
  const double* src; long n; double tmp[2];
  switch (which) {
    case 0: src = d_ph2pr; n = 128; break;
    case 1: src = d_mm; n = O_MM_SIZE; break;
    case 2: src = d_jac; n = O_JAC_SIZE; break;
    case 3: tmp[0] = d_init_const; tmp[1] = d_log10_init; src = tmp; n = 2; break;
    default: return -1;
  }
  if (dst) memcpy(dst, src, sizeof(double) * (size_t)(n < cap ? n : cap));
  return n;
}

static unsigned ftz_on(void) {
#if defined(__x86_64__)
  unsigned old = _mm_getcsr();
  _mm_setcsr(old | 0x8000u); /* FTZ only, like PH/IntelPairHmm.cc:96 */
  return old;
#else
  return 0;
#endif
}
static void ftz_restore(unsigned old) {
#if defined(__x86_64__)
  _mm_setcsr(old);
#else
  (void)old;
#endif
}

/*
 * The forward recurrence, PH/avx-pairhmm-template.h:235-372 stated cell by
 * cell:
 *   rows i=1..R (read), cols j=1..H (hap); quals &127 (:134-136,149)
 *   M[0][*]=X[0][*]=0, Y[0][j]=INIT/H for j=0..H (:110-116,192); col 0 of rows>=1 = 0
 *   prior = (code(r_i)==code(h_j) || either is N) ? 1-eps : eps/3   (:181-183, masks :34,44-49)
 *   M = ((Md*pMM + Xd*pGAPM) + Yd*pGAPM) * prior                    (:213)
 *   X = Mu*pMX + Xu*pXX                                             (:219)
 *   Y = Ml*pMY + Yl*pYY                                             (:222)
 *   result = (sum_j M[R][j]) + (sum_j X[R][j]), ascending j        (:354-369)
 * fma_mode 0: separate mul/add (what the -mavx objects do);
 * fma_mode 1: the contraction gcc-11 applies to the AVX-512 TU:
 *   t=fma(Xd,pGAPM,Md*pMM); M=fma(Yd,pGAPM,t)*prior; X=fma(Xu,pXX,Mu*pMX); Y=fma(Yl,pYY,Ml*pMY)
 */
#define DEFINE_FWD(NAME, T, PH2PR, MMTAB, INITC, FMA)                                   \
  T NAME(const uint8_t* rs, const uint8_t* q, const uint8_t* ins, const uint8_t* del,   \
         const uint8_t* gcp, int R, const uint8_t* hap, int H, int fma_mode) {          \
    oracle_tables_init();                                                               \
    if (R <= 0 || H <= 0) return (T)NAN;                                                \
    unsigned csr = ftz_on();                                                            \
    T* buf = (T*)malloc(sizeof(T) * 6 * (size_t)(H + 1));                               \
    T *Mp = buf, *Xp = Mp + (H + 1), *Yp = Xp + (H + 1);                                \
    T *Mc = Yp + (H + 1), *Xc = Mc + (H + 1), *Yc = Xc + (H + 1);                       \
    const T init_Y = INITC / (T)H;                                                      \
    for (int j = 0; j <= H; j++) { Mp[j] = 0; Xp[j] = 0; Yp[j] = init_Y; }              \
    for (int i = 1; i <= R; i++) {                                                      \
      const int _i = ins[i - 1] & 127, _d = del[i - 1] & 127, _c = gcp[i - 1] & 127;    \
      const int _q = q[i - 1] & 127;                                                    \
      const T pMM = MMTAB[mm_index(_i, _d)];                                            \
      const T pGAPM = (T)1.0 - PH2PR[_c];                                               \
      const T pMX = PH2PR[_i], pXX = PH2PR[_c], pMY = PH2PR[_d], pYY = PH2PR[_c];       \
      const T eps = PH2PR[_q];                                                          \
      const T match = (T)1.0 - eps, mism = eps / (T)3.0;                                \
      const uint8_t rc = conv[rs[i - 1]];                                               \
      Mc[0] = 0; Xc[0] = 0; Yc[0] = 0;                                                  \
      for (int j = 1; j <= H; j++) {                                                    \
        const uint8_t hc = conv[hap[j - 1]];                                            \
        const T prior = (rc == hc || rc == 4 || hc == 4) ? match : mism;                \
        T m, x, y;                                                                      \
        if (fma_mode) {                                                                 \
          T t = FMA(Xp[j - 1], pGAPM, Mp[j - 1] * pMM);                                 \
          m = FMA(Yp[j - 1], pGAPM, t) * prior;                                         \
          x = FMA(Xp[j], pXX, Mp[j] * pMX);                                             \
          y = FMA(Yc[j - 1], pYY, Mc[j - 1] * pMY);                                     \
        } else {                                                                        \
          m = ((Mp[j - 1] * pMM + Xp[j - 1] * pGAPM) + Yp[j - 1] * pGAPM) * prior;      \
          x = Mp[j] * pMX + Xp[j] * pXX;                                                \
          y = Mc[j - 1] * pMY + Yc[j - 1] * pYY;                                        \
        }                                                                               \
        Mc[j] = m; Xc[j] = x; Yc[j] = y;                                                \
      }                                                                                 \
      T* t;                                                                             \
      t = Mp; Mp = Mc; Mc = t; t = Xp; Xp = Xc; Xc = t; t = Yp; Yp = Yc; Yc = t;        \
    }                                                                                   \
    T sumM = 0, sumX = 0;                                                               \
    for (int j = 1; j <= H; j++) { sumM = sumM + Mp[j]; sumX = sumX + Xp[j]; }          \
    T res = sumM + sumX;                                                                \
    free(buf);                                                                          \
    ftz_restore(csr);                                                                   \
    return res;                                                                         \
  }

DEFINE_FWD(oracle_fwd_f32, float, f_ph2pr, f_mm, f_init_const, fmaf)
DEFINE_FWD(oracle_fwd_f64, double, d_ph2pr, d_mm, d_init_const, fma)

/* The final step of PH/IntelPairHmm.cc:159-165, exposed so tests can finalize
 * raw GPU sums exactly like the reference does. */
double oracle_finalize_f32(float raw) {
  oracle_tables_init();
  return (double)(log10f(raw) - f_log10_init);
}
double oracle_finalize_f64(double raw) {
  oracle_tables_init();
  return log10(raw) - d_log10_init;
}

/*
 * Batch loop + precision policy: PH/IntelPairHmm.cc:150-169 over the r-major
 * cross product of PH/JavaData.h:84-105.  Flat layout = include/gkl_hip_pairhmm.h.
 */
void oracle_batch(int n_reads, int n_haps, const int64_t* read_off, const int64_t* hap_off,
                  const uint8_t* read_bases, const uint8_t* read_quals, const uint8_t* ins,
                  const uint8_t* del, const uint8_t* gcp, const uint8_t* hap_bases,
                  int use_double, int fma_mode, int n_threads, double* out, float* raw32,
                  double* raw64, uint8_t* used64) {
  oracle_tables_init();
  const long n = (long)n_reads * n_haps;
  if (n_threads < 1) n_threads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
#endif
  for (long p = 0; p < n; p++) {
    const int r = (int)(p / n_haps), h = (int)(p % n_haps);
    const int R = (int)(read_off[r + 1] - read_off[r]);
    const int H = (int)(hap_off[h + 1] - hap_off[h]);
    const int64_t ro = read_off[r], ho = hap_off[h];
    float rf = use_double ? 0.0f
                          : oracle_fwd_f32(read_bases + ro, read_quals + ro, ins + ro, del + ro,
                                           gcp + ro, R, hap_bases + ho, H, fma_mode);
    if (raw32) raw32[p] = rf;
    double res;
    if (rf < O_MIN_ACCEPTED) { /* NaN compares false -> stays fp32 */
      double rd = oracle_fwd_f64(read_bases + ro, read_quals + ro, ins + ro, del + ro, gcp + ro,
                                 R, hap_bases + ho, H, fma_mode);
      if (raw64) raw64[p] = rd;
      if (used64) used64[p] = 1;
      res = log10(rd) - d_log10_init;
    } else {
      if (used64) used64[p] = 0;
      res = (double)(log10f(rf) - f_log10_init);
    }
    out[p] = res;
  }
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
