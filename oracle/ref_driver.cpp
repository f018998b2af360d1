// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A thin C-callable driver around the *reference's own* PairHMM kernels, which
// oracle/Makefile compiles IN PLACE from /root/reference (avx_impl.cc,
// avx512_impl.cc, pairhmm_common.cc; flags of PH/CMakeLists.txt:7-8 + root
// CMakeLists.txt:14-18).  Nothing from the reference is copied into this repo:
// this file only #includes the reference headers through -I and replays the
// JNI-free part of the batch loop so that tests / the oracle / the
// cpu_baseline leg of bench.py can call the real GKL arithmetic.
//
// What it replays (cited, not copied):
//   * function-pointer choice by CPUID        PH/IntelPairHmm.cc:99-113
//   * FTZ on                                  PH/IntelPairHmm.cc:93-96
//   * ConvertChar::init()                     PH/IntelPairHmm.cc:116
//   * per-pair precision policy + log10       PH/IntelPairHmm.cc:150-169
//   * r-major cross product of testcases      PH/JavaData.h:84-105
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// the resulting oracle/_ref/libgkl_ref_pairhmm.so.
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include <avx.h>             // reference: src/main/native/common/avx.h
#include "pairhmm_common.h"  // reference: testcase, ConvertChar, MIN_ACCEPTED
#include "avx_impl.h"        // reference: compute_fp_avxs / compute_fp_avxd
#include "avx512_impl.h"     // reference: compute_fp_avx512s / compute_fp_avx512d
#include "Context.h"         // reference: Context<float>, Context<double>

namespace {
Context<float>  g_ctxf;  // same static-init objects as PH/IntelPairHmm.cc:44-45
Context<double> g_ctxd;
bool g_inited = false;
float  (*g_f32)(testcase*) = nullptr;
double (*g_f64)(testcase*) = nullptr;
int g_engine = 0;  // 1 = AVX, 2 = AVX-512

void ensure_init(int engine) {
  if (!g_inited) { ConvertChar::init(); g_inited = true; }
  int e = engine;
  if (e == 0) e = is_avx512_supported() ? 2 : 1;
  if (e == 2) { g_f32 = compute_fp_avx512s; g_f64 = compute_fp_avx512d; }
  else        { g_f32 = compute_fp_avxs;    g_f64 = compute_fp_avxd; }
  g_engine = e;
}
}  // namespace

extern "C" {

// engine: 0 = CPUID choice (like the reference), 1 = force AVX objects
// (unfused), 2 = force AVX-512 objects (gcc-contracted FMA). Returns engine.
int ref_init(int engine) { ensure_init(engine); return g_engine; }

int ref_has_avx512(void) { return is_avx512_supported() ? 1 : 0; }
int ref_has_avx(void) { return is_avx_supported() ? 1 : 0; }

// Raw kernel outputs for one pair (likelihood * 2^120 / 2^1020).
void ref_pair_raw(const char* rs, const char* q, const char* i, const char* d,
                  const char* c, int rslen, const char* hap, int haplen,
                  float* raw32, double* raw64) {
  if (!g_f32) ensure_init(0);
  unsigned old = _mm_getcsr();
  _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON);
  testcase tc;
  tc.rslen = rslen; tc.haplen = haplen;
  tc.q = q; tc.i = i; tc.d = d; tc.c = c; tc.hap = hap; tc.rs = rs;
  if (raw32) *raw32 = g_f32(&tc);
  if (raw64) *raw64 = g_f64(&tc);
  _mm_setcsr(old);
}

// Whole batch through the reference's dispatch policy (flat layout of
// include/gkl_hip_pairhmm.h).  out[r*n_haps+h] = log10 likelihood.  Optional
// raw32/raw64/used64 (may be NULL) expose the intermediate values: raw64 is
// only written where the fp64 kernel ran.
void ref_batch(int n_reads, int n_haps, const int64_t* read_off,
               const int64_t* hap_off, const char* read_bases,
               const char* read_quals, const char* ins, const char* del,
               const char* gcp, const char* hap_bases, int use_double,
               int n_threads, double* out, float* raw32, double* raw64,
               uint8_t* used64) {
  if (!g_f32) ensure_init(0);
  const long n = (long)n_reads * n_haps;
  if (n_threads < 1) n_threads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
#endif
  for (long p = 0; p < n; p++) {
    unsigned old = _mm_getcsr();
    _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON);
    const int r = (int)(p / n_haps), h = (int)(p % n_haps);
    testcase tc;
    tc.rslen = (int)(read_off[r + 1] - read_off[r]);
    tc.haplen = (int)(hap_off[h + 1] - hap_off[h]);
    tc.rs = read_bases + read_off[r];
    tc.q = read_quals + read_off[r];
    tc.i = ins + read_off[r];
    tc.d = del + read_off[r];
    tc.c = gcp + read_off[r];
    tc.hap = hap_bases + hap_off[h];
    double result_final;
    float rf = use_double ? 0.0f : g_f32(&tc);
    if (raw32) raw32[p] = rf;
    if (rf < MIN_ACCEPTED) {
      double rd = g_f64(&tc);
      if (raw64) raw64[p] = rd;
      if (used64) used64[p] = 1;
      result_final = log10(rd) - g_ctxd.LOG10_INITIAL_CONSTANT;
    } else {
      if (used64) used64[p] = 0;
      result_final = (double)(log10f(rf) - g_ctxf.LOG10_INITIAL_CONSTANT);
    }
    out[p] = result_final;
    _mm_setcsr(old);
  }
}

// Table dumps, to pin the oracle's restated tables bit-for-bit.
// which: 0 ph2pr[128], 1 matchToMatchProb, 2 jacobianLogTable,
//        3 {INITIAL_CONSTANT, LOG10_INITIAL_CONSTANT}
long ref_table_f32(int which, float* dst, long cap) {
  const float* src = nullptr; long n = 0; float tmp[2];
  switch (which) {
    case 0: src = g_ctxf.ph2pr; n = 128; break;
    case 1: src = g_ctxf.matchToMatchProb; n = ((MAX_QUAL + 1) * (MAX_QUAL + 2)) >> 1; break;
    case 2: src = g_ctxf.jacobianLogTable; n = JACOBIAN_LOG_TABLE_SIZE; break;
    case 3: tmp[0] = g_ctxf.INITIAL_CONSTANT; tmp[1] = g_ctxf.LOG10_INITIAL_CONSTANT; src = tmp; n = 2; break;
    default: return -1;
  }
  if (dst) memcpy(dst, src, sizeof(float) * (size_t)(n < cap ? n : cap));
  return n;
}
long ref_table_f64(int which, double* dst, long cap) {
  const double* src = nullptr; long n = 0; double tmp[2];
  switch (which) {
    case 0: src = g_ctxd.ph2pr; n = 128; break;
    case 1: src = g_ctxd.matchToMatchProb; n = ((MAX_QUAL + 1) * (MAX_QUAL + 2)) >> 1; break;
    case 2: src = g_ctxd.jacobianLogTable; n = JACOBIAN_LOG_TABLE_SIZE; break;
    case 3: tmp[0] = g_ctxd.INITIAL_CONSTANT; tmp[1] = g_ctxd.LOG10_INITIAL_CONSTANT; src = tmp; n = 2; break;
    default: return -1;
  }
  if (dst) memcpy(dst, src, sizeof(double) * (size_t)(n < cap ? n : cap));
  return n;
}

int ref_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"
