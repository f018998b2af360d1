// oracle/ref_pdhmm_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// C-callable driver around the REFERENCE's own PDHMM kernels, compiled in place from
// /root/reference/src/main/native/pdhmm by oracle/Makefile (MathUtils.cc, pdhmm-serial.cc,
// avx2_impl.cc, avx512_impl.cc; flags of pdhmm/CMakeLists.txt:9-12).  It replays the JNI-free
// part of IntelPDHMM.cc (initializeNative + allocateDPTable + computePDHMM,
// pdhmm-implementation.h:287-396) for a chosen engine.  Nothing is copied from the reference.
#include <cstdint>
#include <cstring>

#include "pdhmm-common.h"     // reference
#include "pdhmm-serial.h"     // reference: computePDHMM_serial
#include "avx2_impl.h"        // reference: computePDHMM_fp_avx2, simd_width_avx2
#include "avx512_impl.h"      // reference: computePDHMM_fp_avx512, simd_width_avx512
#include <avx.h>              // reference: is_avx512_supported

extern "C" {

int ref_pdhmm_has_avx512(void) { return is_avx512_supported() ? 1 : 0; }

// engine: 0 = scalar (pdhmm-serial.cc), 1 = AVX2, 2 = AVX-512.  Padded batch layout exactly as
// IntelPDHMM.computePDHMM passes it (IntelPDHMM.java:147-186).  Returns the PDHMM_* status.
int ref_pdhmm_compute(int engine, const int8_t* hap_bases, const int8_t* hap_pdbases, const int8_t* read_bases,
                      const int8_t* read_qual, const int8_t* read_ins_qual, const int8_t* read_del_qual,
                      const int8_t* gcp, double* result, int64_t batch, const int64_t* hap_lengths,
                      const int64_t* read_lengths, int32_t max_read_len, int32_t max_hap_len, int threads) {
  ProbabilityCache& pc = ProbabilityCache::getInstance();
  int32_t st = pc.initialize();
  if (st != PDHMM_SUCCESS) return st;
  const int simd = engine == 2 ? simd_width_avx512 : engine == 1 ? simd_width_avx2 : 1;
  if (threads < 1) threads = 1;
  // pdhmm-implementation.h:298-300
  const size_t dp = (size_t)(max_hap_len + 1) * (size_t)(max_read_len + 1) * simd * threads * sizeof(double);
  const size_t tr = TRANS_PROB_ARRAY_LENGTH * (size_t)(max_read_len + 1) * simd * threads * sizeof(double);
  st = DPTable::getInstance().allocate(dp, tr, dp);
  if (st != PDHMM_SUCCESS) return st;
  switch (engine) {
    case 2: return computePDHMM_fp_avx512(hap_bases, hap_pdbases, read_bases, read_qual, read_ins_qual, read_del_qual,
                                          gcp, result, batch, hap_lengths, read_lengths, max_read_len, max_hap_len, threads);
    case 1: return computePDHMM_fp_avx2(hap_bases, hap_pdbases, read_bases, read_qual, read_ins_qual, read_del_qual,
                                        gcp, result, batch, hap_lengths, read_lengths, max_read_len, max_hap_len, threads);
    default: return computePDHMM_serial(hap_bases, hap_pdbases, read_bases, read_qual, read_ins_qual, read_del_qual,
                                        gcp, result, batch, hap_lengths, read_lengths, max_read_len, max_hap_len, threads);
  }
}

int ref_pdhmm_simd_width(int engine) { return engine == 2 ? simd_width_avx512 : engine == 1 ? simd_width_avx2 : 1; }

// tables, to pin the restated ones: 0 qualToErrorProbCache[255], 1 matchToMatchProb[32640]
long ref_pdhmm_table(int which, double* dst, long cap) {
  ProbabilityCache& pc = ProbabilityCache::getInstance();
  if (pc.initialize() != PDHMM_SUCCESS) return -1;
  const double* src = which == 0 ? pc.getQualToErrorProbCache() : pc.getMatchToMatchProb();
  const long n = which == 0 ? MAX_QUAL + 1 : (((MAX_QUAL + 1) * (MAX_QUAL + 2)) >> 1);
  if (dst) memcpy(dst, src, sizeof(double) * (size_t)(n < cap ? n : cap));
  return n;
}

}  // extern "C"
