// oracle/ref_sw_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// C driver around the REFERENCE's own Smith-Waterman objects (src/main/native/smithwaterman/
// avx2_impl.cc, avx512_impl.cc, smithwaterman_common.cc compiled where they lie by oracle/Makefile):
// what Java_com_intel_gkl_smithwaterman_IntelSmithWaterman_alignNative does after pinning its arrays
// (IntelSmithWaterman.cc:107-110), minus JNI.  Not thread-safe, like the reference (PairWiseSW.h:63
// keeps the matrix width in a mutable static).
#include <stdint.h>
#include <string.h>

#include "avx2_impl.h"
#include "avx512_impl.h"

static bool host_has_avx512() {
  return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") &&
         __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512bw");
}

extern "C" {

int ref_sw_has_avx512(void) { return host_has_avx512() ? 1 : 0; }

// engine 1 = AVX2, 2 = AVX-512.  cigar (cigar_len bytes, zero-filled here like a fresh Java byte[]) receives
// the CIGAR text; returns the reference's status (0 ok, 1 allocation failure), -2 engine unavailable.
int ref_sw_align(int engine, int32_t match, int32_t mismatch, int32_t open, int32_t extend, const uint8_t* seq1,
                 int32_t len1, const uint8_t* seq2, int32_t len2, int32_t strategy, char* cigar, int32_t cigar_len,
                 uint32_t* cigar_count, int32_t* offset) {
  memset(cigar, 0, (size_t)cigar_len);
  *cigar_count = 0;
  *offset = 0;
  if (engine == 2) {
    if (!host_has_avx512()) return -2;
    return runSWOnePairBT_fp_avx512(match, mismatch, open, extend, const_cast<uint8_t*>(seq1), const_cast<uint8_t*>(seq2),
                                    (int16_t)len1, (int16_t)len2, (int8_t)strategy, cigar, cigar_len, cigar_count, offset);
  }
  if (!__builtin_cpu_supports("avx2")) return -2;
  return runSWOnePairBT_fp_avx2(match, mismatch, open, extend, const_cast<uint8_t*>(seq1), const_cast<uint8_t*>(seq2),
                                (int16_t)len1, (int16_t)len2, (int8_t)strategy, cigar, cigar_len, cigar_count, offset);
}

}  // extern "C"
