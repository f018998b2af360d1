/*
 * gkl_hip_sw.h -- C ABI of the MI355X-native pairwise Smith-Waterman (affine gaps, back-track, CIGAR),
 * row f4 of SURVEY.md section 8.
 *
 * Replaces, in the reference (paths under /root/reference/src/main/native/smithwaterman):
 *   gklhip_sw_init         initNative: IntelSmithWaterman.cc:47-66 (engine choice) -- here: device context
 *   gklhip_sw_align        what alignNative calls after pinning its arrays: g_runSWOnePairBT =
 *                          runSWOnePairBT_<engine> (IntelSmithWaterman.cc:107-110, PairWiseSW.h:454-501):
 *                          smithWatermanBackTrack (:65-263) + getCIGAR (:265-452) for ONE pair
 *   gklhip_sw_align_batch  the same for many independent pairs in one launch (no counterpart in the
 *                          reference, whose JNI surface aligns one pair per call; this is the entry point
 *                          that lets a GPU pay off -- see DESIGN.md)
 *   gklhip_sw_done         doneNative (IntelSmithWaterman.cc:130-132, empty there)
 * Results (CIGAR bytes, their count, the alignment offset) are bit-identical to the reference's AVX2 and
 * AVX-512 objects: integer arithmetic only.
 */
#ifndef GKL_HIP_SW_H
#define GKL_HIP_SW_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gklhip_sw_ctx gklhip_sw_ctx;

/* status codes are gklhip_status of gkl_hip_pairhmm.h (0 ok, 1 invalid argument, 2 no device,
 * 3 out of memory, 4 HIP error, 5 unsupported) */

/* SWParameters of gatk-native-bindings, in the order alignNative takes them (IntelSmithWaterman.h:44-46). */
typedef struct {
  int32_t match, mismatch, open, extend;
} gklhip_sw_params;

/* SWOverhangStrategy as the byte IntelSmithWaterman.getStrategy produces (IntelSmithWaterman.java:160-177;
 * smithwaterman_common.h:49-52). */
enum { GKLHIP_SW_SOFTCLIP = 9, GKLHIP_SW_INDEL = 10, GKLHIP_SW_LEADING_INDEL = 11, GKLHIP_SW_IGNORE = 12 };

/* limits checked by the Java wrapper before the native call (IntelSmithWaterman.java:52-55,133-138) */
#define GKLHIP_SW_MAX_SEQUENCE_LENGTH 32767
#define GKLHIP_SW_MAX_MATCH_VALUE 65536

int gklhip_sw_init(int device /* -1 = current */, gklhip_sw_ctx** out_ctx);
int gklhip_sw_done(gklhip_sw_ctx* ctx);

/* One pair.  `cigar` (cigar_len bytes, the Java side passes 2*max(ref_len, alt_len)) is zero-filled and then
 * receives the CIGAR text; *cigar_count = its length; *offset = the alignment offset alignNative returns. */
int gklhip_sw_align(gklhip_sw_ctx* ctx, const gklhip_sw_params* params, int32_t strategy, const uint8_t* ref,
                    int32_t ref_len, const uint8_t* alt, int32_t alt_len, char* cigar, int32_t cigar_len,
                    uint32_t* cigar_count, int32_t* offset);

/* n independent pairs: refs/alts are flat byte arrays addressed by ref_off/alt_off (n + 1 offsets each);
 * cigars is [n][cigar_stride] (each row treated like the `cigar` buffer above with cigar_len = cigar_stride);
 * counts[n], offsets[n]. */
int gklhip_sw_align_batch(gklhip_sw_ctx* ctx, const gklhip_sw_params* params, int32_t strategy, int32_t n,
                          const uint8_t* refs, const int64_t* ref_off, const uint8_t* alts, const int64_t* alt_off,
                          char* cigars, int32_t cigar_stride, uint32_t* counts, int32_t* offsets);

/* HIP-event time of the kernel of the last call, milliseconds. */
float gklhip_sw_last_kernel_ms(gklhip_sw_ctx* ctx);
const char* gklhip_sw_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* GKL_HIP_SW_H */
