/*
 * gkl_hip_pdhmm.h -- C ABI of the MI355X-native PDHMM (partially determined haplotype PairHMM),
 * the "next" row f1 of SURVEY.md section 8 / BASELINE.json config 5.
 *
 * Replaces, in the reference (paths under /root/reference/src/main/native/pdhmm):
 *   gklhip_pdhmm_init     initializeNative: pdhmm-implementation.h:303-322 (ProbabilityCache tables
 *                         of pdhmm-common.h:139-192, engine choice) -- here: tables + device context
 *   gklhip_pdhmm_compute  computePDHMM: pdhmm-implementation.h:361-396 -> computePDHMM_<engine>
 *                         (pdhmm.h:1133-1290), on the padded 1:1 batch IntelPDHMM.computePDHMM passes
 *                         (src/main/java/com/intel/gkl/pdhmm/IntelPDHMM.java:147-186)
 *   gklhip_pdhmm_done     doneNative
 * Arithmetic: GKL's, position by position -- its vector kernels (by default as its AVX-512 object computes, gcc-contracted
 * FMAs; optionally as its AVX2 object does) and its scalar engine for the tail of every batch; see gkl_amd/csrc/pdhmm_kernel.h.
 */
#ifndef GKL_HIP_PDHMM_H
#define GKL_HIP_PDHMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gklhip_pdhmm_ctx gklhip_pdhmm_ctx;

/* status codes are gklhip_status of gkl_hip_pairhmm.h (0 ok, 1 invalid argument, 2 no device,
 * 3 out of memory, 4 HIP error, 5 unsupported) */

typedef struct {
  int32_t batch;             /* number of (read, haplotype) pairs */
  int32_t max_hap_len;       /* row stride of the two haplotype arrays */
  int32_t max_read_len;      /* row stride of the five read arrays */
  const int8_t* hap_bases;   /* [batch][max_hap_len] */
  const int8_t* hap_pdbases; /* [batch][max_hap_len] PD flag bytes: SNP 1, DEL_START 2, DEL_END 4, A 8, C 16, G 32, T 64 */
  const int8_t* read_bases;  /* [batch][max_read_len] */
  const int8_t* read_qual;
  const int8_t* read_ins_qual;
  const int8_t* read_del_qual;
  const int8_t* gcp;
  const int64_t* hap_lengths;  /* [batch], 1..max_hap_len */
  const int64_t* read_lengths; /* [batch], 1..max_read_len */
} gklhip_pdhmm_batch;

/* The reads x haplotypes form of IntelPDHMM.computeLikelihoods (IntelPDHMM.java:83-145): every read against every
 * haplotype, out[r * n_haps + h] (JavaData.h:177-242 builds exactly this cross product as padded PAIRS before calling
 * computePDHMM; here each read and each haplotype crosses PCIe once). */
typedef struct {
  int32_t n_reads, n_haps;
  int32_t max_hap_len;       /* row stride of the two haplotype arrays */
  int32_t max_read_len;      /* row stride of the five read arrays */
  const int8_t* hap_bases;   /* [n_haps][max_hap_len] */
  const int8_t* hap_pdbases;
  const int8_t* read_bases;  /* [n_reads][max_read_len] */
  const int8_t* read_qual;
  const int8_t* read_ins_qual;
  const int8_t* read_del_qual;
  const int8_t* gcp;
  const int64_t* hap_lengths;  /* [n_haps] */
  const int64_t* read_lengths; /* [n_reads] */
} gklhip_pdhmm_cross;

int gklhip_pdhmm_init(int device /* -1 = current */, gklhip_pdhmm_ctx** out_ctx);
/* 1 (default) = bit-identical to GKL's AVX-512 PDHMM object (avx512_impl.cc: a*b + c*d contracted to
 * fma(c, d, a*b)); 0 = bit-identical to its AVX2 object (avx2_impl.cc: no FMA).  Same switch as
 * gklhip_config.fma_mode of the PairHMM. */
int gklhip_pdhmm_set_fma_mode(gklhip_pdhmm_ctx* ctx, int fma_mode);
/* GKL finishes the last `batch mod SIMD width` pairs of every vector batch with its scalar engine (pdhmm.h:1264-1270 ->
 * pdhmm-serial.cc:279-412), whose arithmetic differs from its vector kernels': a pair's value depends on its position
 * in the batch.  mode 1 ("reference", the DEFAULT; GKL_HIP_PDHMM_TAIL=reference): the pairs at positions
 * >= batch - batch mod W (W = 8 with fma_mode 1 = the AVX-512 engine, 4 with fma_mode 0 = AVX2) of a paired batch -- and
 * of every reference batch of a cross product, see gklhip_pdhmm_compute_cross_batched -- take the scalar engine's
 * arithmetic, so that EVERY position matches GKL bit for bit.  mode 0 ("vector"; GKL_HIP_PDHMM_TAIL=vector in the
 * environment at init): the vector arithmetic for every pair -- position-independent results. */
int gklhip_pdhmm_set_tail_mode(gklhip_pdhmm_ctx* ctx, int mode);
/* Host buffers in, out_host[batch] = log10 likelihoods. Negative ins/del/gcp quals ->
 * GKLHIP_ERR_INVALID_ARG (PDHMM_INPUT_DATA_ERROR in the reference). */
int gklhip_pdhmm_compute(gklhip_pdhmm_ctx* ctx, const gklhip_pdhmm_batch* batch, double* out_host);
/* out_host[n_reads * n_haps], read-major. Same arithmetic, errors and tables as gklhip_pdhmm_compute; in tail mode 1 the
 * whole cross product counts as ONE reference batch (= gklhip_pdhmm_compute_cross_batched with ref_batch_pairs 0). */
int gklhip_pdhmm_compute_cross(gklhip_pdhmm_ctx* ctx, const gklhip_pdhmm_cross* batch, double* out_host);
/* The reference's computeLikelihoodsNative cuts the read-major pair list into batches of
 * min(total, maxMemoryInMB / memoryPerPair) pairs (pdhmm/JavaData.h:83-101) and every batch ends in its own scalar tail.
 * ref_batch_pairs replays that cut in tail mode 1 (0: one batch); gklhip_pdhmm_reference_batch_pairs computes the
 * reference's number from the memory limit and the two maximum lengths.  The limit is what initNative kept:
 * gklhip_pdhmm_available_memory_mb(maxMemoryInMB) = min(maxMemoryInMB, the host's free RAM), taken once at
 * initialisation like the reference's getMaxMemoryAvailable (pdhmm-implementation.h:204-235), so that identical calls
 * are cut -- and therefore rounded -- identically.  What the JNI shim's computeLikelihoodsNative calls. */
int gklhip_pdhmm_compute_cross_batched(gklhip_pdhmm_ctx* ctx, const gklhip_pdhmm_cross* batch, int64_t ref_batch_pairs, double* out_host);
int32_t gklhip_pdhmm_available_memory_mb(int32_t max_memory_mb);
int64_t gklhip_pdhmm_reference_batch_pairs(int32_t max_memory_mb, int32_t max_read_len, int32_t max_hap_len, int64_t total_pairs);
/* HIP-event time of the forward kernel of the last call, milliseconds. */
float gklhip_pdhmm_last_kernel_ms(gklhip_pdhmm_ctx* ctx);
/* Diagnostics: bytes of device and pinned host memory the context holds right now.  Its buffers grow with the biggest call
 * and are given back when the last 16 calls each needed less than a quarter of a buffer above 32 MB. */
int64_t gklhip_pdhmm_buffer_bytes(gklhip_pdhmm_ctx* ctx);
/* Diagnostics: how the last cross call's haplotypes were routed: out[0] to the table kernel (at most six classes of
 * (base, SNP alleles, 'N') columns: the match priors come from an LDS table), out[1] to the predicate kernel, out[2] to
 * the byte-comparing kernel (a base outside ACGTN).  After a paired call (gklhip_pdhmm_compute; every pair its own haplotype
 * item, classes and routing found on the device): the call's packed jobs -- wavefront-loads of whole pairs -- by kernel in
 * the same order (a job is the table kernel's when all its haplotypes are eligible).  GKL_HIP_PDHMM_TABLE=0 disables the
 * table kernel.  Results are identical whichever kernel computes a pair. */
int gklhip_pdhmm_last_routing(gklhip_pdhmm_ctx* ctx, int32_t out[3]);
int gklhip_pdhmm_done(gklhip_pdhmm_ctx* ctx);
/* host-built tables as uploaded: 0 qualToErrorProb[255], 1 matchToMatchProb[32640] */
int64_t gklhip_pdhmm_get_table(int which, double* dst, int64_t cap);
const char* gklhip_pdhmm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* GKL_HIP_PDHMM_H */
