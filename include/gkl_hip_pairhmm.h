/*
 * gkl_hip_pairhmm.h -- C ABI of the MI355X-native PairHMM forward hot path.
 *
 * This is the drop-in boundary.  Everything below is plain C: pointers, sizes and
 * status codes -- no JNI, no torch, no C++ types.  The JNI symbols that GKL's Java
 * class com.intel.gkl.pairhmm.IntelPairHmm binds (libgkl_pairhmm.so, see
 * include/gkl_pairhmm_jni.h) are thin shims over these entry points, and so is
 * the ctypes binding in gkl_amd/native.py.
 *
 * What each entry point replaces in the reference (paths under
 * /root/reference/src/main/native/pairhmm):
 *
 *   gklhip_init            initNative: IntelPairHmm.cc:55-118 (globals g_use_double,
 *                          g_max_threads, FTZ, kernel choice) + the static Context<T>
 *                          table objects of IntelPairHmm.cc:44-45 / Context.h:133-189
 *   gklhip_compute         computeLikelihoodsNative after marshalling:
 *                          IntelPairHmm.cc:150-169 (batch loop, fp32->fp64 policy,
 *                          log10) calling compute_full_prob_* of
 *                          avx-pairhmm-template.h:235-372
 *   gklhip_compute_device  same, with the batch already resident in HBM
 *   gklhip_done            doneNative: IntelPairHmm.cc:189-192
 *   gklhip_init_devices    same as gklhip_init for a LIST of devices: every call is then cut into contiguous
 *                          read ranges balanced by cells, one per device -- what the reference's OpenMP
 *                          `schedule(dynamic,1)` loop over pairs (IntelPairHmm.cc:151-154) is to its cores
 *
 * The flat batch replaces the std::vector<testcase> that JavaData::getData builds
 * (JavaData.h:65-111; testcase = pairhmm_common.h:43-47): reads x haplotypes cross
 * product, r-major, result index r*n_haps + h (JavaData.h:94-105, IntelPairHmm.cc:167).
 */
#ifndef GKL_HIP_PAIRHMM_H
#define GKL_HIP_PAIRHMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GKLHIP_ABI_VERSION 2

typedef struct gklhip_ctx gklhip_ctx; /* opaque; one per initNative */

typedef enum {
  GKLHIP_OK = 0,
  GKLHIP_ERR_INVALID_ARG = 1, /* -> java/lang/IllegalArgumentException */
  GKLHIP_ERR_NO_DEVICE = 2,   /* no usable gfx950 device: fails loudly, no CPU fallback */
  GKLHIP_ERR_OOM = 3,         /* -> java/lang/OutOfMemoryError */
  GKLHIP_ERR_HIP = 4,         /* any other HIP runtime failure -> java/lang/RuntimeException */
  GKLHIP_ERR_UNSUPPORTED = 5
} gklhip_status;

/* How raw kernel sums become log10 likelihoods (IntelPairHmm.cc:159-165). */
typedef enum {
  /* Exactly the reference's arithmetic, evaluated on the HOST with the host libm:
   * fp32 pairs (double)(log10f(raw) - log10f(2^120)), fp64 pairs
   * log10(raw) - log10(2^1020).  Bit-identical to GKL given bit-identical sums.
   * Default of gklhip_compute (host buffers). */
  GKLHIP_FINALIZE_REFERENCE_HOST = 0,
  /* Evaluated on the device in double: log10((double)raw) - log10(2^{120|1020}).
   * Never further from the reference's fp64 path than the fp32 recurrence itself
   * (the reference's log10f adds up to 3.8e-6 absolute). Default of
   * gklhip_compute_device. */
  GKLHIP_FINALIZE_DEVICE_F64 = 1,
  /* Device emulation of the reference formula: float-rounded log10 and float
   * subtraction; within 1 float ulp (3.8e-6) of GKLHIP_FINALIZE_REFERENCE_HOST. */
  GKLHIP_FINALIZE_DEVICE_REF32 = 2
} gklhip_finalize;

typedef struct {
  int32_t abi_version;  /* GKLHIP_ABI_VERSION */
  int32_t device;       /* HIP device ordinal, -1 = current device */
  int32_t use_double;   /* PairHMMNativeArguments.useDoublePrecision (IntelPairHmm.cc:70) */
  int32_t max_threads;  /* maxNumberOfThreads (IntelPairHmm.cc:72-89): a CAP on the host threads of the reference-exact
                           finalize, honoured as given -- 1 (the reference's default) = one thread per call;
                           <= 0 = not set: the calls in flight share min(cores, 8).  GKL_HIP_FINALIZE_THREADS overrides */
  int32_t fma_mode;     /* 1 = arithmetic of GKL's AVX-512 objects (gcc-contracted FMA; default),
                           0 = arithmetic of GKL's AVX objects (separate mul/add) */
  int32_t finalize;     /* gklhip_finalize for gklhip_compute_device; -1 = default */
  int32_t record_events;/* 1 = bracket kernels with HIP events and synchronise every call (gklhip_get_stats);
                           2 = record into a ring of 64 event sets WITHOUT synchronising: calls pipeline, the
                               times are read afterwards with gklhip_get_step_times */
  int32_t rows_per_lane;/* fp32 main kernel: 0 = auto (8 rows per lane; 4 or 2 for small batches), 8 = 8-row kernel,
                           4 (or -4) = 4-row kernel, 2 = 2-row kernel when every read has at most 127 bases, else 4 */
} gklhip_config;

/* Flat structure-of-arrays batch. Offsets always live on the host; the byte
 * arrays live on the host for gklhip_compute and in HBM for gklhip_compute_device.
 * Quals are raw Phred bytes; they are masked with &127 like
 * avx-pairhmm-template.h:134-136,149. */
typedef struct {
  int32_t n_reads;
  int32_t n_haps;
  const int64_t* read_off;   /* [n_reads+1], read_off[0] == 0 */
  const int64_t* hap_off;    /* [n_haps+1],  hap_off[0] == 0 */
  const uint8_t* read_bases; /* [read_off[n_reads]]  ReadDataHolder.readBases    */
  const uint8_t* read_quals; /*                      ReadDataHolder.readQuals    */
  const uint8_t* ins_gop;    /*                      ReadDataHolder.insertionGOP */
  const uint8_t* del_gop;    /*                      ReadDataHolder.deletionGOP  */
  const uint8_t* gcp;        /*                      ReadDataHolder.overallGCP   */
  const uint8_t* hap_bases;  /* [hap_off[n_haps]]    HaplotypeDataHolder.haplotypeBases */
} gklhip_batch;

typedef struct {
  int64_t n_pairs;
  int64_t n_fallback;      /* pairs recomputed in fp64 (raw fp32 sum < 1e-28f) */
  int64_t cells;           /* sum of rslen*haplen over all pairs */
  int64_t cells_fp64;      /* same over the fp64-recomputed pairs */
  int32_t n_chunks;        /* 64-lane read packs of the fp32 (or all-fp64) pass */
  int32_t n_hap_groups;
  int32_t rows_per_lane;
  int32_t n_long_pairs;    /* pairs of the main pass routed to the striped long-read kernel */
  float ms_fwd_main;       /* HIP-event time of the main forward kernel (record_events) */
  float ms_fwd_fallback;   /* HIP-event time of the fp64 fallback kernel */
  float ms_total_device;   /* first launch .. last kernel of the call */
  float lane_fill;         /* useful rows / (chunks * 64 * rows_per_lane) of the main pass */
} gklhip_stats;

/* Lifecycle.  gklhip_init uses cfg->device; when that is -1 and the environment names a list
 * (GKL_HIP_DEVICES=0,1,2,...), it is gklhip_init_devices with that list. */
int gklhip_init(const gklhip_config* cfg, gklhip_ctx** out_ctx);
int gklhip_done(gklhip_ctx* ctx);

/* One context over several devices of this node, driven by this one process (one host thread per extra device).
 * Every call is sharded by contiguous read ranges balanced by cells (gklhip_partition_reads), haplotypes and tables
 * replicated.  gklhip_compute: each device copies its range from the caller's host arrays and its results back over
 * its own PCIe link -- no device-to-device step.  gklhip_compute_device: inputs and the output array live on
 * devices[0]; the other devices pull their ranges over xGMI and their results are gathered into the output array --
 * with RCCL (ncclCommInitAll + one ncclSend/ncclRecv pair per device in one group; librccl.so is dlopen()ed by the
 * first multi-device context) when the devices are distinct, with peer copies otherwise (a device may be listed
 * twice: two shards on one GPU, what the one-GPU tests do).  GKL_HIP_GATHER=peer|rccl overrides the choice.
 * devices == NULL or n_devices <= 0: cfg->device alone. */
int gklhip_init_devices(const gklhip_config* cfg, const int32_t* devices, int32_t n_devices, gklhip_ctx** out_ctx);
int gklhip_num_devices(gklhip_ctx* ctx);
/* 0 = single device (no gather), 1 = peer copies, 2 = RCCL (before the first device-resident call: what that call
 * will try -- the communicators are created lazily), 3 = peer copies because RCCL failed (library missing, a device
 * listed twice, ncclCommInitAll or a group call returning an error; gklhip_gather_note says which).  An RCCL failure
 * never fails a call: the shards' results are gathered with peer copies instead, bit-identically.
 * GKL_HIP_RCCL_FAIL=init|group forces such a failure (tests). */
int gklhip_gather_backend(gklhip_ctx* ctx);
const char* gklhip_gather_note(gklhip_ctx* ctx);
/* The sharding rule: bounds_out[0] = 0 <= ... <= bounds_out[n_parts] = n_reads, contiguous read ranges whose summed
 * read lengths (= cells, every range meets every haplotype) are as equal as cut points between reads allow. */
int gklhip_partition_reads(int32_t n_reads, const int64_t* read_off, int32_t n_parts, int32_t* bounds_out);
/* Diagnostics: dlopen RCCL and run the gather's send/recv group on a one-device communicator.  0 = ok. */
int gklhip_rccl_selftest(int32_t device);

/* Host buffers in, host doubles out (n_reads*n_haps). What the JNI shim calls.  The byte arrays
 * are copied to the device asynchronously; when they live in memory from gklhip_host_alloc the
 * copies are true DMA (no bounce through the runtime's staging pages).  Results come back as one
 * 8-byte word per pair through pinned memory owned by the context. */
int gklhip_compute(gklhip_ctx* ctx, const gklhip_batch* host_batch, double* out_host);

/* Page-locked host memory for a binder's marshalling buffers (replaces the per-array
 * Get<T>ArrayElements pins of JavaData.h:135-154 with one flat, DMA-able staging area that is
 * reused across calls).  NULL on failure.  Not tied to a context. */
void* gklhip_host_alloc(size_t bytes);
void gklhip_host_free(void* p);

/* Byte arrays and `out_dev` in HBM; launches on `hip_stream` (a hipStream_t; NULL = HIP's
 * default stream) and returns without a host sync when record_events == 0 (the fp64
 * recomputation pass is planned on the device).  The context's scratch is ordered by that stream; a
 * call on a different stream than the previous one first waits (on the device) for that one's end. */
int gklhip_compute_device(gklhip_ctx* ctx, const gklhip_batch* dev_batch, double* out_dev,
                          void* hip_stream);

/* Introspection (tests, bench). */
int gklhip_get_stats(gklhip_ctx* ctx, gklhip_stats* out);

/* record_events == 2: HIP-event times (ms) of the call `steps_back` calls ago (0 = the last one; at most 63):
 * main forward kernel, fp64 fallback kernel, whole device pipeline.  Waits for that call to finish. */
int gklhip_get_step_times(gklhip_ctx* ctx, int32_t steps_back, float* ms_main, float* ms_fallback, float* ms_total);
/* Raw sums of the last call, copied to host arrays of n_pairs entries (any may be NULL).
 * raw64 is meaningful where used64 != 0. Synchronises the stream. */
int gklhip_get_raw(gklhip_ctx* ctx, float* raw32, double* raw64, uint8_t* used64);
/* Host-built lookup tables exactly as uploaded: which = 0 ph2pr[128],
 * 1 matchToMatch triangle for quals 0..127 [8256], 2 ph2pr/3 [128]. Returns count. */
int64_t gklhip_get_table_f32(int which, float* dst, int64_t cap);
int64_t gklhip_get_table_f64(int which, double* dst, int64_t cap);

/* Host-side work planning only (no device needed): how the reads of a batch would be packed into
 * 64-lane chunks at `rows_per_lane` rows per lane and how many haplotype streams are formed.
 * lanes_out (may be NULL) receives n_chunks*64 pairs {read index or -1, row block}; returns the number
 * of chunks, or a negative status. n_groups_out / n_long_out (may be NULL): stream groups, reads
 * routed to the striped long-read kernel. */
int gklhip_plan_describe(int32_t n_reads, int32_t n_haps, const int64_t* read_off, const int64_t* hap_off,
                         int32_t rows_per_lane, int32_t* lanes_out, int64_t lanes_cap, int32_t* n_groups_out,
                         int32_t* n_long_out);

/* Diagnostics: the VALU issue ceiling of the forward recurrence's instruction mix on the context's first device -- a
 * kernel of nothing but 4 multiplies + 4 fused multiply-adds per cell (fp32, or fp64 with use_double), operands placed
 * so that no op stalls on a register bank, four wavefronts per SIMD on every CU, run for about ms_budget milliseconds.
 * cells_per_s: cells of that mix per second (x 12 FLOP = the ceiling bench.py prints next to the vector peak);
 * clock_ghz (may be NULL): shader cycles counted by the kernel / its duration = the clock sustained under that load. */
int gklhip_measure_issue_ceiling(gklhip_ctx* ctx, int use_double, double ms_budget, double* cells_per_s, double* clock_ghz);

/* Diagnostics: concurrent small host-buffer calls (a GATK region each, from several threads or JNI slots) are launched
 * together when they meet on the device (GKL_HIP_COMBINE).  Process-wide counts for `device`:
 * out[0] calls that took the small-call path, out[1] those of them launched together with other calls, out[2] sets of
 * launches issued.  reset != 0 zeroes the counts after reading. */
int gklhip_small_call_counts(int device, int64_t out[3], int reset);

/* Gives back what an IDLE context holds only for speed: the streams it made for big calls (every stream is a hardware
 * queue the device's scheduler rotates among all processes' -- an idle process should hold one or two, not seven), its
 * twin engines of big host-buffer calls and second engines of two-stream device-resident callers.  Nothing happens while a
 * call is running or work is queued; what went is made again by the first call that needs it.  The JNI library calls
 * this for every slot that has been idle for a second (GKL_HIP_IDLE_RELEASE_MS).  *streams_released: how many streams went. */
int gklhip_release_idle(gklhip_ctx* ctx, int32_t* streams_released);

/* Fault injection for tests of a caller's error handling (the JNI shim retries a failed call once on fresh contexts):
 * spec "compute:N" or "compute:NxK" makes the N-th .. (N+K-1)-th gklhip_compute of the process, counted from this call,
 * fail with GKLHIP_ERR_HIP before any work (output array poisoned with NaN); NULL or "" disarms.  The first gklhip_init of
 * a process arms it from the environment variable GKLHIP_FAULT_INJECT. */
int gklhip_fault_inject(const char* spec);

const char* gklhip_strerror(int status);
/* Thread-local detail message of the last failing call on this thread. */
const char* gklhip_last_error(void);
int gklhip_device_count(void);
int gklhip_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GKL_HIP_PAIRHMM_H */
