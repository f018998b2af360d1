// Small device kernels around the PairHMM forward kernels: stream construction, the fp32 -> fp64 precision
// policy (IntelPairHmm.cc:157-165), log10 finalisation, and the device-side planning of the packed fp64
// recomputation pass (counting sort of the affected reads, window packing, run detection, job ordering).
// Included by pairhmm_api.hip only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gkl_hip_pairhmm.h"
#include "pairhmm_fwd_kernel.h"

namespace gklhip {

// Per-call preparation in ONE launch (it used to be a clear kernel + a stream kernel fed by a host-built source
// index per column): one wavefront per haplotype turns its bases into stream entries (base codes, then the
// separator, then -- behind the last haplotype of a group -- 64 idle entries of drain room), coalesced, and
// flags a haplotype that contains an 'N' (the fp64 kernels route such haplotypes through the general step, see
// WaveJob::kCodes).  The same launch zeroes the small per-call arrays the later kernels count into.
struct PrepArgs {
  const uint8_t* hap_bases;
  const int32_t* hap_src;    // [n_haps] stream order: offset of the haplotype's first base in hap_bases
  const int32_t* hap_len;    // [n_haps] stream order
  const int32_t* hap_pos;    // [n_haps] stream order: stream index of column 1
  const int32_t* hap_group;  // [n_haps] stream order
  const int32_t* hap_pos_flat;  // [n_haps] stream order: position in the flat stream (NULL: none)
  uint32_t* stream_flat;     // all haplotypes back to back (the fp64 recomputation's jobs are runs of it), 64 idle entries at the end
  uint32_t* stream;
  uint8_t* hap_has_n;        // [n_haps] stream order (every entry written: no clear needed)
  int32_t n_haps;
  int32_t* clear_a; int32_t n_a;  // counters
  int32_t* clear_b; int32_t n_b;  // per-read fallback counts
  int32_t* clear_c; int32_t n_c;  // fallback-count histogram
  // small host-buffer calls: the plan block (plan arrays + the call's six input arrays) sits in pinned host memory
  // and this kernel PULLS it into HBM itself -- a copy-engine transfer in front of the first kernel costs ~20 us
  // of queue hand-offs, more than everything this kernel does.  The pointers above then point into the HOST copy
  // (the device copy is complete only when this kernel has finished).
  const uint4* pull_src; uint4* pull_dst; int32_t pull_n16;  // 16-byte words
  int32_t hap_blocks;  // blocks [0, hap_blocks) build the stream, the rest pull
  // The read packing arrives in compact form (pairhmm_plan.h: chunk and first lane per read, lanes taken per chunk --
  // 5 bytes per read instead of 8 bytes per lane, 50 KB instead of 1.5 MB for 10k reads) and is expanded here into the
  // lane map the forward kernels read.
  const int32_t* place_chunk; const uint8_t* place_lane; const uint8_t* chunk_used; const int64_t* read_off;
  LaneSlot* lanes_out;
  int32_t n_reads, n_chunks, rpl;
};
constexpr int kPrepBlock = 256;
// block `block` of `grid` blocks of one call's preparation
__device__ __forceinline__ void prep_block(const PrepArgs& a, int block, int grid) {
  if (block >= a.hap_blocks) {
    const int stride = (grid - a.hap_blocks) * kPrepBlock;
    for (int w = (block - a.hap_blocks) * kPrepBlock + (int)threadIdx.x; w < a.pull_n16; w += stride) a.pull_dst[w] = a.pull_src[w];
    return;
  }
  const int i = block * kPrepBlock + threadIdx.x;
  if (i < a.n_a) a.clear_a[i] = 0;
  if (i < a.n_b) a.clear_b[i] = 0;
  if (i < a.n_c) a.clear_c[i] = 0;
  if (i < a.n_reads) {
    const int chunk = a.place_chunk[i];
    if (chunk >= 0) {
      const int nb = ((int)(a.read_off[i + 1] - a.read_off[i]) + a.rpl) / a.rpl;
      LaneSlot* dst = a.lanes_out + (int64_t)chunk * kLanes + a.place_lane[i];
      for (int b = 0; b < nb; b++) dst[b] = LaneSlot{i, b};
    }
  }
  if (i < a.n_chunks) {
    LaneSlot* dst = a.lanes_out + (int64_t)i * kLanes;
    for (int l = a.chunk_used[i]; l < kLanes; l++) dst[l] = LaneSlot{-1, 0};
  }
  const int lane = threadIdx.x & 63;
  const int k = block * (kPrepBlock / 64) + (threadIdx.x >> 6);
  if (k >= a.n_haps) return;
  const int len = a.hap_len[k], pos = a.hap_pos[k];
  const int posf = a.hap_pos_flat ? a.hap_pos_flat[k] : 0;
  const uint8_t* src = a.hap_bases + a.hap_src[k];
  bool has_n = false;
  for (int c = lane; c < len; c += 64) {
    const uint8_t b = src[c];  // pairhmm_common.h:57-61: A0 C1 T2 G3 N4, anything else 0
    const uint32_t e = b == 'C' ? 1u : b == 'T' ? 2u : b == 'G' ? 3u : b == 'N' ? 4u : 0u;
    a.stream[pos + c] = e;
    if (a.hap_pos_flat) a.stream_flat[posf + c] = e;
    has_n |= b == 'N';
  }
  const bool group_ends = k + 1 == a.n_haps || a.hap_group[k + 1] != a.hap_group[k];
  if (lane == 0) {
    a.stream[pos + len] = kEntSep | (uint32_t)k;
    if (a.hap_pos_flat) a.stream_flat[posf + len] = kEntSep | (uint32_t)k;
    a.hap_has_n[k] = 0;
  }
  if (group_ends) a.stream[pos + len + 1 + lane] = kEntIdle;
  if (a.hap_pos_flat && k + 1 == a.n_haps) a.stream_flat[posf + len + 1 + lane] = kEntIdle;
  if (__ballot(has_n) != 0 && lane == 0) a.hap_has_n[k] = 1;
}
__global__ __launch_bounds__(kPrepBlock) void prep_kernel(PrepArgs a) { prep_block(a, (int)blockIdx.x, (int)gridDim.x); }

// Tables at gklhip_init: pulled from a pinned host block by the device itself, like a small call's plan block -- a
// hipMemcpy would be the process's only reason to open the copy engines' queues (see SmallCombiner::make_streams on why a
// process should hold as few hardware queues as it can).
__global__ __launch_bounds__(256) void pull_words_kernel(const uint32_t* src, uint32_t* dst, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dst[i] = src[i];
}

// ---- several small host-buffer calls in ONE set of launches (pairhmm_api.hip: SmallCombiner) ----
// The device executes the kernels of about four queues at a time (tools/ubench_launch.hip), so sixteen callers with a
// GATK-sized region each get no more through than four.  When calls arrive while others are in flight, their three
// launches (prep, fp32 forward, per-pair policy) are issued once for all of them: a block finds its call from the block
// offsets in the kernel arguments and runs that call's part exactly as the single-call kernel would.  The per-call
// arguments are the call's descriptor, which travels in its plan block.
struct SmallCall {
  PrepArgs prep;
  FwdArgs<float> f;
  FwdArgs<double> d;
  PairPolicyArgs q;
  int32_t prep_grid, rpl_main, main_blocks, rows, n_pairs, fma;
  int32_t fused;  // the whole pair in one wavefront (pair_fused_block) instead of the packed fp32 pass + per-pair policy
  int32_t speculate;  // host side only: the context asked for the fp64 pass beside the fp32 one when the call runs alone
};
constexpr int kMultiMax = 16;
struct MultiArgs {
  const SmallCall* call[kMultiMax];  // prep: the descriptors in the pinned staging blocks; later kernels: the device copies
  int32_t begin[kMultiMax + 1];      // first block of each call in this launch
  int32_t n;
};
__device__ __forceinline__ int multi_find(const MultiArgs& m, int block) {
  int r = 0;
  for (int i = 1; i < m.n; i++) r += block >= m.begin[i] ? 1 : 0;
  return __builtin_amdgcn_readfirstlane(r);
}
__global__ __launch_bounds__(kPrepBlock) void prep_multi_kernel(MultiArgs m) {
  const int r = multi_find(m, (int)blockIdx.x);
  const SmallCall* c = m.call[r];
  const PrepArgs a = c->prep;
  prep_block(a, (int)blockIdx.x - m.begin[r], c->prep_grid);
}
template <bool FMA, int kRplF32>  // kRplF32: the widest fp32 variant (2, 4 or this many rows per lane)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void fwd_stream_multi_kernel(MultiArgs m) {
  constexpr int kLds2 = WaveJob<float, 2, FMA>::kLdsBytes, kLds4 = WaveJob<float, 4, FMA>::kLdsBytes, kLds8 = WaveJob<float, kRplF32, FMA>::kLdsBytes;
  constexpr int kLds = kLds2 > kLds4 ? (kLds2 > kLds8 ? kLds2 : kLds8) : (kLds4 > kLds8 ? kLds4 : kLds8);
  __shared__ __attribute__((aligned(16))) unsigned char lds[kLds];
  const int r = multi_find(m, (int)blockIdx.x);
  const SmallCall* c = m.call[r];
  const int block = (int)blockIdx.x - m.begin[r];
  const FwdArgs<float> a = c->f;
  const int rpl = c->rpl_main;
  if (rpl == 2)      fwd_stream_block<float, 2, FMA>(a, block, lds);
  else if (rpl == 4) fwd_stream_block<float, 4, FMA>(a, block, lds);
  else               fwd_stream_block<float, kRplF32, FMA>(a, block, lds);
}
template <bool FMA, int kRplF64>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void pair_policy_multi_kernel(MultiArgs m) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[WaveJob<double, kRplF64, FMA>::kLdsBytes];
  static_assert(WaveJob<double, 2, FMA>::kLdsBytes <= WaveJob<double, kRplF64, FMA>::kLdsBytes &&
                WaveJob<double, 4, FMA>::kLdsBytes <= WaveJob<double, kRplF64, FMA>::kLdsBytes, "LDS of the widest job");
  const int r = multi_find(m, (int)blockIdx.x);
  const SmallCall* c = m.call[r];
  const int64_t p = (int)blockIdx.x - m.begin[r];
  const FwdArgs<double> d = c->d;
  const PairPolicyArgs q = c->q;
  pair_policy_block<kRplF64, FMA>(d, q, p, lds);  // (rows per lane by the pair's own read)
}

template <bool FMA, int kRplF64>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(kRplF64 <= 4 ? 4 : 3))) void pair_fused_multi_kernel(MultiArgs m) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[PairFusedLds<kRplF64, FMA>::bytes];
  const int r = multi_find(m, (int)blockIdx.x);
  const SmallCall* c = m.call[r];
  const int64_t p = (int)blockIdx.x - m.begin[r];
  const FwdArgs<float> f = c->f;
  const FwdArgs<double> d = c->d;
  const PairPolicyArgs q = c->q;
  pair_fused_block<kRplF64, FMA>(f, d, q, p, lds);
}

constexpr int kModePacked = kModePackedWords;  // FinalizeArgs::mode: `out` receives packed raw sums (kPackedF32Tag, pairhmm_fwd_kernel.h)

struct FinalizeArgs {
  const float* raw32;
  const double* raw64;
  double* out;
  uint8_t* used64;
  int32_t* count;      // [0] number of pairs the policy sent to the fp64 pass
  int32_t* read_fail;  // [n_reads] number of haplotypes each read must be recomputed against
  int32_t n_haps;
  int64_t n;
  int mode;            // gklhip_finalize (device modes only), -1: no output, kModePacked: `out` = packed raw sums
  float log10_init_f;  // log10f(2^120), host libm
  double log10_init32_as_f64;  // log10(2^120) in double
  double log10_init_d;         // log10(2^1020)
};

// log10 of the fp32 sums the policy kept (device finalisation modes).  Launched on the context's side stream
// right after the policy, so its ~1.3 M double-precision log10 overlap the fp64 pass.
__global__ void finalize32_kernel(FinalizeArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n || a.used64[i]) return;
  const float v = a.raw32[i];
  if (a.mode == GKLHIP_FINALIZE_DEVICE_F64) a.out[i] = log10((double)v) - a.log10_init32_as_f64;
  else if (a.mode == GKLHIP_FINALIZE_DEVICE_REF32) a.out[i] = (double)((float)log10((double)v) - a.log10_init_f);
}

// log10 of the fp64 sums: all pairs (useDoublePrecision: all_pairs != 0) or the ones the policy flagged.
__global__ void finalize64_kernel(FinalizeArgs a, int all_pairs) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.n) return;
  if (all_pairs) a.used64[p] = 1;
  else if (!a.used64[p]) return;
  if (a.mode >= 0) a.out[p] = log10(a.raw64[p]) - a.log10_init_d;
  if (a.mode == kModePacked) reinterpret_cast<uint64_t*>(a.out)[p] = packed_word(a.raw64[p]);
}

// ---- precision policy + planning of the packed fp64 recomputation pass: THREE stream-ordered launches -------
// Phases:
//   P  policy of IntelPairHmm.cc:157-165 on the raw fp32 sums: keep pairs with sum >= 1e-28f, flag the rest for
//      the fp64 kernel and count, per read, how many haplotypes it failed against;
//   H  histogram of those counts over the reads that fit a chunk;   S  bucket starts for DESCENDING count;
//   C  counting-sort scatter: affected reads ordered by how many haplotypes they failed against;
//   W  window by window (kPackWindow reads, one wavefront each) best-fit-decreasing packing into 64-lane chunks --
//      the device twin of pack_reads_windowed().  Reads with similar fallback counts share chunks; in nested
//      patterns (a read underflows against every haplotype shorter than some length) a chunk then needs one
//      contiguous run of the length-sorted haplotype stream;
//   J  per chunk, the runs of consecutive haplotypes (stream order, never across a stream group) that at least
//      one of its reads must be recomputed against: one wave job per run (also for the pseudo-chunks of reads too
//      long for a chunk, which feed the striped kernel);
//   O  the job list ordered by decreasing length (counting sort on columns / 128, one block), so that the
//      persistent wavefronts of the jobs kernel start the long runs first and the kernel's tail is short.
// Launches: plan_policy_kernel = P on the whole grid, then H S C by the block that finishes P LAST;
// plan_pack_kernel = W; plan_jobs_kernel = J on the whole grid, then O by the block that finishes J last.
// No block ever waits for another one: the single-block phases run in whichever block finds, on an arrival counter,
// that every other block has already left its grid-wide phase (last_block_done).  Nothing therefore depends on the
// blocks being resident together, on how many other launches share the device, or on the size of the device
// (rounds 2-3 ran all seven phases in ONE launch with spinning grid barriers, which needed a process-wide gate on
// concurrent launches and still could not cover several processes or a partitioned device).
// None of the arrays written here is declared const/__restrict__: data produced in one phase is read in the next,
// and the compiler must not move such reads to the scalar cache, which the fences do not invalidate.
#ifndef GKL_PACK_WINDOW
#define GKL_PACK_WINDOW 96
#endif
constexpr int kPackWindow = GKL_PACK_WINDOW;   // (r05 A/B on the bench batch: 64 / 96 / 128 reads per window, see docs/NOTES.md)
constexpr int kJobClasses = 64;
constexpr int kPlanBlock = 1024;

struct PlanArgs {
  FinalizeArgs fa;
  int32_t n_reads, n_haps, n_pairs_i;
  const int64_t* read_off;
  int32_t rpl;           // rows per lane of the fp64 kernel
  int32_t max_len;       // longest read that fits a chunk at that rpl
  int32_t* cnts;         // [0] fp64 pairs [2] jobs [3] next job [4] affected reads [5] chunks [7] next long job (main pass) [8] long jobs [9] next long job [10] [11] arrival counters of the plan launches [12] [13] next long job of the striped launch behind a super-stripe launch (main / fp64 pass) [16..] stamps
                         // [10] [11] arrival counters of plan_policy_kernel / plan_jobs_kernel
  int32_t* hist;         // [n_haps + 2]
  int32_t* pos;          // [n_haps + 2]
  int32_t* order;        // [n_reads]
  LaneSlot* lanes2;      // [n_reads * 64] worst case
  const int32_t* hap_orig;
  const int32_t* hap_group;
  const int32_t* hap_pos;   // positions in the FLAT stream (the one the fp64 kernels read)
  const int32_t* hap_len;
  FwdJob* jobs;          // as built
  FwdJob* sorted;        // by decreasing length
  const LaneSlot* long_lanes;  // pseudo-chunks (lane 0 names the read) of reads too long for a chunk
  int32_t n_long;
  FwdJob* jobs_long;     // [n_long * n_haps]
  int32_t* long_chunk_jobs;  // [n_long]
  // job length: runs are cut after `wanted` columns' worth ... see phase J
  int32_t total_cols;    // columns + separators of all haplotypes
  int32_t wanted_jobs;   // the pass is cut into about this many jobs (when the runs allow it)
  int32_t min_job_cols;
  int32_t packed_by_kernels;  // the forward kernels wrote the packed words themselves (FwdArgs::packed_out)
};

// Every thread of every block calls this once, after the block's share of a grid-wide phase.  True in exactly one
// block: the one that arrives last, which then sees everything the other blocks wrote before arriving.
__device__ __forceinline__ bool last_block_done(int32_t* arrivals) {
  __shared__ int32_t s_is_last;
  __syncthreads();  // every store of the block has reached this XCD's L2 (workgroup-scope release: vmcnt(0) before the barrier)
  if (threadIdx.x == 0) {
    __threadfence();  // release at agent scope: ONE write-back of the L2 per block (1024 of them cost the policy launch ~20 us)
    const int32_t before = atomicAdd(arrivals, 1);
    s_is_last = before == (int32_t)gridDim.x - 1 ? 1 : 0;
    __threadfence();  // acquire the other blocks' writes (invalidates this CU's vector cache and the non-local L2 lines)
  }
  __syncthreads();
  return s_is_last != 0;
}

// counters are written with atomics (performed at the L2) and read back through it
__device__ __forceinline__ int32_t ld_cnt(int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int job_class(const FwdJob& j, const int32_t* hap_pos, const int32_t* hap_len) {
  const int cols = hap_pos[j.hap_end - 1] + hap_len[j.hap_end - 1] - hap_pos[j.hap_begin];
  const int c = cols >> 7;
  return kJobClasses - 1 - (c < kJobClasses - 1 ? c : kJobClasses - 1);  // class 0 = longest
}

// Run detection for one (pseudo-)chunk by ONE wavefront, 64 haplotypes (stream order) at a time: lane = haplotype.
// A needed haplotype starts a job if its predecessor is not needed, or lies in another cut_cols-wide window of the
// flat stream (jobs are cut on a fixed grid of stream positions -- no prefix sums), or -- `by_group`, the striped
// long-read kernel whose carry rows are sized for one stream group -- in another stream group; the job ends at the
// next start or the next haplotype that is not needed.  Everything is ballots and bit tricks on the two masks.  A run
// that continues from one 64-haplotype segment into the next extends the job it already has.
// The jobs of chunk c go to jobs[c * n_haps ...] (a job holds at least one haplotype), their number to chunk_jobs[c]:
// no atomics at all (thousands of atomics on ONE job counter retire one per ~7 ns and were most of this phase).
__device__ __forceinline__ void build_jobs_for_chunk(const PlanArgs& a, const LaneSlot* lanes, int c, FwdJob* jobs,
                                                     int32_t* chunk_jobs, int cut_cols, bool by_group, int lane) {
  const LaneSlot sl = lanes[(int64_t)c * kLanes + lane];
  jobs += (int64_t)c * a.n_haps;
  int n_jobs = 0;
  bool prev_need = false;  // the last haplotype of the previous segment is needed (wave-uniform)
  uint64_t reads = __ballot(sl.read >= 0 && sl.block == 0);  // lanes that name a read of the chunk
  for (int k0 = 0; k0 < a.n_haps; k0 += kLanes) {
    const int k = k0 + lane;
    const bool in = k < a.n_haps;
    const int h = in ? a.hap_orig[k] : 0;
    uint8_t nd = 0;
    for (uint64_t m = reads; m; m &= m - 1) {
      const int r = __builtin_amdgcn_readlane(sl.read, __builtin_ctzll(m));
      if (in) nd |= a.fa.used64[(int64_t)r * a.n_haps + h];
    }
    const uint64_t need = __ballot(in && nd != 0);
    if (!need) { prev_need = false; continue; }
    bool start = false;
    if (in && nd) {
      const bool pred_needed = lane == 0 ? prev_need : ((need >> (lane - 1)) & 1ull) != 0;
      start = !pred_needed || a.hap_pos[k - 1] / cut_cols != a.hap_pos[k] / cut_cols ||
              (by_group && a.hap_group[k - 1] != a.hap_group[k]);
    }
    const uint64_t starts = __ballot(start);
    if ((need & 1ull) && !(starts & 1ull)) {
      // the run at the head of this segment continues the previous segment's last job: extend it
      const uint64_t stop = starts | ~need;  // bit 0 is clear in both
      const int e = stop ? __builtin_ctzll(stop) : kLanes;
      if (lane == 0) jobs[n_jobs - 1].hap_end = min(k0 + e, a.n_haps);
    }
    const int base = n_jobs;
    n_jobs += __builtin_popcountll(starts);
    if (start) {
      const uint64_t after = lane == kLanes - 1 ? 0ull : (~0ull << (lane + 1));
      const uint64_t stop = (starts | ~need) & after;
      const int e = stop ? __builtin_ctzll(stop) : kLanes;
      FwdJob j;
      j.chunk = c; j.hap_begin = k; j.hap_end = min(k0 + e, a.n_haps);
      j.klass = 0;
      jobs[base + __builtin_popcountll(starts & ((1ull << lane) - 1ull))] = j;
    }
    prev_need = ((need >> (kLanes - 1)) & 1ull) != 0;
  }
  // the length class of every job rides along for phase O; only now are the ends final
  __threadfence_block();
  for (int i = lane; i < n_jobs; i += kLanes) {
    FwdJob j = jobs[i];
    jobs[i].klass = job_class(j, a.hap_pos, a.hap_len);
  }
  if (lane == 0) chunk_jobs[c] = n_jobs;
}

// stamps for GKLHIP_TIMING: cnts[16 + k] = 10 ns ticks since the start of the phase's own kernel
#define GKLHIP_PLAN_STAMP(k) do { if (tid == 0) a.cnts[16 + (k)] = (int32_t)(wall_clock64() - clk0); } while (0)

// Phases H S C by ONE block (the last one to leave the policy): counting sort of the affected reads by DESCENDING
// fallback count.  `hist` / `pos` are LDS arrays when n_haps + 2 entries fit (kInLds: cleared here; every atomic with a
// returned value is then an LDS operation of ~0.1 us instead of a ~2 us round trip to the L2) and the context's global
// arrays (cleared by prep_kernel) otherwise.  Loads are issued four reads per thread at a time so their latencies overlap.
constexpr int kPlanLdsBins = 4096;
template <bool kInLds>
__device__ __forceinline__ void order_affected_reads(const PlanArgs& a, int32_t* hist, int32_t* pos, int32_t* s_wave_tot,
                                                     int tid, uint64_t clk0) {
  constexpr int kBatch = 4;
  const int n = a.n_haps, lane = tid & 63, wave = tid >> 6;
  if (kInLds) {
    for (int c = tid; c < n + 2; c += kPlanBlock) hist[c] = 0;
    __syncthreads();
  }
  // ---- H: histogram of fallback counts (reads longer than max_len take the striped long-read path) ----
  for (int r0 = tid; r0 < a.n_reads; r0 += kBatch * kPlanBlock) {
    int f[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; k++) {
      const int r = r0 + k * kPlanBlock;
      const bool in = r < a.n_reads;
      const int fr = in ? a.fa.read_fail[r] : 0;
      const int64_t len = in ? a.read_off[r + 1] - a.read_off[r] : 0;
      f[k] = (fr > 0 && len <= a.max_len) ? fr : 0;
    }
#pragma unroll
    for (int k = 0; k < kBatch; k++)
      if (f[k] > 0) atomicAdd(hist + f[k], 1);
  }
  __threadfence();
  __syncthreads();
  GKLHIP_PLAN_STAMP(0);

  // ---- S: bucket starts for DESCENDING count: pos[c] = #reads with count > c ----
  {
    // per-thread contiguous segments of the count range [1, n_haps], highest counts first
    const int per = (n + kPlanBlock - 1) / kPlanBlock;
    const int hi = n - tid * per, lo = max(hi - per, 0);  // this thread owns counts (lo, hi]
    int sum = 0;
    for (int c = hi; c > lo; c--) sum += hist[c];
    int inc = sum;  // inclusive scan over the wavefront, then over the wavefronts' totals
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave_tot[wave] = inc;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < kPlanBlock / 64; w++) {
      const int t = s_wave_tot[w];
      base += w < wave ? t : 0;
      total += t;
    }
    if (tid == 0) __hip_atomic_store(a.cnts + 4, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int acc = base + inc - sum;
    for (int c = hi; c > lo; c--) { pos[c] = acc; acc += hist[c]; }
  }
  __threadfence();
  __syncthreads();
  GKLHIP_PLAN_STAMP(1);

  // ---- C: scatter ----
  for (int r0 = tid; r0 < a.n_reads; r0 += kBatch * kPlanBlock) {
    int f[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; k++) {
      const int r = r0 + k * kPlanBlock;
      const bool in = r < a.n_reads;
      const int fr = in ? a.fa.read_fail[r] : 0;
      const int64_t len = in ? a.read_off[r + 1] - a.read_off[r] : 0;
      f[k] = (fr > 0 && len <= a.max_len) ? fr : 0;
    }
#pragma unroll
    for (int k = 0; k < kBatch; k++)
      if (f[k] > 0) a.order[atomicAdd(pos + f[k], 1)] = r0 + k * kPlanBlock;
  }
}

__global__ __launch_bounds__(kPlanBlock) void plan_policy_kernel(PlanArgs a) {
  __shared__ int32_t s_i32[kPlanBlock / 64];  // phase S: scan
  __shared__ int32_t s_wave[kPlanBlock / 64];
  __shared__ int32_t s_hist[2 * kPlanLdsBins];  // phases H S C: histogram and bucket positions when they fit
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nblk = (int)gridDim.x, blk = (int)blockIdx.x;

  // ---- P: policy ----
  {
    int32_t my_fails = 0;
    for (int64_t base = (int64_t)blk * kPlanBlock; base < a.fa.n; base += (int64_t)nblk * kPlanBlock) {
      const int64_t i = base + tid;
      const bool in_range = i < a.fa.n;
      const float v = in_range ? a.fa.raw32[i] : 1.0f;
      const bool fails = in_range && v < 1e-28f;  // NaN compares false and stays fp32, like the reference
      const uint64_t mask = __ballot(fails);
      if (mask) {
        // r-major pairs: the failing lanes of a wavefront usually belong to one read (n_haps >= 64) or a few
        const int leader = __builtin_ctzll(mask);
        const int32_t read = fails ? (int32_t)((uint32_t)i / (uint32_t)a.n_haps) : -1;  // (pairs < 2^31: a 32-bit division, not the 64-bit one of `i / n_haps`)
        const int32_t lead_read = __shfl(read, leader, 64);
        const uint64_t same = __ballot(fails && read == lead_read);
        if (lane == leader) atomicAdd(a.fa.read_fail + lead_read, __builtin_popcountll(same));
        if (fails && read != lead_read) atomicAdd(a.fa.read_fail + read, 1);
        if (lane == 0) my_fails += __builtin_popcountll(mask);
      }
      if (in_range) {
        a.fa.used64[i] = fails ? 1 : 0;
        // "pending" (0) for the flagged pairs: finalize64_kernel fills them in; the device log10 of the kept pairs
        // is finalize32_kernel's job, which runs beside the fp64 pass
        if (a.fa.mode == kModePacked && !a.packed_by_kernels)
          reinterpret_cast<uint64_t*>(a.fa.out)[i] = fails ? 0ull : (kPackedF32Tag | (uint64_t)__float_as_uint(v));
      }
    }
    // one atomic per block on the pair counter (same-address atomics retire one per ~7 ns at the L2)
    if (lane == 0) s_wave[wave] = my_fails;
    __syncthreads();
    if (tid == 0) {
      int total = 0;
      for (int w = 0; w < kPlanBlock / 64; w++) total += s_wave[w];
      if (total) atomicAdd(a.cnts + 0, total);
    }
  }
  if (!last_block_done(a.cnts + 10)) return;
  // ---- this block left the policy last: it alone sorts the affected reads (a few thousand of them) ----
  const int n_fail = ld_cnt(a.cnts + 0);
  if (n_fail == 0) return;  // nothing underflowed: the next two launches see the same count and leave at once
  const uint64_t clk0 = wall_clock64();

  if (a.n_haps + 2 <= kPlanLdsBins) order_affected_reads<true>(a, s_hist, s_hist + kPlanLdsBins, s_i32, tid, clk0);
  else                             order_affected_reads<false>(a, a.hist, a.pos, s_i32, tid, clk0);
  GKLHIP_PLAN_STAMP(2);
}

__global__ __launch_bounds__(kPlanBlock) void plan_pack_kernel(PlanArgs a) {
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nblk = (int)gridDim.x, blk = (int)blockIdx.x;
  if (ld_cnt(a.cnts + 0) == 0) return;
  const uint64_t clk0 = wall_clock64();

  // ---- W: pack windows, one wavefront per window ----
  {
    // One wavefront per window of <= kPackWindow reads: rank sort by lanes needed (descending, stable), then best fit.
    static_assert(kPackWindow <= 2 * kLanes, "the window's lane counts sit in two registers per lane");
    __shared__ int32_t s_pack[kPlanBlock / 64][4][kPackWindow];
    int32_t* s_read = s_pack[wave][0]; int32_t* s_needl = s_pack[wave][1]; int32_t* s_sread = s_pack[wave][2];
    int32_t* s_sneed = s_pack[wave][3];
    const int n = ld_cnt(a.cnts + 4);
    const int n_win = (n + kPackWindow - 1) / kPackWindow;
    for (int w = blk * (kPlanBlock / 64) + wave; w < n_win; w += nblk * (kPlanBlock / 64)) {
      const int w0 = w * kPackWindow;
      const int cnt = min(kPackWindow, n - w0);
      for (int i = lane; i < cnt; i += kLanes) {
        const int r = a.order[w0 + i];
        s_read[i] = r;
        s_needl[i] = (int)((a.read_off[r + 1] - a.read_off[r] + a.rpl) / a.rpl);  // blocks_for()
      }
      __builtin_amdgcn_wave_barrier();
      if (w == 0 && lane == 0) a.cnts[22] = (int32_t)(wall_clock64() - clk0);
      for (int i = lane; i < cnt; i += kLanes) {
        const int ni = s_needl[i];
        int rank = 0;
        for (int j = 0; j < cnt; j++) {
          const int nj = s_needl[j];
          rank += (nj > ni) || (nj == ni && j < i);
        }
        s_sread[rank] = s_read[i];
        s_sneed[rank] = ni;
      }
      __builtin_amdgcn_wave_barrier();
      if (w == 0 && lane == 0) a.cnts[23] = (int32_t)(wall_clock64() - clk0);
      // best fit: lane f owns the stack of open bins with exactly f free lanes; the stacks are linked through
      // next_lo/next_hi (lane b holds the successor of bins b and b + 64), a wave-uniform bit mask says which stacks
      // are non-empty -- the smallest free size that fits is one ctz away.  All in registers: readlane/select only.
      int head = -1, next_lo = -1, next_hi = -1;
      int bin_lo = 0, off_lo = 0, bin_hi = 0, off_hi = 0;  // where reads `lane` and `lane + 64` (sorted order) went
      uint64_t avail = 0;
      int nb = 0;  // bins opened so far
      const int need_lo = lane < cnt ? s_sneed[lane] : 0, need_hi = lane + kLanes < cnt ? s_sneed[lane + kLanes] : 0;
      for (int i = 0; i < cnt; i++) {
        const int nn = __builtin_amdgcn_readlane(i < kLanes ? need_lo : need_hi, i & 63);  // 1..64
        const uint64_t fits = nn >= kLanes ? 0ull : (avail >> nn) << nn;
        int bin, f;
        if (fits) {
          f = __builtin_ctzll(fits);
          bin = __builtin_amdgcn_readlane(head, f);
          const int nxt = __builtin_amdgcn_readlane(bin < kLanes ? next_lo : next_hi, bin & 63);
          if (lane == f) head = nxt;
          if (nxt < 0) avail &= ~(1ull << f);
        } else {
          bin = nb++;
          f = kLanes;
        }
        const int left = f - nn;
        if (left > 0) {
          const int old = __builtin_amdgcn_readlane(head, left);
          if (lane == (bin & 63)) { if (bin < kLanes) next_lo = old; else next_hi = old; }
          if (lane == left) head = bin;
          avail |= 1ull << left;
        }
        if (lane == (i & 63)) { if (i < kLanes) { bin_lo = bin; off_lo = kLanes - f; } else { bin_hi = bin; off_hi = kLanes - f; } }
      }
      if (w == 0 && lane == 0) a.cnts[24] = (int32_t)(wall_clock64() - clk0);
      int base = 0;
      if (lane == 0) base = atomicAdd(a.cnts + 5, nb);
      base = __shfl(base, 0, kLanes);
      LaneSlot idle; idle.read = -1; idle.block = 0;
      for (int i = lane; i < nb * kLanes; i += kLanes) a.lanes2[(int64_t)base * kLanes + i] = idle;
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();  // the idle slots before the real ones (other lanes of this wavefront wrote them)
      if (w == 0 && lane == 0) a.cnts[25] = (int32_t)(wall_clock64() - clk0);
      // every lane writes the slots of its (up to two) reads
      if (lane < cnt) {
        LaneSlot sl; sl.read = s_sread[lane];
        LaneSlot* dst = a.lanes2 + (int64_t)(base + bin_lo) * kLanes + off_lo;
        for (int b = 0; b < need_lo; b++) { sl.block = b; dst[b] = sl; }
      }
      if (lane + kLanes < cnt) {
        LaneSlot sl; sl.read = s_sread[lane + kLanes];
        LaneSlot* dst = a.lanes2 + (int64_t)(base + bin_hi) * kLanes + off_hi;
        for (int b = 0; b < need_hi; b++) { sl.block = b; dst[b] = sl; }
      }
      __builtin_amdgcn_wave_barrier();
      if (w == 0 && lane == 0) a.cnts[26] = (int32_t)(wall_clock64() - clk0);
    }
  }
  if (blk == 0) GKLHIP_PLAN_STAMP(3);
}

__global__ __launch_bounds__(kPlanBlock) void plan_jobs_kernel(PlanArgs a) {
  __shared__ int32_t s_i32[kPlanBlock];  // phase O: class counters
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nblk = (int)gridDim.x, blk = (int)blockIdx.x;
  const int n_fail = ld_cnt(a.cnts + 0);
  if (n_fail == 0) return;
  const uint64_t clk0 = wall_clock64();

  // ---- J: jobs = needed haplotype runs per chunk, one wavefront per chunk ----
  {
    const int total = ld_cnt(a.cnts + 5);
    // Job length.  A run is at most a stream group (~2048 columns); a big batch has thousands of them and the
    // longest-first order keeps the tail short.  A small one (an eighth of a batch on each of 8 GPUs, a GATK region)
    // would hand a few hundred long jobs to 3072 wavefront slots -- one wavefront per SIMD issues an instruction only
    // every ~6 cycles -- so runs are cut to about (estimated wave-steps of the pass / wanted_jobs) columns:
    // wave-steps ~ chunks x all columns x the fraction of (affected read, haplotype) pairs that were flagged.
    const int n_fail_reads = ld_cnt(a.cnts + 4);
    const double density = (double)n_fail / ((double)max(n_fail_reads, 1) * (double)a.n_haps);
    const double est_steps = (double)max(total, 1) * (double)a.total_cols * fmin(density, 1.0);
    const int cut_cols = max(a.min_job_cols, (int)fmin(est_steps / (double)a.wanted_jobs, 1e9));
    const int n_waves = nblk * (kPlanBlock / 64), w = blk * (kPlanBlock / 64) + wave;
    // (`order` has done its duty in phase W: it now takes the job count of every chunk)
    for (int c = w; c < total; c += n_waves) build_jobs_for_chunk(a, a.lanes2, c, a.jobs, a.order, cut_cols, false, lane);
    for (int c = w; c < a.n_long; c += n_waves) build_jobs_for_chunk(a, a.long_lanes, c, a.jobs_long, a.long_chunk_jobs, 0x7fffffff, true, lane);
  }
  if (!last_block_done(a.cnts + 11)) return;
  GKLHIP_PLAN_STAMP(4);

  // ---- O: order the jobs longest first (block 0): counting sort over the chunks' job lists ----
  {
    int32_t* cnt = s_i32;                // [kJobClasses]
    int32_t* base = s_i32 + kJobClasses; // [kJobClasses]
    const int total = ld_cnt(a.cnts + 5);
    __syncthreads();
    if (tid < kJobClasses) cnt[tid] = 0;
    __syncthreads();
    for (int c = tid; c < total; c += kPlanBlock) {
      const FwdJob* mine = a.jobs + (int64_t)c * a.n_haps;
      const int n = a.order[c];
      for (int i = 0; i < n; i++) atomicAdd(&cnt[mine[i].klass], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int acc = 0;
      for (int c = 0; c < kJobClasses; c++) { base[c] = acc; acc += cnt[c]; }
      a.cnts[2] = acc;  // read by the next kernel
    }
    __syncthreads();
    for (int c = tid; c < total; c += kPlanBlock) {
      const FwdJob* mine = a.jobs + (int64_t)c * a.n_haps;
      const int n = a.order[c];
      for (int i = 0; i < n; i++) {
        const FwdJob j = mine[i];
        a.sorted[atomicAdd(&base[j.klass], 1)] = j;
      }
    }
    // the striped long-read kernel takes its jobs in one list: compact the pseudo-chunks' lists in place
    if (tid == 0 && a.n_long > 0) {
      int at = 0;
      for (int c = 0; c < a.n_long; c++) {
        const int n = a.long_chunk_jobs[c];
        for (int i = 0; i < n; i++) a.jobs_long[at++] = a.jobs_long[(int64_t)c * a.n_haps + i];  // at <= c * n_haps + i
      }
      a.cnts[8] = at;
    }
  }
  GKLHIP_PLAN_STAMP(5);
}
#undef GKLHIP_PLAN_STAMP

// ---- build self-test: the kernels of this library must run with fp32 AND fp64 denormals flushed (Makefile: -fgpu-flush-
// denormals-to-zero, -fdenormal-fp-math=preserve-sign; MXCSR.FTZ in the reference).  The generated fp64 programs zero a
// value by masking its high half and rely on the flush for the rest, so a library built without the flags must not start.
// out[0] = the bits of (smallest fp32 denormal * 1), out[1] = the high word of (an fp64 denormal * 1): both 0 when flushed.
__global__ void flush_selftest_kernel(uint32_t* out, float f, double d) {
  float fm;
  double dm;
  // (real multiplies: the compiler folds x * 1 away)
  asm volatile("v_mul_f32 %0, 1.0, %1" : "=v"(fm) : "v"(f));
  asm volatile("v_mul_f64 %0, 1.0, %1" : "=v"(dm) : "v"(d));
  out[0] = __float_as_uint(fm);
  out[1] = (uint32_t)((uint64_t)__double_as_longlong(dm) >> 32) | (uint32_t)(uint64_t)__double_as_longlong(dm);
}

// The fp32 whole-job programs (tools/gen_fwd_asm.py: OOB_PLANE) give a lane on a separator-type entry the priors of a
// "plane" that starts 512 row-blocks behind its own -- 256 KB or more, beyond the CU's whole LDS -- and rely on a DS read
// beyond the workgroup's allocation returning 0 (ISA manuals since GCN3).  Asked of the device once per process (dev_init):
// workgroups with the forward kernels' kind of allocation, every byte non-zero, neighbours resident on the same CU, read
// what such a lane reads at the three plane sizes (8 / 4 / 2 rows per lane: ds_read_b128 at the base and one plane
// further, ds_read_b64) and OR the bits into out[0].  Anything but 0 and the context keeps the fp32 general steps in C++
// (the round-3 arrangement, FwdArgs::asm_general = 0), where a separator lane's priors are selected, not fetched.
__global__ __launch_bounds__(256) void lds_oob_selftest_kernel(uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[4 * 2560];   // four wavefronts' 10 KB tables
  for (int i = threadIdx.x; i < 4 * 2560; i += 256) lds[i] = 0xA5A50000u | (uint32_t)i;
  __syncthreads();
  const uint32_t loff = (uint32_t)(uintptr_t)lds + (threadIdx.x >> 6) * 10240u + (threadIdx.x & 63u) * 16u;
  uint32_t acc = 0;
#pragma unroll
  for (int shift = 9; shift <= 11; shift++) {
    const uint32_t addr = (512u << shift) + loff;   // v_lshl_add_u32 ADDR, min(entry, 512), code_shift, LOFF
    uint32_t v0, v1, v2, v3, w0, w1, w2, w3, x0, x1;
    asm volatile("ds_read_b128 v[40:43], %10\n\tds_read_b128 v[44:47], %10 offset:1024\n\tds_read_b64 v[48:49], %10\n\ts_waitcnt lgkmcnt(0)\n\t"
                 "v_mov_b32 %0, v40\n\tv_mov_b32 %1, v41\n\tv_mov_b32 %2, v42\n\tv_mov_b32 %3, v43\n\t"
                 "v_mov_b32 %4, v44\n\tv_mov_b32 %5, v45\n\tv_mov_b32 %6, v46\n\tv_mov_b32 %7, v47\n\tv_mov_b32 %8, v48\n\tv_mov_b32 %9, v49"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(x0), "=v"(x1)
                 : "v"(addr) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "memory");
    acc |= v0 | v1 | v2 | v3 | w0 | w1 | w2 | w3 | x0 | x1;
  }
  if (acc) atomicOr(out, acc);
  if (lds[threadIdx.x] == 0) atomicOr(out, 0x80000000u);   // (keeps the stores above alive)
}

// The same question in the shape of the kernels with the biggest allocations (r05 advisor): the super-stripe kernel's
// 512 threads and ~78 KB of LDS (two such workgroups fill a CU's 160 KB, so a neighbour's LDS lies right behind this one's),
// the wide kernels' 128-256 threads and up to 128 KB.  kWords 32-bit words, every byte non-zero; each
// wavefront reads what a separator lane of its 10 KB table region reads.  Launched with enough workgroups to fill every
// CU several times over.  (Static allocation, like the kernels it stands for.)
template <int kWords, int kThreads>
__global__ __launch_bounds__(kThreads) void lds_oob_selftest_big_kernel(uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[kWords];
  constexpr int words = kWords;
  for (int i = threadIdx.x; i < words; i += blockDim.x) lds_dyn[i] = 0x5A5A0000u | (uint32_t)(i & 0xffff) | 1u;
  __syncthreads();
  const uint32_t wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const uint32_t region = ((uint32_t)words * 4u / waves) & ~15u;
  const uint32_t loff = (uint32_t)(uintptr_t)lds_dyn + wave * region + (threadIdx.x & 63u) * 16u;
  uint32_t acc = 0;
#pragma unroll
  for (int shift = 9; shift <= 11; shift++) {
    const uint32_t addr = (512u << shift) + loff;
    uint32_t v0, v1, v2, v3, w0, w1, w2, w3, x0, x1;
    asm volatile("ds_read_b128 v[40:43], %10\n\tds_read_b128 v[44:47], %10 offset:1024\n\tds_read_b64 v[48:49], %10\n\ts_waitcnt lgkmcnt(0)\n\t"
                 "v_mov_b32 %0, v40\n\tv_mov_b32 %1, v41\n\tv_mov_b32 %2, v42\n\tv_mov_b32 %3, v43\n\t"
                 "v_mov_b32 %4, v44\n\tv_mov_b32 %5, v45\n\tv_mov_b32 %6, v46\n\tv_mov_b32 %7, v47\n\tv_mov_b32 %8, v48\n\tv_mov_b32 %9, v49"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(x0), "=v"(x1)
                 : "v"(addr) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "memory");
    acc |= v0 | v1 | v2 | v3 | w0 | w1 | w2 | w3 | x0 | x1;
  }
  if (acc) atomicOr(out, acc);
  if (lds_dyn[threadIdx.x] == 0) atomicOr(out, 0x80000000u);   // (keeps the stores above alive)
}

// ---- diagnostics: the VALU issue ceiling of the forward recurrence's instruction mix ----------------------------
// Eight "cells" of 4 multiplies + 4 fused multiply-adds per loop iteration, operands in registers chosen so that no
// three-source op has all sources in one VGPR bank (even / odd), four wavefronts per SIMD on every CU: what the chip
// issues when nothing but the recurrence's arithmetic is in the way.  gklhip_measure_issue_ceiling times it and
// bench.py prices the forward kernel against it (roofline.issue_ceiling_tflops).  GENERATED text (same generator as
// tools/gen_ubench_banks2.py's "cell mix").
__global__ __launch_bounds__(256) void issue_mix_f32_kernel(int iters, uint64_t* cycles) {
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  asm volatile(
      "v_mov_b32 v24, 1.0\n\tv_mov_b32 v25, 1.0\n\tv_mov_b32 v26, 1.0\n\tv_mov_b32 v27, 1.0\n\tv_mov_b32 v28, 1.0\n\tv_mov_b32 v29, 1.0\n\tv_mov_b32 v30, 1.0\n\tv_mov_b32 v31, 1.0\n\t"
      "v_mov_b32 v32, 1.0\n\tv_mov_b32 v33, 1.0\n\tv_mov_b32 v34, 1.0\n\tv_mov_b32 v35, 1.0\n\tv_mov_b32 v36, 1.0\n\tv_mov_b32 v37, 1.0\n\tv_mov_b32 v38, 1.0\n\tv_mov_b32 v39, 1.0\n\t"
      "v_mov_b32 v40, 1.0\n\tv_mov_b32 v41, 1.0\n\tv_mov_b32 v42, 1.0\n\tv_mov_b32 v43, 1.0\n\tv_mov_b32 v44, 1.0\n\tv_mov_b32 v45, 1.0\n\tv_mov_b32 v46, 1.0\n\tv_mov_b32 v47, 1.0\n\t"
      "v_mov_b32 v48, 1.0\n\tv_mov_b32 v49, 1.0\n\tv_mov_b32 v50, 1.0\n\tv_mov_b32 v51, 1.0\n\tv_mov_b32 v52, 1.0\n\tv_mov_b32 v53, 1.0\n\tv_mov_b32 v54, 1.0\n\tv_mov_b32 v55, 1.0\n\t"
      "s_mov_b32 s6, %0\n\t"
      "1:\n\t"
      "v_mul_f32 v27, v24, v25\n\t"
      "v_fmac_f32 v27, v25, v26\n\t"
      "v_fmac_f32 v27, v24, v26\n\t"
      "v_mul_f32 v27, v27, v26\n\t"
      "v_mul_f32 v24, v24, v25\n\t"
      "v_fmac_f32 v24, v26, v25\n\t"
      "v_mul_f32 v26, v26, v25\n\t"
      "v_fmac_f32 v26, v27, v25\n\t"
      "v_mul_f32 v31, v28, v29\n\t"
      "v_fmac_f32 v31, v29, v30\n\t"
      "v_fmac_f32 v31, v28, v30\n\t"
      "v_mul_f32 v31, v31, v30\n\t"
      "v_mul_f32 v28, v28, v29\n\t"
      "v_fmac_f32 v28, v30, v29\n\t"
      "v_mul_f32 v30, v30, v29\n\t"
      "v_fmac_f32 v30, v31, v29\n\t"
      "v_mul_f32 v35, v32, v33\n\t"
      "v_fmac_f32 v35, v33, v34\n\t"
      "v_fmac_f32 v35, v32, v34\n\t"
      "v_mul_f32 v35, v35, v34\n\t"
      "v_mul_f32 v32, v32, v33\n\t"
      "v_fmac_f32 v32, v34, v33\n\t"
      "v_mul_f32 v34, v34, v33\n\t"
      "v_fmac_f32 v34, v35, v33\n\t"
      "v_mul_f32 v39, v36, v37\n\t"
      "v_fmac_f32 v39, v37, v38\n\t"
      "v_fmac_f32 v39, v36, v38\n\t"
      "v_mul_f32 v39, v39, v38\n\t"
      "v_mul_f32 v36, v36, v37\n\t"
      "v_fmac_f32 v36, v38, v37\n\t"
      "v_mul_f32 v38, v38, v37\n\t"
      "v_fmac_f32 v38, v39, v37\n\t"
      "v_mul_f32 v43, v40, v41\n\t"
      "v_fmac_f32 v43, v41, v42\n\t"
      "v_fmac_f32 v43, v40, v42\n\t"
      "v_mul_f32 v43, v43, v42\n\t"
      "v_mul_f32 v40, v40, v41\n\t"
      "v_fmac_f32 v40, v42, v41\n\t"
      "v_mul_f32 v42, v42, v41\n\t"
      "v_fmac_f32 v42, v43, v41\n\t"
      "v_mul_f32 v47, v44, v45\n\t"
      "v_fmac_f32 v47, v45, v46\n\t"
      "v_fmac_f32 v47, v44, v46\n\t"
      "v_mul_f32 v47, v47, v46\n\t"
      "v_mul_f32 v44, v44, v45\n\t"
      "v_fmac_f32 v44, v46, v45\n\t"
      "v_mul_f32 v46, v46, v45\n\t"
      "v_fmac_f32 v46, v47, v45\n\t"
      "v_mul_f32 v51, v48, v49\n\t"
      "v_fmac_f32 v51, v49, v50\n\t"
      "v_fmac_f32 v51, v48, v50\n\t"
      "v_mul_f32 v51, v51, v50\n\t"
      "v_mul_f32 v48, v48, v49\n\t"
      "v_fmac_f32 v48, v50, v49\n\t"
      "v_mul_f32 v50, v50, v49\n\t"
      "v_fmac_f32 v50, v51, v49\n\t"
      "v_mul_f32 v55, v52, v53\n\t"
      "v_fmac_f32 v55, v53, v54\n\t"
      "v_fmac_f32 v55, v52, v54\n\t"
      "v_mul_f32 v55, v55, v54\n\t"
      "v_mul_f32 v52, v52, v53\n\t"
      "v_fmac_f32 v52, v54, v53\n\t"
      "v_mul_f32 v54, v54, v53\n\t"
      "v_fmac_f32 v54, v55, v53\n\t"
      "s_sub_u32 s6, s6, 1\n\t"
      "s_cmp_lg_u32 s6, 0\n\t"
      "s_cbranch_scc1 1b\n\t"
      :: "s"(iters) : "s6", "scc", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void issue_mix_f64_kernel(int iters, uint64_t* cycles) {
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  asm volatile(
      "v_mov_b32 v24, 0\n\tv_mov_b32 v25, 0x3ff00000\n\tv_mov_b32 v26, 0\n\tv_mov_b32 v27, 0x3ff00000\n\tv_mov_b32 v28, 0\n\tv_mov_b32 v29, 0x3ff00000\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0x3ff00000\n\t"
      "v_mov_b32 v32, 0\n\tv_mov_b32 v33, 0x3ff00000\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v35, 0x3ff00000\n\tv_mov_b32 v36, 0\n\tv_mov_b32 v37, 0x3ff00000\n\tv_mov_b32 v38, 0\n\tv_mov_b32 v39, 0x3ff00000\n\t"
      "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0x3ff00000\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0x3ff00000\n\tv_mov_b32 v44, 0\n\tv_mov_b32 v45, 0x3ff00000\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0x3ff00000\n\t"
      "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0x3ff00000\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0x3ff00000\n\tv_mov_b32 v52, 0\n\tv_mov_b32 v53, 0x3ff00000\n\tv_mov_b32 v54, 0\n\tv_mov_b32 v55, 0x3ff00000\n\t"
      "v_mov_b32 v56, 0\n\tv_mov_b32 v57, 0x3ff00000\n\tv_mov_b32 v58, 0\n\tv_mov_b32 v59, 0x3ff00000\n\tv_mov_b32 v60, 0\n\tv_mov_b32 v61, 0x3ff00000\n\tv_mov_b32 v62, 0\n\tv_mov_b32 v63, 0x3ff00000\n\t"
      "v_mov_b32 v64, 0\n\tv_mov_b32 v65, 0x3ff00000\n\tv_mov_b32 v66, 0\n\tv_mov_b32 v67, 0x3ff00000\n\tv_mov_b32 v68, 0\n\tv_mov_b32 v69, 0x3ff00000\n\tv_mov_b32 v70, 0\n\tv_mov_b32 v71, 0x3ff00000\n\t"
      "v_mov_b32 v72, 0\n\tv_mov_b32 v73, 0x3ff00000\n\tv_mov_b32 v74, 0\n\tv_mov_b32 v75, 0x3ff00000\n\tv_mov_b32 v76, 0\n\tv_mov_b32 v77, 0x3ff00000\n\tv_mov_b32 v78, 0\n\tv_mov_b32 v79, 0x3ff00000\n\t"
      "v_mov_b32 v80, 0\n\tv_mov_b32 v81, 0x3ff00000\n\tv_mov_b32 v82, 0\n\tv_mov_b32 v83, 0x3ff00000\n\tv_mov_b32 v84, 0\n\tv_mov_b32 v85, 0x3ff00000\n\tv_mov_b32 v86, 0\n\tv_mov_b32 v87, 0x3ff00000\n\t"
      "s_mov_b32 s6, %0\n\t"
      "1:\n\t"
      "v_mul_f64 v[30:31], v[24:25], v[26:27]\n\t"
      "v_fma_f64 v[30:31], v[26:27], v[28:29], v[30:31]\n\t"
      "v_fma_f64 v[30:31], v[24:25], v[28:29], v[30:31]\n\t"
      "v_mul_f64 v[30:31], v[30:31], v[28:29]\n\t"
      "v_mul_f64 v[24:25], v[24:25], v[26:27]\n\t"
      "v_fma_f64 v[24:25], v[28:29], v[26:27], v[24:25]\n\t"
      "v_mul_f64 v[28:29], v[28:29], v[26:27]\n\t"
      "v_fma_f64 v[28:29], v[30:31], v[26:27], v[28:29]\n\t"
      "v_mul_f64 v[38:39], v[32:33], v[34:35]\n\t"
      "v_fma_f64 v[38:39], v[34:35], v[36:37], v[38:39]\n\t"
      "v_fma_f64 v[38:39], v[32:33], v[36:37], v[38:39]\n\t"
      "v_mul_f64 v[38:39], v[38:39], v[36:37]\n\t"
      "v_mul_f64 v[32:33], v[32:33], v[34:35]\n\t"
      "v_fma_f64 v[32:33], v[36:37], v[34:35], v[32:33]\n\t"
      "v_mul_f64 v[36:37], v[36:37], v[34:35]\n\t"
      "v_fma_f64 v[36:37], v[38:39], v[34:35], v[36:37]\n\t"
      "v_mul_f64 v[46:47], v[40:41], v[42:43]\n\t"
      "v_fma_f64 v[46:47], v[42:43], v[44:45], v[46:47]\n\t"
      "v_fma_f64 v[46:47], v[40:41], v[44:45], v[46:47]\n\t"
      "v_mul_f64 v[46:47], v[46:47], v[44:45]\n\t"
      "v_mul_f64 v[40:41], v[40:41], v[42:43]\n\t"
      "v_fma_f64 v[40:41], v[44:45], v[42:43], v[40:41]\n\t"
      "v_mul_f64 v[44:45], v[44:45], v[42:43]\n\t"
      "v_fma_f64 v[44:45], v[46:47], v[42:43], v[44:45]\n\t"
      "v_mul_f64 v[54:55], v[48:49], v[50:51]\n\t"
      "v_fma_f64 v[54:55], v[50:51], v[52:53], v[54:55]\n\t"
      "v_fma_f64 v[54:55], v[48:49], v[52:53], v[54:55]\n\t"
      "v_mul_f64 v[54:55], v[54:55], v[52:53]\n\t"
      "v_mul_f64 v[48:49], v[48:49], v[50:51]\n\t"
      "v_fma_f64 v[48:49], v[52:53], v[50:51], v[48:49]\n\t"
      "v_mul_f64 v[52:53], v[52:53], v[50:51]\n\t"
      "v_fma_f64 v[52:53], v[54:55], v[50:51], v[52:53]\n\t"
      "v_mul_f64 v[62:63], v[56:57], v[58:59]\n\t"
      "v_fma_f64 v[62:63], v[58:59], v[60:61], v[62:63]\n\t"
      "v_fma_f64 v[62:63], v[56:57], v[60:61], v[62:63]\n\t"
      "v_mul_f64 v[62:63], v[62:63], v[60:61]\n\t"
      "v_mul_f64 v[56:57], v[56:57], v[58:59]\n\t"
      "v_fma_f64 v[56:57], v[60:61], v[58:59], v[56:57]\n\t"
      "v_mul_f64 v[60:61], v[60:61], v[58:59]\n\t"
      "v_fma_f64 v[60:61], v[62:63], v[58:59], v[60:61]\n\t"
      "v_mul_f64 v[70:71], v[64:65], v[66:67]\n\t"
      "v_fma_f64 v[70:71], v[66:67], v[68:69], v[70:71]\n\t"
      "v_fma_f64 v[70:71], v[64:65], v[68:69], v[70:71]\n\t"
      "v_mul_f64 v[70:71], v[70:71], v[68:69]\n\t"
      "v_mul_f64 v[64:65], v[64:65], v[66:67]\n\t"
      "v_fma_f64 v[64:65], v[68:69], v[66:67], v[64:65]\n\t"
      "v_mul_f64 v[68:69], v[68:69], v[66:67]\n\t"
      "v_fma_f64 v[68:69], v[70:71], v[66:67], v[68:69]\n\t"
      "v_mul_f64 v[78:79], v[72:73], v[74:75]\n\t"
      "v_fma_f64 v[78:79], v[74:75], v[76:77], v[78:79]\n\t"
      "v_fma_f64 v[78:79], v[72:73], v[76:77], v[78:79]\n\t"
      "v_mul_f64 v[78:79], v[78:79], v[76:77]\n\t"
      "v_mul_f64 v[72:73], v[72:73], v[74:75]\n\t"
      "v_fma_f64 v[72:73], v[76:77], v[74:75], v[72:73]\n\t"
      "v_mul_f64 v[76:77], v[76:77], v[74:75]\n\t"
      "v_fma_f64 v[76:77], v[78:79], v[74:75], v[76:77]\n\t"
      "v_mul_f64 v[86:87], v[80:81], v[82:83]\n\t"
      "v_fma_f64 v[86:87], v[82:83], v[84:85], v[86:87]\n\t"
      "v_fma_f64 v[86:87], v[80:81], v[84:85], v[86:87]\n\t"
      "v_mul_f64 v[86:87], v[86:87], v[84:85]\n\t"
      "v_mul_f64 v[80:81], v[80:81], v[82:83]\n\t"
      "v_fma_f64 v[80:81], v[84:85], v[82:83], v[80:81]\n\t"
      "v_mul_f64 v[84:85], v[84:85], v[82:83]\n\t"
      "v_fma_f64 v[84:85], v[86:87], v[82:83], v[84:85]\n\t"
      "s_sub_u32 s6, s6, 1\n\t"
      "s_cmp_lg_u32 s6, 0\n\t"
      "s_cbranch_scc1 1b\n\t"
      :: "s"(iters) : "s6", "scc", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87");
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = t1 - t0;
}

}  // namespace gklhip
