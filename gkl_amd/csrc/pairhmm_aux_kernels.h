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

// Zeroes the small per-call arrays (counters, per-read fail counts, fail histogram, haplotype flags) in one launch;
// hipMemsetAsync blits cost a barrier bubble of ~50 us per step between back-to-back batches.
__global__ void clear_kernel(int32_t* a, int na, int32_t* b, int nb, int32_t* c, int nc, int32_t* d, int nd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < na) a[i] = 0;
  if (i < nb) b[i] = 0;
  if (i < nc) c[i] = 0;
  if (i < nd) d[i] = 0;
}

// stream_src (host plan) -> stream entries: haplotype base codes / separators / idle.
// A thread that meets an 'N' also flags its haplotype (hap_has_n, zeroed beforehand: the fp64 kernels route such
// haplotypes through the general step, see WaveJob::kCodes); it finds the haplotype by bisecting hap_pos.
__global__ void build_stream_kernel(const int32_t* __restrict__ src, const uint8_t* __restrict__ hap_bases,
                                    uint32_t* __restrict__ stream, int n, const int32_t* __restrict__ hap_pos,
                                    int n_haps, uint8_t* __restrict__ hap_has_n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t s = src[i];
  uint32_t e;
  if (s >= 0) {
    const uint8_t b = hap_bases[s];  // pairhmm_common.h:57-61: A0 C1 T2 G3 N4, anything else 0
    e = b == 'C' ? 1u : b == 'T' ? 2u : b == 'G' ? 3u : b == 'N' ? 4u : 0u;
    if (b == 'N') {
      int lo = 0, hi = n_haps - 1;  // largest stream-order haplotype whose first column is at or before i
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (hap_pos[mid] <= i) lo = mid; else hi = mid - 1;
      }
      hap_has_n[lo] = 1;
    }
  } else if (s == -1) {
    e = kEntIdle;
  } else {
    e = kEntSep | (uint32_t)(-2 - s);
  }
  stream[i] = e;
}

// Host finalisation (reference-exact log10f / log10 of the host libm) needs, per pair, either the raw
// fp32 sum or the raw fp64 sum: one 8-byte word carries both cases -- the double's bits, or
// 0xFFFFFFFF:float bits (a double whose high word is all ones is a NaN no computation here produces; if
// one ever does it is replaced by the default NaN, which finalises to NaN all the same).
constexpr int kModePacked = -2;
constexpr uint64_t kPackedF32Tag = 0xFFFFFFFF00000000ull;

struct FinalizeArgs {
  const float* raw32;
  const double* raw64;
  double* out;
  uint8_t* used64;
  int32_t* list;
  int32_t* count;
  int32_t* read_fail;  // [n_reads] number of haplotypes each read must be recomputed against
  int32_t n_haps;
  int64_t n;
  int mode;            // gklhip_finalize (device modes only), -1: no output, kModePacked: `out` = packed raw sums
  float log10_init_f;  // log10f(2^120), host libm
  double log10_init32_as_f64;  // log10(2^120) in double
  double log10_init_d;         // log10(2^1020)
};

// Precision policy of IntelPairHmm.cc:157-165 on the raw fp32 sums: keep (and
// finalise) pairs with sum >= 1e-28f, queue the rest for the fp64 kernel.
constexpr int kPolicyBlock = 1024;
__global__ __launch_bounds__(kPolicyBlock) void policy_kernel(FinalizeArgs a) {
  __shared__ int32_t wave_cnt[kPolicyBlock / 64], wave_base[kPolicyBlock / 64];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool in_range = i < a.n;
  const float v = in_range ? a.raw32[i] : 1.0f;
  const bool fails = in_range && v < 1e-28f;  // NaN compares false and stays fp32, like the reference
  // One atomic per BLOCK on the queue counter: atomics on a single address retire one per ~7 ns at the L2, and
  // one per wavefront (20 k of them for the bench batch) made this kernel take 150 us.
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  const uint64_t mask = __ballot(fails);
  if (lane == 0) wave_cnt[wave] = __builtin_popcountll(mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    int total = 0;
    for (int w = 0; w < kPolicyBlock / 64; w++) { wave_base[w] = total; total += wave_cnt[w]; }
    const int base = total ? atomicAdd(a.count, total) : 0;
    for (int w = 0; w < kPolicyBlock / 64; w++) wave_base[w] += base;
  }
  __syncthreads();
  if (mask) {
    const int leader = __builtin_ctzll(mask);
    const int32_t read = fails ? (int32_t)(i / a.n_haps) : -1;
    if (fails) a.list[wave_base[wave] + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (int32_t)i;
    // r-major pairs: the failing lanes of a wavefront usually belong to one read (n_haps >= 64) or a few
    const int32_t lead_read = __shfl(read, leader, 64);
    const uint64_t same = __ballot(fails && read == lead_read);
    if (lane == leader) atomicAdd(a.read_fail + lead_read, __builtin_popcountll(same));
    if (fails && read != lead_read) atomicAdd(a.read_fail + read, 1);
  }
  if (!in_range) return;
  if (fails) {
    a.used64[i] = 1;
    if (a.mode == kModePacked) reinterpret_cast<uint64_t*>(a.out)[i] = 0;  // "pending": filled in by finalize64_kernel
  } else {
    a.used64[i] = 0;
    // the device log10 of the kept pairs is finalize32_kernel's job: it runs beside the fp64 pass
    if (a.mode == kModePacked) reinterpret_cast<uint64_t*>(a.out)[i] = kPackedF32Tag | (uint64_t)__float_as_uint(v);
  }
}

// log10 of the fp32 sums the policy kept (device finalisation modes).  Launched on the context's side stream
// right after policy_kernel, so its ~1.3 M double-precision log10 overlap the planning kernels and the fp64 pass.
__global__ void finalize32_kernel(FinalizeArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n || a.used64[i]) return;
  const float v = a.raw32[i];
  if (a.mode == GKLHIP_FINALIZE_DEVICE_F64) a.out[i] = log10((double)v) - a.log10_init32_as_f64;
  else if (a.mode == GKLHIP_FINALIZE_DEVICE_REF32) a.out[i] = (double)((float)log10((double)v) - a.log10_init_f);
}

// log10 of the fp64 sums: all pairs (useDoublePrecision) or the queued ones.
__global__ void finalize64_kernel(FinalizeArgs a, int use_list) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = use_list ? (int64_t)*a.count : a.n;
  if (i >= n) return;
  const int64_t p = use_list ? (int64_t)a.list[i] : i;
  if (!use_list) a.used64[p] = 1;
  if (a.mode >= 0) a.out[p] = log10(a.raw64[p]) - a.log10_init_d;
  if (a.mode == kModePacked) {
    uint64_t bits = (uint64_t)__double_as_longlong(a.raw64[p]);
    if ((bits & kPackedF32Tag) == kPackedF32Tag) bits = 0x7FF8000000000000ull;
    reinterpret_cast<uint64_t*>(a.out)[p] = bits;
  }
}

// Packed fp64 fallback, step 2 (device): for every chunk of the second read packing, find the
// runs of consecutive haplotypes (stream order, never across a stream group) that at least one
// of its reads must be recomputed against, and queue one wave job per run.
__global__ void build_jobs_kernel(const LaneSlot* __restrict__ lanes, const int32_t* __restrict__ n_chunks,
                                  const uint8_t* __restrict__ used64, int n_haps,
                                  const int32_t* __restrict__ hap_orig, const int32_t* __restrict__ hap_group,
                                  FwdJob* __restrict__ jobs, int32_t* __restrict__ job_count) {
  extern __shared__ int32_t smem[];
  int32_t* s_reads = smem;                                   // [64] distinct reads of the chunk
  uint8_t* need = reinterpret_cast<uint8_t*>(smem + kLanes + 1);  // [n_haps]
  const int total = *n_chunks;
  for (int c = blockIdx.x; c < total; c += gridDim.x) {
  __syncthreads();
  if (threadIdx.x == 0) smem[kLanes] = 0;
  __syncthreads();
  if (threadIdx.x < kLanes) {
    const LaneSlot sl = lanes[(int64_t)c * kLanes + threadIdx.x];
    if (sl.read >= 0 && sl.block == 0) s_reads[atomicAdd(&smem[kLanes], 1)] = sl.read;
  }
  __syncthreads();
  const int nr = smem[kLanes];
  for (int k = threadIdx.x; k < n_haps; k += blockDim.x) {
    const int h = hap_orig[k];
    uint8_t nd = 0;
    for (int i = 0; i < nr; i++) nd |= used64[(int64_t)s_reads[i] * n_haps + h];
    need[k] = nd;
  }
  __syncthreads();
  // a needed haplotype starts a run if its predecessor is not needed or lies in another stream group
  for (int k = threadIdx.x; k < n_haps; k += blockDim.x) {
    if (!need[k]) continue;
    if (k > 0 && need[k - 1] && hap_group[k - 1] == hap_group[k]) continue;
    int e = k + 1;
    while (e < n_haps && need[e] && hap_group[e] == hap_group[k]) e++;
    FwdJob j;
    j.chunk = c; j.hap_begin = k; j.hap_end = e; j.pad_ = 0;
    jobs[atomicAdd(job_count, 1)] = j;
  }
  }  // chunk loop
}

// Packed fp64 fallback, step 3 (device): order the job list by decreasing length (counting sort on
// columns / 128, one block), so that the persistent wavefronts of the jobs kernel start the long runs
// first and the kernel's tail is made of short ones.
constexpr int kJobClasses = 64;
__global__ __launch_bounds__(1024) void sort_jobs_kernel(const FwdJob* __restrict__ jobs, const int32_t* __restrict__ job_count,
                                                         const int32_t* __restrict__ hap_pos,
                                                         const int32_t* __restrict__ hap_len, FwdJob* __restrict__ sorted) {
  __shared__ int32_t cnt[kJobClasses], base[kJobClasses];
  const int n = *job_count;
  if (threadIdx.x < kJobClasses) cnt[threadIdx.x] = 0;
  __syncthreads();
  auto cls_of = [&](const FwdJob& j) {
    const int cols = hap_pos[j.hap_end - 1] + hap_len[j.hap_end - 1] - hap_pos[j.hap_begin];
    const int c = cols >> 7;
    return kJobClasses - 1 - (c < kJobClasses - 1 ? c : kJobClasses - 1);  // class 0 = longest
  };
  for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&cnt[cls_of(jobs[i])], 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int c = 0; c < kJobClasses; c++) { base[c] = acc; acc += cnt[c]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const FwdJob j = jobs[i];
    sorted[atomicAdd(&base[cls_of(j)], 1)] = j;
  }
}

// ---- packed fp64 fallback, step 1 (device): order the affected reads by how many haplotypes
// they failed against (counting sort, most first) and pack them, window by window, into 64-lane
// chunks with best-fit-decreasing -- the device twin of pack_reads_windowed(), so the pass needs
// no host round trip.  Reads with similar fallback counts share chunks; in nested patterns (a
// read underflows against every haplotype shorter than some length) a chunk then needs one
// contiguous run of the length-sorted haplotype stream.
constexpr int kPackWindow = 96;

// (reads longer than max_len bases do not fit a chunk: they take the striped long-read path)
__global__ void fail_hist_kernel(const int32_t* __restrict__ read_fail, int n_reads, int32_t* __restrict__ hist,
                                 const int64_t* __restrict__ read_off, int max_len) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_reads && read_fail[r] > 0 && read_off[r + 1] - read_off[r] <= max_len) atomicAdd(hist + read_fail[r], 1);
}

// one block: bucket start positions for DESCENDING fail count; pos[c] = #reads with count > c
__global__ void fail_scan_kernel(const int32_t* __restrict__ hist, int n_haps, int32_t* __restrict__ pos,
                                 int32_t* __restrict__ n_fail_reads) {
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int c = n_haps; c >= 1; c--) { pos[c] = acc; acc += hist[c]; }
    *n_fail_reads = acc;
  }
}

__global__ void fail_scatter_kernel(const int32_t* __restrict__ read_fail, int n_reads, int32_t* __restrict__ pos,
                                    int32_t* __restrict__ order, const int64_t* __restrict__ read_off, int max_len) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_reads && read_fail[r] > 0 && read_off[r + 1] - read_off[r] <= max_len)
    order[atomicAdd(pos + read_fail[r], 1)] = r;
}

__global__ __launch_bounds__(64) void pack_windows_kernel(const int32_t* __restrict__ order,
                                                          const int32_t* __restrict__ n_fail_reads,
                                                          const int64_t* __restrict__ read_off, int rpl,
                                                          LaneSlot* __restrict__ lanes, int32_t* __restrict__ n_chunks) {
  // One wavefront per window of <= 96 reads; everything is wave-parallel: rank sort by lanes
  // needed (descending, stable), then best fit where the 64 lanes each watch up to two bins and
  // a shuffle reduction picks the fullest bin that still fits.
  __shared__ int32_t s_read[kPackWindow], s_need[kPackWindow], s_sread[kPackWindow], s_sneed[kPackWindow];
  __shared__ int32_t s_bin[kPackWindow], s_off[kPackWindow];
  const int lane = threadIdx.x;
  const int n = *n_fail_reads;
  const int w0 = blockIdx.x * kPackWindow;
  if (w0 >= n) return;
  const int cnt = min(kPackWindow, n - w0);
  for (int i = lane; i < cnt; i += kLanes) {
    const int r = order[w0 + i];
    s_read[i] = r;
    s_need[i] = (int)((read_off[r + 1] - read_off[r] + rpl) / rpl);  // blocks_for()
  }
  __syncthreads();
  for (int i = lane; i < cnt; i += kLanes) {
    const int ni = s_need[i];
    int rank = 0;
    for (int j = 0; j < cnt; j++) {
      const int nj = s_need[j];
      rank += (nj > ni) || (nj == ni && j < i);
    }
    s_sread[rank] = s_read[i];
    s_sneed[rank] = ni;
  }
  __syncthreads();
  int free0 = 0, free1 = 0;  // free lanes of bins `lane` and `lane + 64` (0 = bin not open)
  int nb = 0;                // bins opened so far (wave-uniform)
  for (int i = 0; i < cnt; i++) {
    const int nn = s_sneed[i];
    // key = free*256 + bin for bins that fit, smallest free wins (best fit); none -> large
    int key = 0x7fffffff;
    if (free0 >= nn) key = free0 * 256 + lane;
    if (free1 >= nn && free1 * 256 + lane + kLanes < key) key = free1 * 256 + lane + kLanes;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) key = min(key, __shfl_xor(key, d, kLanes));
    int bin, off;
    if (key == 0x7fffffff) { bin = nb++; off = 0; }
    else { bin = key & 255; off = kLanes - (key >> 8); }
    if (bin == lane) free0 = (key == 0x7fffffff ? kLanes : free0) - nn;
    if (bin == lane + kLanes) free1 = (key == 0x7fffffff ? kLanes : free1) - nn;
    if (lane == 0) { s_bin[i] = bin; s_off[i] = off; }
  }
  int base = 0;
  if (lane == 0) base = atomicAdd(n_chunks, nb);
  base = __shfl(base, 0, kLanes);
  __syncthreads();
  LaneSlot idle; idle.read = -1; idle.block = 0;
  for (int i = lane; i < nb * kLanes; i += kLanes) lanes[(int64_t)base * kLanes + i] = idle;
  __syncthreads();
  for (int i = 0; i < cnt; i++)
    for (int b = lane; b < s_sneed[i]; b += kLanes) {
      LaneSlot sl; sl.read = s_sread[i]; sl.block = b;
      lanes[(int64_t)(base + s_bin[i]) * kLanes + s_off[i] + b] = sl;
    }
}

}  // namespace gklhip
