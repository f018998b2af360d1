// One pass of the hot path on one device: table upload, plan block layout, kernel launch helpers, run_device (prep -> fp32 forward -> policy +
// device-side planning -> fp64 recomputation -> finalisation).
// Part of the ONE translation unit gkl_amd/csrc/pairhmm_api.hip (included there, in this order: pairhmm_ctx.h, pairhmm_device_pass.h,
// pairhmm_ctx_lifecycle.h, pairhmm_host_call.h, pairhmm_multi_device.h, pairhmm_diagnostics.h); not a stand-alone header.
#pragma once

namespace {

// The context's third stream, made on first use together with a padding stream (see dev_init: how many streams a process
// holds decides how the device's scheduler treats it next to other processes; two and four are good numbers, three is not).
int aux_streams(DevCtx* c) {
  if (!c->upload_stream) HIP_TRY(hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking));
  if (!c->copy_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->pad_stream) HIP_TRY(hipStreamCreateWithFlags(&c->pad_stream, hipStreamNonBlocking));
  }
  return GKLHIP_OK;
}

template <typename T>
int upload_tables(DevCtx* c, const HostTables<T>& h, DevBuf* buf, DevTables<T>* dt) {
  const size_t n = (size_t)kQuals * 2 + kMmEntries;
  int st = buf->reserve(n * sizeof(T));
  if (st) return st;
  T* base = buf->as<T>();
  // (on the context's own stream -- the null stream would be one more hardware queue per process -- and pulled by a kernel
  //  from a pinned block instead of copied: no copy-engine queue either)
  {
    T* pin = nullptr;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&pin), n * sizeof(T), hipHostMallocDefault));
    memcpy(pin, h.ph2pr.data(), kQuals * sizeof(T));
    memcpy(pin + kQuals, h.div3.data(), kQuals * sizeof(T));
    memcpy(pin + 2 * kQuals, h.mm.data(), kMmEntries * sizeof(T));
    void* pin_dev = nullptr;
    hipError_t e = hipHostGetDevicePointer(&pin_dev, pin, 0);
    if (e == hipSuccess) {
      static_assert(sizeof(T) % 4 == 0, "whole words");
      hipLaunchKernelGGL(pull_words_kernel, dim3(64), dim3(256), 0, c->stream, static_cast<const uint32_t*>(pin_dev),
                         reinterpret_cast<uint32_t*>(base), (int)(n * sizeof(T) / 4));
      e = hipGetLastError();
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    (void)hipHostFree(pin);
    HIP_TRY(e);
  }
  dt->ph2pr = base;
  dt->div3 = base + kQuals;
  dt->mm = base + 2 * kQuals;
  return GKLHIP_OK;
}

size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Layout of the per-call plan block (identical in pinned staging and on the device).  A small host-buffer call
// appends its six input arrays (`batch`: 5 read arrays at `batch_stride`, then the haplotype bases), so that plan
// and inputs travel in ONE copy.
struct PlanLayout {
  size_t place_chunk, place_lane, chunk_used, groups, hap_len, hap_pos, hap_pos_flat, hap_orig, hap_sidx, hap_group, hap_src, y0_32, y0_64, read_off, long_lanes, long_jobs, long_count,
      batch, batch_stride, stream, stream_flat, has_n, desc, total;
};
PlanLayout layout_for(const Plan& p, int n_reads, int n_haps, size_t n_long_lanes, size_t n_long_jobs, size_t inline_read_bytes,
                      size_t inline_hap_bytes) {
  PlanLayout l;
  size_t o = 0;
  l.place_chunk = o; o = align_up(o + (size_t)n_reads * 4);
  l.place_lane = o; o = align_up(o + (size_t)n_reads);
  l.chunk_used = o; o = align_up(o + (size_t)p.n_chunks);
  l.groups = o; o = align_up(o + p.groups.size() * sizeof(PlanGroup));
  l.hap_len = o; o = align_up(o + (size_t)n_haps * 4);
  l.hap_pos = o; o = align_up(o + (size_t)n_haps * 4);
  l.hap_pos_flat = o; o = align_up(o + (size_t)n_haps * 4);
  l.hap_orig = o; o = align_up(o + (size_t)n_haps * 4);
  l.hap_sidx = o; o = align_up(o + (size_t)n_haps * 4);
  l.hap_group = o; o = align_up(o + (size_t)n_haps * 4);
  l.hap_src = o; o = align_up(o + (size_t)n_haps * 4);
  l.y0_32 = o; o = align_up(o + (size_t)n_haps * 4);
  l.y0_64 = o; o = align_up(o + (size_t)n_haps * 8);
  l.read_off = o; o = align_up(o + (size_t)(n_reads + 1) * 8);
  l.long_lanes = o; o = align_up(o + n_long_lanes * sizeof(PlanLane));
  l.long_jobs = o; o = align_up(o + n_long_jobs * sizeof(FwdJob));
  l.long_count = o; o = align_up(o + 16);
  l.batch = o;
  l.batch_stride = align_up(inline_read_bytes);
  if (inline_read_bytes) o = o + 5 * l.batch_stride + align_up(inline_hap_bytes);
  // ... and, when the host holds the haplotype bases anyway, the two haplotype streams and the 'N' flags, built on
  // the host: the first kernel then only pulls the block (its wavefronts would otherwise chase three dependent reads
  // of pinned host memory per haplotype before the first forward kernel can start)
  l.stream = o; if (inline_read_bytes) o = align_up(o + (size_t)p.n_stream * 4);
  l.stream_flat = o; if (inline_read_bytes) o = align_up(o + (size_t)p.n_stream_flat * 4);
  l.has_n = o; if (inline_read_bytes) o = align_up(o + (size_t)n_haps);
  // ... and the call's descriptor for the combined launches of several small calls (SmallCombiner)
  l.desc = o; if (inline_read_bytes) o = align_up(o + sizeof(SmallCall));
  l.total = o;
  return l;
}

int validate(const gklhip_batch* b) {
  if (!b) return fail(GKLHIP_ERR_INVALID_ARG, "batch is NULL");
  if (b->n_reads < 0 || b->n_haps < 0) return fail(GKLHIP_ERR_INVALID_ARG, "negative batch size");
  if (b->n_reads == 0 || b->n_haps == 0) return GKLHIP_OK;
  if (!b->read_off || !b->hap_off) return fail(GKLHIP_ERR_INVALID_ARG, "offset arrays are NULL");
  if (!b->read_bases || !b->read_quals || !b->ins_gop || !b->del_gop || !b->gcp || !b->hap_bases)
    return fail(GKLHIP_ERR_INVALID_ARG, "a batch byte array is NULL");
  if (b->read_off[0] != 0 || b->hap_off[0] != 0)
    return fail(GKLHIP_ERR_INVALID_ARG, "offset arrays must start at 0");
  // The reference does not guard empty reads/haplotypes (division by zero / negative index,
  // SURVEY appendix A.10); this boundary rejects them.
  for (int r = 0; r < b->n_reads; r++)
    if (b->read_off[r + 1] <= b->read_off[r])
      return fail(GKLHIP_ERR_INVALID_ARG, "read %d is empty or offsets are not increasing", r);
  for (int h = 0; h < b->n_haps; h++)
    if (b->hap_off[h + 1] <= b->hap_off[h])
      return fail(GKLHIP_ERR_INVALID_ARG, "haplotype %d is empty or offsets are not increasing", h);
  if ((int64_t)b->n_reads * b->n_haps >= (int64_t)1 << 31)
    return fail(GKLHIP_ERR_UNSUPPORTED, "more than 2^31 pairs in one call");
  if (b->hap_off[b->n_haps] + b->n_haps + 4096 >= (int64_t)1 << 31)
    return fail(GKLHIP_ERR_UNSUPPORTED, "haplotype bases exceed 2^31");
  return GKLHIP_OK;
}

template <typename T, int RPL>
void launch_stream(const FwdArgs<T>& a, int fma, int n_blocks, hipStream_t s) {
  if (fma) hipLaunchKernelGGL((pairhmm_fwd_stream_kernel<T, RPL, true>), dim3(n_blocks), dim3(64), 0, s, a);
  else     hipLaunchKernelGGL((pairhmm_fwd_stream_kernel<T, RPL, false>), dim3(n_blocks), dim3(64), 0, s, a);
}
template <typename T, int RPL>
void launch_jobs(const FwdArgs<T>& a, int fma, int n_blocks, hipStream_t s) {
  if (fma) hipLaunchKernelGGL((pairhmm_fwd_jobs_kernel<T, RPL, true>), dim3(n_blocks), dim3(64), 0, s, a);
  else     hipLaunchKernelGGL((pairhmm_fwd_jobs_kernel<T, RPL, false>), dim3(n_blocks), dim3(64), 0, s, a);
}

template <typename T, int RPL>
void launch_long(const FwdArgs<T>& a, int fma, int n_blocks, T* carry, int carry_len, hipStream_t s) {
  if (fma) hipLaunchKernelGGL((pairhmm_fwd_long_kernel<T, RPL, true>), dim3(n_blocks), dim3(64), 0, s, a, carry, carry_len);
  else     hipLaunchKernelGGL((pairhmm_fwd_long_kernel<T, RPL, false>), dim3(n_blocks), dim3(64), 0, s, a, carry, carry_len);
}

// long reads: workgroups of kWideWaves wavefronts per (read, haplotype run) job (pairhmm_fwd_wide_kernel: the asm programs'
// arithmetic only); the unfused arithmetic keeps the one-wavefront striped kernel
// compute wavefronts of a super-stripe workgroup (+ 1 helper): 7 + 1 in both precisions = 512 threads, two wavefronts per
// SIMD (amdgpu_waves_per_eu(4): 128 VGPRs) -- the first version's 5 + 1 put four compute wavefronts on SIMD 0 and two on SIMDs
// 2, 3, and a pipeline of wavefronts advances at the speed of its slowest (docs/NOTES.md 44).  LDS: ~78 KB per fp32
// workgroup (two fit a CU's 160 KB), fp64 with its 16 KB tables one per CU; super_blocks_max = the workgroups that can be
// resident at once (256 CUs x 2 / x 1): the carry rows (two per workgroup) are sized for that many.
template <typename T> constexpr int super_waves() { return 7; }
template <typename T> constexpr int super_blocks_max() { return sizeof(T) == 8 ? 256 : 512; }
// carry rows of the super-stripe kernel: two per workgroup, one 32-byte slot per step of the deepest array's longest stream
inline int64_t super_steps(int carry_len, int max_read_len, int rpl) { return (int64_t)carry_len + 64 * (int64_t)((blocks_for(max_read_len, rpl) + kLanes - 1) / kLanes); }
template <typename T, int RPL, int RPL_STRIPED>
void launch_long_jobs(const FwdArgs<T>& a, int fma, int n_blocks, int max_read_len, T* carry, int carry_len, hipStream_t s,
                      unsigned char* xcarry = nullptr, int64_t xsteps = 0, int32_t* next2 = nullptr) {
  const bool wide_env = g_env.wide_long, super_env = g_env.super_long;
  if (!wide_env) { launch_long<T, RPL_STRIPED>(a, fma, n_blocks, carry, carry_len, s); return; }
  // a read that needs more wavefronts than a wide workgroup holds: super-stripes of super_waves<T>() wavefronts, the carry row through HBM
  if (super_env && xcarry && next2 && (blocks_for(max_read_len, RPL) + kLanes - 1) / kLanes > kWideWavesMax) {
    static_assert(RPL == kRplSuper, "the super-stripe kernel's array depth");
    FwdArgs<T> sa = a;
    sa.super_steps = xsteps;
    if (fma) hipLaunchKernelGGL((pairhmm_fwd_super_kernel<T, RPL, super_waves<T>(), true>), dim3(std::min(n_blocks, super_blocks_max<T>())), dim3(64 * (super_waves<T>() + 1)), 0, s, sa, xcarry);
    else     hipLaunchKernelGGL((pairhmm_fwd_super_kernel<T, RPL, super_waves<T>(), false>), dim3(std::min(n_blocks, super_blocks_max<T>())), dim3(64 * (super_waves<T>() + 1)), 0, s, sa, xcarry);
    // ... and the jobs it leaves (a haplotype no longer than a wavefront is deep, fp64: an N haplotype): one-wavefront stripes
    sa.long_filter = 2;
    sa.job_next = next2;
    launch_long<T, RPL_STRIPED>(sa, fma, n_blocks, carry, carry_len, s);
    return;
  }
  // wavefronts per workgroup: what the call's longest read needs, at most kWideWavesMax (longer reads are striped in-kernel)
  const int waves = std::max(2, std::min(kWideWavesMax, (blocks_for(max_read_len, RPL) + kLanes - 1) / kLanes));
  if (fma) {
    if (waves == 2)      hipLaunchKernelGGL((pairhmm_fwd_wide_kernel<T, RPL, true, 2>), dim3(n_blocks), dim3(128), 0, s, a, carry, carry_len);
    else if (waves == 3) hipLaunchKernelGGL((pairhmm_fwd_wide_kernel<T, RPL, true, 3>), dim3(n_blocks), dim3(192), 0, s, a, carry, carry_len);
    else                 hipLaunchKernelGGL((pairhmm_fwd_wide_kernel<T, RPL, true, 4>), dim3(n_blocks), dim3(256), 0, s, a, carry, carry_len);
  } else {   // the unfused arithmetic (fma_mode 0): the same kernels over the "...n" programs
    if (waves == 2)      hipLaunchKernelGGL((pairhmm_fwd_wide_kernel<T, RPL, false, 2>), dim3(n_blocks), dim3(128), 0, s, a, carry, carry_len);
    else if (waves == 3) hipLaunchKernelGGL((pairhmm_fwd_wide_kernel<T, RPL, false, 3>), dim3(n_blocks), dim3(192), 0, s, a, carry, carry_len);
    else                 hipLaunchKernelGGL((pairhmm_fwd_wide_kernel<T, RPL, false, 4>), dim3(n_blocks), dim3(256), 0, s, a, carry, carry_len);
  }
}

// A small host-buffer call, planned and staged but not launched: SmallCombiner decides how it reaches the device (on
// its own, or in one set of launches with the calls of other threads).
struct SmallLaunch {
  bool filled = false;
  SmallCall call;                          // the descriptor
  const SmallCall* desc_pinned = nullptr;  // ... as the device sees it in the pinned staging block (the prep kernel reads this one)
  const SmallCall* desc_dev = nullptr;     // ... in the device copy of the plan block (which the prep kernel pulls)
};

// Rows per lane.  fp32 main pass: 8 (4 or 2 for small batches).  fp64: 10 in the streaming and job-list kernels, 6 (kRplF64) in the
// one-pair-per-wavefront and striped long-read kernels.  A read of length R needs R+1 rows; reads that exceed 64*RPL rows go to the
// striped long-read kernel.
#ifndef GKL_RPL_F64
#define GKL_RPL_F64 6
#endif
constexpr int kRplF64 = GKL_RPL_F64;
// The streaming and job-list fp64 kernels run two wavefronts per SIMD (256 VGPRs, 8 x 20 KB of LDS): 10 rows per lane
// (20 spilled registers, none in the unrolled loop) -- fewer hand-offs per cell and shorter general-step windows than 6
// or 8: the packed fp64 pass of the precision policy takes 3.23 (6 rows) / 2.91 (8) / 2.77 ms (10), the all-fp64 mode
// 18.2 / 17.8 / 17.1 ms (A/B on one box; 12 rows would leave LDS for three wavefronts per CU pair only).  kRplF64 (6)
// remains the row count of the one-pair-per-wavefront kernel (three wavefronts per SIMD) and of the striped long-read kernel.
#ifndef GKL_RPL_F64_JOBS
#define GKL_RPL_F64_JOBS 10
#endif
constexpr int kRplF64Jobs = GKL_RPL_F64_JOBS;
// The wide long-read kernel (several wavefronts of a workgroup per read) runs fp64 at 8 rows per lane: 16 KB of prior planes
// per wavefront instead of 20 -- four / three / two workgroups per CU at two / three / four wavefronts each instead of three / two / one.
constexpr int kRplF64Wide = 8;
constexpr size_t kSmallBatchBytes = 1 << 20;  // host-buffer calls up to this size send their inputs inside the plan block
constexpr int64_t kDirectPairs = 65536;        // calls up to this many pairs: policy + fp64 recomputation of one pair per wavefront (host calls of 24k / 38k / 50k pairs: 0.64 / 0.74 / 0.96 ms against 0.81 / 0.82 / 1.09 through the planned fp64 pass; equal at 80k)
constexpr int64_t kTwoStepFrom = 2048;         // ... from this many pairs in two launches: policy + list of the failing pairs, then their recomputation (10k / 16k / 32k pairs: 0.37 / 0.45-0.48 / 0.72-0.84 ms against 0.43 / 0.49-0.54 / 0.76-0.97 in one)
constexpr int kPlanBlocks = 64;                // 1024-thread blocks of the packing / run-detection launches of the fp64 plan
constexpr int kFallbackWantedJobs = 12288;     // the packed fp64 pass is cut into about this many jobs (4 per wavefront slot)
constexpr int64_t kHostShardPairs = 400000;    // single-device host-buffer calls from this many pairs run as two half-batches (see gklhip_ctx::host_dev)
constexpr int64_t kOnePassPairs = 65536;      // host-buffer calls up to this many pairs finalise in one pass after the last kernel
constexpr int kTargetCols = 2048;  // columns of a full-size haplotype group (sweep 1024..4096: flat within 2 %, optimum 1800..2600)
#ifndef GKL_RPL_F32
#define GKL_RPL_F32 8
#endif
constexpr int kRplF32 = GKL_RPL_F32;
std::atomic<int> g_host_calls_in_flight{0};  // host-buffer calls inside the library right now, process-wide
// fp32 main pass: which kernel.  rows_per_lane of the config: 0 = choose, 8 / 4 / 2 = that many rows per lane.
// Choosing: a small batch (one GATK active region) gives the 8-row kernel fewer jobs than the chip has wavefront
// slots worth filling (< 2 per SIMD), and a lone wavefront issues one instruction per ~6 cycles; fewer rows per
// lane mean more chunks and a shorter step (2 rows: reads of up to 127 bases).  `load`: small host calls in flight in
// this process -- they share the chip (and leave in combined launches, SmallCombiner), so their jobs count together
// and the wider, cheaper-per-cell kernels pay from fewer jobs per call.
int pick_f32_rpl(int forced, int n_reads, int n_haps, const int64_t* read_off, const int64_t* hap_off, int load = 1) {
  int max_len = 0;
  for (int r = 0; r < n_reads; r++) max_len = std::max(max_len, (int)(read_off[r + 1] - read_off[r]));
  if (forced == 2 && max_len <= 2 * kLanes - 1) return 2;
  if (forced == 4 || forced == -4 || forced == 2) return 4;
  if (forced == 8) return kRplF32;
  const int64_t total_cols = hap_off[n_haps] + n_haps;
  auto jobs_at = [&](int rpl) {
    int64_t blocks = 0;
    for (int r = 0; r < n_reads; r++) {
      const int nb = blocks_for((int)(read_off[r + 1] - read_off[r]), rpl);
      if (nb <= kLanes) blocks += nb;
    }
    const int64_t chunks = std::max<int64_t>(1, (blocks + kLanes - 1) / kLanes);
    const int64_t groups = std::min<int64_t>(n_haps, std::max<int64_t>((total_cols + kTargetCols - 1) / kTargetCols,
                                                                        (4096 + chunks - 1) / chunks));
    return chunks * groups;
  };
  // (a read of 256 bases or more does not fit 64 lanes x 4 rows: it would take the striped long-read kernel)
  if (jobs_at(kRplF32) * load >= 2048 || max_len > 4 * kLanes - 1) return kRplF32;
  if (jobs_at(4) * load >= 1024 || max_len > 2 * kLanes - 1) return 4;
  return 2;
}

// the per-pair policy of a mid-size call in two launches (pairhmm_pair_flag_kernel): `list` holds n_pairs entries
void launch_pair_policy_two_step(const FwdArgs<double>& d, const PairPolicyArgs& q, int rows, int fma, int64_t n_pairs, int32_t* list, hipStream_t s) {
  hipLaunchKernelGGL(pairhmm_pair_flag_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, s, q, (int32_t)n_pairs, list);
  const dim3 grid((unsigned)std::max<int64_t>(256, n_pairs / 2)), block(64);
  if (fma) {
    if (rows == 2)      hipLaunchKernelGGL((pairhmm_pair_recompute_kernel<2, true>), grid, block, 0, s, d, q, list);
    else if (rows == 4) hipLaunchKernelGGL((pairhmm_pair_recompute_kernel<4, true>), grid, block, 0, s, d, q, list);
    else                hipLaunchKernelGGL((pairhmm_pair_recompute_kernel<kRplF64, true>), grid, block, 0, s, d, q, list);
  } else {
    if (rows == 2)      hipLaunchKernelGGL((pairhmm_pair_recompute_kernel<2, false>), grid, block, 0, s, d, q, list);
    else if (rows == 4) hipLaunchKernelGGL((pairhmm_pair_recompute_kernel<4, false>), grid, block, 0, s, d, q, list);
    else                hipLaunchKernelGGL((pairhmm_pair_recompute_kernel<kRplF64, false>), grid, block, 0, s, d, q, list);
  }
}
void launch_main_f32(const FwdArgs<float>& a, int rpl_main, int fma, int n_blocks, hipStream_t s) {
  if (rpl_main == 2)      launch_stream<float, 2>(a, fma, n_blocks, s);
  else if (rpl_main == 4) launch_stream<float, 4>(a, fma, n_blocks, s);
  else                    launch_stream<float, kRplF32>(a, fma, n_blocks, s);
}
void launch_pair_policy(const FwdArgs<double>& d, const PairPolicyArgs& q, int rows, int fma, int64_t n_pairs, hipStream_t s) {
  const dim3 grid((unsigned)n_pairs), block(64);
  if (fma) {
    if (rows == 2)      hipLaunchKernelGGL((pairhmm_pair_policy_kernel<2, true>), grid, block, 0, s, d, q);
    else if (rows == 4) hipLaunchKernelGGL((pairhmm_pair_policy_kernel<4, true>), grid, block, 0, s, d, q);
    else                hipLaunchKernelGGL((pairhmm_pair_policy_kernel<kRplF64, true>), grid, block, 0, s, d, q);
  } else {
    if (rows == 2)      hipLaunchKernelGGL((pairhmm_pair_policy_kernel<2, false>), grid, block, 0, s, d, q);
    else if (rows == 4) hipLaunchKernelGGL((pairhmm_pair_policy_kernel<4, false>), grid, block, 0, s, d, q);
    else                hipLaunchKernelGGL((pairhmm_pair_policy_kernel<kRplF64, false>), grid, block, 0, s, d, q);
  }
}

// tiny calls: fp32 + policy + fp64 of ONE pair per wavefront in one launch (pairhmm_pair_fused_kernel)
// `alone`: nothing else is on the device -- the fp64 recomputation of every pair runs beside its fp32 recurrence
// (pairhmm_pair_spec_kernel) and the call takes max(fp32, fp64) instead of fp32 + fp64
void launch_pair_fused(const FwdArgs<float>& f, const FwdArgs<double>& d, const PairPolicyArgs& q, int rows, int fma, int64_t n_pairs,
                       hipStream_t s, bool speculate = false) {
  // Opt-in (GKLHIP_SPECULATE_FP64=1, read when the context is made, and only for a call that is alone on the device): it pays when a good share of the pairs fails the policy (100 x 10 with
  // 16 % failing: 0.151 -> 0.130 ms per call) and costs when none does (0.100 -> 0.130: the fp64 wavefront of a pair takes
  // twice as long as its fp32 one) -- and real active regions are mostly of the second kind.
  if (speculate) {
    const dim3 grid((unsigned)n_pairs), block(128);
    if (fma) hipLaunchKernelGGL((pairhmm_pair_spec_kernel<kRplF64, true>), grid, block, 0, s, f, d, q);
    else     hipLaunchKernelGGL((pairhmm_pair_spec_kernel<kRplF64, false>), grid, block, 0, s, f, d, q);
    return;
  }
  const dim3 grid((unsigned)n_pairs), block(64);
  if (rows <= 4) {   // every read of the call has at most 255 bases: the four-wavefronts-per-SIMD variant
    if (fma) hipLaunchKernelGGL((pairhmm_pair_fused_kernel<4, true>), grid, block, 0, s, f, d, q);
    else     hipLaunchKernelGGL((pairhmm_pair_fused_kernel<4, false>), grid, block, 0, s, f, d, q);
    return;
  }
  if (fma) hipLaunchKernelGGL((pairhmm_pair_fused_kernel<kRplF64, true>), grid, block, 0, s, f, d, q);
  else     hipLaunchKernelGGL((pairhmm_pair_fused_kernel<kRplF64, false>), grid, block, 0, s, f, d, q);
}

// The whole device-side pipeline on stream `s`: 7 launches in the policy mode (prep, fp32 forward, the three launches of
// policy + planning of the fp64 pass, fp64 forward over the job list, log10 / packed words of the recomputed pairs; + the
// log10 of the kept pairs on a side stream in the device finalisation modes), 3-4 for calls of up to 65 536 pairs (prep,
// fp32 forward, per-pair policy in one or two launches), 2 for up to 2048 (prep, the fused per-pair kernel); no host
// synchronisation.  `db` holds host offsets and DEVICE byte
// arrays -- or, with `inline_host`, HOST byte arrays that travel inside the plan block (small host-buffer calls: one
// copy for plan and inputs).
// `defer` (host-buffer calls on an idle context only): a call that takes the small-call path -- pulled plan block,
// per-pair policy -- is planned and staged but NOT launched; its descriptor is returned in *defer (filled = true).
int run_device(DevCtx* c, const gklhip_batch* db, double* out_dev, int finalize_mode, hipStream_t s, bool inline_host,
               SmallLaunch* defer = nullptr) {
  const int n_reads = db->n_reads, n_haps = db->n_haps;
  const int64_t n_pairs = (int64_t)n_reads * n_haps;
  gklhip_stats& st = c->stats;
  memset(&st, 0, sizeof st);
  st.n_pairs = n_pairs;
  c->have_last = false;
  if (n_pairs == 0) return GKLHIP_OK;
  const bool use_double = c->cfg.use_double != 0;
  const int fma = c->cfg.fma_mode != 0;

  // ---- plan (host) ----
  const auto t_plan0 = std::chrono::steady_clock::now();
  Plan& plan = c->plan;
  const int rpl64 = kRplF64Jobs;
  const int load_env = g_env.combine_load;
  // (about half of the calls inside the library are on the device at any moment, the others are being staged or
  //  finalised: 16 callers of 100 x 10 regions keep the 4-row kernel -- the 8-row one needs three wavefronts per SIMD
  //  to pay, tools/small_scaling.py -- and 32 callers get the 8-row one)
  const int load = !defer ? 1 : load_env > 0 ? load_env : std::max(1, g_host_calls_in_flight.load(std::memory_order_relaxed) / 2);
  const int rpl_main = use_double ? rpl64 : pick_f32_rpl(c->cfg.rows_per_lane, n_reads, n_haps, db->read_off, db->hap_off, load);
  const int target_cols_env = g_env.target_cols;
  build_plan(n_reads, n_haps, db->read_off, db->hap_off, rpl_main, target_cols_env > 0 ? target_cols_env : kTargetCols, &plan);
  // Long reads: pseudo-chunks (lane 0 names the read) + one striped job per (read, stream group)
  // for the main pass; for the fp64 fallback the same pseudo-chunks feed the run detection.
  std::vector<PlanLane>& long_lanes = c->long_lanes;
  std::vector<FwdJob>& long_jobs = c->long_jobs;
  long_lanes.clear(); long_jobs.clear();
  const std::vector<int32_t>& long_main = plan.long_reads;
  const int n_long_main = (int)long_main.size();
  // pseudo-chunk index space: [0, n_long_main) main-pass reads, then [n_long_main, +n_long64) fp64-pass reads
  for (int32_t r : long_main) { long_lanes.resize(long_lanes.size() + kLanes, PlanLane{-1, 0}); long_lanes[long_lanes.size() - kLanes] = PlanLane{r, 0}; }
  int n_long64 = 0;  // reads too long for the packed fp64 pass
  if (!use_double && plan.max_read_len > kLanes * kRplF64Jobs - 1)
    for (int r = 0; r < n_reads; r++)
      if (blocks_for((int)(db->read_off[r + 1] - db->read_off[r]), kRplF64Jobs) > kLanes) {
        long_lanes.resize(long_lanes.size() + kLanes, PlanLane{-1, 0});
        long_lanes[long_lanes.size() - kLanes] = PlanLane{r, 0};
        n_long64++;
      }
  for (int i = 0; i < n_long_main; i++)
    for (const PlanGroup& g : plan.groups) long_jobs.push_back(FwdJob{i, g.hap_begin, g.hap_end, 0});
  int carry_len = 0;
  for (const PlanGroup& g : plan.groups) {
    const int last = g.hap_end - 1;
    carry_len = std::max(carry_len, plan.hap_pos[last] + plan.hap_len[last] - plan.hap_pos[g.hap_begin] + 3 * kLanes);
  }
  carry_len = (carry_len + 63) / 64 * 64;
  const size_t rl = (size_t)db->read_off[n_reads], hl = (size_t)db->hap_off[n_haps];
  const PlanLayout L = layout_for(plan, n_reads, n_haps, long_lanes.size(), long_jobs.size(), inline_host ? rl : 0, inline_host ? hl : 0);

  // ---- stage + upload plan ----
  int rc;
  const int slot = c->plan_slot ^= 1;
  PinBuf& stage = c->stage_slot[slot];
  DevBuf& plan_dev = c->plan_dev_slot[slot];
  HIP_TRY(hipEventSynchronize(c->stage_free_slot[slot]));
  if (L.total > stage.cap || L.total > plan_dev.cap) HIP_TRY(hipEventSynchronize(c->plan_unused_slot[slot]));  // about to reallocate
  if ((rc = stage.reserve(L.total))) return rc;
  if ((rc = plan_dev.reserve(L.total))) return rc;
  unsigned char* hs = stage.as<unsigned char>();
  memcpy(hs + L.place_chunk, plan.place_chunk.data(), (size_t)n_reads * 4);
  memcpy(hs + L.place_lane, plan.place_lane.data(), (size_t)n_reads);
  memcpy(hs + L.chunk_used, plan.chunk_used.data(), (size_t)plan.n_chunks);
  memcpy(hs + L.groups, plan.groups.data(), plan.groups.size() * sizeof(PlanGroup));
  memcpy(hs + L.hap_len, plan.hap_len.data(), (size_t)n_haps * 4);
  memcpy(hs + L.hap_pos, plan.hap_pos.data(), (size_t)n_haps * 4);
  memcpy(hs + L.hap_pos_flat, plan.hap_pos_flat.data(), (size_t)n_haps * 4);
  memcpy(hs + L.hap_orig, plan.hap_orig.data(), (size_t)n_haps * 4);
  memcpy(hs + L.hap_sidx, plan.hap_sidx.data(), (size_t)n_haps * 4);
  memcpy(hs + L.hap_group, plan.hap_group.data(), (size_t)n_haps * 4);
  memcpy(hs + L.hap_src, plan.hap_src.data(), (size_t)n_haps * 4);
  {
    // Y[0][j] = INITIAL_CONSTANT / (NUMBER)haplen, divided on the host (template.h:110,176)
    float* y32 = reinterpret_cast<float*>(hs + L.y0_32);
    double* y64 = reinterpret_cast<double*>(hs + L.y0_64);
    const float i32 = host_tables_f32().initial_constant;
    const double i64 = host_tables_f64().initial_constant;
    for (int k = 0; k < n_haps; k++) {
      y32[k] = i32 / (float)plan.hap_len[k];
      y64[k] = i64 / (double)plan.hap_len[k];
    }
  }
  memcpy(hs + L.read_off, db->read_off, (size_t)(n_reads + 1) * 8);
  if (!long_lanes.empty()) memcpy(hs + L.long_lanes, long_lanes.data(), long_lanes.size() * sizeof(PlanLane));
  if (!long_jobs.empty()) memcpy(hs + L.long_jobs, long_jobs.data(), long_jobs.size() * sizeof(FwdJob));
  {
    int32_t lc[4] = {(int32_t)long_jobs.size(), n_long_main, n_long64, 0};
    memcpy(hs + L.long_count, lc, sizeof lc);
  }
  unsigned char* dp = plan_dev.as<unsigned char>();
  gklhip_batch dbi = *db;  // device pointers of the six byte arrays
  if (inline_host) {
    const uint8_t* srcs[5] = {db->read_bases, db->read_quals, db->ins_gop, db->del_gop, db->gcp};
    for (int i = 0; i < 5; i++) memcpy(hs + L.batch + i * L.batch_stride, srcs[i], rl);
    memcpy(hs + L.batch + 5 * L.batch_stride, db->hap_bases, hl);
    unsigned char* d = dp + L.batch;
    dbi.read_bases = d; dbi.read_quals = d + L.batch_stride; dbi.ins_gop = d + 2 * L.batch_stride;
    dbi.del_gop = d + 3 * L.batch_stride; dbi.gcp = d + 4 * L.batch_stride; dbi.hap_bases = d + 5 * L.batch_stride;
    // the haplotype streams (what prep_kernel builds on the device for resident batches)
    uint32_t* sg = reinterpret_cast<uint32_t*>(hs + L.stream);
    uint32_t* sf = reinterpret_cast<uint32_t*>(hs + L.stream_flat);
    uint8_t* hn = hs + L.has_n;
    for (int k = 0; k < n_haps; k++) {
      const uint8_t* src = db->hap_bases + plan.hap_src[k];
      const int len = plan.hap_len[k], pg = plan.hap_pos[k], pf = plan.hap_pos_flat[k];
      bool has_n = false;
      for (int col = 0; col < len; col++) {
        const uint8_t bb = src[col];  // pairhmm_common.h:57-61: A0 C1 T2 G3 N4, anything else 0
        const uint32_t e = bb == 'C' ? 1u : bb == 'T' ? 2u : bb == 'G' ? 3u : bb == 'N' ? 4u : 0u;
        sg[pg + col] = e; sf[pf + col] = e;
        has_n |= bb == 'N';
      }
      sg[pg + len] = kEntSep | (uint32_t)k;
      sf[pf + len] = kEntSep | (uint32_t)k;
      hn[k] = has_n ? 1 : 0;
      if (k + 1 == n_haps || plan.hap_group[k + 1] != plan.hap_group[k])
        for (int i = 0; i < kLanes; i++) sg[pg + len + 1 + i] = kEntIdle;
      if (k + 1 == n_haps)
        for (int i = 0; i < kLanes; i++) sf[pf + len + 1 + i] = kEntIdle;
    }
  }
  // scratch is shared by the calls of a context: one on another stream than the last one waits for that one's end
  if (c->have_call_done && c->last_stream != s) HIP_TRY(hipStreamWaitEvent(s, c->call_done, 0));
  // Big plans ride the upload stream (the copy overlaps the previous call's kernels); a small plan (GATK-sized
  // call) is PULLED from the pinned staging block by the prep kernel itself: no copy-engine hop at all.
  const bool pull = L.total < (1u << 20);  // (256 KB .. 2 MB measure within 2 % on calls of 4k-50k pairs, 1 MB best)
  // (the one-pair-per-wavefront policy kernel holds at most 64 x kRplF64 - 1 rows)
  const bool per_pair_call = !use_double && n_pairs <= kDirectPairs && n_long64 == 0 && plan.max_read_len <= kLanes * kRplF64 - 1;
  // ... the tiny ones (one GATK active region) with the fp32 recurrence in the same wavefront and launch as the policy
  const bool fused_env = g_env.fused_pairs;
  const int64_t fused_max = g_env.fused_max >= 0 ? g_env.fused_max : (int64_t)kTwoStepFrom;
  const bool fused_call = per_pair_call && fused_env && n_pairs <= fused_max && n_long_main == 0 && c->cfg.rows_per_lane == 0;
  const bool deferred_launch = defer && pull && inline_host && c->cfg.record_events == 0 && per_pair_call && n_long_main == 0 &&
                               finalize_mode == kModePacked && plan.n_chunks > 0 && n_pairs <= kTwoStepFrom;
  const unsigned char* hs_dev = nullptr;  // the staging block as the device sees it
  if (pull) {
    void* p = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&p, hs, 0));
    hs_dev = static_cast<const unsigned char*>(p);
    if (!deferred_launch) HIP_TRY(hipStreamWaitEvent(s, c->plan_unused_slot[slot], 0));
  } else {
    if ((rc = aux_streams(c))) return rc;
    HIP_TRY(hipStreamWaitEvent(c->upload_stream, c->plan_unused_slot[slot], 0));  // readers of the old contents are done
    HIP_TRY(hipMemcpyAsync(dp, hs, L.total, hipMemcpyHostToDevice, c->upload_stream));
    HIP_TRY(hipEventRecord(c->stage_free_slot[slot], c->upload_stream));
    HIP_TRY(hipStreamWaitEvent(s, c->stage_free_slot[slot], 0));                    // kernels below read the new plan
  }
  const bool timing = g_env.timing;
  if (timing)
    fprintf(stderr, "[gklhip] host plan + staging: %.3f ms (%d chunks, %d stream entries, %zu plan bytes)\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_plan0).count(),
            plan.n_chunks, plan.n_stream, L.total);

  // ---- scratch ----
  if ((rc = c->raw32.reserve((size_t)n_pairs * 4))) return rc;
  if ((rc = c->raw64.reserve((size_t)n_pairs * 8))) return rc;
  if ((rc = c->used64.reserve((size_t)n_pairs))) return rc;
  if ((rc = c->counters.reserve(128))) return rc;
  if ((rc = c->read_fail.reserve((size_t)n_reads * 4))) return rc;
  if ((rc = c->stream_buf.reserve(((size_t)plan.n_stream + (size_t)plan.n_stream_flat) * 4))) return rc;
  const int n_hist = use_double ? 0 : 2 * (n_haps + 2);
  if (!use_double && (rc = c->fail_hist.reserve((size_t)n_hist * 4))) return rc;
  if ((rc = c->hap_flags.reserve((size_t)n_haps))) return rc;
  if ((rc = c->lanes_main.reserve((size_t)std::max(plan.n_chunks, 1) * kLanes * sizeof(LaneSlot)))) return rc;

  const bool ev = c->cfg.record_events != 0;
  const bool deferred = c->cfg.record_events == 2;
  if (ev) {
    c->ev = c->ev_ring[deferred ? c->calls % DevCtx::kEventRing : 0];
    c->ring_double[deferred ? c->calls % DevCtx::kEventRing : 0] = use_double;
    c->calls++;
  }
  if (ev) HIP_TRY(hipEventRecord(c->ev[0], s));

  // ---- haplotype streams + clears: one launch ----
  uint32_t *stream_grouped = nullptr, *stream_flat = nullptr;
  uint8_t* hap_has_n = nullptr;
  {
    PrepArgs pa;
    const unsigned char* pb = pull ? hs_dev : dp;  // pulling: this kernel reads the HOST copy of the plan
    pa.hap_bases = (pull && inline_host) ? pb + L.batch + 5 * L.batch_stride : dbi.hap_bases;
    pa.hap_src = reinterpret_cast<const int32_t*>(pb + L.hap_src);
    pa.hap_len = reinterpret_cast<const int32_t*>(pb + L.hap_len);
    pa.hap_pos = reinterpret_cast<const int32_t*>(pb + L.hap_pos);
    pa.hap_group = reinterpret_cast<const int32_t*>(pb + L.hap_group);
    // (host-built streams: they arrive with the pulled block; this kernel then only pulls and clears)
    const bool host_streams = inline_host;
    stream_grouped = host_streams ? reinterpret_cast<uint32_t*>(dp + L.stream) : c->stream_buf.as<uint32_t>();
    stream_flat = host_streams ? reinterpret_cast<uint32_t*>(dp + L.stream_flat) : c->stream_buf.as<uint32_t>() + plan.n_stream;
    hap_has_n = host_streams ? dp + L.has_n : c->hap_flags.as<uint8_t>();
    pa.stream = stream_grouped;
    // the flat stream (no gaps between groups): the fp64 recomputation's jobs are arbitrary runs of it
    pa.hap_pos_flat = use_double ? nullptr : reinterpret_cast<const int32_t*>(pb + L.hap_pos_flat);
    pa.stream_flat = stream_flat;
    pa.hap_has_n = hap_has_n;
    pa.n_haps = host_streams ? 0 : n_haps;
    pa.clear_a = c->counters.as<int32_t>(); pa.n_a = 32;
    pa.clear_b = c->read_fail.as<int32_t>(); pa.n_b = use_double ? 0 : n_reads;
    pa.clear_c = c->fail_hist.as<int32_t>(); pa.n_c = n_hist;
    pa.place_chunk = reinterpret_cast<const int32_t*>(pb + L.place_chunk);
    pa.place_lane = pb + L.place_lane;
    pa.chunk_used = pb + L.chunk_used;
    pa.read_off = reinterpret_cast<const int64_t*>(pb + L.read_off);
    pa.lanes_out = c->lanes_main.as<LaneSlot>();
    pa.n_reads = n_reads; pa.n_chunks = plan.n_chunks; pa.rpl = rpl_main;
    const int threads_needed = std::max({pa.n_haps * 64, 32, pa.n_b, pa.n_c, n_reads});
    pa.hap_blocks = (threads_needed + kPrepBlock - 1) / kPrepBlock;
    pa.pull_src = reinterpret_cast<const uint4*>(hs_dev);
    pa.pull_dst = reinterpret_cast<uint4*>(dp);
    pa.pull_n16 = pull ? (int32_t)(L.total / 16) : 0;
    const int pull_blocks = pull ? (int)std::min<size_t>(64, (L.total / 16 + kPrepBlock * 4 - 1) / (kPrepBlock * 4)) : 0;
    if (deferred_launch) {
      defer->call.prep = pa;
      defer->call.prep_grid = pa.hap_blocks + pull_blocks;
    } else {
      hipLaunchKernelGGL(prep_kernel, dim3((unsigned)(pa.hap_blocks + pull_blocks)), dim3(kPrepBlock), 0, s, pa);
      if (pull) HIP_TRY(hipEventRecord(c->stage_free_slot[slot], s));
    }
  }

  DevBatch b;
  b.read_bases = dbi.read_bases; b.read_quals = dbi.read_quals; b.ins = dbi.ins_gop;
  b.del = dbi.del_gop; b.gcp = dbi.gcp;
  b.read_off = reinterpret_cast<const int64_t*>(dp + L.read_off);
  b.n_reads = n_reads; b.n_haps = n_haps;

  // XCD-aware grid of the streaming kernels (fwd_stream_block): a chunk's jobs all land on one XCD
  const bool xcd_env = g_env.xcd_aware;
  // (c->n_xcds: what the device reports -- 8 on an MI355X in SPX mode; a partitioned device shows fewer and gets no padding it cannot use)
  const int xq = c->n_xcds;
  const int chunk_stride = (xcd_env && xq > 1 && plan.n_chunks >= 64) ? (plan.n_chunks + xq - 1) / xq * xq : plan.n_chunks;
  auto fill_common = [&](auto& a) {
    a.b = b;
    a.stream = stream_grouped;
    a.hap_len = reinterpret_cast<const int32_t*>(dp + L.hap_len);
    a.hap_pos = reinterpret_cast<const int32_t*>(dp + L.hap_pos);
    a.hap_orig = reinterpret_cast<const int32_t*>(dp + L.hap_orig);
    a.hap_has_n = hap_has_n;
    a.groups = reinterpret_cast<const HapGroup*>(dp + L.groups);
    a.n_groups = (int)plan.groups.size();
    a.chunk_lanes = c->lanes_main.as<LaneSlot>();
    a.n_chunks = plan.n_chunks;
    a.chunk_stride = chunk_stride;
    a.jobs = c->jobs.as<FwdJob>();
    a.job_count = c->counters.as<int32_t>() + 2;
    a.job_next = c->counters.as<int32_t>() + 3;
    // the fp32 programs fetch a separator lane's priors from beyond the LDS allocation: only where that reads 0 (dev_init)
    constexpr bool is_f32 = std::is_same<typename std::decay<decltype(a)>::type, FwdArgs<float>>::value;
    a.asm_general = (c->asm_general && (!is_f32 || c->lds_oob_zero)) ? 1 : 0;
  };

  FinalizeArgs fa;
  fa.raw32 = c->raw32.as<float>(); fa.raw64 = c->raw64.as<double>(); fa.out = out_dev;
  fa.used64 = c->used64.as<uint8_t>();
  fa.count = c->counters.as<int32_t>(); fa.n = n_pairs; fa.mode = finalize_mode;
  fa.read_fail = c->read_fail.as<int32_t>(); fa.n_haps = n_haps;
  fa.log10_init_f = host_tables_f32().log10_initial;
  fa.log10_init32_as_f64 = std::log10(std::ldexp(1.0, 120));
  fa.log10_init_d = host_tables_f64().log10_initial;

  const int n_main_blocks = chunk_stride * (int)plan.groups.size();
  // persistent wavefronts of the striped long-read kernel: one per job up to two per SIMD (each owns two carry rows of
  // the longest stream group: ~110 KB)
  const int n_long_waves = (int)std::min<size_t>(2048, std::max<size_t>(512, std::max(long_jobs.size(), (size_t)n_long64 * plan.groups.size())));
  // ... and, when a read needs more wavefronts than a wide workgroup holds, the super-stripe kernel's carry rows behind them
  const size_t striped_carry_bytes = (size_t)n_long_waves * 2 * (3 * (size_t)carry_len + 64) * sizeof(double);
  const bool super_long = (blocks_for(plan.max_read_len, kRplF32) + kLanes - 1) / kLanes > kWideWavesMax;
  const int64_t xsteps = super_long ? super_steps(carry_len, plan.max_read_len, kRplF32) : 0;   // (fp32 and fp64 both run the long reads at 8 rows per lane)
  static_assert(kRplF32 == kRplF64Wide, "one array depth for the long reads of both precisions");
  unsigned char* xcarry = nullptr;
  if (n_long_main > 0 || n_long64 > 0) {
    if ((rc = c->carry.reserve(striped_carry_bytes + (size_t)super_blocks_max<float>() * 2 * (size_t)xsteps * 32))) return rc;
    if (super_long) xcarry = c->carry.as<unsigned char>() + striped_carry_bytes;
  }
  st.n_long_pairs = (int32_t)std::min<int64_t>((int64_t)n_long_main * n_haps, 0x7fffffff);
  st.n_chunks = plan.n_chunks;
  st.n_hap_groups = (int)plan.groups.size();
  st.rows_per_lane = rpl_main;
  st.lane_fill = plan.n_chunks ? (float)((double)plan.useful_rows / ((double)plan.n_chunks * 64 * rpl_main)) : 0.f;
  st.cells = (int64_t)rl * (int64_t)hl;

  if (ev) HIP_TRY(hipEventRecord(c->ev[1], s));
  if (use_double) {
    FwdArgs<double> a{};
    fill_common(a);
    a.tab = c->dt64;
    a.y0 = reinterpret_cast<const double*>(dp + L.y0_64);
    a.raw = c->raw64.as<double>();
    if (n_main_blocks > 0) launch_stream<double, kRplF64Jobs>(a, fma, n_main_blocks, s);
    if (n_long_main > 0) {
      FwdArgs<double> la = a;
      la.chunk_lanes = reinterpret_cast<const LaneSlot*>(dp + L.long_lanes);
      la.jobs = reinterpret_cast<const FwdJob*>(dp + L.long_jobs);
      la.job_count = reinterpret_cast<const int32_t*>(dp + L.long_count);
      la.job_next = c->counters.as<int32_t>() + 7;
      launch_long_jobs<double, kRplF64Wide, kRplF64>(la, fma, n_long_waves, plan.max_read_len, c->carry.as<double>(), carry_len, s, xcarry, xsteps, c->counters.as<int32_t>() + 12);
    }
    if (ev) HIP_TRY(hipEventRecord(c->ev[2], s));
    hipLaunchKernelGGL(finalize64_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, s, fa, 1);
    if (ev) { HIP_TRY(hipEventRecord(c->ev[3], s)); HIP_TRY(hipEventRecord(c->ev[4], s)); }
    HIP_TRY(hipEventRecord(c->policy_done, s));  // (host path: "results are final from here")
  } else {
    FwdArgs<float> a{};
    fill_common(a);
    a.tab = c->dt32;
    a.y0 = reinterpret_cast<const float*>(dp + L.y0_32);
    a.raw = c->raw32.as<float>();
    // (the small-call path applies the policy per pair and writes the words itself)
    const bool fold_packed = finalize_mode == kModePacked && !per_pair_call;
    a.packed_out = fold_packed ? reinterpret_cast<uint64_t*>(out_dev) : nullptr;
    if (deferred_launch) {
      defer->call.f = a;
      defer->call.rpl_main = rpl_main;
      defer->call.main_blocks = n_main_blocks;
      defer->call.fused = fused_call ? 1 : 0;
    } else if (n_main_blocks > 0 && !fused_call) {
      launch_main_f32(a, rpl_main, fma, n_main_blocks, s);
    }
    if (n_long_main > 0) {
      FwdArgs<float> la = a;
      la.chunk_lanes = reinterpret_cast<const LaneSlot*>(dp + L.long_lanes);
      la.jobs = reinterpret_cast<const FwdJob*>(dp + L.long_jobs);
      la.job_count = reinterpret_cast<const int32_t*>(dp + L.long_count);
      la.job_next = c->counters.as<int32_t>() + 7;
      if (rpl_main <= 4) launch_long<float, 4>(la, fma, n_long_waves, c->carry.as<float>(), carry_len, s);  // (2 is only chosen without long reads)
      else               launch_long_jobs<float, kRplF32, kRplF32>(la, fma, n_long_waves, plan.max_read_len, c->carry.as<float>(), carry_len, s, xcarry, xsteps, c->counters.as<int32_t>() + 12);
    }
    if (ev) HIP_TRY(hipEventRecord(c->ev[2], s));

    // fp64 arguments shared by the two ways of recomputing (the flat stream: a job may run across stream groups)
    FwdArgs<double> d{};
    fill_common(d);
    d.tab = c->dt64;
    d.y0 = reinterpret_cast<const double*>(dp + L.y0_64);
    d.raw = c->raw64.as<double>();
    d.stream = stream_flat;
    d.hap_pos = reinterpret_cast<const int32_t*>(dp + L.hap_pos_flat);
    // (the planned fp64 pass leaves the packed words of the recomputed pairs to finalize64_kernel: its jobs run as whole-job
    //  asm programs that store the raw sums only)
    d.packed_out = nullptr;
    d.packed_only_flagged = c->used64.as<uint8_t>();
    int32_t* cnts = c->counters.as<int32_t>();
    // Small calls (one GATK region): policy + fp64 recomputation + finalisation of one pair per wavefront in ONE launch
    // (pairhmm_pair_policy_kernel); rows per lane by the longest read.
    const bool per_pair = per_pair_call;
    if (per_pair) {
      PairPolicyArgs q;
      q.raw32 = c->raw32.as<float>(); q.out = out_dev; q.used64 = c->used64.as<uint8_t>(); q.count = cnts;
      q.hap_sidx = reinterpret_cast<const int32_t*>(dp + L.hap_sidx);
      q.mode = finalize_mode;
      q.log10_init_f = fa.log10_init_f; q.log10_init32_as_f64 = fa.log10_init32_as_f64; q.log10_init_d = fa.log10_init_d;
      const int rows = plan.max_read_len <= 2 * kLanes - 1 ? 2 : plan.max_read_len <= 4 * kLanes - 1 ? 4 : kRplF64;
      if (deferred_launch) {
        SmallCall& k = defer->call;
        k.d = d; k.q = q; k.rows = rows; k.n_pairs = (int32_t)n_pairs; k.fma = fma; k.speculate = c->speculate_fp64;
        memcpy(hs + L.desc, &k, sizeof k);  // nothing has been launched yet: the block is still ours to write
        defer->desc_pinned = reinterpret_cast<const SmallCall*>(hs_dev + L.desc);
        defer->desc_dev = reinterpret_cast<const SmallCall*>(dp + L.desc);
        defer->filled = true;
        c->last_pairs = n_pairs;
        c->last_stream = s;
        c->have_last = true;
        st.n_fallback = -1;
        return GKLHIP_OK;
      }
      if (ev) HIP_TRY(hipEventRecord(c->ev[3], s));
      if (fused_call) {
        launch_pair_fused(a, d, q, rows, fma, n_pairs, s, c->speculate_fp64 && g_host_calls_in_flight.load(std::memory_order_relaxed) <= 1);
      } else if (n_pairs > kTwoStepFrom) {
        if ((rc = c->fail_order.reserve((size_t)n_pairs * 4))) return rc;
        launch_pair_policy_two_step(d, q, rows, fma, n_pairs, c->fail_order.as<int32_t>(), s);
      } else {
        launch_pair_policy(d, q, rows, fma, n_pairs, s);
      }
      if (ev) HIP_TRY(hipEventRecord(c->ev[4], s));
      HIP_TRY(hipEventRecord(c->policy_done, s));
    } else {
    // ---- precision policy + device-side planning of the fp64 recomputation (three launches, no host round trip) ----
    const size_t jobs_per_chunk = (size_t)n_haps;  // a job holds at least one haplotype and the jobs of a chunk do not overlap
    const size_t max_jobs = (size_t)n_reads * jobs_per_chunk;
    if ((rc = c->fail_order.reserve(((size_t)n_reads + (size_t)n_long64) * 4))) return rc;
    if ((rc = c->lanes2.reserve((size_t)n_reads * kLanes * sizeof(LaneSlot)))) return rc;
    if ((rc = c->jobs.reserve(2 * max_jobs * sizeof(FwdJob)))) return rc;  // as built + sorted by length
    if (n_long64 > 0 && (rc = c->jobs_long.reserve((size_t)n_long64 * jobs_per_chunk * sizeof(FwdJob)))) return rc;
    const LaneSlot* pl = reinterpret_cast<const LaneSlot*>(dp + L.long_lanes);
    {
      PlanArgs pa;
      pa.fa = fa;
      pa.n_reads = n_reads; pa.n_haps = n_haps; pa.n_pairs_i = (int32_t)n_pairs;
      pa.read_off = b.read_off;
      pa.rpl = kRplF64Jobs; pa.max_len = kLanes * kRplF64Jobs - 1;
      pa.cnts = cnts;
      pa.hist = c->fail_hist.as<int32_t>();
      pa.pos = pa.hist + (n_haps + 2);
      pa.order = c->fail_order.as<int32_t>();
      pa.lanes2 = c->lanes2.as<LaneSlot>();
      pa.hap_orig = reinterpret_cast<const int32_t*>(dp + L.hap_orig);
      pa.hap_group = reinterpret_cast<const int32_t*>(dp + L.hap_group);
      pa.hap_pos = reinterpret_cast<const int32_t*>(dp + L.hap_pos_flat);
      pa.hap_len = reinterpret_cast<const int32_t*>(dp + L.hap_len);
      pa.jobs = c->jobs.as<FwdJob>();
      pa.sorted = c->jobs.as<FwdJob>() + max_jobs;
      pa.long_lanes = pl + (size_t)n_long_main * kLanes;
      pa.n_long = n_long64;
      pa.jobs_long = c->jobs_long.as<FwdJob>();
      pa.long_chunk_jobs = c->fail_order.as<int32_t>() + n_reads;
      pa.total_cols = (int32_t)std::min<int64_t>((int64_t)hl + n_haps, 0x7fffffff);
      const int wanted_env = g_env.fb_wanted_jobs;
      // (a shard of the batch wants fewer, longer jobs: 4096 for an eighth, measured on the 1250 x 128 shard)
      pa.wanted_jobs = wanted_env > 0 ? wanted_env : (int)std::min<int64_t>(kFallbackWantedJobs, std::max<int64_t>(4096, n_pairs / 100));
      pa.min_job_cols = 256;
      pa.packed_by_kernels = fold_packed ? 1 : 0;
      // Three stream-ordered launches (pairhmm_aux_kernels.h): no block waits for another, so nothing limits how many
      // of these are in flight per device or process.  The policy takes a block per 4096 pairs (up to one per CU), the
      // packing a wavefront per window of affected reads, the run detection a wavefront per chunk (grid-stride).
      const int blocks_env = g_env.plan_blocks;
      const int policy_grid = std::max(1, blocks_env > 0 ? blocks_env : (int)std::min<int64_t>(c->n_cus, std::max<int64_t>(16, n_pairs / 4096)));
      const int64_t max_windows = ((int64_t)n_reads + kPackWindow - 1) / kPackWindow;
      const int pack_grid = (int)std::max<int64_t>(1, std::min<int64_t>(kPlanBlocks, (max_windows + kPlanBlock / 64 - 1) / (kPlanBlock / 64)));
      const int jobs_grid = std::max(1, std::min(c->n_cus, blocks_env > 0 ? blocks_env : (int)std::min<int64_t>(kPlanBlocks, std::max<int64_t>(16, n_pairs / 8192))));
      hipLaunchKernelGGL(plan_policy_kernel, dim3((unsigned)policy_grid), dim3(kPlanBlock), 0, s, pa);
      // The policy's flags and the kept pairs' words are final here: the log10 of the kept pairs (side stream below; the
      // host's early pass in host-buffer calls) starts now and overlaps the two small planning launches -- behind them it
      // would queue up against the fp64 pass, whose persistent wavefronts leave it no registers until they drain.
      HIP_TRY(hipEventRecord(c->policy_done, s));
      hipLaunchKernelGGL(plan_pack_kernel, dim3((unsigned)pack_grid), dim3(kPlanBlock), 0, s, pa);
      hipLaunchKernelGGL(plan_jobs_kernel, dim3((unsigned)jobs_grid), dim3(kPlanBlock), 0, s, pa);
    }
    const bool side_finalize = finalize_mode == GKLHIP_FINALIZE_DEVICE_F64 || finalize_mode == GKLHIP_FINALIZE_DEVICE_REF32;
    if (side_finalize) {
      if ((rc = aux_streams(c))) return rc;
      HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->policy_done, 0));
      hipLaunchKernelGGL(finalize32_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, c->copy_stream, fa);
      HIP_TRY(hipEventRecord(c->early_copy_done, c->copy_stream));
    }
    // ---- fp64 recomputation of the underflowed pairs: persistent wavefronts stream the job list -- same WaveJob
    // template as the main pass, T = double (no jobs: the kernel's wavefronts leave at once) ----
    d.chunk_lanes = c->lanes2.as<LaneSlot>();
    d.n_chunks = n_reads;  // upper bound; the job list only names packed chunks
    d.jobs = c->jobs.as<FwdJob>() + max_jobs;
    if (ev) HIP_TRY(hipEventRecord(c->ev[3], s));
    launch_jobs<double, kRplF64Jobs>(d, fma, (int)std::min<int64_t>(n_pairs, (int64_t)c->n_cus * 16), s);
    if (n_long64 > 0) {
      // reads too long for a chunk: one pseudo-chunk each, same run detection, striped kernel
      FwdArgs<double> ld = d;
      ld.chunk_lanes = pl + (size_t)n_long_main * kLanes;
      ld.jobs = c->jobs_long.as<FwdJob>();
      ld.job_count = cnts + 8;
      ld.job_next = cnts + 9;
      launch_long_jobs<double, kRplF64Wide, kRplF64>(ld, fma, n_long_waves, plan.max_read_len, c->carry.as<double>(), carry_len, s, xcarry, xsteps, cnts + 13);
    }
    if (ev) HIP_TRY(hipEventRecord(c->ev[4], s));
    // (log10 of the recomputed pairs / host-buffer calls: their packed words)
    hipLaunchKernelGGL(finalize64_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, s, fa, 0);
    if (side_finalize) HIP_TRY(hipStreamWaitEvent(s, c->early_copy_done, 0));  // join the side stream
    }  // !per_pair
  }
  if (ev) HIP_TRY(hipEventRecord(c->ev[5], s));
  HIP_TRY(hipGetLastError());

  HIP_TRY(hipEventRecord(c->plan_unused_slot[slot], s));
  HIP_TRY(hipEventRecord(c->call_done, s));
  c->have_call_done = true;
  c->last_pairs = n_pairs;
  c->last_stream = s;
  c->have_last = true;

  if (ev && !deferred) {
    HIP_TRY(hipEventSynchronize(c->ev[5]));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); st.ms_fwd_main = ms;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[3], c->ev[4])); st.ms_fwd_fallback = use_double ? 0.f : ms;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[5])); st.ms_total_device = ms;
    int32_t cnt[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(cnt, c->counters.p, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    st.n_fallback = use_double ? n_pairs : cnt[0];
    if (timing && !use_double) {
      int32_t k[32];
      HIP_TRY(hipMemcpy(k, c->counters.p, sizeof k, hipMemcpyDeviceToHost));
      fprintf(stderr, "[gklhip] policy+plan phases, each from the start of its own launch (us): hist %.1f scan %.1f scatter %.1f pack %.1f jobs %.1f sort %.1f | "
              "%d affected reads, %d chunks, %d jobs | window 0: loaded %.1f ranked %.1f fitted %.1f cleared %.1f written %.1f\n",
              k[16] * 0.01, k[17] * 0.01, k[18] * 0.01, k[19] * 0.01, k[20] * 0.01, k[21] * 0.01,
              k[4], k[5], k[2], k[22] * 0.01, k[23] * 0.01, k[24] * 0.01, k[25] * 0.01, k[26] * 0.01);
    }
  } else {
    st.n_fallback = use_double ? n_pairs : -1;  // unknown without a sync; gklhip_get_raw fills it in
  }
  return GKLHIP_OK;
}

}  // namespace
