// C-ABI implementation (include/gkl_hip_pairhmm.h) of the MI355X PairHMM forward path:
// context / tables / planning / kernel launches / precision policy / finalisation.
//
// Reference counterparts (src/main/native/pairhmm): IntelPairHmm.cc:55-118 (init),
// :150-169 (batch loop + fp32->fp64 policy + log10), :189-192 (done).  There is no
// CPU compute path in this library: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gkl_hip_pairhmm.h"
#include "pairhmm_fwd_kernel.h"
#include "pairhmm_plan.h"
#include "pairhmm_tables.h"
#include "pairhmm_aux_kernels.h"
#include "pairhmm_host_finalize.h"

using namespace gklhip;

// ------------------------------------------------------------------ errors
namespace {
thread_local std::string g_err;

int fail(int status, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return status;
}

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      (void)hipGetLastError();                                                               \
      return fail(e__ == hipErrorOutOfMemory ? GKLHIP_ERR_OOM : GKLHIP_ERR_HIP, "%s: %s",    \
                  #expr, hipGetErrorString(e__));                                            \
    }                                                                                        \
  } while (0)

// Grow-only device / pinned-host buffers.
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t n) {
    if (n <= cap) return GKLHIP_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    const size_t want = n + n / 4 + 256;
    HIP_TRY(hipMalloc(&p, want));
    cap = want;
    return GKLHIP_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <typename T> T* as() const { return static_cast<T*>(p); }
};
struct PinBuf {
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t n) {
    if (n <= cap) return GKLHIP_OK;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    const size_t want = n + n / 4 + 256;
    HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
    cap = want;
    return GKLHIP_OK;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
  template <typename T> T* as() const { return static_cast<T*>(p); }
};
}  // namespace

// ------------------------------------------------------------------ context
struct gklhip_ctx {
  gklhip_config cfg;
  int device = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;
  // tables
  DevBuf tab32, tab64;
  DevTables<float> dt32;
  DevTables<double> dt64;
  // per-call plan uploads (pinned staging -> device)
  // Two slots alternate from call to call: the plan of call k+1 is staged and uploaded (own stream) while the
  // kernels of call k still read theirs -- back-to-back batches then never wait for the 1.7 MB plan block.
  PinBuf stage_slot[2];
  DevBuf plan_dev_slot[2];
  hipEvent_t stage_free_slot[2] = {nullptr, nullptr};   // the slot's upload has left the staging buffer
  hipEvent_t plan_unused_slot[2] = {nullptr, nullptr};  // the last call that used the slot's device copy has finished
  hipStream_t upload_stream = nullptr;
  int plan_slot = 0;
  // small host-buffer calls: after the policy kernel the packed results and the fallback count come back at once;
  // a call without underflowed pairs (the usual GATK region) then skips the whole fp64 stage
  bool peek_after_policy = false;   // in: set by gklhip_compute
  bool fallback_skipped = false;    // out
  PinBuf peek_count;
  // per-call device scratch
  DevBuf raw32, raw64, used64, list, counters, stream_buf, read_off_dev, out_dev;
  DevBuf read_fail, lanes2, jobs, jobs_long, fail_order, fail_hist, hap_flags;
  // host-API device copies of the batch, packed results (device + pinned), finalisation workers
  DevBuf batch_dev, res_dev;
  PinBuf res_pin, res_pin2, batch_stage;
  WorkerPool workers;
  hipStream_t copy_stream = nullptr;  // early D2H of the fp32 results while the fp64 pass runs
  hipEvent_t policy_done = nullptr, early_copy_done = nullptr;
  // events: kEventRing sets of 6 (call start, main begin/end, fallback begin/end, call end); record_events == 1 uses
  // set 0 and synchronises every call, record_events == 2 rotates through the ring and never synchronises
  // (gklhip_get_step_times reads a set later)
  static constexpr int kEventRing = 64;
  hipEvent_t ev_ring[kEventRing][6] = {};
  hipEvent_t* ev = ev_ring[0];
  int64_t calls = 0;
  bool ring_double[kEventRing] = {};
  // last call
  gklhip_stats stats;
  int64_t last_pairs = 0;
  hipStream_t last_stream = nullptr;
  bool have_last = false;
  Plan plan;
  std::vector<PlanLane> long_lanes;
  std::vector<FwdJob> long_jobs;
  DevBuf carry;
};

namespace {

template <typename T>
int upload_tables(gklhip_ctx* c, const HostTables<T>& h, DevBuf* buf, DevTables<T>* dt) {
  const size_t n = (size_t)kQuals * 2 + kMmEntries;
  int st = buf->reserve(n * sizeof(T));
  if (st) return st;
  T* base = buf->as<T>();
  HIP_TRY(hipMemcpy(base, h.ph2pr.data(), kQuals * sizeof(T), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(base + kQuals, h.div3.data(), kQuals * sizeof(T), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(base + 2 * kQuals, h.mm.data(), kMmEntries * sizeof(T), hipMemcpyHostToDevice));
  dt->ph2pr = base;
  dt->div3 = base + kQuals;
  dt->mm = base + 2 * kQuals;
  (void)c;
  return GKLHIP_OK;
}

size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Layout of the per-call plan block (identical in pinned staging and on the device).
struct PlanLayout {
  size_t lanes, groups, hap_len, hap_pos, hap_orig, hap_sidx, hap_group, stream_src, y0_32, y0_64, read_off, long_lanes, long_jobs, long_count, total;
};
PlanLayout layout_for(const Plan& p, int n_reads, int n_haps, size_t n_long_lanes, size_t n_long_jobs) {
  PlanLayout l;
  size_t o = 0;
  l.lanes = o; o = align_up(o + p.lanes.size() * sizeof(PlanLane));
  l.groups = o; o = align_up(o + p.groups.size() * sizeof(PlanGroup));
  l.hap_len = o; o = align_up(o + (size_t)n_haps * 4);
  l.hap_pos = o; o = align_up(o + (size_t)n_haps * 4);
  l.hap_orig = o; o = align_up(o + (size_t)n_haps * 4);
  l.hap_sidx = o; o = align_up(o + (size_t)n_haps * 4);
  l.hap_group = o; o = align_up(o + (size_t)n_haps * 4);
  l.stream_src = o; o = align_up(o + p.stream_src.size() * 4);
  l.y0_32 = o; o = align_up(o + (size_t)n_haps * 4);
  l.y0_64 = o; o = align_up(o + (size_t)n_haps * 8);
  l.read_off = o; o = align_up(o + (size_t)(n_reads + 1) * 8);
  l.long_lanes = o; o = align_up(o + n_long_lanes * sizeof(PlanLane));
  l.long_jobs = o; o = align_up(o + n_long_jobs * sizeof(FwdJob));
  l.long_count = o; o = align_up(o + 16);
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
template <int RPL>
void launch_stream2(const FwdArgs<float>& a, int fma, int n_blocks, hipStream_t s) {
  if (fma) hipLaunchKernelGGL((pairhmm_fwd_stream2_kernel<RPL, true>), dim3(n_blocks), dim3(64), 0, s, a);
  else     hipLaunchKernelGGL((pairhmm_fwd_stream2_kernel<RPL, false>), dim3(n_blocks), dim3(64), 0, s, a);
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

// Rows per lane.  fp32 main pass: 8 (one chunk per wavefront; 4 = the dual-chunk packed-math
// kernel, opt-in).  fp64 passes: 6.  A read of length R needs R+1 rows; reads that exceed
// 64*RPL rows go to the striped long-read kernel of the same RPL.
#ifndef GKL_RPL_F64
#define GKL_RPL_F64 6
#endif
constexpr int kRplF64 = GKL_RPL_F64;
constexpr int64_t kPeekPairs = 65536;          // host-buffer calls up to this many pairs look at the fallback count before the fp64 stage
constexpr size_t kSmallBatchBytes = 1 << 20;  // host-buffer calls up to this size stage their inputs in one block
constexpr int kTargetCols = 2048;  // columns of a full-size haplotype group (sweep 1024..4096: flat within 2 %, optimum 1800..2600)
#ifndef GKL_RPL_F32
#define GKL_RPL_F32 8
#endif
constexpr int kRplF32 = GKL_RPL_F32;
// fp32 main pass: which kernel.  rows_per_lane of the config: 0 = choose, 8 = the 8-row kernel, 4 = the dual-chunk
// packed-math kernel (2 x 4 rows), -4 = the single-chunk 4-row kernel.  Choosing: a small batch (one GATK active
// region) gives the 8-row kernel fewer jobs than the chip has wavefront slots worth filling (< 2 per SIMD), and a
// lone wavefront issues one instruction per ~6 cycles; 4 rows per lane doubles the chunks and halves the step.
struct F32Kernel { int rpl; bool dual; };
F32Kernel pick_f32_kernel(int forced, int n_reads, int n_haps, const int64_t* read_off, const int64_t* hap_off) {
  if (forced == 4) return {4, true};
  if (forced == -4) return {4, false};
  if (forced == 8) return {kRplF32, false};
  int64_t blocks = 0;
  for (int r = 0; r < n_reads; r++) {
    const int nb = blocks_for((int)(read_off[r + 1] - read_off[r]), kRplF32);
    if (nb <= kLanes) blocks += nb;
  }
  const int64_t chunks = std::max<int64_t>(1, (blocks + kLanes - 1) / kLanes);
  const int64_t total_cols = hap_off[n_haps] + n_haps;
  const int64_t groups = std::min<int64_t>(n_haps, std::max<int64_t>((total_cols + kTargetCols - 1) / kTargetCols,
                                                                      (4096 + chunks - 1) / chunks));
  return chunks * groups < 2048 ? F32Kernel{4, false} : F32Kernel{kRplF32, false};
}

// The whole device-side pipeline on stream `s`; `db` holds DEVICE byte arrays, host offsets.
int run_device(gklhip_ctx* c, const gklhip_batch* db, double* out_dev, int finalize_mode, hipStream_t s) {
  const int n_reads = db->n_reads, n_haps = db->n_haps;
  const int64_t n_pairs = (int64_t)n_reads * n_haps;
  gklhip_stats& st = c->stats;
  memset(&st, 0, sizeof st);
  st.n_pairs = n_pairs;
  c->have_last = false;
  c->fallback_skipped = false;
  if (n_pairs == 0) return GKLHIP_OK;
  const bool use_double = c->cfg.use_double != 0;
  const int fma = c->cfg.fma_mode != 0;

  // ---- plan (host) ----
  const auto t_plan0 = std::chrono::steady_clock::now();
  Plan& plan = c->plan;
  const int rpl64 = kRplF64;
  const F32Kernel f32k = pick_f32_kernel(c->cfg.rows_per_lane, n_reads, n_haps, db->read_off, db->hap_off);
  const int rpl_main = use_double ? rpl64 : f32k.rpl;
  static const int target_cols_env = [] { const char* v = getenv("GKLHIP_TARGET_COLS"); return v ? atoi(v) : 0; }();
  build_plan(n_reads, n_haps, db->read_off, db->hap_off, rpl_main, target_cols_env > 0 ? target_cols_env : kTargetCols, &plan);
  // Long reads: pseudo-chunks (lane 0 names the read) + one striped job per (read, stream group)
  // for the main pass; for the fp64 fallback the same pseudo-chunks feed build_jobs_kernel.
  std::vector<PlanLane>& long_lanes = c->long_lanes;
  std::vector<FwdJob>& long_jobs = c->long_jobs;
  std::vector<int32_t> long64;  // reads too long for the packed fp64 pass
  long_lanes.clear(); long_jobs.clear();
  for (int r = 0; r < n_reads; r++)
    if (blocks_for((int)(db->read_off[r + 1] - db->read_off[r]), rpl64) > kLanes) long64.push_back(r);
  const std::vector<int32_t>& long_main = plan.long_reads;
  const int n_long_main = (int)long_main.size();
  const int n_long64 = use_double ? 0 : (int)long64.size();
  // pseudo-chunk index space: [0, n_long_main) main-pass reads, then [n_long_main, +n_long64) fp64-pass reads
  for (int32_t r : long_main) { long_lanes.resize(long_lanes.size() + kLanes, PlanLane{-1, 0}); long_lanes[long_lanes.size() - kLanes] = PlanLane{r, 0}; }
  if (!use_double)
    for (int32_t r : long64) { long_lanes.resize(long_lanes.size() + kLanes, PlanLane{-1, 0}); long_lanes[long_lanes.size() - kLanes] = PlanLane{r, 0}; }
  for (int i = 0; i < n_long_main; i++)
    for (const PlanGroup& g : plan.groups) long_jobs.push_back(FwdJob{i, g.hap_begin, g.hap_end, 0});
  int carry_len = 0;
  for (const PlanGroup& g : plan.groups) {
    const int last = g.hap_end - 1;
    carry_len = std::max(carry_len, plan.hap_pos[last] + plan.hap_len[last] - plan.hap_pos[g.hap_begin] + 3 * kLanes);
  }
  carry_len = (carry_len + 63) / 64 * 64;
  const PlanLayout L = layout_for(plan, n_reads, n_haps, long_lanes.size(), long_jobs.size());

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
  memcpy(hs + L.lanes, plan.lanes.data(), plan.lanes.size() * sizeof(PlanLane));
  memcpy(hs + L.groups, plan.groups.data(), plan.groups.size() * sizeof(PlanGroup));
  memcpy(hs + L.hap_len, plan.hap_len.data(), (size_t)n_haps * 4);
  memcpy(hs + L.hap_pos, plan.hap_pos.data(), (size_t)n_haps * 4);
  memcpy(hs + L.hap_orig, plan.hap_orig.data(), (size_t)n_haps * 4);
  memcpy(hs + L.hap_sidx, plan.hap_sidx.data(), (size_t)n_haps * 4);
  memcpy(hs + L.hap_group, plan.hap_group.data(), (size_t)n_haps * 4);
  memcpy(hs + L.stream_src, plan.stream_src.data(), plan.stream_src.size() * 4);
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
  if (L.total >= (256u << 10)) {
    HIP_TRY(hipStreamWaitEvent(c->upload_stream, c->plan_unused_slot[slot], 0));  // readers of the old contents are done
    HIP_TRY(hipMemcpyAsync(dp, hs, L.total, hipMemcpyHostToDevice, c->upload_stream));
    HIP_TRY(hipEventRecord(c->stage_free_slot[slot], c->upload_stream));
    HIP_TRY(hipStreamWaitEvent(s, c->stage_free_slot[slot], 0));                    // kernels below read the new plan
  } else {
    // a small plan (GATK-sized call): the cross-stream hand-off would cost more than the copy
    HIP_TRY(hipStreamWaitEvent(s, c->plan_unused_slot[slot], 0));
    HIP_TRY(hipMemcpyAsync(dp, hs, L.total, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(c->stage_free_slot[slot], s));
  }
  if (getenv("GKLHIP_TIMING"))
    fprintf(stderr, "[gklhip] host plan + staging: %.3f ms (%d chunks, %zu stream entries, %zu plan bytes)\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_plan0).count(),
            plan.n_chunks, plan.stream_src.size(), L.total);

  // ---- scratch ----
  if ((rc = c->raw32.reserve((size_t)n_pairs * 4))) return rc;
  if ((rc = c->raw64.reserve((size_t)n_pairs * 8))) return rc;
  if ((rc = c->used64.reserve((size_t)n_pairs))) return rc;
  if ((rc = c->list.reserve((size_t)n_pairs * 4))) return rc;
  if ((rc = c->counters.reserve(64))) return rc;
  if ((rc = c->read_fail.reserve((size_t)n_reads * 4))) return rc;
  if ((rc = c->stream_buf.reserve(plan.stream_src.size() * 4))) return rc;
  const int n_hist = use_double ? 0 : 2 * (n_haps + 2);
  if (!use_double && (rc = c->fail_hist.reserve((size_t)n_hist * 4))) return rc;
  const int n_flag_words = (n_haps + 3) / 4;
  if ((rc = c->hap_flags.reserve((size_t)n_flag_words * 4))) return rc;
  {
    const int n_clear = std::max({16, use_double ? 0 : n_reads, n_hist, n_flag_words});
    hipLaunchKernelGGL(clear_kernel, dim3((unsigned)((n_clear + 255) / 256)), dim3(256), 0, s, c->counters.as<int32_t>(), 16,
                       c->read_fail.as<int32_t>(), use_double ? 0 : n_reads, c->fail_hist.as<int32_t>(), n_hist,
                       c->hap_flags.as<int32_t>(), n_flag_words);
  }

  const bool ev = c->cfg.record_events != 0;
  const bool deferred = c->cfg.record_events == 2;
  if (ev) {
    c->ev = c->ev_ring[deferred ? c->calls % gklhip_ctx::kEventRing : 0];
    c->ring_double[deferred ? c->calls % gklhip_ctx::kEventRing : 0] = use_double;
    c->calls++;
  }
  if (ev) HIP_TRY(hipEventRecord(c->ev[0], s));

  // ---- haplotype streams ----
  const int n_stream = (int)plan.stream_src.size();
  hipLaunchKernelGGL(build_stream_kernel, dim3((n_stream + 255) / 256), dim3(256), 0, s,
                     reinterpret_cast<const int32_t*>(dp + L.stream_src), db->hap_bases,
                     c->stream_buf.as<uint32_t>(), n_stream, reinterpret_cast<const int32_t*>(dp + L.hap_pos), n_haps,
                     c->hap_flags.as<uint8_t>());

  DevBatch b;
  b.read_bases = db->read_bases; b.read_quals = db->read_quals; b.ins = db->ins_gop;
  b.del = db->del_gop; b.gcp = db->gcp;
  b.read_off = reinterpret_cast<const int64_t*>(dp + L.read_off);
  b.n_reads = n_reads; b.n_haps = n_haps;

  auto fill_common = [&](auto& a) {
    a.b = b;
    a.stream = c->stream_buf.as<uint32_t>();
    a.hap_len = reinterpret_cast<const int32_t*>(dp + L.hap_len);
    a.hap_pos = reinterpret_cast<const int32_t*>(dp + L.hap_pos);
    a.hap_orig = reinterpret_cast<const int32_t*>(dp + L.hap_orig);
    a.hap_sidx = reinterpret_cast<const int32_t*>(dp + L.hap_sidx);
    a.hap_has_n = c->hap_flags.as<uint8_t>();
    a.groups = reinterpret_cast<const HapGroup*>(dp + L.groups);
    a.n_groups = (int)plan.groups.size();
    a.chunk_lanes = reinterpret_cast<const LaneSlot*>(dp + L.lanes);
    a.n_chunks = plan.n_chunks;
    a.jobs = c->jobs.as<FwdJob>();
    a.job_count = c->counters.as<int32_t>() + 2;
    a.job_next = c->counters.as<int32_t>() + 3;
  };

  FinalizeArgs fa;
  fa.raw32 = c->raw32.as<float>(); fa.raw64 = c->raw64.as<double>(); fa.out = out_dev;
  fa.used64 = c->used64.as<uint8_t>(); fa.list = c->list.as<int32_t>();
  fa.count = c->counters.as<int32_t>(); fa.n = n_pairs; fa.mode = finalize_mode;
  fa.read_fail = c->read_fail.as<int32_t>(); fa.n_haps = n_haps;
  fa.log10_init_f = host_tables_f32().log10_initial;
  fa.log10_init32_as_f64 = std::log10(std::ldexp(1.0, 120));
  fa.log10_init_d = host_tables_f64().log10_initial;

  const int n_main_blocks = plan.n_chunks * (int)plan.groups.size();
  const int n_long_waves = 512;  // persistent wavefronts of the striped long-read kernel
  if (n_long_main > 0 || n_long64 > 0) {
    if ((rc = c->carry.reserve((size_t)n_long_waves * 2 * (3 * (size_t)carry_len + 64) * sizeof(double)))) return rc;
  }
  st.n_long_pairs = (int32_t)std::min<int64_t>((int64_t)n_long_main * n_haps, 0x7fffffff);
  st.n_chunks = plan.n_chunks;
  st.n_hap_groups = (int)plan.groups.size();
  st.rows_per_lane = rpl_main;
  st.lane_fill = plan.n_chunks ? (float)((double)plan.useful_rows / ((double)plan.n_chunks * 64 * rpl_main)) : 0.f;
  {
    int64_t rl = db->read_off[n_reads], hl = db->hap_off[n_haps];
    st.cells = rl * hl;
  }

  if (ev) HIP_TRY(hipEventRecord(c->ev[1], s));
  if (use_double) {
    FwdArgs<double> a{};
    fill_common(a);
    a.tab = c->dt64;
    a.y0 = reinterpret_cast<const double*>(dp + L.y0_64);
    a.raw = c->raw64.as<double>();
    if (n_main_blocks > 0) launch_stream<double, kRplF64>(a, fma, n_main_blocks, s);
    if (n_long_main > 0) {
      FwdArgs<double> la = a;
      la.chunk_lanes = reinterpret_cast<const LaneSlot*>(dp + L.long_lanes);
      la.jobs = reinterpret_cast<const FwdJob*>(dp + L.long_jobs);
      la.job_count = reinterpret_cast<const int32_t*>(dp + L.long_count);
      la.job_next = c->counters.as<int32_t>() + 7;
      launch_long<double, kRplF64>(la, fma, n_long_waves, c->carry.as<double>(), carry_len, s);
    }
    if (ev) HIP_TRY(hipEventRecord(c->ev[2], s));
    hipLaunchKernelGGL(finalize64_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, s, fa, 0);
    if (ev) { HIP_TRY(hipEventRecord(c->ev[3], s)); HIP_TRY(hipEventRecord(c->ev[4], s)); }
  } else {
    FwdArgs<float> a{};
    fill_common(a);
    a.tab = c->dt32;
    a.y0 = reinterpret_cast<const float*>(dp + L.y0_32);
    a.raw = c->raw32.as<float>();
    if (n_main_blocks > 0) {
      if (f32k.dual)          launch_stream2<4>(a, fma, ((plan.n_chunks + 1) / 2) * (int)plan.groups.size(), s);
      else if (rpl_main == 4) launch_stream<float, 4>(a, fma, n_main_blocks, s);
      else               launch_stream<float, kRplF32>(a, fma, n_main_blocks, s);
    }
    if (n_long_main > 0) {
      FwdArgs<float> la = a;
      la.chunk_lanes = reinterpret_cast<const LaneSlot*>(dp + L.long_lanes);
      la.jobs = reinterpret_cast<const FwdJob*>(dp + L.long_jobs);
      la.job_count = reinterpret_cast<const int32_t*>(dp + L.long_count);
      la.job_next = c->counters.as<int32_t>() + 7;
      if (rpl_main == 4) launch_long<float, 4>(la, fma, n_long_waves, c->carry.as<float>(), carry_len, s);
      else               launch_long<float, kRplF32>(la, fma, n_long_waves, c->carry.as<float>(), carry_len, s);
    }
    if (ev) HIP_TRY(hipEventRecord(c->ev[2], s));
    hipLaunchKernelGGL(policy_kernel, dim3((unsigned)((n_pairs + kPolicyBlock - 1) / kPolicyBlock)), dim3(kPolicyBlock), 0, s, fa);
    HIP_TRY(hipEventRecord(c->policy_done, s));
    const bool side_finalize = finalize_mode == GKLHIP_FINALIZE_DEVICE_F64 || finalize_mode == GKLHIP_FINALIZE_DEVICE_REF32;
    if (side_finalize) {
      HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->policy_done, 0));
      hipLaunchKernelGGL(finalize32_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, c->copy_stream, fa);
      HIP_TRY(hipEventRecord(c->early_copy_done, c->copy_stream));
    }
    if (c->peek_after_policy) {
      if ((rc = c->peek_count.reserve(64))) return rc;
      HIP_TRY(hipMemcpyAsync(c->res_pin.p, out_dev, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(c->peek_count.p, c->counters.p, 4, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      if (*c->peek_count.as<int32_t>() == 0) {
        c->fallback_skipped = true;
        if (ev) { HIP_TRY(hipEventRecord(c->ev[3], s)); HIP_TRY(hipEventRecord(c->ev[4], s)); }
      }
    }
    if (!c->fallback_skipped) {
    // ---- fp64 recomputation of the underflowed pairs ----
    FwdArgs<double> d{};
    fill_common(d);
    d.tab = c->dt64;
    d.y0 = reinterpret_cast<const double*>(dp + L.y0_64);
    d.raw = c->raw64.as<double>();
    if (ev) HIP_TRY(hipEventRecord(c->ev[3], s));
    // Plan the pass on the device (no host round trip): order affected reads by fallback count,
    // pack them into chunks, queue one job per (chunk, needed haplotype run), then let persistent
    // wavefronts stream those jobs -- same WaveJob template as the main pass, T = double.
    const size_t n_groups = plan.groups.size();
    if ((rc = c->fail_order.reserve((size_t)n_reads * 4))) return rc;
    if ((rc = c->lanes2.reserve((size_t)n_reads * kLanes * sizeof(LaneSlot)))) return rc;
    const size_t max_jobs = (size_t)n_reads * ((size_t)(n_haps + 1) / 2 + n_groups);
    if ((rc = c->jobs.reserve(2 * max_jobs * sizeof(FwdJob)))) return rc;  // as built + sorted by length
    int32_t* hist = c->fail_hist.as<int32_t>();
    int32_t* pos = hist + (n_haps + 2);
    int32_t* cnts = c->counters.as<int32_t>();  // [0] pairs [2] jobs [3] next job [4] fail reads [5] chunks
    const unsigned rb = (unsigned)((n_reads + 255) / 256);
    hipLaunchKernelGGL(fail_hist_kernel, dim3(rb), dim3(256), 0, s, c->read_fail.as<int32_t>(), n_reads, hist,
                       b.read_off, kLanes * rpl64 - 1);
    hipLaunchKernelGGL(fail_scan_kernel, dim3(1), dim3(64), 0, s, hist, n_haps, pos, cnts + 4);
    hipLaunchKernelGGL(fail_scatter_kernel, dim3(rb), dim3(256), 0, s, c->read_fail.as<int32_t>(), n_reads, pos,
                       c->fail_order.as<int32_t>(), b.read_off, kLanes * rpl64 - 1);
    hipLaunchKernelGGL(pack_windows_kernel, dim3((unsigned)((n_reads + kPackWindow - 1) / kPackWindow)), dim3(64), 0, s,
                       c->fail_order.as<int32_t>(), cnts + 4, b.read_off, rpl64, c->lanes2.as<LaneSlot>(), cnts + 5);
    const int jb_threads = n_haps <= 64 ? 64 : n_haps <= 128 ? 128 : 256;
    hipLaunchKernelGGL(build_jobs_kernel, dim3((unsigned)std::min(n_reads, 2048)), dim3(jb_threads),
                       (size_t)(kLanes + 1) * 4 + (size_t)n_haps, s, c->lanes2.as<LaneSlot>(), cnts + 5,
                       c->used64.as<uint8_t>(), n_haps, reinterpret_cast<const int32_t*>(dp + L.hap_orig),
                       reinterpret_cast<const int32_t*>(dp + L.hap_group), c->jobs.as<FwdJob>(), cnts + 2);
    hipLaunchKernelGGL(sort_jobs_kernel, dim3(1), dim3(1024), 0, s, c->jobs.as<FwdJob>(), cnts + 2,
                       reinterpret_cast<const int32_t*>(dp + L.hap_pos), reinterpret_cast<const int32_t*>(dp + L.hap_len),
                       c->jobs.as<FwdJob>() + max_jobs);
    d.chunk_lanes = c->lanes2.as<LaneSlot>();
    d.n_chunks = n_reads;  // upper bound; the job list only names packed chunks
    d.jobs = c->jobs.as<FwdJob>() + max_jobs;
    launch_jobs<double, kRplF64>(d, fma, (int)std::min<int64_t>(n_pairs, 256 * 16), s);
    if (n_long64 > 0) {
      // reads too long for a chunk: one pseudo-chunk each, same run detection, striped kernel
      if ((rc = c->jobs_long.reserve((size_t)n_long64 * ((size_t)(n_haps + 1) / 2 + n_groups) * sizeof(FwdJob)))) return rc;
      const LaneSlot* pl = reinterpret_cast<const LaneSlot*>(dp + L.long_lanes);
      hipLaunchKernelGGL(build_jobs_kernel, dim3((unsigned)n_long64), dim3(jb_threads),
                         (size_t)(kLanes + 1) * 4 + (size_t)n_haps, s, pl + (size_t)n_long_main * kLanes,
                         reinterpret_cast<const int32_t*>(dp + L.long_count) + 2, c->used64.as<uint8_t>(), n_haps,
                         reinterpret_cast<const int32_t*>(dp + L.hap_orig),
                         reinterpret_cast<const int32_t*>(dp + L.hap_group), c->jobs_long.as<FwdJob>(), cnts + 8);
      FwdArgs<double> ld = d;
      ld.chunk_lanes = pl + (size_t)n_long_main * kLanes;
      ld.jobs = c->jobs_long.as<FwdJob>();
      ld.job_count = cnts + 8;
      ld.job_next = cnts + 9;
      launch_long<double, kRplF64>(ld, fma, n_long_waves, c->carry.as<double>(), carry_len, s);
    }
    if (ev) HIP_TRY(hipEventRecord(c->ev[4], s));
    hipLaunchKernelGGL(finalize64_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, s, fa, 1);
    }  // !fallback_skipped
    if (side_finalize) HIP_TRY(hipStreamWaitEvent(s, c->early_copy_done, 0));  // join the side stream
  }
  if (ev) HIP_TRY(hipEventRecord(c->ev[5], s));
  HIP_TRY(hipGetLastError());

  HIP_TRY(hipEventRecord(c->plan_unused_slot[slot], s));
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
  } else {
    st.n_fallback = use_double ? n_pairs : -1;  // unknown without a sync; gklhip_get_raw fills it in
  }
  return GKLHIP_OK;
}

}  // namespace

// ------------------------------------------------------------------ C ABI
extern "C" {

int gklhip_abi_version(void) { return GKLHIP_ABI_VERSION; }

const char* gklhip_last_error(void) { return g_err.c_str(); }

const char* gklhip_strerror(int status) {
  switch (status) {
    case GKLHIP_OK: return "ok";
    case GKLHIP_ERR_INVALID_ARG: return "invalid argument";
    case GKLHIP_ERR_NO_DEVICE: return "no usable HIP device";
    case GKLHIP_ERR_OOM: return "out of memory";
    case GKLHIP_ERR_HIP: return "HIP runtime error";
    case GKLHIP_ERR_UNSUPPORTED: return "unsupported input";
    default: return "unknown status";
  }
}

int gklhip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

int gklhip_init(const gklhip_config* cfg, gklhip_ctx** out_ctx) {
  if (!out_ctx) return fail(GKLHIP_ERR_INVALID_ARG, "out_ctx is NULL");
  *out_ctx = nullptr;
  gklhip_config c0;
  memset(&c0, 0, sizeof c0);
  c0.abi_version = GKLHIP_ABI_VERSION; c0.device = -1; c0.max_threads = 1; c0.fma_mode = 1; c0.finalize = -1;
  if (cfg) {
    if (cfg->abi_version != GKLHIP_ABI_VERSION)
      return fail(GKLHIP_ERR_INVALID_ARG, "ABI version %d, library is %d", cfg->abi_version, GKLHIP_ABI_VERSION);
    c0 = *cfg;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return fail(GKLHIP_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU compute path)");
  }
  int dev = c0.device;
  if (dev < 0) { HIP_TRY(hipGetDevice(&dev)); }
  if (dev >= ndev) return fail(GKLHIP_ERR_INVALID_ARG, "device %d of %d", dev, ndev);
  HIP_TRY(hipSetDevice(dev));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(GKLHIP_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", dev, prop.gcnArchName);
  gklhip_ctx* c = new (std::nothrow) gklhip_ctx();
  if (!c) return fail(GKLHIP_ERR_OOM, "context allocation failed");
  c->cfg = c0;
  c->device = dev;
  memset(&c->stats, 0, sizeof c->stats);
  int rc = GKLHIP_OK;
  auto bail = [&](int status) { gklhip_done(c); return status; };
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(GKLHIP_ERR_HIP, "hipStreamCreate failed"));
  if (hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(GKLHIP_ERR_HIP, "hipStreamCreate failed"));
  for (int k = 0; k < 2; k++)
    if (hipEventCreateWithFlags(&c->stage_free_slot[k], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->plan_unused_slot[k], hipEventDisableTiming) != hipSuccess)
      return bail(fail(GKLHIP_ERR_HIP, "hipEventCreate failed"));
  if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(GKLHIP_ERR_HIP, "hipStreamCreate failed"));
  if (hipEventCreateWithFlags(&c->policy_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->early_copy_done, hipEventDisableTiming) != hipSuccess)
    return bail(fail(GKLHIP_ERR_HIP, "hipEventCreate failed"));
  for (int k = 0; k < 2; k++)
    if (hipEventRecord(c->stage_free_slot[k], c->stream) != hipSuccess || hipEventRecord(c->plan_unused_slot[k], c->stream) != hipSuccess)
      return bail(fail(GKLHIP_ERR_HIP, "hipEventRecord failed"));
  {
    const int sets = c->cfg.record_events == 2 ? gklhip_ctx::kEventRing : 1;
    for (int k = 0; k < sets; k++)
      for (auto& e : c->ev_ring[k])
        if (hipEventCreate(&e) != hipSuccess) return bail(fail(GKLHIP_ERR_HIP, "hipEventCreate failed"));
  }
  if ((rc = upload_tables(c, host_tables_f32(), &c->tab32, &c->dt32))) return bail(rc);
  if ((rc = upload_tables(c, host_tables_f64(), &c->tab64, &c->dt64))) return bail(rc);
  *out_ctx = c;
  return GKLHIP_OK;
}

int gklhip_done(gklhip_ctx* c) {
  if (!c) return GKLHIP_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (DevBuf* b : {&c->tab32, &c->tab64, &c->plan_dev_slot[0], &c->plan_dev_slot[1], &c->raw32, &c->raw64, &c->used64, &c->list,
                    &c->counters, &c->stream_buf, &c->read_off_dev, &c->out_dev, &c->batch_dev, &c->read_fail,
                    &c->lanes2, &c->jobs, &c->jobs_long, &c->fail_order, &c->fail_hist, &c->carry, &c->res_dev, &c->hap_flags})
    b->release();
  c->stage_slot[0].release();
  c->stage_slot[1].release();
  c->res_pin.release();
  c->res_pin2.release();
  c->batch_stage.release();
  c->peek_count.release();
  if (c->policy_done) (void)hipEventDestroy(c->policy_done);
  if (c->early_copy_done) (void)hipEventDestroy(c->early_copy_done);
  if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
  for (auto& set : c->ev_ring)
    for (auto& e : set) if (e) (void)hipEventDestroy(e);
  for (int k = 0; k < 2; k++) {
    if (c->stage_free_slot[k]) (void)hipEventDestroy(c->stage_free_slot[k]);
    if (c->plan_unused_slot[k]) (void)hipEventDestroy(c->plan_unused_slot[k]);
  }
  if (c->upload_stream) { (void)hipStreamSynchronize(c->upload_stream); (void)hipStreamDestroy(c->upload_stream); }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return GKLHIP_OK;
}

int gklhip_compute_device(gklhip_ctx* c, const gklhip_batch* dev_batch, double* out_dev, void* hip_stream) {
  if (!c) return fail(GKLHIP_ERR_INVALID_ARG, "context is NULL (initNative not called)");
  int rc = validate(dev_batch);
  if (rc) return rc;
  if (!out_dev && (int64_t)dev_batch->n_reads * dev_batch->n_haps > 0)
    return fail(GKLHIP_ERR_INVALID_ARG, "output array is NULL");
  std::lock_guard<std::mutex> lock(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  int mode = c->cfg.finalize;
  if (mode != GKLHIP_FINALIZE_DEVICE_F64 && mode != GKLHIP_FINALIZE_DEVICE_REF32) mode = GKLHIP_FINALIZE_DEVICE_F64;
  hipStream_t s = static_cast<hipStream_t>(hip_stream);  // NULL = HIP's default stream
  return run_device(c, dev_batch, out_dev, mode, s);
}

int gklhip_compute(gklhip_ctx* c, const gklhip_batch* hb, double* out_host) {
  if (!c) return fail(GKLHIP_ERR_INVALID_ARG, "context is NULL (initNative not called)");
  int rc = validate(hb);
  if (rc) return rc;
  const int64_t n_pairs = (int64_t)hb->n_reads * hb->n_haps;
  if (n_pairs == 0) return GKLHIP_OK;
  if (!out_host) return fail(GKLHIP_ERR_INVALID_ARG, "output array is NULL");
  std::lock_guard<std::mutex> lock(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  // H2D of the six byte arrays (one allocation, 256-byte aligned sub-buffers)
  const size_t rl = (size_t)hb->read_off[hb->n_reads], hl = (size_t)hb->hap_off[hb->n_haps];
  const size_t stride = align_up(rl);
  if ((rc = c->batch_dev.reserve(5 * stride + align_up(hl)))) return rc;
  unsigned char* d = c->batch_dev.as<unsigned char>();
  const uint8_t* srcs[5] = {hb->read_bases, hb->read_quals, hb->ins_gop, hb->del_gop, hb->gcp};
  const size_t all_bytes = 5 * stride + align_up(hl);
  if (all_bytes <= kSmallBatchBytes) {
    // a GATK-sized call: gather the six arrays in one pinned block and pay ONE copy launch instead of six
    // (each costs the host ~6 us; the block is tiny).  The previous call has completed (this entry point
    // synchronises before it returns), so the block is free.
    if ((rc = c->batch_stage.reserve(all_bytes))) return rc;
    unsigned char* hs = c->batch_stage.as<unsigned char>();
    for (int i = 0; i < 5; i++) memcpy(hs + i * stride, srcs[i], rl);
    memcpy(hs + 5 * stride, hb->hap_bases, hl);
    HIP_TRY(hipMemcpyAsync(d, hs, all_bytes, hipMemcpyHostToDevice, s));
  } else {
    for (int i = 0; i < 5; i++) HIP_TRY(hipMemcpyAsync(d + i * stride, srcs[i], rl, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d + 5 * stride, hb->hap_bases, hl, hipMemcpyHostToDevice, s));
  }
  gklhip_batch db = *hb;
  db.read_bases = d; db.read_quals = d + stride; db.ins_gop = d + 2 * stride;
  db.del_gop = d + 3 * stride; db.gcp = d + 4 * stride; db.hap_bases = d + 5 * stride;
  const int mode = c->cfg.finalize;
  const bool on_device = (mode == GKLHIP_FINALIZE_DEVICE_F64 || mode == GKLHIP_FINALIZE_DEVICE_REF32);
  if (on_device) {
    if ((rc = c->out_dev.reserve((size_t)n_pairs * 8))) return rc;
    if ((rc = run_device(c, &db, c->out_dev.as<double>(), mode, s))) return rc;
    HIP_TRY(hipMemcpyAsync(out_host, c->out_dev.p, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return GKLHIP_OK;
  }
  // reference-exact finalisation on the host: one packed 8-byte word per pair comes back through pinned
  // memory.  The fp32 results are final as soon as the policy kernel has run, so they are copied out on a
  // second stream and finalised by the host WHILE the fp64 recomputation pass runs; only the recomputed
  // pairs are left for after the last kernel.
  int threads = c->cfg.max_threads;
  if (threads <= 0) threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  const size_t bytes = (size_t)n_pairs * 8;
  if ((rc = c->res_dev.reserve(bytes))) return rc;
  if ((rc = c->res_pin.reserve(bytes))) return rc;
  HostFinalizer fin;
  if (c->cfg.use_double) {
    if ((rc = run_device(c, &db, c->res_dev.as<double>(), kModePacked, s))) return rc;
    HIP_TRY(hipMemcpyAsync(c->res_pin.p, c->res_dev.p, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    c->stats.n_fallback = fin.all(&c->workers, c->res_pin.as<uint64_t>(), out_host, n_pairs, threads);
    return GKLHIP_OK;
  }
  if (n_pairs <= kPeekPairs) {
    // GATK-sized call: one D2H after the policy kernel; no underflowed pair (the usual case) = done
    c->peek_after_policy = true;
    rc = run_device(c, &db, c->res_dev.as<double>(), kModePacked, s);
    c->peek_after_policy = false;
    if (rc) return rc;
    if (c->fallback_skipped) {
      c->stats.n_fallback = fin.all(&c->workers, c->res_pin.as<uint64_t>(), out_host, n_pairs, threads);
      return GKLHIP_OK;
    }
    fin.early(&c->workers, c->res_pin.as<uint64_t>(), out_host, n_pairs, threads);  // overlaps the fp64 pass
    if ((rc = c->res_pin2.reserve(bytes))) return rc;
    HIP_TRY(hipMemcpyAsync(c->res_pin2.p, c->res_dev.p, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    c->stats.n_fallback = fin.late(&c->workers, c->res_pin2.as<uint64_t>(), out_host, threads);
    return GKLHIP_OK;
  }
  if ((rc = c->res_pin2.reserve(bytes))) return rc;
  if ((rc = run_device(c, &db, c->res_dev.as<double>(), kModePacked, s))) return rc;  // records policy_done
  HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->policy_done, 0));
  HIP_TRY(hipMemcpyAsync(c->res_pin.p, c->res_dev.p, bytes, hipMemcpyDeviceToHost, c->copy_stream));
  HIP_TRY(hipEventRecord(c->early_copy_done, c->copy_stream));
  HIP_TRY(hipMemcpyAsync(c->res_pin2.p, c->res_dev.p, bytes, hipMemcpyDeviceToHost, s));  // after the last kernel
  HIP_TRY(hipEventSynchronize(c->early_copy_done));
  fin.early(&c->workers, c->res_pin.as<uint64_t>(), out_host, n_pairs, threads);
  HIP_TRY(hipStreamSynchronize(s));
  c->stats.n_fallback = fin.late(&c->workers, c->res_pin2.as<uint64_t>(), out_host, threads);
  return GKLHIP_OK;
}

void* gklhip_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

void gklhip_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int gklhip_get_step_times(gklhip_ctx* c, int32_t steps_back, float* ms_main, float* ms_fallback, float* ms_total) {
  if (!c) return fail(GKLHIP_ERR_INVALID_ARG, "context is NULL");
  std::lock_guard<std::mutex> lock(c->mu);
  if (c->cfg.record_events != 2) return fail(GKLHIP_ERR_INVALID_ARG, "context was not created with record_events = 2");
  if (steps_back < 0 || steps_back >= gklhip_ctx::kEventRing || steps_back >= c->calls)
    return fail(GKLHIP_ERR_INVALID_ARG, "steps_back %d outside the %d recorded calls", steps_back,
                (int)std::min<int64_t>(c->calls, gklhip_ctx::kEventRing));
  HIP_TRY(hipSetDevice(c->device));
  const int slot = (int)((c->calls - 1 - steps_back) % gklhip_ctx::kEventRing);
  hipEvent_t* e = c->ev_ring[slot];
  HIP_TRY(hipEventSynchronize(e[5]));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, e[1], e[2])); if (ms_main) *ms_main = ms;
  HIP_TRY(hipEventElapsedTime(&ms, e[3], e[4])); if (ms_fallback) *ms_fallback = c->ring_double[slot] ? 0.f : ms;
  HIP_TRY(hipEventElapsedTime(&ms, e[0], e[5])); if (ms_total) *ms_total = ms;
  return GKLHIP_OK;
}

int gklhip_get_stats(gklhip_ctx* c, gklhip_stats* out) {
  if (!c || !out) return fail(GKLHIP_ERR_INVALID_ARG, "NULL argument");
  std::lock_guard<std::mutex> lock(c->mu);
  *out = c->stats;
  return GKLHIP_OK;
}

int gklhip_get_raw(gklhip_ctx* c, float* raw32, double* raw64, uint8_t* used64) {
  if (!c) return fail(GKLHIP_ERR_INVALID_ARG, "context is NULL");
  std::lock_guard<std::mutex> lock(c->mu);
  if (!c->have_last) return fail(GKLHIP_ERR_INVALID_ARG, "no completed call to read back");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = c->last_stream;
  const size_t n = (size_t)c->last_pairs;
  if (raw32 && !c->cfg.use_double) HIP_TRY(hipMemcpyAsync(raw32, c->raw32.p, n * 4, hipMemcpyDeviceToHost, s));
  if (raw64) HIP_TRY(hipMemcpyAsync(raw64, c->raw64.p, n * 8, hipMemcpyDeviceToHost, s));
  if (used64) HIP_TRY(hipMemcpyAsync(used64, c->used64.p, n, hipMemcpyDeviceToHost, s));
  int32_t cnt[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(cnt, c->counters.p, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  c->stats.n_fallback = c->cfg.use_double ? (int64_t)n : cnt[0];
  return GKLHIP_OK;
}

int gklhip_plan_describe(int32_t n_reads, int32_t n_haps, const int64_t* read_off, const int64_t* hap_off,
                         int32_t rows_per_lane, int32_t* lanes_out, int64_t lanes_cap, int32_t* n_groups_out,
                         int32_t* n_long_out) {
  if (n_reads < 0 || n_haps < 0 || !read_off || !hap_off || (rows_per_lane != 4 && rows_per_lane != 8))
    return -fail(GKLHIP_ERR_INVALID_ARG, "bad arguments to gklhip_plan_describe");
  Plan p;
  build_plan(n_reads, n_haps, read_off, hap_off, rows_per_lane, kTargetCols, &p);
  if (n_groups_out) *n_groups_out = (int32_t)p.groups.size();
  if (n_long_out) *n_long_out = (int32_t)p.long_reads.size();
  if (lanes_out) {
    const int64_t n = std::min<int64_t>(lanes_cap, (int64_t)p.lanes.size());
    for (int64_t i = 0; i < n; i++) { lanes_out[2 * i] = p.lanes[i].read; lanes_out[2 * i + 1] = p.lanes[i].block; }
  }
  return p.n_chunks;
}

int64_t gklhip_get_table_f32(int which, float* dst, int64_t cap) {
  const HostTables<float>& t = host_tables_f32();
  const std::vector<float>* v = which == 0 ? &t.ph2pr : which == 1 ? &t.mm : which == 2 ? &t.div3 : nullptr;
  if (!v) return -1;
  if (dst) memcpy(dst, v->data(), sizeof(float) * (size_t)std::min<int64_t>(cap, (int64_t)v->size()));
  return (int64_t)v->size();
}
int64_t gklhip_get_table_f64(int which, double* dst, int64_t cap) {
  const HostTables<double>& t = host_tables_f64();
  const std::vector<double>* v = which == 0 ? &t.ph2pr : which == 1 ? &t.mm : which == 2 ? &t.div3 : nullptr;
  if (!v) return -1;
  if (dst) memcpy(dst, v->data(), sizeof(double) * (size_t)std::min<int64_t>(cap, (int64_t)v->size()));
  return (int64_t)v->size();
}

}  // extern "C"
