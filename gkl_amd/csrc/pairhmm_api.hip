// C-ABI implementation (include/gkl_hip_pairhmm.h) of the MI355X PairHMM forward path:
// context / tables / planning / kernel launches / precision policy / finalisation.
//
// Reference counterparts (src/main/native/pairhmm): IntelPairHmm.cc:55-118 (init),
// :150-169 (batch loop + fp32->fp64 policy + log10), :189-192 (done).  There is no
// CPU compute path in this library: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <rccl/rccl.h>  // types only: the library is dlopen()ed by multi-device contexts
#include <dlfcn.h>

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
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <deque>
#include <type_traits>
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

// No C++ exception leaves the C ABI or a helper thread of this library (a std::bad_alloc from a plan vector inside a
// JVM would otherwise be std::terminate): entry points and thread bodies run their work through guarded().
int fail_noexcept(int status, const char* msg) noexcept {
  try { g_err = msg; } catch (...) {}
  return status;
}
template <typename F>
int guarded(F&& body) noexcept {
  try { return body(); }
  catch (const std::bad_alloc&) { return fail_noexcept(GKLHIP_ERR_OOM, "host memory allocation failed"); }
  catch (const std::exception& e) { return fail_noexcept(GKLHIP_ERR_HIP, e.what()); }
  catch (...) { return fail_noexcept(GKLHIP_ERR_HIP, "unexpected C++ exception"); }
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

// Device / pinned-host buffers of a context: they grow with the biggest call and shrink again when the calls stay small --
// a buffer above kTrimFloor that the last kTrimCalls calls each needed less than a quarter of is given back and re-made at
// the size in use (one 1.28 M-pair call must not pin ~100 MB per slot for the life of the JVM).  hipFree / hipHostFree wait
// for the device to finish with the memory, exactly as on the grow path.
constexpr size_t kTrimFloor = (size_t)32 << 20;
constexpr int kTrimCalls = 16;
inline bool trim_due(size_t n, size_t cap, int* small_uses) {
  if (cap <= kTrimFloor || n >= cap / 4) { *small_uses = 0; return false; }
  return ++*small_uses >= kTrimCalls;
}
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int small_uses = 0;
  int reserve(size_t n) {
    if (n <= cap && !trim_due(n, cap, &small_uses)) return GKLHIP_OK;
    small_uses = 0;
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
  int small_uses = 0;
  int reserve(size_t n) {
    if (n <= cap && !trim_due(n, cap, &small_uses)) return GKLHIP_OK;
    small_uses = 0;
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
// One device's engine: streams, tables, grow-only scratch.  The public gklhip_ctx owns one of these per
// device of its list (one for the usual single-device context).
struct DevCtx {
  gklhip_config cfg;
  int device = 0;
  int n_cus = 256;
  int n_xcds = 8;   // hipDeviceAttributeNumberOfXccs: workgroups go to the XCDs round-robin by index
  // development / cross-check switches, read from the environment ONCE per context (dev_init), never on a call path
  // (getenv is not safe against a concurrent setenv in the host JVM): GKLHIP_ASM_GENERAL=0 (round-3 arrangement: C++
  // general steps), GKLHIP_SPECULATE_FP64=1 (fp64 beside fp32 for a lone tiny call)
  int asm_general = 1;
  int speculate_fp64 = 0;
  int lds_oob_zero = 1;   // dev_init's self-test: a DS read beyond the allocation returns 0 here (the fp32 programs' separator priors)
  hipStream_t stream = nullptr;
  // tables
  DevBuf tab32, tab64;
  DevTables<float> dt32;
  DevTables<double> dt64;
  // per-call plan uploads (pinned staging -> device)
  // Two slots alternate from call to call: the plan of call k+1 is staged and uploaded (own stream) while the
  // kernels of call k still read theirs -- back-to-back batches then never wait for the plan block.
  PinBuf stage_slot[2];
  DevBuf plan_dev_slot[2];
  hipEvent_t stage_free_slot[2] = {nullptr, nullptr};   // the slot's upload has left the staging buffer
  hipEvent_t plan_unused_slot[2] = {nullptr, nullptr};  // the last call that used the slot's device copy has finished
  hipStream_t upload_stream = nullptr;
  hipStream_t pad_stream = nullptr;   // never used: keeps the context's stream count at four once copy_stream exists (aux_streams)
  int plan_slot = 0;
  // per-call device scratch
  DevBuf raw32, raw64, used64, counters, stream_buf, out_dev;
  DevBuf lanes_main;  // the main pass's lane map, expanded by prep_kernel from the plan's compact read packing
  DevBuf read_fail, lanes2, jobs, jobs_long, fail_order, fail_hist, hap_flags;
  // host-API device copies of the batch, packed results (device + pinned), finalisation workers
  DevBuf batch_dev;
  PinBuf res_pin;
  WorkerPool workers;
  hipStream_t copy_stream = nullptr;  // early D2H of the fp32 results / device log10 of the kept pairs while the fp64 pass runs
  hipEvent_t policy_done = nullptr, early_copy_done = nullptr;
  // scratch is per context and ordered by the stream of the call that uses it: a call on a different stream than
  // the previous one first waits for that one's end
  hipEvent_t call_done = nullptr;
  bool have_call_done = false;
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
  std::vector<int64_t> sub_read_off;  // multi-device: this device's read range, offsets rebased to 0
  DevBuf carry;
};

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
// compute wavefronts of a super-stripe workgroup (+ 1 helper): fp32 (149 VGPRs: three wavefronts per SIMD) 5 + 1, two workgroups
// per CU; fp64 (252 VGPRs: two per SIMD, 16 KB tables) 7 + 1, one per CU
template <typename T> constexpr int super_waves() { return 7; }
template <typename T> constexpr int super_blocks_max() { return sizeof(T) == 8 ? 256 : 512; }
// carry rows of the super-stripe kernel: two per workgroup, one 32-byte slot per step of the deepest array's longest stream
inline int64_t super_steps(int carry_len, int max_read_len, int rpl) { return (int64_t)carry_len + 64 * (int64_t)((blocks_for(max_read_len, rpl) + kLanes - 1) / kLanes); }
template <typename T, int RPL, int RPL_STRIPED>
void launch_long_jobs(const FwdArgs<T>& a, int fma, int n_blocks, int max_read_len, T* carry, int carry_len, hipStream_t s,
                      unsigned char* xcarry = nullptr, int64_t xsteps = 0, int32_t* next2 = nullptr) {
  static const bool wide_env = [] { const char* v = getenv("GKLHIP_WIDE_LONG"); return !v || atoi(v) != 0; }();
  static const bool super_env = [] { const char* v = getenv("GKLHIP_SUPER_LONG"); return !v || atoi(v) != 0; }();
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
  static const int load_env = [] { const char* v = getenv("GKL_HIP_COMBINE_LOAD"); return v ? atoi(v) : 0; }();
  // (about half of the calls inside the library are on the device at any moment, the others are being staged or
  //  finalised: 16 callers of 100 x 10 regions keep the 4-row kernel -- the 8-row one needs three wavefronts per SIMD
  //  to pay, tools/small_scaling.py -- and 32 callers get the 8-row one)
  const int load = !defer ? 1 : load_env > 0 ? load_env : std::max(1, g_host_calls_in_flight.load(std::memory_order_relaxed) / 2);
  const int rpl_main = use_double ? rpl64 : pick_f32_rpl(c->cfg.rows_per_lane, n_reads, n_haps, db->read_off, db->hap_off, load);
  static const int target_cols_env = [] { const char* v = getenv("GKLHIP_TARGET_COLS"); return v ? atoi(v) : 0; }();
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
  static const bool fused_env = [] { const char* v = getenv("GKLHIP_FUSED_PAIRS"); return !v || atoi(v) != 0; }();
  static const int64_t fused_max = [] { const char* v = getenv("GKLHIP_FUSED_MAX_PAIRS"); return v ? atoll(v) : (long long)kTwoStepFrom; }();
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
  static const bool timing = getenv("GKLHIP_TIMING") != nullptr;
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
  static const bool xcd_env = [] { const char* v = getenv("GKLHIP_XCD_AWARE"); return !v || atoi(v) != 0; }();
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
      static const int wanted_env = [] { const char* v = getenv("GKLHIP_FB_WANTED_JOBS"); return v ? atoi(v) : 0; }();
      // (a shard of the batch wants fewer, longer jobs: 4096 for an eighth, measured on the 1250 x 128 shard)
      pa.wanted_jobs = wanted_env > 0 ? wanted_env : (int)std::min<int64_t>(kFallbackWantedJobs, std::max<int64_t>(4096, n_pairs / 100));
      pa.min_job_cols = 256;
      pa.packed_by_kernels = fold_packed ? 1 : 0;
      // Three stream-ordered launches (pairhmm_aux_kernels.h): no block waits for another, so nothing limits how many
      // of these are in flight per device or process.  The policy takes a block per 4096 pairs (up to one per CU), the
      // packing a wavefront per window of affected reads, the run detection a wavefront per chunk (grid-stride).
      static const int blocks_env = [] { const char* v = getenv("GKLHIP_PLAN_BLOCKS"); return v ? atoi(v) : 0; }();
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

// ------------------------------------------------------------------ one device: lifecycle + host-buffer call
void dev_done(DevCtx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->last_stream && c->have_last) (void)hipStreamSynchronize(c->last_stream);
  for (DevBuf* b : {&c->tab32, &c->tab64, &c->plan_dev_slot[0], &c->plan_dev_slot[1], &c->raw32, &c->raw64, &c->used64,
                    &c->counters, &c->stream_buf, &c->out_dev, &c->batch_dev, &c->read_fail, &c->lanes_main,
                    &c->lanes2, &c->jobs, &c->jobs_long, &c->fail_order, &c->fail_hist, &c->carry, &c->hap_flags})
    b->release();
  c->stage_slot[0].release();
  c->stage_slot[1].release();
  c->res_pin.release();
  if (c->policy_done) (void)hipEventDestroy(c->policy_done);
  if (c->early_copy_done) (void)hipEventDestroy(c->early_copy_done);
  if (c->call_done) (void)hipEventDestroy(c->call_done);
  if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
  for (auto& set : c->ev_ring)
    for (auto& e : set) if (e) (void)hipEventDestroy(e);
  for (int k = 0; k < 2; k++) {
    if (c->stage_free_slot[k]) (void)hipEventDestroy(c->stage_free_slot[k]);
    if (c->plan_unused_slot[k]) (void)hipEventDestroy(c->plan_unused_slot[k]);
  }
  if (c->upload_stream) { (void)hipStreamSynchronize(c->upload_stream); (void)hipStreamDestroy(c->upload_stream); }
  if (c->pad_stream) (void)hipStreamDestroy(c->pad_stream);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int dev_init(const gklhip_config& cfg, int dev, int ndev, DevCtx** out) {
  *out = nullptr;
  if (dev < 0 || dev >= ndev) return fail(GKLHIP_ERR_INVALID_ARG, "device %d of %d", dev, ndev);
  HIP_TRY(hipSetDevice(dev));
  {
    // GKL_HIP_SCHEDULE=spin|yield|blocking: how host threads wait for the device (hipSetDeviceFlags); default: HIP's own
    static const char* sched = getenv("GKL_HIP_SCHEDULE");
    if (sched && *sched) {
      const unsigned f = strcmp(sched, "yield") == 0 ? hipDeviceScheduleYield : strcmp(sched, "blocking") == 0 ? hipDeviceScheduleBlockingSync
                         : strcmp(sched, "spin") == 0 ? hipDeviceScheduleSpin : hipDeviceScheduleAuto;
      (void)hipSetDeviceFlags(f);
      (void)hipGetLastError();
    }
  }
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(GKLHIP_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", dev, prop.gcnArchName);
  DevCtx* c = new (std::nothrow) DevCtx();
  if (!c) return fail(GKLHIP_ERR_OOM, "context allocation failed");
  c->cfg = cfg;
  c->device = dev;
  c->n_cus = std::max(1, prop.multiProcessorCount);
  {
    int xccs = 0;
    if (hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess) { (void)hipGetLastError(); xccs = 8; }
    c->n_xcds = std::max(1, std::min(xccs, 64));
  }
  memset(&c->stats, 0, sizeof c->stats);
  int rc = GKLHIP_OK;
  auto bail = [&](int status) { dev_done(c); return status; };
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(GKLHIP_ERR_HIP, "hipStreamCreate failed"));
  {
    // once per process and device: is this build's denormal mode the one the kernels (and the reference) assume?
    static std::mutex mu;
    static std::vector<int> checked;   // 0 unknown, 1 good, -1 bad
    std::lock_guard<std::mutex> l(mu);
    if ((int)checked.size() <= dev) checked.resize((size_t)dev + 1, 0);
    static std::vector<int> oob_checked;   // the same for "a DS read beyond the LDS allocation returns 0"
    if ((int)oob_checked.size() <= dev) oob_checked.resize((size_t)dev + 1, 0);
    if (checked[(size_t)dev] == 0 || oob_checked[(size_t)dev] == 0) {
      uint32_t* d_out = nullptr;
      uint32_t h_out[3] = {1u, 1u, 1u};
      if (hipMalloc(reinterpret_cast<void**>(&d_out), 12) != hipSuccess) return bail(fail(GKLHIP_ERR_OOM, "hipMalloc failed"));
      float f_den; double d_den;
      { const uint32_t fb = 1u; memcpy(&f_den, &fb, 4); const uint64_t db = 0x0000000100000001ull; memcpy(&d_den, &db, 8); }
      bool ok = hipMemsetAsync(d_out, 0, 12, c->stream) == hipSuccess;
      hipLaunchKernelGGL(flush_selftest_kernel, dim3(1), dim3(1), 0, c->stream, d_out, f_den, d_den);
      hipLaunchKernelGGL(lds_oob_selftest_kernel, dim3(1024), dim3(256), 0, c->stream, d_out + 2);
      ok = ok && hipGetLastError() == hipSuccess && hipMemcpyAsync(h_out, d_out, 12, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
           hipStreamSynchronize(c->stream) == hipSuccess;
      (void)hipFree(d_out);
      // a HIP failure here says nothing about the build or the chip: the verdicts stay open and the error goes to the caller
      if (!ok) { (void)hipGetLastError(); return bail(fail(GKLHIP_ERR_HIP, "the start-up self-tests could not run on device %d", dev)); }
      checked[(size_t)dev] = h_out[0] == 0u && h_out[1] == 0u ? 1 : -1;
      oob_checked[(size_t)dev] = h_out[2] == 0u ? 1 : -1;
      if (oob_checked[(size_t)dev] < 0)
        fprintf(stderr, "[gklhip] pairhmm: LDS reads beyond the allocation do not return 0 on device %d (%08x): the fp32 general steps stay in C++\n", dev, h_out[2]);
    }
    if (checked[(size_t)dev] < 0)
      return bail(fail(GKLHIP_ERR_HIP, "this library was built without the denormal-flush flags its kernels depend on (gkl_amd/csrc/Makefile: HIPFLAGS)"));
    c->lds_oob_zero = oob_checked[(size_t)dev] > 0 ? 1 : 0;
  }
  {
    const char* ag = getenv("GKLHIP_ASM_GENERAL");
    c->asm_general = ag ? (atoi(ag) != 0) : 1;
    const char* sp = getenv("GKLHIP_SPECULATE_FP64");
    c->speculate_fp64 = sp ? (atoi(sp) != 0) : 0;
    // tests only: GKLHIP_SELFTEST_FAIL=lds_oob makes this context behave as if the self-test above had failed
    const char* sf = getenv("GKLHIP_SELFTEST_FAIL");
    if (sf && strcmp(sf, "lds_oob") == 0) {
      c->lds_oob_zero = 0;
      fprintf(stderr, "[gklhip] pairhmm: GKLHIP_SELFTEST_FAIL=lds_oob: the fp32 general steps stay in C++ for this context\n");
    }
  }
  // A context starts with TWO streams: its own and upload_stream.  copy_stream (device-side finalisation of big device-resident
  // calls) and the combiner's flight streams are made by the first call that needs them.  Why the count matters: a process with
  // one caller of GATK-sized regions (a HaplotypeCaller JVM) only ever uses the first stream, but every stream is a hardware
  // queue, and how many queues each process holds decides how the device's scheduler shares the chip among processes --
  // measured with P such processes on one GPU (tools/proc_scaling.py, docs/NOTES.md 48; GCUPS at 4 / 8 / 16 processes):
  // 1 stream 810 / 1055 / 1325, **2 streams 974 / 1571 / 1596** (p99 of a call 0.19 / 0.26 / 11.6 ms), 3 streams
  // 572 / 707 / 755, 4 streams 979 / 1101 / 1174, the 7 of round 4 965 / 1100 / 1130 (p99 0.19 / 11 / 25-43 ms).
  // ... per PROCESS: the first context of a process opens upload_stream with its own; the contexts after it (the JNI shim's
  // slots of further Java threads) open only their own -- the runtime deals streams onto the process's (four) hardware queues
  // in the order they are made, and with a second stream per context the own streams of four callers shared two queues
  // (4 callers 0.72 -> 0.9 TCUPS with the pool widened to eight queues; this order gets them onto different ones as is).
  static std::atomic<int> contexts_made{0};
  if (contexts_made.fetch_add(1) == 0 &&
      hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(GKLHIP_ERR_HIP, "hipStreamCreate failed"));
  if (const char* v = getenv("GKL_HIP_EAGER_STREAMS")) {   // A/B: 7 = the r04 arrangement (every stream at init); 1..3 = that many spare streams on top of the two
    const int k = atoi(v);
    if (k >= 7) { if (aux_streams(c) != GKLHIP_OK) return bail(GKLHIP_ERR_HIP); }
    else for (int i = 0; i < k; i++) { hipStream_t d = nullptr; (void)hipStreamCreateWithFlags(&d, hipStreamNonBlocking); }   // (leaked on purpose: an experiment)
  }
  for (int k = 0; k < 2; k++)
    if (hipEventCreateWithFlags(&c->stage_free_slot[k], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->plan_unused_slot[k], hipEventDisableTiming) != hipSuccess)
      return bail(fail(GKLHIP_ERR_HIP, "hipEventCreate failed"));
  if (hipEventCreateWithFlags(&c->policy_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->early_copy_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->call_done, hipEventDisableTiming) != hipSuccess)
    return bail(fail(GKLHIP_ERR_HIP, "hipEventCreate failed"));
  for (int k = 0; k < 2; k++)
    if (hipEventRecord(c->stage_free_slot[k], c->stream) != hipSuccess || hipEventRecord(c->plan_unused_slot[k], c->stream) != hipSuccess)
      return bail(fail(GKLHIP_ERR_HIP, "hipEventRecord failed"));
  {
    const int sets = c->cfg.record_events == 2 ? DevCtx::kEventRing : 1;
    for (int k = 0; k < sets; k++)
      for (auto& e : c->ev_ring[k])
        if (hipEventCreate(&e) != hipSuccess) return bail(fail(GKLHIP_ERR_HIP, "hipEventCreate failed"));
  }
  if ((rc = upload_tables(c, host_tables_f32(), &c->tab32, &c->dt32))) return bail(rc);
  if ((rc = upload_tables(c, host_tables_f64(), &c->tab64, &c->dt64))) return bail(rc);
  *out = c;
  return GKLHIP_OK;
}

// Host threads of the reference-exact finalisation.  maxNumberOfThreads caps the OpenMP compute threads of the
// reference's OMP build (IntelPairHmm.cc:72-89; 1 is the default of PairHMMNativeArguments, IntelPairHmm.java:86-90);
// here the compute is on the device and the only host work it can cap is log10f/log10 over the results.  It IS a cap:
// a value >= 1 is honoured as given -- an explicit 1 means ONE finalisation thread per call (bench.py reports what
// that costs a 1.28 M-pair call in host_path.max_threads_1).  Only <= 0 (C ABI: "not set") picks a number here: the
// host-buffer calls in flight in this process then SHARE a budget of min(cores, 8) threads -- one call alone takes all
// of it, the two engines of a pipelined or twin-engine call half each, eight concurrent slots one each.
// GKL_HIP_FINALIZE_THREADS overrides both (per call).
struct HostCallInFlight {
  int share;
  HostCallInFlight() : share(g_host_calls_in_flight.fetch_add(1) + 1) {}
  ~HostCallInFlight() { g_host_calls_in_flight.fetch_sub(1); }
};
int finalize_threads(const DevCtx* c, int share) {
  static const int env = [] { const char* v = getenv("GKL_HIP_FINALIZE_THREADS"); return v ? atoi(v) : 0; }();
  if (env > 0) return env;
  const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
  int threads = c->cfg.max_threads;
  if (threads <= 0) threads = std::max(1, std::min(hw, 8) / std::max(1, share));
  return std::max(1, std::min(threads, 64));
}

// ---- small host-buffer calls of several threads: combined launches ----
// The device executes the kernels of about four hardware queues at a time (tools/ubench_launch.hip: 16 threads with a
// stream each get 4 x the kernel rate of one, not 16 x), so GATK-sized calls from many threads queue up behind each
// other however many streams they use.  A call that arrives while others are in flight therefore waits for a flight
// slot, and the thread that gets the slot launches ALL waiting calls in one set of three kernels (prep_multi_kernel,
// fwd_stream_multi_kernel, pair_policy_multi_kernel: a block finds its call through block offsets in the kernel
// arguments).  A call that finds a free slot and nobody waiting goes out on its own stream exactly as before.
constexpr int kFlightSlots = 4;
struct SmallCombiner {
  struct Ticket {
    const SmallLaunch* sl = nullptr;
    int state = 0;  // 0 queued, 4 taken by a leader, 1 launched (wait for `ev`), 2 failed
    hipEvent_t ev = nullptr;
    int rc = GKLHIP_OK;
    std::string err;
    int64_t t_in = 0;
  };
  struct Slot {
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;
    bool busy = false;
  };
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Ticket*> queue;
  Slot slot[kFlightSlots];
  int device = 0;
  bool streams_made = false;       // the flight streams are created by the first COMBINED launch (make_streams)
  int flights = 0;
  int max_flights = 4;   // (r05, alternating on one box: 4 callers 698-723 -> 740-757 GCUPS, 16 callers 1941-2010 -> 2127-2133 with four instead of three)
  int min_batch = 0;               // 0: by load (see run())
  int64_t batch_wait_ns = 50000;
  int64_t n_calls = 0, n_combined = 0, n_launch_sets = 0;  // diagnostics (gklhip_small_call_counts)
  int64_t ns_queued = 0, ns_launch = 0, ns_sync = 0;
  std::atomic<int64_t> ns_stage{0}, ns_run{0}, ns_finalize{0};  // per call, outside the lock
  static int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

  // The flight streams, created together -- the runtime deals streams round-robin onto the process's hardware queues, so
  // consecutive ones land on different queues and the sets in flight really run side by side -- but only when two calls
  // first meet: a process with ONE caller (a HaplotypeCaller JVM) never needs them, and every stream it does not create is a
  // hardware queue the device's scheduler does not have to rotate in -- with sixteen such processes on one GPU that is the
  // difference between 0.9 and 1.5 TCUPS (docs/NOTES.md 48).  Called with the combiner's lock held.
  void make_streams() {
    if (streams_made) return;
    streams_made = true;
    int prev = 0;
    const bool have_dev = hipGetDevice(&prev) == hipSuccess;
    if (hipSetDevice(device) == hipSuccess) {
      for (auto& sl : slot)
        if (hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) != hipSuccess) {
          sl.stream = nullptr;  // (a set that gets this slot reports the failure)
          (void)hipGetLastError();
        }
    }
    if (have_dev) (void)hipSetDevice(prev);
  }

  int launch_single(const SmallCall& k, hipStream_t s, bool alone) {
    hipLaunchKernelGGL(prep_kernel, dim3((unsigned)k.prep_grid), dim3(kPrepBlock), 0, s, k.prep);
    if (k.fused) {
      launch_pair_fused(k.f, k.d, k.q, k.rows, k.fma, k.n_pairs, s, alone && k.speculate);
    } else {
      launch_main_f32(k.f, k.rpl_main, k.fma, k.main_blocks, s);
      launch_pair_policy(k.d, k.q, k.rows, k.fma, k.n_pairs, s);
    }
    HIP_TRY(hipGetLastError());
    return GKLHIP_OK;
  }
  int launch_multi(Ticket* const* batch, int n, int fma, Slot& sl) {
    MultiArgs mp{}, mf{}, mq{};
    mp.n = mf.n = mq.n = n;
    for (int i = 0; i < n; i++) {
      const SmallLaunch& L = *batch[i]->sl;
      mp.call[i] = L.desc_pinned; mf.call[i] = L.desc_dev; mq.call[i] = L.desc_dev;
      mp.begin[i + 1] = mp.begin[i] + L.call.prep_grid;
      mf.begin[i + 1] = mf.begin[i] + L.call.main_blocks;
      mq.begin[i + 1] = mq.begin[i] + L.call.n_pairs;
    }
    hipLaunchKernelGGL(prep_multi_kernel, dim3((unsigned)mp.begin[n]), dim3(kPrepBlock), 0, sl.stream, mp);
    if (batch[0]->sl->call.fused) {  // (every call of a set is of one kind: the leader only takes calls like its own)
      bool narrow = true;   // reads of at most 255 bases in every call of the set: the four-wavefronts-per-SIMD variant
      for (int i = 0; i < n; i++) narrow = narrow && batch[i]->sl->call.rows <= 4;
      const dim3 grid((unsigned)mq.begin[n]), block(64);
      if (narrow && fma)  hipLaunchKernelGGL((pair_fused_multi_kernel<true, 4>), grid, block, 0, sl.stream, mq);
      else if (narrow)    hipLaunchKernelGGL((pair_fused_multi_kernel<false, 4>), grid, block, 0, sl.stream, mq);
      else if (fma)       hipLaunchKernelGGL((pair_fused_multi_kernel<true, kRplF64>), grid, block, 0, sl.stream, mq);
      else                hipLaunchKernelGGL((pair_fused_multi_kernel<false, kRplF64>), grid, block, 0, sl.stream, mq);
    } else if (fma) {
      hipLaunchKernelGGL((fwd_stream_multi_kernel<true, kRplF32>), dim3((unsigned)mf.begin[n]), dim3(64), 0, sl.stream, mf);
      hipLaunchKernelGGL((pair_policy_multi_kernel<true, kRplF64>), dim3((unsigned)mq.begin[n]), dim3(64), 0, sl.stream, mq);
    } else {
      hipLaunchKernelGGL((fwd_stream_multi_kernel<false, kRplF32>), dim3((unsigned)mf.begin[n]), dim3(64), 0, sl.stream, mf);
      hipLaunchKernelGGL((pair_policy_multi_kernel<false, kRplF64>), dim3((unsigned)mq.begin[n]), dim3(64), 0, sl.stream, mq);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(sl.ev, sl.stream));
    return GKLHIP_OK;
  }

  // Runs one staged call to completion (its packed words are in the caller's pinned result buffer on return).
  int run(const SmallLaunch& mine, hipStream_t own_stream) {
    Ticket t;
    t.sl = &mine;
    const int64_t t_in = t.t_in = now_ns();
    std::unique_lock<std::mutex> l(mu);
    n_calls++;
    queue.push_back(&t);
    while (t.state == 0 || t.state == 4) {
      if (t.state == 4 || flights >= max_flights) { cv.wait(l); continue; }
      // Under load (other sets are in the air) a set is worth more the more calls it carries -- its kernels take as long
      // as their slowest pair whatever their size -- so a would-be leader that finds fewer than `min_batch` calls waiting
      // gives the others `batch_wait_ns` to arrive (GKL_HIP_COMBINE_MIN / GKL_HIP_COMBINE_WAIT_US; 1 / 0 = lead at once).
      // The number to wait for follows the load: a quarter of the host calls inside the library right now, at most 4
      // (16 callers: 4, 8: 2, up to 7: none -- with few callers the wait only adds latency; measured with 50 us: 16 callers
      // 1.42 -> 2.14 TCUPS, while a fixed minimum of 4 cost 4 callers 0.90 -> 0.71).
      {
        const int want = min_batch > 0 ? min_batch : std::min(4, g_host_calls_in_flight.load(std::memory_order_relaxed) / 4);
        if (flights > 0 && (int)queue.size() < want && now_ns() - t_in < batch_wait_ns) {
          cv.wait_for(l, std::chrono::microseconds(5));
          continue;
        }
      }
      // lead: this call first, then the waiting calls of the same arithmetic mode
      const int64_t t_lead = now_ns();
      ns_queued += t_lead - t_in;
      Ticket* batch[kMultiMax];
      int n = 0;
      batch[n++] = &t;
      for (auto it = queue.begin(); it != queue.end();) {
        if (*it == &t) { it = queue.erase(it); continue; }
        if (n < kMultiMax && (*it)->sl->call.fma == mine.call.fma && (*it)->sl->call.fused == mine.call.fused) {
          (*it)->state = 4;  // taken: its owner keeps sleeping until this thread reports the launch (or the end)
          ns_queued += t_lead - (*it)->t_in;
          batch[n++] = *it;
          it = queue.erase(it);
          continue;
        }
        ++it;
      }
      int si = 0;
      while (slot[si].busy) si++;
      Slot& sl = slot[si];
      sl.busy = true;
      const bool alone = flights == 0 && queue.empty() && n == 1;   // no other small call on the device or waiting for it
      flights++;
      n_launch_sets++;
      if (n > 1) n_combined += n;
      int rc = GKLHIP_OK;
      if (n > 1) make_streams();
      if (n > 1 && !sl.stream) rc = fail(GKLHIP_ERR_HIP, "no stream for combined small calls");
      l.unlock();
      if (rc == GKLHIP_OK) rc = n == 1 ? launch_single(mine.call, own_stream, alone) : launch_multi(batch, n, mine.call.fma, sl);
      const std::string err = rc == GKLHIP_OK ? std::string() : g_err;
      const int64_t t_launched = now_ns();
      // a launch that failed part-way may have left kernels on the stream that still read the calls' staging blocks and
      // write their result buffers: drain it BEFORE any of the calls is told about the failure (and returns to a caller
      // that is free to reuse those buffers)
      if (rc != GKLHIP_OK) { (void)(n == 1 ? hipStreamSynchronize(own_stream) : hipStreamSynchronize(sl.stream)); (void)hipGetLastError(); }
      if (n > 1) {
        l.lock();
        // (the others wait on the set's event themselves; letting them sleep until this thread has seen the end
        //  measured the same)
        for (int i = 1; i < n; i++) {
          batch[i]->rc = rc; batch[i]->err = err; batch[i]->ev = sl.ev;
          batch[i]->state = rc == GKLHIP_OK ? 1 : 2;
        }
        cv.notify_all();
        l.unlock();
      }
      hipError_t e = hipSuccess;
      if (rc == GKLHIP_OK) e = n == 1 ? hipStreamSynchronize(own_stream) : hipEventSynchronize(sl.ev);
      l.lock();
      {
        const int64_t t_end = now_ns();
        ns_launch += t_launched - t_lead; ns_sync += t_end - t_launched;
      }
      sl.busy = false;  // (the event is recorded again only from here on: a late waiter of this flight then waits a little longer)
      flights--;
      cv.notify_all();
      l.unlock();
      if (rc != GKLHIP_OK) { g_err = err; return rc; }
      if (e != hipSuccess) return fail(GKLHIP_ERR_HIP, "%s (combined small calls)", hipGetErrorString(e));
      return GKLHIP_OK;
    }
    l.unlock();
    if (t.state == 2) { g_err = t.err; return t.rc; }
    if (t.state == 1) HIP_TRY(hipEventSynchronize(t.ev));
    return GKLHIP_OK;
  }
};
SmallCombiner* small_combiner(int device) {
  static std::mutex mu;
  static std::vector<SmallCombiner*> all;
  std::lock_guard<std::mutex> l(mu);
  if ((int)all.size() <= device) all.resize((size_t)device + 1, nullptr);
  if (!all[(size_t)device]) {
    SmallCombiner* k = all[(size_t)device] = new SmallCombiner();  // lives as long as the process (a handful of streams and events)
    k->device = device;   // (its flight streams: SmallCombiner::make_streams, when two calls first meet)
    if (const char* v = getenv("GKL_HIP_EAGER_STREAMS")) if (atoi(v) >= 7) k->make_streams();   // A/B: the r04 arrangement
    if (const char* v = getenv("GKL_HIP_COMBINE_FLIGHTS")) all[(size_t)device]->max_flights = std::max(1, std::min(kFlightSlots, atoi(v)));
    if (const char* v = getenv("GKL_HIP_COMBINE_MIN")) all[(size_t)device]->min_batch = std::max(0, std::min(kMultiMax, atoi(v)));
    if (const char* v = getenv("GKL_HIP_COMBINE_WAIT_US")) all[(size_t)device]->batch_wait_ns = (int64_t)std::max(0, atoi(v)) * 1000;
  }
  return all[(size_t)device];
}
bool combine_enabled() {
  static const bool on = [] { const char* v = getenv("GKL_HIP_COMBINE"); return !(v && v[0] == '0'); }();
  return on;
}

int dev_compute_host_impl(DevCtx* c, const gklhip_batch* hb, double* out_host) {
  const int64_t n_pairs = (int64_t)hb->n_reads * hb->n_haps;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  int rc;
  const size_t rl = (size_t)hb->read_off[hb->n_reads], hl = (size_t)hb->hap_off[hb->n_haps];
  const size_t stride = align_up(rl);
  const size_t all_bytes = 5 * stride + align_up(hl);
  // a GATK-sized call: the six arrays travel inside the plan block (ONE copy launch for plan + inputs)
  const bool inline_inputs = all_bytes <= kSmallBatchBytes;
  gklhip_batch db = *hb;
  if (!inline_inputs) {
    // H2D of the six byte arrays (one allocation, 256-byte aligned sub-buffers)
    if ((rc = c->batch_dev.reserve(all_bytes))) return rc;
    unsigned char* d = c->batch_dev.as<unsigned char>();
    if (c->have_call_done && c->last_stream != s) HIP_TRY(hipStreamWaitEvent(s, c->call_done, 0));
    const uint8_t* srcs[5] = {hb->read_bases, hb->read_quals, hb->ins_gop, hb->del_gop, hb->gcp};
    for (int i = 0; i < 5; i++) HIP_TRY(hipMemcpyAsync(d + i * stride, srcs[i], rl, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d + 5 * stride, hb->hap_bases, hl, hipMemcpyHostToDevice, s));
    db.read_bases = d; db.read_quals = d + stride; db.ins_gop = d + 2 * stride;
    db.del_gop = d + 3 * stride; db.gcp = d + 4 * stride; db.hap_bases = d + 5 * stride;
  }
  const int mode = c->cfg.finalize;
  const bool on_device = (mode == GKLHIP_FINALIZE_DEVICE_F64 || mode == GKLHIP_FINALIZE_DEVICE_REF32);
  // The kernels store their results straight into pinned host memory (posted writes over PCIe, 8 bytes per pair):
  // a copy-engine transfer behind the last kernel costs a small call ~15 us of queue hand-offs, and in a big call
  // the runtime's copy kernel for the early results slowed the fp64 pass it was meant to overlap with by a third.
  if ((rc = c->res_pin.reserve((size_t)n_pairs * 8))) return rc;
  double* pin_out = nullptr;
  {
    void* p = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&p, c->res_pin.p, 0));
    pin_out = static_cast<double*>(p);
  }
  if (on_device) {
    if ((rc = run_device(c, &db, pin_out, mode, s, inline_inputs))) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    memcpy(out_host, c->res_pin.p, (size_t)n_pairs * 8);
    return GKLHIP_OK;
  }
  // Reference-exact finalisation on the host: one packed 8-byte word per pair.  The fp32 results are final as soon
  // as the policy has run, so in a big call the host finalises them WHILE the fp64 recomputation pass runs; only the
  // recomputed pairs are left for after the last kernel (their words are rewritten in place by finalize64_kernel;
  // the early pass skips every word that is not fp32-tagged, whatever it holds at that moment).
  const HostCallInFlight in_flight;
  const int threads = finalize_threads(c, in_flight.share);
  HostFinalizer fin;
  // (a context with an asynchronous device-resident call still in flight keeps the stream-ordered path)
  SmallLaunch small;
  const bool may_defer = inline_inputs && combine_enabled() && (!c->have_call_done || hipEventQuery(c->call_done) == hipSuccess);
  (void)hipGetLastError();  // (hipErrorNotReady of the query)
  const int64_t t_call = SmallCombiner::now_ns();
  if ((rc = run_device(c, &db, pin_out, kModePacked, s, inline_inputs, may_defer ? &small : nullptr))) return rc;  // records policy_done
  if (small.filled) {
    SmallCombiner* k = small_combiner(c->device);
    const int64_t t_staged = SmallCombiner::now_ns();
    if ((rc = k->run(small, s))) return rc;
    const int64_t t_done = SmallCombiner::now_ns();
    c->stats.n_fallback = fin.all(&c->workers, c->res_pin.as<uint64_t>(), out_host, n_pairs, threads);
    k->ns_stage.fetch_add(t_staged - t_call, std::memory_order_relaxed);
    k->ns_run.fetch_add(t_done - t_staged, std::memory_order_relaxed);
    k->ns_finalize.fetch_add(SmallCombiner::now_ns() - t_done, std::memory_order_relaxed);
    return GKLHIP_OK;
  }
  if (c->cfg.use_double || n_pairs <= kOnePassPairs) {
    // all-fp64 mode, or a GATK-sized call (the fp64 stage of a region without underflowed pairs -- the usual case --
    // is two launches that find nothing to do): one pass over the words once the last kernel is done
    HIP_TRY(hipStreamSynchronize(s));
    c->stats.n_fallback = fin.all(&c->workers, c->res_pin.as<uint64_t>(), out_host, n_pairs, threads);
    return GKLHIP_OK;
  }
  HIP_TRY(hipEventSynchronize(c->policy_done));
  fin.early(&c->workers, c->res_pin.as<uint64_t>(), out_host, n_pairs, threads);
  HIP_TRY(hipStreamSynchronize(s));
  c->stats.n_fallback = fin.late(&c->workers, c->res_pin.as<uint64_t>(), out_host, threads);
  return GKLHIP_OK;
}

// Host buffers in, host doubles out on one device.  An error return must not leave copies from the caller's
// arrays (or into them) in flight: drain the streams first.
int dev_compute_host(DevCtx* c, const gklhip_batch* hb, double* out_host) {
  // ... and neither must a C++ exception on its way to the entry point's guarded() (bad_alloc from a plan vector, a
  // finalisation worker's rethrow): the same drain, then the exception goes on
  auto drain = [c]() noexcept {
    (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->upload_stream) (void)hipStreamSynchronize(c->upload_stream);
    (void)hipGetLastError();
  };
  int rc;
  try {
    rc = dev_compute_host_impl(c, hb, out_host);
  } catch (...) {
    drain();
    throw;
  }
  if (rc != GKLHIP_OK) {
    const std::string keep = g_err;
    drain();
    g_err = keep;
  }
  return rc;
}

// ------------------------------------------------------------------ several devices behind one context
// One host thread per extra device: plans and enqueues that device's shard while the caller's thread does
// device 0's.
class DevWorker {
 public:
  DevWorker() : th_([this] { loop(); }) {}
  ~DevWorker() {
    {
      std::lock_guard<std::mutex> l(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    th_.join();
  }
  void submit(std::function<int()> f) {
    {
      std::lock_guard<std::mutex> l(mu_);
      task_ = std::move(f);
      pending_ = true;
      done_ = false;
    }
    cv_.notify_all();
  }
  int wait(std::string* err) {
    std::unique_lock<std::mutex> l(mu_);
    done_cv_.wait(l, [&] { return done_; });
    if (rc_ != GKLHIP_OK && err) *err = err_;
    return rc_;
  }

 private:
  void loop() {
    std::unique_lock<std::mutex> l(mu_);
    for (;;) {
      cv_.wait(l, [&] { return quit_ || pending_; });
      if (quit_) return;
      pending_ = false;
      std::function<int()> f = std::move(task_);
      l.unlock();
      const int rc = guarded(f);
      std::string e;
      try { e = g_err; } catch (...) {}  // the detail message is thread-local: carry it to the caller
      l.lock();
      rc_ = rc;
      err_.swap(e);
      done_ = true;
      done_cv_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  std::function<int()> task_;
  bool pending_ = false, done_ = true, quit_ = false;
  int rc_ = GKLHIP_OK;
  std::string err_;
  std::thread th_;  // last member: the thread starts with everything above constructed
};

// RCCL, loaded on first use (a single-device context never touches it): the gather of the shards' results on
// device 0 over xGMI, one ncclSend/ncclRecv pair per extra device inside ONE group, driven by this one process
// (ncclCommInitAll) -- SURVEY 5.8 / 8(e).
struct RcclApi {
  void* h = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load() {
    if (h) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
    if (!h) return false;
    auto sym = [&](const char* n) { return dlsym(h, n); };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
    Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) {
      dlclose(h);
      h = nullptr;
      return false;
    }
    return true;
  }
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

#define NCCL_TRY(expr)                                                                              \
  do {                                                                                              \
    ncclResult_t r__ = (expr);                                                                      \
    if (r__ != ncclSuccess) return fail(GKLHIP_ERR_HIP, "%s: %s", #expr, g_rccl.GetErrorString(r__)); \
  } while (0)

}  // namespace

struct gklhip_ctx {
  std::mutex mu;
  gklhip_config cfg;
  std::vector<DevCtx*> dev;                          // dev[0]: where the device-resident entry point gathers
  // Engines of the host-buffer path: `dev`, or -- a single-device context serving a BIG host call -- that device's
  // engine plus a twin on the same GPU: two half-batches whose copies and host-side log10 passes overlap each other's
  // kernels (15.0 instead of 15.6 ms per 10k x 128 batch; smaller calls are better off whole).  Twins are created on
  // first use and owned here.
  std::vector<DevCtx*> host_dev;
  std::vector<DevCtx*> twins;
  const std::vector<DevCtx*>* last = nullptr;        // the engine list of the last call (gklhip_get_raw)
  int host_shards = 2;                               // GKL_HIP_HOST_SHARDS
  std::vector<std::unique_ptr<DevWorker>> workers;   // workers[d-1] drives shard d
  std::vector<int32_t> bounds;                       // read-range boundaries of the last call, [n_dev + 1]
  std::vector<std::vector<int64_t>> sub_off;         // per device: its read range's offsets rebased to 0
  // Gather of the device-resident path: 1 = peer copies, 2 = RCCL, 3 = peer copies after RCCL failed (why: rccl_note).
  // The communicators are created by the FIRST multi-device gklhip_compute_device call (the host path never gathers,
  // and every JNI slot is a context of its own: none of them should pay for, or fail on, communicators it never uses).
  bool want_rccl = false, use_rccl = false, rccl_failed = false;
  std::string rccl_note;
  std::vector<int> rccl_devs;
  std::vector<ncclComm_t> comms;
  hipEvent_t inputs_ready = nullptr;                 // device 0: the caller's stream has reached this call
  std::vector<hipEvent_t> shard_done;                // [n_dev]: device d's results have landed on device 0
  // Second engine per device for the device-resident entry point: a caller that issues consecutive calls on TWO streams
  // (what bench.py does per rank for N > 1: the tail and the planning kernel of one step run under the next step's
  // kernels, 1.91 -> 1.69 ms per eighth-shard step) gets an engine per stream, so the calls do not wait for each other's
  // scratch.  Created by the first call that arrives on another stream than the previous one (GKL_HIP_DEVICE_ENGINES=1:
  // never); a caller with one stream never pays for it.
  std::vector<DevCtx*> dev_alt;
  hipEvent_t inputs_ready_alt = nullptr;
  std::vector<hipEvent_t> shard_done_alt;
  hipStream_t stream_of[2] = {nullptr, nullptr};     // the caller stream each engine set served last
  bool used_set[2] = {false, false};
  int last_set = 0;
  gklhip_stats stats;
  int32_t last_reads = 0, last_haps = 0;
  ~gklhip_ctx() {
    workers.clear();  // joins the threads
    for (size_t d = 0; d < comms.size(); d++)
      if (comms[d]) { (void)hipSetDevice(dev[d]->device); (void)g_rccl.CommDestroy(comms[d]); }
    for (size_t d = 0; d < shard_done.size(); d++)
      if (shard_done[d]) { (void)hipSetDevice(dev[d]->device); (void)hipEventDestroy(shard_done[d]); }
    for (size_t d = 0; d < shard_done_alt.size(); d++)
      if (shard_done_alt[d]) { (void)hipSetDevice(dev[d]->device); (void)hipEventDestroy(shard_done_alt[d]); }
    if (inputs_ready) { (void)hipSetDevice(dev[0]->device); (void)hipEventDestroy(inputs_ready); }
    if (inputs_ready_alt) { (void)hipSetDevice(dev[0]->device); (void)hipEventDestroy(inputs_ready_alt); }
    for (DevCtx* d : dev_alt) dev_done(d);
    for (DevCtx* d : dev) dev_done(d);
    for (DevCtx* d : twins) dev_done(d);
  }
};

namespace {

// Contiguous read ranges balanced by cells: a read's work is its length (every shard sees all haplotypes).  Same
// rule as gkl_amd/shard.py:partition_reads (the cut point closest to p/n of the total, the lower one on a tie).
void partition_reads(int n_reads, const int64_t* read_off, int n_parts, int32_t* bounds) {
  const int64_t total = read_off[n_reads];
  bounds[0] = 0;
  for (int p = 1; p < n_parts; p++) {
    const double target = (double)total * p / n_parts;
    int i = (int)(std::lower_bound(read_off, read_off + n_reads + 1, target, [](int64_t v, double t) { return (double)v < t; }) - read_off);
    if (i > 0 && (i > n_reads || std::fabs((double)read_off[i - 1] - target) <= std::fabs((double)read_off[std::min(i, n_reads)] - target))) i--;
    bounds[p] = std::min(std::max(i, bounds[p - 1]), n_reads);
  }
  bounds[n_parts] = n_reads;
}

// Shard d of `b` (contiguous read range, every haplotype): pointers into the same arrays, offsets rebased.
gklhip_batch shard_view(gklhip_ctx* c, const gklhip_batch* b, int d) {
  const int r0 = c->bounds[d], r1 = c->bounds[d + 1];
  std::vector<int64_t>& off = c->sub_off[(size_t)d];
  off.resize((size_t)(r1 - r0) + 1);
  const int64_t base = b->read_off[r0];
  for (int r = r0; r <= r1; r++) off[(size_t)(r - r0)] = b->read_off[r] - base;
  gklhip_batch v = *b;
  v.n_reads = r1 - r0;
  v.read_off = off.data();
  v.read_bases += base; v.read_quals += base; v.ins_gop += base; v.del_gop += base; v.gcp += base;
  return v;
}

void merge_stats(gklhip_ctx* c, const std::vector<DevCtx*>& list) {
  gklhip_stats t;
  memset(&t, 0, sizeof t);
  bool unknown = false;
  c->last = &list;
  for (size_t d = 0; d < list.size(); d++) {
    if (c->bounds[d + 1] == c->bounds[d]) continue;
    const gklhip_stats& s = list[d]->stats;
    t.n_pairs += s.n_pairs; t.cells += s.cells; t.cells_fp64 += s.cells_fp64;
    if (s.n_fallback < 0) unknown = true; else t.n_fallback += s.n_fallback;
    t.n_chunks += s.n_chunks; t.n_long_pairs += s.n_long_pairs;
    t.n_hap_groups = std::max(t.n_hap_groups, s.n_hap_groups);
    t.rows_per_lane = std::max(t.rows_per_lane, s.rows_per_lane);
    t.ms_fwd_main = std::max(t.ms_fwd_main, s.ms_fwd_main);
    t.ms_fwd_fallback = std::max(t.ms_fwd_fallback, s.ms_fwd_fallback);
    t.ms_total_device = std::max(t.ms_total_device, s.ms_total_device);
    t.lane_fill += s.lane_fill * (float)s.n_chunks;
  }
  if (t.n_chunks) t.lane_fill /= (float)t.n_chunks;
  if (unknown) t.n_fallback = -1;
  c->stats = t;
}

// Run fn(d) for every device with a non-empty shard: device 0 on this thread, the others on their workers.
template <typename F>
int for_each_shard(gklhip_ctx* c, int n, F fn) {
  while ((int)c->workers.size() < n - 1) c->workers.emplace_back(new DevWorker());
  for (int d = 1; d < n; d++)
    if (c->bounds[d + 1] > c->bounds[d]) c->workers[(size_t)d - 1]->submit([=] { return fn(d); });
  int rc = c->bounds[1] > c->bounds[0] ? fn(0) : GKLHIP_OK;
  const std::string err0 = g_err;
  for (int d = 1; d < n; d++)
    if (c->bounds[d + 1] > c->bounds[d]) {
      std::string e;
      const int r = c->workers[(size_t)d - 1]->wait(&e);
      if (r != GKLHIP_OK && rc == GKLHIP_OK) { rc = r; g_err = e; }
    }
  if (rc != GKLHIP_OK && !err0.empty() && g_err.empty()) g_err = err0;
  return rc;
}

int multi_compute_host(gklhip_ctx* c, const std::vector<DevCtx*>& list, const gklhip_batch* hb, double* out_host) {
  const int n = (int)list.size();
  c->sub_off.resize((size_t)n);
  c->bounds.assign((size_t)n + 1, 0);
  partition_reads(hb->n_reads, hb->read_off, n, c->bounds.data());
  // every device copies its own read range straight from the caller's arrays (its own PCIe link) and its
  // results straight back: the host path needs no device-to-device step at all
  const std::vector<DevCtx*>* lp = &list;
  const int rc = for_each_shard(c, n, [=](int d) {
    const gklhip_batch v = shard_view(c, hb, d);
    return dev_compute_host((*lp)[(size_t)d], &v, out_host + (int64_t)c->bounds[(size_t)d] * hb->n_haps);
  });
  merge_stats(c, list);
  return rc;
}

// Device-resident call on several devices: inputs and `out_dev` live on device 0.  Device d > 0 pulls its read
// range and the haplotypes over xGMI (peer copies on its own stream), computes, and its results are gathered into
// out_dev: RCCL send/recv in one group (distinct devices) or a peer copy.  Nothing synchronises with the host;
// the caller's stream `s` ends up waiting for every shard.
// GKL_HIP_RCCL_FAIL=init|group: pretend that RCCL fails there (tests of the fall-back to peer copies on one-GPU boxes).
bool rccl_forced_failure(const char* where) {
  const char* v = getenv("GKL_HIP_RCCL_FAIL");
  return v && strcmp(v, where) == 0;
}

void rccl_give_up(gklhip_ctx* c, const std::string& why) {
  c->use_rccl = false;
  c->rccl_failed = true;
  c->rccl_note = why;
  static const bool quiet = getenv("GKL_HIP_QUIET") != nullptr;
  if (!quiet) fprintf(stderr, "[gklhip] RCCL gather unavailable (%s): gathering with peer copies\n", why.c_str());
}

// First multi-device device-resident call of a context that wants RCCL: load the library, create the communicators.
// Any failure (library missing, a device listed twice, ncclCommInitAll error) selects the peer-copy gather.
void rccl_lazy_init(gklhip_ctx* c) {
  if (!c->want_rccl || c->use_rccl || c->rccl_failed) return;
  std::lock_guard<std::mutex> l(g_rccl_mu);
  if (rccl_forced_failure("init")) return rccl_give_up(c, "forced by GKL_HIP_RCCL_FAIL=init");
  if (!g_rccl.load()) return rccl_give_up(c, "librccl.so cannot be loaded");
  const int n = (int)c->dev.size();
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++)
      if (c->rccl_devs[(size_t)i] == c->rccl_devs[(size_t)j]) return rccl_give_up(c, "a device is listed twice (a communicator holds a device once)");
  c->comms.assign((size_t)n, nullptr);
  const ncclResult_t r = g_rccl.CommInitAll(c->comms.data(), n, c->rccl_devs.data());
  if (r != ncclSuccess) {
    c->comms.clear();
    return rccl_give_up(c, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
  }
  c->use_rccl = true;
}

int multi_compute_device(gklhip_ctx* c, int set, const gklhip_batch* db, double* out_dev, int mode, hipStream_t s) {
  const std::vector<DevCtx*>& devs = set ? c->dev_alt : c->dev;
  const std::vector<hipEvent_t>& shard_done = set ? c->shard_done_alt : c->shard_done;
  hipEvent_t inputs_ready = set ? c->inputs_ready_alt : c->inputs_ready;
  const int n = (int)devs.size();
  rccl_lazy_init(c);
  const bool use_rccl = c->use_rccl;
  c->bounds.assign((size_t)n + 1, 0);
  partition_reads(db->n_reads, db->read_off, n, c->bounds.data());
  DevCtx* root = devs[0];
  HIP_TRY(hipSetDevice(root->device));
  HIP_TRY(hipEventRecord(inputs_ready, s));
  const int n_haps = db->n_haps;
  const size_t hl = (size_t)db->hap_off[n_haps];
  const std::vector<DevCtx*>* dp = &devs;
  const std::vector<hipEvent_t>* sdp = &shard_done;
  int rc = for_each_shard(c, n, [=](int d) -> int {
    DevCtx* dc = (*dp)[(size_t)d];
    const gklhip_batch v = shard_view(c, db, d);
    if (d == 0) return run_device(dc, &v, out_dev, mode, s, false);
    HIP_TRY(hipSetDevice(dc->device));
    hipStream_t sd = dc->stream;
    const size_t rl = (size_t)v.read_off[v.n_reads], stride = align_up(rl);
    int r;
    if ((r = dc->batch_dev.reserve(5 * stride + align_up(hl)))) return r;
    if ((r = dc->out_dev.reserve((size_t)v.n_reads * n_haps * 8))) return r;
    unsigned char* dst = dc->batch_dev.as<unsigned char>();
    HIP_TRY(hipStreamWaitEvent(sd, inputs_ready, 0));
    const uint8_t* srcs[5] = {v.read_bases, v.read_quals, v.ins_gop, v.del_gop, v.gcp};
    for (int i = 0; i < 5; i++)
      HIP_TRY(hipMemcpyPeerAsync(dst + i * stride, dc->device, srcs[i], root->device, rl, sd));
    HIP_TRY(hipMemcpyPeerAsync(dst + 5 * stride, dc->device, v.hap_bases, root->device, hl, sd));
    gklhip_batch lv = v;
    lv.read_bases = dst; lv.read_quals = dst + stride; lv.ins_gop = dst + 2 * stride;
    lv.del_gop = dst + 3 * stride; lv.gcp = dst + 4 * stride; lv.hap_bases = dst + 5 * stride;
    if ((r = run_device(dc, &lv, dc->out_dev.as<double>(), mode, sd, false))) return r;
    if (!use_rccl) {
      HIP_TRY(hipMemcpyPeerAsync(out_dev + (int64_t)c->bounds[(size_t)d] * n_haps, root->device, dc->out_dev.p, dc->device,
                                 (size_t)v.n_reads * n_haps * 8, sd));
      HIP_TRY(hipEventRecord((*sdp)[(size_t)d], sd));
    }
    return GKLHIP_OK;
  });
  if (rc == GKLHIP_OK && use_rccl) {
    // the one exchange step: every extra device sends its slice, device 0 receives them, all in one group.  Group
    // submission is serialised process-wide (several contexts over the same devices must not interleave their
    // groups), the group is always closed, and a failure at any point degrades THIS and all later calls of the
    // context to peer copies -- the shards' results are still sitting in their devices' buffers.
    std::string why;
    {
      std::lock_guard<std::mutex> gl(g_rccl_mu);
      ncclResult_t bad = rccl_forced_failure("group") ? ncclInternalError : ncclSuccess;
      const char* what = "forced by GKL_HIP_RCCL_FAIL=group";
      if (bad == ncclSuccess) {
        ncclResult_t r = g_rccl.GroupStart();
        if (r != ncclSuccess) { bad = r; what = "ncclGroupStart"; }
        else {
          for (int d = 1; d < n && bad == ncclSuccess; d++) {
            const size_t cnt = (size_t)(c->bounds[(size_t)d + 1] - c->bounds[(size_t)d]) * n_haps;
            if (!cnt) continue;
            r = g_rccl.Send(devs[(size_t)d]->out_dev.p, cnt, ncclDouble, 0, c->comms[(size_t)d], devs[(size_t)d]->stream);
            if (r != ncclSuccess) { bad = r; what = "ncclSend"; break; }
            r = g_rccl.Recv(out_dev + (int64_t)c->bounds[(size_t)d] * n_haps, cnt, ncclDouble, d, c->comms[0], s);
            if (r != ncclSuccess) { bad = r; what = "ncclRecv"; }
          }
          r = g_rccl.GroupEnd();  // always: an open group would swallow every later RCCL call of this thread
          if (r != ncclSuccess && bad == ncclSuccess) { bad = r; what = "ncclGroupEnd"; }
        }
      }
      if (bad != ncclSuccess) why = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(bad) : "error");
    }
    if (why.empty()) {
      for (int d = 1; d < n; d++)
        if (c->bounds[(size_t)d + 1] > c->bounds[(size_t)d]) {
          HIP_TRY(hipSetDevice(devs[(size_t)d]->device));
          HIP_TRY(hipEventRecord(shard_done[(size_t)d], devs[(size_t)d]->stream));
        }
    } else {
      rccl_give_up(c, why);
      for (int d = 1; d < n; d++) {
        const size_t cnt = (size_t)(c->bounds[(size_t)d + 1] - c->bounds[(size_t)d]) * n_haps;
        if (!cnt) continue;
        DevCtx* dc = devs[(size_t)d];
        HIP_TRY(hipSetDevice(dc->device));
        HIP_TRY(hipMemcpyPeerAsync(out_dev + (int64_t)c->bounds[(size_t)d] * n_haps, root->device, dc->out_dev.p, dc->device, cnt * 8, dc->stream));
        HIP_TRY(hipEventRecord(shard_done[(size_t)d], dc->stream));
      }
    }
  }
  HIP_TRY(hipSetDevice(root->device));
  if (rc == GKLHIP_OK)
    for (int d = 1; d < n; d++)
      if (c->bounds[(size_t)d + 1] > c->bounds[(size_t)d]) HIP_TRY(hipStreamWaitEvent(s, shard_done[(size_t)d], 0));
  merge_stats(c, devs);
  return rc;
}

int parse_device_list(const char* v, std::vector<int32_t>* out) {
  out->clear();
  if (!v) return GKLHIP_OK;
  const char* p = v;
  while (*p) {
    while (*p == ' ' || *p == ',') p++;
    if (!*p) break;
    char* end = nullptr;
    const long d = strtol(p, &end, 10);
    if (end == p || d < 0 || d > 1023) return fail(GKLHIP_ERR_INVALID_ARG, "GKL_HIP_DEVICES: cannot parse \"%s\"", v);
    out->push_back((int32_t)d);
    p = end;
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

static int init_devices_impl(const gklhip_config* cfg, const int32_t* devices, int32_t n_devices, gklhip_ctx** out_ctx) {
  if (!out_ctx) return fail(GKLHIP_ERR_INVALID_ARG, "out_ctx is NULL");
  *out_ctx = nullptr;
  gklhip_config c0;
  memset(&c0, 0, sizeof c0);
  c0.abi_version = GKLHIP_ABI_VERSION; c0.device = -1; c0.max_threads = 0; c0.fma_mode = 1; c0.finalize = -1;
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
  std::vector<int32_t> list;
  if (devices && n_devices > 0) list.assign(devices, devices + n_devices);
  if (list.empty()) {
    int dev = c0.device;
    if (dev < 0) { HIP_TRY(hipGetDevice(&dev)); }
    list.push_back(dev);
  }
  if (list.size() > 64) return fail(GKLHIP_ERR_INVALID_ARG, "%zu devices in the list (at most 64)", list.size());
  std::unique_ptr<gklhip_ctx> c(new (std::nothrow) gklhip_ctx());
  if (!c) return fail(GKLHIP_ERR_OOM, "context allocation failed");
  c->cfg = c0;
  memset(&c->stats, 0, sizeof c->stats);
  for (int32_t d : list) {
    DevCtx* dc = nullptr;
    gklhip_config dcfg = c0;
    dcfg.device = d;
    const int rc = dev_init(dcfg, d, ndev, &dc);
    if (rc) return rc;
    c->dev.push_back(dc);
  }
  const int n = (int)c->dev.size();
  c->sub_off.resize((size_t)n);
  c->bounds.assign((size_t)n + 1, 0);
  if (n > 1) {
    bool distinct = true;
    for (int i = 0; i < n; i++)
      for (int j = i + 1; j < n; j++) distinct &= list[(size_t)i] != list[(size_t)j];
    c->shard_done.assign((size_t)n, nullptr);
    HIP_TRY(hipSetDevice(c->dev[0]->device));
    HIP_TRY(hipEventCreateWithFlags(&c->inputs_ready, hipEventDisableTiming));
    for (int d = 1; d < n; d++) {
      HIP_TRY(hipSetDevice(c->dev[(size_t)d]->device));
      HIP_TRY(hipEventCreateWithFlags(&c->shard_done[(size_t)d], hipEventDisableTiming));
      if (c->dev[(size_t)d]->device != c->dev[0]->device) {
        // direct xGMI loads/stores between device 0 and this one (already enabled / unsupported: the copies then bounce)
        (void)hipDeviceEnablePeerAccess(c->dev[0]->device, 0);
        (void)hipGetLastError();
        (void)hipSetDevice(c->dev[0]->device);
        (void)hipDeviceEnablePeerAccess(c->dev[(size_t)d]->device, 0);
        (void)hipGetLastError();
      }
    }
    // Gather over RCCL when the devices are distinct (a communicator cannot hold a device twice; the 0,0 list of
    // the one-GPU tests gathers with plain copies).  GKL_HIP_GATHER=peer|rccl overrides.
    // Communicators are created by the first device-resident call (rccl_lazy_init); RCCL that cannot be had there --
    // library missing, a device listed twice, ncclCommInitAll failing -- degrades to peer copies, never to an error.
    const char* g = getenv("GKL_HIP_GATHER");
    c->want_rccl = g ? strcmp(g, "rccl") == 0 : distinct;
    c->rccl_devs.assign(list.begin(), list.end());
  }
  {
    const char* hs = getenv("GKL_HIP_HOST_SHARDS");
    c->host_shards = hs && *hs ? std::max(1, std::min(atoi(hs), 8)) : 2;
  }
  c->host_dev = c->dev;
  c->last = &c->dev;
  HIP_TRY(hipSetDevice(c->dev[0]->device));
  *out_ctx = c.release();
  return GKLHIP_OK;
}

int gklhip_fault_inject(const char* spec);
static int init_impl(const gklhip_config* cfg, gklhip_ctx** out_ctx) {
  static std::once_flag fault_env;
  std::call_once(fault_env, [] { if (const char* v = getenv("GKLHIP_FAULT_INJECT")) (void)gklhip_fault_inject(v); });
  // GKL_HIP_DEVICES=0,1,...: shard every call over these devices (used when the config does not pin one)
  std::vector<int32_t> list;
  if (!cfg || cfg->device < 0) {
    const int rc = parse_device_list(getenv("GKL_HIP_DEVICES"), &list);
    if (rc) { if (out_ctx) *out_ctx = nullptr; return rc; }
  }
  return init_devices_impl(cfg, list.empty() ? nullptr : list.data(), (int32_t)list.size(), out_ctx);
}

int gklhip_done(gklhip_ctx* c) {
  if (!c) return GKLHIP_OK;
  delete c;
  return GKLHIP_OK;
}

int gklhip_num_devices(gklhip_ctx* c) { return c ? (int)c->dev.size() : 0; }

int gklhip_gather_backend(gklhip_ctx* c) {
  if (!c || c->dev.size() < 2) return 0;
  std::lock_guard<std::mutex> lock(c->mu);
  // (before the first device-resident call: what that call will try)
  return c->rccl_failed ? 3 : (c->use_rccl || c->want_rccl) ? 2 : 1;
}

const char* gklhip_gather_note(gklhip_ctx* c) {
  // a copy per calling thread, valid until that thread's next call (the context's own string may be rewritten by a
  // concurrent compute call as soon as the lock is dropped)
  static thread_local char note[512];
  note[0] = 0;
  if (!c) return note;
  std::lock_guard<std::mutex> lock(c->mu);
  snprintf(note, sizeof note, "%s", c->rccl_note.c_str());
  return note;
}

int gklhip_partition_reads(int32_t n_reads, const int64_t* read_off, int32_t n_parts, int32_t* bounds_out) {
  if (n_reads < 0 || n_parts <= 0 || !read_off || !bounds_out) return fail(GKLHIP_ERR_INVALID_ARG, "bad arguments to gklhip_partition_reads");
  partition_reads(n_reads, read_off, n_parts, bounds_out);
  return GKLHIP_OK;
}

static int compute_device_impl(gklhip_ctx* c, const gklhip_batch* dev_batch, double* out_dev, void* hip_stream) {
  if (!c) return fail(GKLHIP_ERR_INVALID_ARG, "context is NULL (initNative not called)");
  int rc = validate(dev_batch);
  if (rc) return rc;
  if (!out_dev && (int64_t)dev_batch->n_reads * dev_batch->n_haps > 0)
    return fail(GKLHIP_ERR_INVALID_ARG, "output array is NULL");
  std::lock_guard<std::mutex> lock(c->mu);
  HIP_TRY(hipSetDevice(c->dev[0]->device));
  int mode = c->cfg.finalize;
  if (mode != GKLHIP_FINALIZE_DEVICE_F64 && mode != GKLHIP_FINALIZE_DEVICE_REF32) mode = GKLHIP_FINALIZE_DEVICE_F64;
  hipStream_t s = static_cast<hipStream_t>(hip_stream);  // NULL = HIP's default stream
  c->last_reads = dev_batch->n_reads; c->last_haps = dev_batch->n_haps;
  // which engine set: the one that served this stream last; a call on a NEW stream while the other set is busy with
  // another stream's work takes (first: creates) the second set
  int set = 0;
  static const bool one_engine = [] { const char* v = getenv("GKL_HIP_DEVICE_ENGINES"); return v && atoi(v) == 1; }();
  if (!one_engine && c->used_set[0] && c->stream_of[0] != s && c->cfg.record_events == 0) {
    if (c->used_set[1] && c->stream_of[1] != s) set = 1 - c->last_set;   // a third stream: the set used longest ago
    else set = 1;
    if (set == 1 && c->dev_alt.empty()) {
      int ndev = 0;
      HIP_TRY(hipGetDeviceCount(&ndev));
      std::vector<DevCtx*> made;
      for (DevCtx* d : c->dev) {
        DevCtx* twin = nullptr;
        if (dev_init(d->cfg, d->device, ndev, &twin) != GKLHIP_OK) break;   // e.g. out of memory: stay with one set
        made.push_back(twin);
      }
      bool ok = made.size() == c->dev.size();
      if (ok && c->dev.size() > 1) {
        c->shard_done_alt.assign(c->dev.size(), nullptr);
        ok = hipSetDevice(c->dev[0]->device) == hipSuccess && hipEventCreateWithFlags(&c->inputs_ready_alt, hipEventDisableTiming) == hipSuccess;
        for (size_t d = 1; ok && d < c->dev.size(); d++)
          ok = hipSetDevice(c->dev[d]->device) == hipSuccess && hipEventCreateWithFlags(&c->shard_done_alt[d], hipEventDisableTiming) == hipSuccess;
        (void)hipSetDevice(c->dev[0]->device);
      }
      if (ok) c->dev_alt = made;
      else { for (DevCtx* d : made) dev_done(d); (void)hipGetLastError(); set = 0; }
    }
  }
  c->stream_of[set] = s; c->used_set[set] = true; c->last_set = set;
  const std::vector<DevCtx*>& devs = set ? c->dev_alt : c->dev;
  if (devs.size() == 1 || (int64_t)dev_batch->n_reads * dev_batch->n_haps == 0) {
    c->bounds.assign(devs.size() + 1, dev_batch->n_reads);
    c->bounds[0] = 0;
    rc = run_device(devs[0], dev_batch, out_dev, mode, s, false);
    c->stats = devs[0]->stats;
    c->last = &devs;
    return rc;
  }
  return multi_compute_device(c, set, dev_batch, out_dev, mode, s);
}

// Fault injection (tests of the callers' error handling; nothing in the reference): "compute:N" or "compute:NxK" makes
// the N-th .. (N+K-1)-th gklhip_compute of the process -- counted from the arming -- return GKLHIP_ERR_HIP before any
// work, with the output array poisoned.  Armed by gklhip_fault_inject(), or once from GKLHIP_FAULT_INJECT by the first
// gklhip_init of the process (read there, never on a call path).
static std::atomic<int64_t> g_fault_from{0}, g_fault_count{0}, g_fault_calls{0};
int gklhip_fault_inject(const char* spec) {
  long from = 0, count = 0;
  if (spec && *spec) {
    if (strncmp(spec, "compute:", 8) != 0) return fail(GKLHIP_ERR_INVALID_ARG, "fault spec: compute:N or compute:NxK");
    char* end = nullptr;
    from = strtol(spec + 8, &end, 10);
    count = (end && *end == 'x') ? strtol(end + 1, nullptr, 10) : 1;
    if (from <= 0 || count <= 0) return fail(GKLHIP_ERR_INVALID_ARG, "fault spec: compute:N or compute:NxK");
  }
  g_fault_calls = 0; g_fault_count = count; g_fault_from = from;
  return GKLHIP_OK;
}
static bool fault_due() {
  const int64_t from = g_fault_from.load(std::memory_order_relaxed);
  if (from <= 0) return false;
  const int64_t nth = g_fault_calls.fetch_add(1) + 1;
  return nth >= from && nth < from + g_fault_count.load();
}

static int compute_impl(gklhip_ctx* c, const gklhip_batch* hb, double* out_host) {
  if (!c) return fail(GKLHIP_ERR_INVALID_ARG, "context is NULL (initNative not called)");
  int rc = validate(hb);
  if (rc) return rc;
  const int64_t n_pairs = (int64_t)hb->n_reads * hb->n_haps;
  if (n_pairs == 0) return GKLHIP_OK;
  if (!out_host) return fail(GKLHIP_ERR_INVALID_ARG, "output array is NULL");
  if (fault_due()) {
    for (int64_t i = 0; i < n_pairs; i++) out_host[i] = std::numeric_limits<double>::quiet_NaN();
    return fail(GKLHIP_ERR_HIP, "injected fault (GKLHIP_FAULT_INJECT)");
  }
  std::lock_guard<std::mutex> lock(c->mu);
  c->last_reads = hb->n_reads; c->last_haps = hb->n_haps;
  if (c->dev.size() == 1 && c->host_shards > 1 && n_pairs >= kHostShardPairs && hb->n_reads >= 2 * c->host_shards) {
    // a big call on one device: two (GKL_HIP_HOST_SHARDS) half-batches on twin engines
    while ((int)c->host_dev.size() < c->host_shards) {
      DevCtx* twin = nullptr;
      int ndev = 0;
      HIP_TRY(hipGetDeviceCount(&ndev));
      if ((rc = dev_init(c->dev[0]->cfg, c->dev[0]->device, ndev, &twin))) { c->host_shards = (int)c->host_dev.size(); break; }  // e.g. out of memory: stay whole
      c->twins.push_back(twin);
      c->host_dev.push_back(twin);
    }
    if (c->host_dev.size() > 1) return multi_compute_host(c, c->host_dev, hb, out_host);
  }
  if (c->dev.size() == 1) {
    c->bounds.assign(2, hb->n_reads);
    c->bounds[0] = 0;
    rc = dev_compute_host(c->dev[0], hb, out_host);
    c->stats = c->dev[0]->stats;
    c->last = &c->dev;
    return rc;
  }
  return multi_compute_host(c, c->dev, hb, out_host);
}

void* gklhip_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault | hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

void gklhip_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

static int get_step_times_impl(gklhip_ctx* ctx, int32_t steps_back, float* ms_main, float* ms_fallback, float* ms_total) {
  if (!ctx) return fail(GKLHIP_ERR_INVALID_ARG, "context is NULL");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DevCtx* c = ctx->dev[0];  // (several devices: device 0's shard)
  if (c->cfg.record_events != 2) return fail(GKLHIP_ERR_INVALID_ARG, "context was not created with record_events = 2");
  if (steps_back < 0 || steps_back >= DevCtx::kEventRing || steps_back >= c->calls)
    return fail(GKLHIP_ERR_INVALID_ARG, "steps_back %d outside the %d recorded calls", steps_back,
                (int)std::min<int64_t>(c->calls, DevCtx::kEventRing));
  HIP_TRY(hipSetDevice(c->device));
  const int slot = (int)((c->calls - 1 - steps_back) % DevCtx::kEventRing);
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

static int get_raw_impl(gklhip_ctx* ctx, float* raw32, double* raw64, uint8_t* used64) {
  if (!ctx) return fail(GKLHIP_ERR_INVALID_ARG, "context is NULL");
  std::lock_guard<std::mutex> lock(ctx->mu);
  int64_t n_fallback = 0;
  bool any = false;
  const std::vector<DevCtx*>& list = ctx->last ? *ctx->last : ctx->dev;
  for (size_t d = 0; d < list.size(); d++) {
    DevCtx* c = list[d];
    if (list.size() > 1 && ctx->bounds[d + 1] == ctx->bounds[d]) continue;
    if (!c->have_last) continue;
    any = true;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->last_stream;
    const size_t n = (size_t)c->last_pairs;
    const size_t at = (size_t)ctx->bounds[d] * (size_t)ctx->last_haps;
    if (raw32 && !c->cfg.use_double) HIP_TRY(hipMemcpyAsync(raw32 + at, c->raw32.p, n * 4, hipMemcpyDeviceToHost, s));
    if (raw64) HIP_TRY(hipMemcpyAsync(raw64 + at, c->raw64.p, n * 8, hipMemcpyDeviceToHost, s));
    if (used64) HIP_TRY(hipMemcpyAsync(used64 + at, c->used64.p, n, hipMemcpyDeviceToHost, s));
    int32_t cnt[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(cnt, c->counters.p, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    c->stats.n_fallback = c->cfg.use_double ? (int64_t)n : cnt[0];
    n_fallback += c->stats.n_fallback;
  }
  if (!any) return fail(GKLHIP_ERR_INVALID_ARG, "no completed call to read back");
  ctx->stats.n_fallback = n_fallback;
  HIP_TRY(hipSetDevice(ctx->dev[0]->device));
  return GKLHIP_OK;
}

int gklhip_plan_describe(int32_t n_reads, int32_t n_haps, const int64_t* read_off, const int64_t* hap_off,
                         int32_t rows_per_lane, int32_t* lanes_out, int64_t lanes_cap, int32_t* n_groups_out,
                         int32_t* n_long_out) {
  if (n_reads < 0 || n_haps < 0 || !read_off || !hap_off || (rows_per_lane != 4 && rows_per_lane != 8))
    return -fail(GKLHIP_ERR_INVALID_ARG, "bad arguments to gklhip_plan_describe");
  Plan p;
  build_plan(n_reads, n_haps, read_off, hap_off, rows_per_lane, kTargetCols, &p, /*want_lanes=*/true);
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

int gklhip_small_call_counts(int device, int64_t out[3], int reset) {
  if (!out || device < 0) return fail(GKLHIP_ERR_INVALID_ARG, "NULL argument or negative device");
  SmallCombiner* k = small_combiner(device);
  std::lock_guard<std::mutex> l(k->mu);
  out[0] = k->n_calls; out[1] = k->n_combined; out[2] = k->n_launch_sets;
  if (getenv("GKLHIP_TIMING"))
    fprintf(stderr, "[gklhip] small calls: %lld calls, %lld combined, %lld launch sets; per set: queued %.1f us (sum over its calls), launch %.1f us, sync %.1f us\n",
            (long long)k->n_calls, (long long)k->n_combined, (long long)k->n_launch_sets, k->ns_queued * 1e-3 / std::max<int64_t>(1, k->n_launch_sets),
            k->ns_launch * 1e-3 / std::max<int64_t>(1, k->n_launch_sets), k->ns_sync * 1e-3 / std::max<int64_t>(1, k->n_launch_sets));
  if (getenv("GKLHIP_TIMING"))
    fprintf(stderr, "[gklhip] small calls, per call: plan + staging %.1f us, queued + launches + wait %.1f us, host log10 %.1f us\n",
            k->ns_stage.load() * 1e-3 / std::max<int64_t>(1, k->n_calls), k->ns_run.load() * 1e-3 / std::max<int64_t>(1, k->n_calls),
            k->ns_finalize.load() * 1e-3 / std::max<int64_t>(1, k->n_calls));
  if (reset) {
    k->n_calls = k->n_combined = k->n_launch_sets = k->ns_queued = k->ns_launch = k->ns_sync = 0;
    k->ns_stage = 0; k->ns_run = 0; k->ns_finalize = 0;
  }
  return GKLHIP_OK;
}

// Diagnostics: the VALU issue ceiling of the recurrence's instruction mix on this device (issue_mix_*_kernel: 4 multiplies
// + 4 FMAs per "cell", four wavefronts per SIMD, every CU) over about `ms_budget` milliseconds.  cells_per_s x 12 FLOP is
// what roofline.issue_ceiling_tflops reports; clock_ghz = shader cycles the kernel counted / its HIP-event time, i.e. the
// clock the chip sustains under this load (it clocks to its power budget).
int gklhip_measure_issue_ceiling(gklhip_ctx* ctx, int use_double, double ms_budget, double* cells_per_s, double* clock_ghz) {
  if (!ctx || !cells_per_s) return fail(GKLHIP_ERR_INVALID_ARG, "NULL argument");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DevCtx* c = ctx->dev[0];
  HIP_TRY(hipSetDevice(c->device));
  uint64_t* cyc = nullptr;
  HIP_TRY(hipMalloc(&cyc, 8));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  const int blocks = c->n_cus * 4;  // 4 x 256 threads per CU = four wavefronts per SIMD
  auto run = [&](int iters, float* ms) -> int {
    HIP_TRY(hipEventRecord(e0, c->stream));
    if (use_double) hipLaunchKernelGGL(issue_mix_f64_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, iters, cyc);
    else            hipLaunchKernelGGL(issue_mix_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, iters, cyc);
    HIP_TRY(hipEventRecord(e1, c->stream));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(ms, e0, e1));
    return GKLHIP_OK;
  };
  float ms = 0;
  int rc = run(2000, &ms);   // warm-up + calibration (~1 ms)
  int iters = (int)std::max(2000.0, std::min(4.0e6, 2000.0 * std::max(1.0, ms_budget) / std::max(ms, 0.05f)));
  if (!rc) rc = run(iters, &ms);
  uint64_t cycles = 0;
  if (!rc && hipMemcpy(&cycles, cyc, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(GKLHIP_ERR_HIP, "hipMemcpy failed");
  (void)hipFree(cyc);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc) return rc;
  const double cells = (double)blocks * 4 /*wavefronts*/ * 64 /*lanes*/ * 8 /*cells per iteration*/ * (double)iters;
  *cells_per_s = cells / (ms * 1e-3);
  if (clock_ghz) *clock_ghz = (double)cycles / (ms * 1e-3) * 1e-9;
  return GKLHIP_OK;
}

// Diagnostics: load RCCL and run one send/recv pair inside one group on a one-device communicator (what the
// multi-device gather does per extra device).  0 = ok.
static int rccl_selftest_impl(int32_t device) {
  std::lock_guard<std::mutex> l(g_rccl_mu);
  if (!g_rccl.load()) return fail(GKLHIP_ERR_HIP, "librccl.so cannot be loaded: %s", dlerror());
  HIP_TRY(hipSetDevice(device));
  ncclComm_t comm = nullptr;
  const int devs[1] = {device};
  NCCL_TRY(g_rccl.CommInitAll(&comm, 1, devs));
  const size_t n = 4096;
  double *src = nullptr, *dst = nullptr;
  HIP_TRY(hipMalloc(&src, n * 8));
  HIP_TRY(hipMalloc(&dst, n * 8));
  std::vector<double> h(n), back(n, 0.0);
  for (size_t i = 0; i < n; i++) h[i] = (double)i * 0.5 - 7.0;
  HIP_TRY(hipMemcpy(src, h.data(), n * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(dst, 0, n * 8));
  hipStream_t s;
  HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  NCCL_TRY(g_rccl.GroupStart());
  NCCL_TRY(g_rccl.Send(src, n, ncclDouble, 0, comm, s));
  NCCL_TRY(g_rccl.Recv(dst, n, ncclDouble, 0, comm, s));
  NCCL_TRY(g_rccl.GroupEnd());
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipMemcpy(back.data(), dst, n * 8, hipMemcpyDeviceToHost));
  (void)hipStreamDestroy(s);
  (void)hipFree(src);
  (void)hipFree(dst);
  (void)g_rccl.CommDestroy(comm);
  if (memcmp(h.data(), back.data(), n * 8) != 0) return fail(GKLHIP_ERR_HIP, "RCCL self send/recv returned different data");
  return GKLHIP_OK;
}

// ---- the guarded entry points (see guarded()) ----
int gklhip_init_devices(const gklhip_config* cfg, const int32_t* devices, int32_t n_devices, gklhip_ctx** out_ctx) { return guarded([&] { return init_devices_impl(cfg, devices, n_devices, out_ctx); }); }
int gklhip_compute_device(gklhip_ctx* c, const gklhip_batch* dev_batch, double* out_dev, void* hip_stream) { return guarded([&] { return compute_device_impl(c, dev_batch, out_dev, hip_stream); }); }
int gklhip_compute(gklhip_ctx* c, const gklhip_batch* hb, double* out_host) { return guarded([&] { return compute_impl(c, hb, out_host); }); }
int gklhip_get_raw(gklhip_ctx* ctx, float* raw32, double* raw64, uint8_t* used64) { return guarded([&] { return get_raw_impl(ctx, raw32, raw64, used64); }); }
int gklhip_get_step_times(gklhip_ctx* ctx, int32_t steps_back, float* ms_main, float* ms_fallback, float* ms_total) { return guarded([&] { return get_step_times_impl(ctx, steps_back, ms_main, ms_fallback, ms_total); }); }
int gklhip_rccl_selftest(int32_t device) { return guarded([&] { return rccl_selftest_impl(device); }); }
int gklhip_init(const gklhip_config* cfg, gklhip_ctx** out_ctx) { return guarded([&] { return init_impl(cfg, out_ctx); }); }

}  // extern "C"
