// Process-wide switches, error plumbing, the grow-and-trim device / pinned buffers and DevCtx: one device's engine state.
// Part of the ONE translation unit gkl_amd/csrc/pairhmm_api.hip (included there, in this order: pairhmm_ctx.h, pairhmm_device_pass.h,
// pairhmm_ctx_lifecycle.h, pairhmm_host_call.h, pairhmm_multi_device.h, pairhmm_diagnostics.h); not a stand-alone header.
#pragma once

using namespace gklhip;

// Development / A-B switches, read from the environment ONCE -- when the library is loaded -- and never on a call path
// (getenv is not safe against a concurrent setenv in the host JVM).  Context-level settings (GKL_HIP_DEVICES, GKL_HIP_GATHER,
// GKL_HIP_HOST_SHARDS, GKLHIP_ASM_GENERAL, ...) are read by the init functions; the test hooks GKL_HIP_RCCL_FAIL /
// GKLHIP_SELFTEST_FAIL where a context or a communicator is made.
namespace {
struct EnvSwitches {
  static bool on_unless_zero(const char* name) { const char* v = getenv(name); return !v || atoi(v) != 0; }
  static int integer(const char* name) { const char* v = getenv(name); return v ? atoi(v) : 0; }
  const bool wide_long = on_unless_zero("GKLHIP_WIDE_LONG");        // long reads: workgroups of several wavefronts (0: one-wavefront stripes)
  const bool super_long = on_unless_zero("GKLHIP_SUPER_LONG");      // reads beyond a workgroup: super-stripes
  const bool fused_pairs = on_unless_zero("GKLHIP_FUSED_PAIRS");    // small calls: fp32 + policy + fp64 of a pair in one wavefront
  const int64_t fused_max = [] { const char* v = getenv("GKLHIP_FUSED_MAX_PAIRS"); return v ? atoll(v) : -1LL; }();
  const bool xcd_aware = on_unless_zero("GKLHIP_XCD_AWARE");
  const bool timing = getenv("GKLHIP_TIMING") != nullptr;
  const int combine_load = integer("GKL_HIP_COMBINE_LOAD");
  const int target_cols = integer("GKLHIP_TARGET_COLS");
  const int fb_wanted_jobs = integer("GKLHIP_FB_WANTED_JOBS");
  const int plan_blocks = integer("GKLHIP_PLAN_BLOCKS");
  const int finalize_threads = integer("GKL_HIP_FINALIZE_THREADS");
  const int finalize_min = integer("GKL_HIP_FINALIZE_MIN");   // one-pass host log10: pairs from which it is spread over the workers (0: the built-in 8192)
  const bool combine = [] { const char* v = getenv("GKL_HIP_COMBINE"); return !(v && v[0] == '0'); }();
  const bool quiet = getenv("GKL_HIP_QUIET") != nullptr;
  const bool one_device_engine = integer("GKL_HIP_DEVICE_ENGINES") == 1;
};
const EnvSwitches g_env;
}  // namespace

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
// Hysteresis (r05 advisor): hipFree / hipHostFree synchronise the whole device -- every other context's work in flight
// waits -- so a buffer that GREW less than kTrimQuiet calls ago is left alone: a workload that alternates one big call with
// sixteen small ones keeps its buffers instead of freeing and re-making 100 MB every round.  `small_uses` counts the small
// calls in a row, `since_grow` the calls since the buffer last grew.
constexpr int kTrimQuiet = 64;
inline bool trim_due(size_t n, size_t cap, int* small_uses, int* since_grow) {
  if (*since_grow < kTrimQuiet) ++*since_grow;
  if (cap <= kTrimFloor || n >= cap / 4) { *small_uses = 0; return false; }
  if (*small_uses < kTrimCalls) ++*small_uses;
  return *small_uses >= kTrimCalls && *since_grow >= kTrimQuiet;
}
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int small_uses = 0, since_grow = kTrimQuiet;
  int reserve(size_t n) {
    if (n <= cap && !trim_due(n, cap, &small_uses, &since_grow)) return GKLHIP_OK;
    if (n > cap) since_grow = 0;
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
  int small_uses = 0, since_grow = kTrimQuiet;
  int reserve(size_t n) {
    if (n <= cap && !trim_due(n, cap, &small_uses, &since_grow)) return GKLHIP_OK;
    if (n > cap) since_grow = 0;
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
  bool upload_eager = false;          // opened with the context (the first one of the process: see dev_init) -- trim_streams keeps it
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
