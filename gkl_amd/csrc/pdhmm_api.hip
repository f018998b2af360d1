// C-ABI implementation (include/gkl_hip_pdhmm.h) of the MI355X PDHMM path.  Compiled WITHOUT the
// fp64 flush-to-zero flag of the PairHMM translation unit: the reference's PDHMM never touches
// MXCSR (no _MM_SET_FLUSH_ZERO_MODE anywhere under src/main/native/pdhmm).
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <sys/sysinfo.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

#include "../../include/gkl_hip_pairhmm.h"  // status codes
#include "../../include/gkl_hip_pdhmm.h"
#include "pdhmm_kernel.h"
#include "pairhmm_plan.h"
#include "pairhmm_host_finalize.h"   // gklhip::WorkerPool: persistent threads for the host log10

using namespace gklhip;

namespace {
thread_local std::string g_pd_err;

int pd_fail(int status, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_pd_err = buf;
  return status;
}

#define PD_HIP_TRY(expr)                                                                          \
  do {                                                                                            \
    hipError_t e__ = (expr);                                                                      \
    if (e__ != hipSuccess) {                                                                      \
      (void)hipGetLastError();                                                                    \
      return pd_fail(e__ == hipErrorOutOfMemory ? GKLHIP_ERR_OOM : GKLHIP_ERR_HIP, "%s: %s", #expr, \
                     hipGetErrorString(e__));                                                     \
    }                                                                                             \
  } while (0)

// ---- host tables: ProbabilityCache of pdhmm-common.h:139-192 (exact 1/ln10 here, unlike PairHMM) ----
constexpr int kPdMaxQual = 254;
constexpr int kPdMmSize = ((kPdMaxQual + 1) * (kPdMaxQual + 2)) >> 1;

struct PdTables {
  std::vector<double> q2err, mm;
  double initial_condition, initial_condition_log10;
};

const PdTables& pd_tables() {
  static const PdTables t = [] {
    PdTables r;
    std::vector<double> jac(80001);
    for (int k = 0; k < 80001; k++) jac[k] = std::log10(1.0 + std::pow(10.0, -k * 0.0001));  // MathUtils.cc:84-87
    auto round_half_away = [](double d) { return d > 0.0 ? (int)(d + 0.5) : (int)(d - 0.5); };
    auto log10_sum = [&](double a, double b) {  // MathUtils.cc:91-109 (a <= b after the swap)
      if (a > b) std::swap(a, b);
      if (a == -1e10) return b;
      const double diff = b - a;
      return b + (diff < 8.0 ? jac[round_half_away(diff * (1.0 / 0.0001))] : 0.0);
    };
    const double inv_ln10 = 1.0 / std::log(10);
    r.mm.resize(kPdMmSize);
    for (int i = 0, offset = 0; i <= kPdMaxQual; offset += ++i)
      for (int j = 0; j <= i; j++) {
        const double l10 = std::log1p(-std::min(1.0, std::pow(10, log10_sum(-0.1 * i, -0.1 * j)))) * inv_ln10;
        r.mm[offset + j] = std::pow(10, l10);
      }
    r.q2err.resize(kPdMaxQual + 1);
    for (int q = 0; q <= kPdMaxQual; q++) r.q2err[q] = std::pow(10.0, (double)q / -10.0);
    r.initial_condition = std::pow(2, 1020);                       // MathUtils.cc:31
    r.initial_condition_log10 = std::log10(r.initial_condition);  // :32
    return r;
  }();
  return t;
}

// (like libgklhip_pairhmm's buffers: grown by the biggest call, given back when the last 16 calls each needed less than a
//  quarter of a buffer above 32 MB -- a 424k-pair call holds ~3 GB of streams and tables -- but not within 64 calls of the
//  buffer's last growth: hipFree / hipHostFree synchronise the whole device, and a workload that alternates one big call
//  with a few small ones must not free and re-make its buffers every round)
constexpr int kPdTrimCalls = 16, kPdTrimQuiet = 64;
inline bool pd_trim_due(size_t n, size_t cap, int* small_uses, int* since_grow) {
  if (*since_grow < kPdTrimQuiet) ++*since_grow;
  if (cap <= ((size_t)32 << 20) || n >= cap / 4) { *small_uses = 0; return false; }
  if (*small_uses < kPdTrimCalls) ++*small_uses;
  return *small_uses >= kPdTrimCalls && *since_grow >= kPdTrimQuiet;
}
struct Buf {
  void* p = nullptr;
  size_t cap = 0;
  int small_uses = 0, since_grow = kPdTrimQuiet;
  int reserve(size_t n) {
    if (n <= cap && !pd_trim_due(n, cap, &small_uses, &since_grow)) return GKLHIP_OK;
    if (n > cap) since_grow = 0;
    small_uses = 0;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    const size_t want = n + n / 4 + 256;
    PD_HIP_TRY(hipMalloc(&p, want));
    cap = want;
    return GKLHIP_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <typename T> T* as() const { return static_cast<T*>(p); }
};
struct PinBuf {  // page-locked host memory
  void* p = nullptr;
  size_t cap = 0;
  int small_uses = 0, since_grow = kPdTrimQuiet;
  int reserve(size_t n) {
    if (n <= cap && !pd_trim_due(n, cap, &small_uses, &since_grow)) return GKLHIP_OK;
    if (n > cap) since_grow = 0;
    small_uses = 0;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    const size_t want = n + n / 4 + 256;
    PD_HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
    cap = want;
    return GKLHIP_OK;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
  template <typename T> T* as() const { return static_cast<T*>(p); }
};
}  // namespace

struct gklhip_pdhmm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // big paired calls are cut into slices of pairs whose kernels run while later slices still cross PCIe (pd_run_locked)
  static constexpr int kMaxSlices = 8;
  hipStream_t up_stream = nullptr;
  gklhip::WorkerPool workers;           // host log10 of the sums: three helpers from 4096 pairs on
  std::vector<uint32_t> class_stamp;    // cross layout's class discovery: last haplotype that showed a (base, flags) pair
  uint32_t class_stamp_id = 0;
  hipEvent_t up_ev[kMaxSlices] = {}, sl_ev0[kMaxSlices] = {}, sl_ev1[kMaxSlices] = {};
  int pipeline = 1;                     // GKL_HIP_PDHMM_PIPELINE=0: one slice whatever the size
  std::mutex mu;
  Buf tables, inputs, entries, entries_tab, sums, misc, carry, jobs, tabx;
  PackScratch pack_scratch;
  PinBuf stage_in, stage_jobs, sums_pin;   // small calls: ONE copy per device buffer instead of one per array (9 + 17 of them)
  float last_ms = 0.f;
  int32_t last_routing[3] = {0, 0, 0};  // last cross call: haplotype items by kernel (table / predicate / byte-comparing); last paired call: packed jobs by kernel
  int use_table = 1;                    // GKL_HIP_PDHMM_TABLE=0: never route to the table kernel
  int fma_mode = 1;  // 1 = arithmetic of GKL's AVX-512 object (default), 0 = of its AVX2 object
  int tail_mode = 1; // 1 (default) = the last `batch mod SIMD width` pairs of every reference batch take the scalar engine's arithmetic, like GKL; 0 = vector arithmetic everywhere
};

extern "C" {

const char* gklhip_pdhmm_last_error(void) { return g_pd_err.c_str(); }

int64_t gklhip_pdhmm_get_table(int which, double* dst, int64_t cap) {
  const PdTables& t = pd_tables();
  const std::vector<double>* v = which == 0 ? &t.q2err : which == 1 ? &t.mm : nullptr;
  if (!v) return -1;
  if (dst) memcpy(dst, v->data(), sizeof(double) * (size_t)std::min<int64_t>(cap, (int64_t)v->size()));
  return (int64_t)v->size();
}

int gklhip_pdhmm_init(int device, gklhip_pdhmm_ctx** out_ctx) {
  if (!out_ctx) return pd_fail(GKLHIP_ERR_INVALID_ARG, "out_ctx is NULL");
  *out_ctx = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return pd_fail(GKLHIP_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU compute path)");
  }
  if (device < 0) PD_HIP_TRY(hipGetDevice(&device));
  if (device >= ndev) return pd_fail(GKLHIP_ERR_INVALID_ARG, "device %d of %d", device, ndev);
  PD_HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  PD_HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return pd_fail(GKLHIP_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
  gklhip_pdhmm_ctx* c = new (std::nothrow) gklhip_pdhmm_ctx();
  if (!c) return pd_fail(GKLHIP_ERR_OOM, "context allocation failed");
  c->device = device;
  auto bail = [&](int st) { gklhip_pdhmm_done(c); return st; };
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(pd_fail(GKLHIP_ERR_HIP, "hipStreamCreate failed"));
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) return bail(pd_fail(GKLHIP_ERR_HIP, "hipEventCreate failed"));
  if (hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking) != hipSuccess) return bail(pd_fail(GKLHIP_ERR_HIP, "hipStreamCreate failed"));
  for (int k = 0; k < gklhip_pdhmm_ctx::kMaxSlices; k++)
    if (hipEventCreateWithFlags(&c->up_ev[k], hipEventDisableTiming) != hipSuccess || hipEventCreate(&c->sl_ev0[k]) != hipSuccess ||
        hipEventCreate(&c->sl_ev1[k]) != hipSuccess)
      return bail(pd_fail(GKLHIP_ERR_HIP, "hipEventCreate failed"));
  const PdTables& t = pd_tables();
  int rc = c->tables.reserve((t.q2err.size() + t.mm.size()) * sizeof(double));
  if (rc) return bail(rc);
  if (hipMemcpy(c->tables.p, t.q2err.data(), t.q2err.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(c->tables.as<double>() + t.q2err.size(), t.mm.data(), t.mm.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
    return bail(pd_fail(GKLHIP_ERR_HIP, "table upload failed"));
  {
    const char* tm = getenv("GKL_HIP_PDHMM_TAIL");
    c->tail_mode = (tm && (strcmp(tm, "vector") == 0 || strcmp(tm, "0") == 0)) ? 0 : 1;  // "reference" (default) | "vector"
    const char* tb = getenv("GKL_HIP_PDHMM_TABLE");
    c->use_table = (tb && tb[0] == '0') ? 0 : 1;
    const char* pl = getenv("GKL_HIP_PDHMM_PIPELINE");
    c->pipeline = (pl && pl[0] == '0') ? 0 : 1;
  }
  if (c->use_table) {
    // once per process and device: does a DS read beyond the workgroup's LDS allocation return 0 here?  The table
    // kernel's idle entries depend on it (pdhmm_kernel.h: kPdTabIdle); if not, every haplotype takes the other kernels.
    static std::mutex mu;
    static std::vector<int> checked;   // 0 unknown, 1 good, -1 bad
    std::lock_guard<std::mutex> l(mu);
    if ((int)checked.size() <= device) checked.resize((size_t)device + 1, 0);
    if (checked[(size_t)device] == 0) {
      uint32_t* d_out = nullptr;
      uint32_t h_out = 1u;
      if (hipMalloc(reinterpret_cast<void**>(&d_out), 4) != hipSuccess) return bail(pd_fail(GKLHIP_ERR_OOM, "hipMalloc failed"));
      bool ok = hipMemsetAsync(d_out, 0, 4, c->stream) == hipSuccess;
      hipLaunchKernelGGL(pdhmm_idle_class_selftest_kernel, dim3(64), dim3(64), 0, c->stream, d_out);
      ok = ok && hipMemcpyAsync(&h_out, d_out, 4, hipMemcpyDeviceToHost, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
      (void)hipFree(d_out);
      checked[(size_t)device] = ok && h_out == 0u ? 1 : -1;
      if (checked[(size_t)device] < 0)
        fprintf(stderr, "[gklhip] pdhmm: LDS reads beyond the allocation do not return 0 on device %d (%08x): the table kernel is off\n", device, h_out);
    }
    if (checked[(size_t)device] < 0) c->use_table = 0;
  }
  *out_ctx = c;
  return GKLHIP_OK;
}

int gklhip_pdhmm_set_tail_mode(gklhip_pdhmm_ctx* c, int mode) {
  if (!c) return pd_fail(GKLHIP_ERR_INVALID_ARG, "context is NULL");
  if (mode != 0 && mode != 1) return pd_fail(GKLHIP_ERR_INVALID_ARG, "tail mode %d (0 or 1)", mode);
  std::lock_guard<std::mutex> lock(c->mu);
  c->tail_mode = mode;
  return GKLHIP_OK;
}

int gklhip_pdhmm_done(gklhip_pdhmm_ctx* c) {
  if (!c) return GKLHIP_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->up_stream) (void)hipStreamSynchronize(c->up_stream);
  for (int k = 0; k < gklhip_pdhmm_ctx::kMaxSlices; k++)
    for (hipEvent_t e : {c->up_ev[k], c->sl_ev0[k], c->sl_ev1[k]})
      if (e) (void)hipEventDestroy(e);
  if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
  for (Buf* b : {&c->tables, &c->inputs, &c->entries, &c->entries_tab, &c->sums, &c->misc, &c->carry, &c->jobs, &c->tabx}) b->release();
  for (PinBuf* b : {&c->stage_in, &c->stage_jobs, &c->sums_pin}) b->release();
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return GKLHIP_OK;
}

int gklhip_pdhmm_set_fma_mode(gklhip_pdhmm_ctx* c, int fma_mode) {
  if (!c) return pd_fail(GKLHIP_ERR_INVALID_ARG, "context is NULL");
  if (fma_mode != 0 && fma_mode != 1) return pd_fail(GKLHIP_ERR_INVALID_ARG, "fma_mode %d (0 or 1)", fma_mode);
  std::lock_guard<std::mutex> lock(c->mu);
  c->fma_mode = fma_mode;
  return GKLHIP_OK;
}

float gklhip_pdhmm_last_kernel_ms(gklhip_pdhmm_ctx* c) { return c ? c->last_ms : 0.f; }
int64_t gklhip_pdhmm_buffer_bytes(gklhip_pdhmm_ctx* c) {
  if (!c) return 0;
  std::lock_guard<std::mutex> lock(c->mu);
  size_t total = 0;
  for (const Buf* b : {&c->tables, &c->inputs, &c->entries, &c->entries_tab, &c->sums, &c->misc, &c->carry, &c->jobs, &c->tabx}) total += b->cap;
  for (const PinBuf* b : {&c->stage_in, &c->stage_jobs, &c->sums_pin}) total += b->cap;
  return (int64_t)total;
}
int gklhip_pdhmm_last_routing(gklhip_pdhmm_ctx* c, int32_t out[3]) {
  if (!c || !out) return pd_fail(GKLHIP_ERR_INVALID_ARG, "NULL argument");
  std::lock_guard<std::mutex> lock(c->mu);
  for (int i = 0; i < 3; i++) out[i] = c->last_routing[i];
  return GKLHIP_OK;
}

namespace {
// Shared by the two entry points.  Paired layout: n_read_items == n_hap_items == n_pairs, cross_haps = 0.
// Cross layout: n_pairs == n_read_items * n_hap_items, cross_haps = n_hap_items, pair p = (p / n_haps, p % n_haps).
struct PdProblem {
  int64_t n_pairs;
  int32_t n_read_items, n_hap_items, cross_haps, max_hap_len, max_read_len;
  const int8_t *hap_bases, *hap_pdbases, *read_bases, *read_qual, *read_ins_qual, *read_del_qual, *gcp;
  const int64_t *hap_lengths, *read_lengths;
  // cross layout, reference-tail mode: the reference cuts the read-major pair list into batches of this many pairs
  // (JavaData.h:83-101) and each batch has its own scalar tail; 0 = the whole cross product is one batch
  int64_t ref_batch_pairs;
};

int pd_validate(const PdProblem& q, const double* out_host) {
  if (q.max_hap_len <= 0 || q.max_read_len <= 0)
    return pd_fail(GKLHIP_ERR_INVALID_ARG, "maxHapLength / maxReadLength must be greater than 0");
  if (!q.hap_bases || !q.hap_pdbases || !q.read_bases || !q.read_qual || !q.read_ins_qual || !q.read_del_qual ||
      !q.gcp || !q.hap_lengths || !q.read_lengths || !out_host)
    return pd_fail(GKLHIP_ERR_INVALID_ARG, "Input arrays aren't valid.");
  if (q.n_pairs > 0x7fffffffLL) return pd_fail(GKLHIP_ERR_INVALID_ARG, "more than 2^31 pairs");
  for (int i = 0; i < q.n_hap_items; i++)
    if (q.hap_lengths[i] < 1 || q.hap_lengths[i] > q.max_hap_len)
      return pd_fail(GKLHIP_ERR_INVALID_ARG, "hap_lengths[%d] = %lld outside 1..%d", i, (long long)q.hap_lengths[i], q.max_hap_len);
  for (int i = 0; i < q.n_read_items; i++)
    if (q.read_lengths[i] < 1 || q.read_lengths[i] > q.max_read_len)
      return pd_fail(GKLHIP_ERR_INVALID_ARG, "read_lengths[%d] = %lld outside 1..%d", i, (long long)q.read_lengths[i], q.max_read_len);
  return GKLHIP_OK;
}

int pd_run_locked(gklhip_pdhmm_ctx* c, const PdProblem& q, double* out_host);

// Does the haplotype hold a base outside ACGTN?  (Such columns need the byte-comparing step: the job goes to the full kernel.)
bool has_odd_base_scalar(const int8_t* b, int64_t n) {
  unsigned ok = 1;
  for (int64_t j = 0; j < n; j++)
    ok &= (unsigned)(b[j] == 'A') | (unsigned)(b[j] == 'C') | (unsigned)(b[j] == 'G') | (unsigned)(b[j] == 'T') | (unsigned)(b[j] == 'N');
  return !ok;
}
__attribute__((target("avx2"))) bool has_odd_base_avx2(const int8_t* b, int64_t n) {
  const __m256i cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T'),
                cN = _mm256_set1_epi8('N');
  __m256i all = _mm256_set1_epi8((char)0xff);
  int64_t j = 0;
  for (; j + 32 <= n; j += 32) {
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(b + j));
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, cA), _mm256_cmpeq_epi8(v, cC)),
                                                       _mm256_or_si256(_mm256_cmpeq_epi8(v, cG), _mm256_cmpeq_epi8(v, cT))),
                                       _mm256_cmpeq_epi8(v, cN));
    all = _mm256_and_si256(all, ok);
  }
  if (_mm256_movemask_epi8(all) != -1) return true;
  return has_odd_base_scalar(b + j, n - j);
}
bool has_odd_base(const int8_t* b, int64_t n) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  return avx2 ? has_odd_base_avx2(b, n) : has_odd_base_scalar(b, n);
}

// An error return must not leave asynchronous copies from this call's host vectors (or the caller's arrays) in
// flight when those go out of scope: drain the stream first.
int pd_run(gklhip_pdhmm_ctx* c, const PdProblem& q, double* out_host) {
  std::lock_guard<std::mutex> lock(c->mu);
  int rc;
  // no C++ exception leaves the C ABI (a host vector that cannot grow, a helper thread that cannot start): it becomes a
  // status like any other error -- after the same drain
  try { rc = pd_run_locked(c, q, out_host); }
  catch (const std::bad_alloc&) { rc = pd_fail(GKLHIP_ERR_OOM, "host memory allocation failed"); }
  catch (const std::exception& e) { rc = pd_fail(GKLHIP_ERR_HIP, "%s", e.what()); }
  catch (...) { rc = pd_fail(GKLHIP_ERR_HIP, "unexpected C++ exception"); }
  if (rc != GKLHIP_OK) {
    const std::string keep = g_pd_err;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->up_stream) (void)hipStreamSynchronize(c->up_stream);
    (void)hipGetLastError();
    g_pd_err = keep;
  }
  return rc;
}

constexpr size_t kPdStageBytes = (size_t)4 << 20;
// Cross layout: reads are packed into 64-lane chunks by best fit within windows of this many reads.  The chunks are reused
// for every haplotype, so a fuller chunk pays off nh times: 2048 fills 99.2 % of the lanes on the reference's fixture
// (192, the paired layout's window: 97.7 %).
constexpr int kPdCrossWindow = 2048;
int pd_run_locked(gklhip_pdhmm_ctx* c, const PdProblem& q, double* out_host) {
  static const bool timing = getenv("GKLHIP_TIMING") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  auto ms_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  PD_HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const size_t n = (size_t)q.n_pairs;
  const size_t nh = (size_t)q.n_hap_items, nr = (size_t)q.n_read_items;
  const size_t hap_bytes = nh * (size_t)q.max_hap_len, read_bytes = nr * (size_t)q.max_read_len;
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t o_hb = 0, o_hp = up(hap_bytes), o_rb = o_hp + up(hap_bytes), o_rq = o_rb + up(read_bytes),
               o_ri = o_rq + up(read_bytes), o_rd = o_ri + up(read_bytes), o_gc = o_rd + up(read_bytes),
               o_hl = o_gc + up(read_bytes), o_rl = o_hl + up(nh * 8), total = o_rl + up(nr * 8);
  int rc;
  if ((rc = c->inputs.reserve(total))) return rc;
  unsigned char* d = c->inputs.as<unsigned char>();
  // The nine input copies (the paired layout of a big batch is hundreds of MB of padded [pair][maxLen] arrays: the
  // calls alone -- pinning the caller's pages -- take milliseconds): a helper thread issues them while this one builds
  // the jobs; both meet before anything else goes onto the stream.
  // A big PAIRED call (computePDHMMNative: ~1.2 KB of padded input per pair -- the x32 fixture is 498 MB, 9.5 ms of PCIe
  // against 5 ms of kernels) is cut into slices of consecutive pairs, biggest first: the helper thread sends slice after
  // slice on the upload stream and records an event behind each; the kernels of a slice -- its pairs are packed among
  // themselves -- wait for that event only, so they run while the later slices still cross the bus, and the last,
  // smallest slice leaves little to do once the bus is done.  Everything else about the call is one problem: one set of
  // device arrays, one job list (slice k = a range of listed jobs), the rare striped / odd-base / tail jobs at the end.
  int n_slices = 1;
  size_t slice_lo[gklhip_pdhmm_ctx::kMaxSlices + 1] = {0, n};
  if (!q.cross_haps && c->pipeline && n >= 65536 && hap_bytes + 5 * read_bytes >= ((size_t)64 << 20)) {
    static const double kShare[] = {0.28, 0.24, 0.19, 0.13, 0.09, 0.05, 0.02};
    n_slices = (int)(sizeof kShare / sizeof kShare[0]);
    double acc = 0.0;
    for (int k = 0; k < n_slices; k++) { slice_lo[k] = (size_t)(acc * (double)n) / 64 * 64; acc += kShare[k]; }
    slice_lo[n_slices] = n;
  }
  struct Uploads {
    hipError_t err = hipSuccess;
    std::thread th;
    std::atomic<int> recorded{0};   // slices whose upload event has been recorded (or given up on)
    ~Uploads() { if (th.joinable()) th.join(); }
  } up_th;
  auto do_uploads = [&, d]() {
    (void)hipSetDevice(c->device);
    hipStream_t us = n_slices > 1 ? c->up_stream : s;
    auto cp = [&](size_t off, const void* src, size_t bytes) {
      if (up_th.err != hipSuccess || bytes == 0) return;
      up_th.err = hipMemcpyAsync(d + off, src, bytes, hipMemcpyHostToDevice, us);
    };
    for (int k = 0; k < n_slices; k++) {
      // (paired layout: item i of every array belongs to pair i; cross layout: one slice, all items)
      const size_t h0 = n_slices > 1 ? slice_lo[k] : 0, h1 = n_slices > 1 ? slice_lo[k + 1] : nh;
      const size_t r0 = n_slices > 1 ? slice_lo[k] : 0, r1 = n_slices > 1 ? slice_lo[k + 1] : nr;
      const size_t mh = (size_t)q.max_hap_len, mr = (size_t)q.max_read_len;
      cp(o_hl + h0 * 8, q.hap_lengths + h0, (h1 - h0) * 8); cp(o_rl + r0 * 8, q.read_lengths + r0, (r1 - r0) * 8);
      cp(o_hb + h0 * mh, q.hap_bases + h0 * mh, (h1 - h0) * mh); cp(o_hp + h0 * mh, q.hap_pdbases + h0 * mh, (h1 - h0) * mh);
      cp(o_rb + r0 * mr, q.read_bases + r0 * mr, (r1 - r0) * mr); cp(o_rq + r0 * mr, q.read_qual + r0 * mr, (r1 - r0) * mr);
      cp(o_ri + r0 * mr, q.read_ins_qual + r0 * mr, (r1 - r0) * mr); cp(o_rd + r0 * mr, q.read_del_qual + r0 * mr, (r1 - r0) * mr);
      cp(o_gc + r0 * mr, q.gcp + r0 * mr, (r1 - r0) * mr);
      if (n_slices > 1) {
        if (up_th.err == hipSuccess) up_th.err = hipEventRecord(c->up_ev[k], us);
        up_th.recorded.store(k + 1, std::memory_order_release);
      }
    }
  };
  // A call of the fixture's size (276 reads x 48 haplotypes) spent half its time in two dozen small copies from
  // pageable memory (~10 us each): up to kPdStageBytes the arrays are gathered in a pinned block and travel in ONE copy.
  const bool staged = total <= kPdStageBytes;
  if (staged) {
    if ((rc = c->stage_in.reserve(total))) return rc;
    unsigned char* h = c->stage_in.as<unsigned char>();
    memcpy(h + o_hb, q.hap_bases, hap_bytes); memcpy(h + o_hp, q.hap_pdbases, hap_bytes);
    memcpy(h + o_rb, q.read_bases, read_bytes); memcpy(h + o_rq, q.read_qual, read_bytes); memcpy(h + o_ri, q.read_ins_qual, read_bytes);
    memcpy(h + o_rd, q.read_del_qual, read_bytes); memcpy(h + o_gc, q.gcp, read_bytes);
    memcpy(h + o_hl, q.hap_lengths, nh * 8); memcpy(h + o_rl, q.read_lengths, nr * 8);
    PD_HIP_TRY(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, s));
  } else if (n_slices > 1 || hap_bytes + 5 * read_bytes >= ((size_t)8 << 20)) {
    up_th.th = std::thread(do_uploads);
  } else {
    do_uploads();
  }
  const double ms_uploads = ms_since(t_begin);
  const int cross = q.cross_haps;
  auto read_len_of = [&](size_t p) { return (int)q.read_lengths[cross ? p / (size_t)cross : p]; };
  auto hap_len_of = [&](size_t p) { return (int)q.hap_lengths[cross ? p % (size_t)cross : p]; };

  // ---- jobs ----
  std::vector<int32_t> place_chunk;            // paired layout: compact packing of the pairs (pack_reads_place)
  std::vector<uint8_t> place_lane, chunk_used;
  size_t n_striped = 0;
  std::vector<int32_t> job_pair, job_steps;
  std::vector<uint8_t> job_striped;
  std::vector<PlanLane> cross_lanes;           // cross layout: [chunk][64] = {read item, block}
  std::vector<int32_t> hap_order, chunk_steps, chunk_rep;
  int32_t slice_chunk0[gklhip_pdhmm_ctx::kMaxSlices + 1] = {0, 0};   // paired layout: slice k = chunks [slice_chunk0[k], slice_chunk0[k + 1]) = listed jobs n_striped + those
  size_t n_tail = 0;                           // paired layout, tail mode: the last n_tail pairs
  std::vector<PlanLane> tail_lanes;
  std::vector<int32_t> tail_pair, tail_steps;
  std::vector<uint8_t> tail_striped;
  if (cross) {
    // reads are packed into 64-lane chunks ONCE; every chunk meets every haplotype (longest haplotypes first).
    std::vector<int64_t> read_off(nr + 1, 0);
    for (size_t r = 0; r < nr; r++) read_off[r + 1] = read_off[r] + q.read_lengths[r];
    std::vector<int32_t> shorts;
    shorts.reserve(nr);
    for (size_t r = 0; r < nr; r++) {
      if (blocks_for((int)q.read_lengths[r], kPdRpl) <= kLanes) { shorts.push_back((int32_t)r); continue; }
      for (size_t h = 0; h < nh; h++) {  // a read that needs more than 64 lanes (64 x kPdRpl = 384 bases or more): one striped job per haplotype
        job_pair.push_back((int32_t)(r * nh + h)); job_striped.push_back(1); job_steps.push_back(0);
      }
    }
    const int made = pack_reads_windowed(shorts.data(), (int)shorts.size(), read_off.data(), kPdRpl, kPdCrossWindow, &cross_lanes, nullptr);
    chunk_steps.assign((size_t)made, 0);
    chunk_rep.assign((size_t)made, 0);
    for (int k = 0; k < made; k++) {
      const PlanLane* row = cross_lanes.data() + (size_t)k * kLanes;
      int32_t rep = -1, top = 0;
      for (int l = 0; l < kLanes; l++) {
        if (row[l].read < 0) continue;
        if (rep < 0) rep = row[l].read;
        top = std::max(top, row[l].block);
      }
      chunk_steps[(size_t)k] = top;
      chunk_rep[(size_t)k] = rep;
    }
    hap_order.resize(nh);
    for (size_t h = 0; h < nh; h++) hap_order[h] = (int32_t)h;
    std::stable_sort(hap_order.begin(), hap_order.end(),
                     [&](int32_t x, int32_t y) { return q.hap_lengths[x] > q.hap_lengths[y]; });
    // "Reference tail" of the cross product: computeLikelihoods expands it into read-major pairs, batch by batch
    // (JavaData.h:177-242), and computePDHMM finishes the last `batch mod SIMD width` pairs of EVERY batch with the
    // scalar engine (pdhmm.h:1264-1268).  The main launch computes all pairs with the vector arithmetic; the pairs at
    // those positions are then recomputed by the scalar-arithmetic instantiation and overwrite their sums.
    if (c->tail_mode == 1) {
      const size_t width = c->fma_mode ? 8 : 4;
      const size_t per = q.ref_batch_pairs > 0 ? (size_t)std::min<int64_t>(q.ref_batch_pairs, (int64_t)n) : n;
      for (size_t start = 0; start < n; start += per) {
        const size_t nb_pairs = std::min(per, n - start);
        for (size_t i = start + nb_pairs / width * width; i < start + nb_pairs; i++) {
          const int nb = blocks_for(read_len_of(i), kPdRpl);
          tail_pair.push_back((int32_t)i);
          tail_striped.push_back(nb > kLanes ? 1 : 0);
          tail_steps.push_back(hap_len_of(i) + std::min(nb, kLanes) - 1);
          tail_lanes.resize(tail_lanes.size() + kLanes, PlanLane{-1, 0});
          if (nb <= kLanes)
            for (int b = 0; b < nb; b++) tail_lanes[tail_lanes.size() - kLanes + (size_t)b] = PlanLane{(int32_t)i, b};
        }
      }
      n_tail = tail_pair.size();
    }
  } else {
    // "Reference tail": GKL finishes the last `batch mod SIMD width` pairs of every vector batch with its SCALAR
    // engine (pdhmm.h:1264-1270; 8 doubles per AVX-512 vector, 4 per AVX2 vector), whose arithmetic differs in the
    // last bits (and, with deletions at a haplotype's end, by more).  In that mode those pairs get jobs of their own,
    // run by the scalar-arithmetic instantiation of the kernel.
    if (c->tail_mode == 1) n_tail = n % (size_t)(c->fma_mode ? 8 : 4);
    const size_t n_vec = n - n_tail;
    for (size_t i = n_vec; i < n; i++) {
      const int nb = blocks_for(read_len_of(i), kPdRpl);
      tail_pair.push_back((int32_t)i);
      tail_striped.push_back(nb > kLanes ? 1 : 0);
      tail_steps.push_back(hap_len_of(i) + std::min(nb, kLanes) - 1);
      tail_lanes.resize(tail_lanes.size() + kLanes, PlanLane{-1, 0});
      if (nb <= kLanes)
        for (int b = 0; b < nb; b++) tail_lanes[tail_lanes.size() - kLanes + (size_t)b] = PlanLane{(int32_t)i, b};
    }
    // short pairs, ordered by haplotype length so that wavefront mates finish together, are packed best-fit
    // into 64-lane chunks; a read that needs more than 64 lanes becomes a striped job
    std::vector<int64_t> pair_off(n + 1, 0);  // pack_reads_windowed() addresses reads through offsets
    for (size_t i = 0; i < n; i++) pair_off[i + 1] = pair_off[i] + read_len_of(i);
    for (size_t i = 0; i < n_vec; i++) {
      if (blocks_for(read_len_of(i), kPdRpl) <= kLanes) continue;
      job_pair.push_back((int32_t)i); job_striped.push_back(1); job_steps.push_back(0);  // (a striped job's lane row stays unused)
    }
    // compact packing (5 bytes per pair); pdhmm_expand_kernel turns it into the lane rows on the device.  Slice by slice
    // (one slice: the whole batch): the pairs of a slice share chunks among themselves only, chunk numbers run on.
    n_striped = job_pair.size();
    place_chunk.assign(n, -1);
    place_lane.assign(n, 0);
    std::vector<int32_t> shorts, cnt;
    shorts.reserve(n);
    for (int k = 0; k < n_slices; k++) {
      const size_t lo = std::min(slice_lo[k], n_vec), hi = std::min(slice_lo[k + 1], n_vec);
      slice_chunk0[k] = (int32_t)chunk_used.size();
      // counting sort by haplotype length, longest first (the big jobs start first)
      cnt.assign((size_t)q.max_hap_len + 2, 0);
      size_t n_short = 0;
      for (size_t i = lo; i < hi; i++)
        if (blocks_for(read_len_of(i), kPdRpl) <= kLanes) { cnt[(size_t)hap_len_of(i)]++; n_short++; }
      int32_t acc = 0;
      for (int64_t h = q.max_hap_len; h >= 0; h--) { const int32_t m = cnt[(size_t)h]; cnt[(size_t)h] = acc; acc += m; }
      shorts.resize(n_short);
      for (size_t i = lo; i < hi; i++)
        if (blocks_for(read_len_of(i), kPdRpl) <= kLanes) shorts[(size_t)cnt[(size_t)hap_len_of(i)]++] = (int32_t)i;
      const int made = pack_reads_place(shorts.data(), (int)shorts.size(), pair_off.data(), kPdRpl, 192, place_chunk.data(),
                                        place_lane.data(), &chunk_used, nullptr, &c->pack_scratch);
      const size_t j0 = n_striped + (size_t)slice_chunk0[k];
      job_pair.resize(j0 + (size_t)made, -1);
      job_steps.resize(j0 + (size_t)made, 0);
      job_striped.resize(j0 + (size_t)made, 0);
      for (const int32_t i : shorts) {
        const size_t j = n_striped + (size_t)place_chunk[(size_t)i];
        if (job_pair[j] < 0) job_pair[j] = i;
        job_steps[j] = std::max(job_steps[j], (int32_t)(hap_len_of((size_t)i) + blocks_for(read_len_of((size_t)i), kPdRpl) - 1));
      }
    }
    slice_chunk0[n_slices] = (int32_t)chunk_used.size();
  }
  const double ms_jobs = ms_since(t_begin);
  // ---- routing: the hot launch (only the two in-place step loops, see pdhmm_fwd_kernel) takes every job without a
  // striped read and without a haplotype that has a base outside ACGTN (such columns need the byte-comparing step);
  // the rest -- rare -- go to a second launch of the full kernel.  Cross layout: the host looks at the (few)
  // haplotypes and orders them by kernel.  Listed jobs (paired layout: every job): routed on the device
  // (PdArgs::job_flags) -- a host scan of a haplotype per PAIR is ~100 MB for 400k pairs.
  std::vector<uint8_t> hap_odd;
  if (cross) {
    hap_odd.assign(nh, 0);
    for (size_t h = 0; h < nh; h++) hap_odd[h] = has_odd_base(q.hap_bases + h * (size_t)q.max_hap_len, q.hap_lengths[h]) ? 1 : 0;
  }
  size_t n_clean_haps = nh, n_tab_haps = 0;
  // cross layout: a clean haplotype whose columns fall into at most kPdTabClasses classes of (base, SNP alleles, 'N')
  // goes to the table kernel (pdhmm_fwd_tab_kernel); class c of haplotype h has the match bits class_codes[8 h + c]
  std::vector<uint8_t> hap_ncls;
  std::vector<uint32_t> class_codes;
  if (cross) {
    hap_ncls.assign(nh, 0);
    class_codes.assign(nh * 8, 0u);
    for (size_t h = 0; h < nh && c->use_table; h++) {
      if (hap_odd[h]) continue;
      const int8_t* hb = q.hap_bases + h * (size_t)q.max_hap_len;
      const int8_t* pd = q.hap_pdbases + h * (size_t)q.max_hap_len;
      uint32_t* codes = class_codes.data() + h * 8;
      int ncls = 0;
      // (a column's class is a function of its (base, flags) pair: 2^15 of them, a haplotype shows a handful -- a stamp
      //  per pair and haplotype skips the columns whose pair has been seen; this loop was 0.06 of a region call's 0.33 ms)
      if (c->class_stamp.empty()) c->class_stamp.assign(1u << 15, 0u);
      if (++c->class_stamp_id == 0u) { std::fill(c->class_stamp.begin(), c->class_stamp.end(), 0u); c->class_stamp_id = 1u; }
      const uint32_t stamp_id = c->class_stamp_id;
      uint32_t* const stamp = c->class_stamp.data();
      for (int64_t j = 0; j < q.hap_lengths[h] && ncls <= kPdTabClasses; j++) {
        // (as pdhmm_entries_kernel builds the entry's match bits)
        const uint32_t yb = (uint32_t)hb[j] & 0xffu, flags = (uint32_t)pd[j] & 0x7fu;
        uint32_t& seen = stamp[(yb << 7) | flags];
        if (seen == stamp_id) continue;
        seen = stamp_id;
        const uint32_t hot = yb == (uint32_t)'A' ? 1u : yb == (uint32_t)'C' ? 2u : yb == (uint32_t)'G' ? 4u : yb == (uint32_t)'T' ? 8u : 0u;
        const uint32_t allele = (flags & kPdSnp) ? ((flags >> 3) & 0xfu) : 0u;
        const uint32_t code = (hot << 20) | (allele << 24) | (1u << 28) | (yb == (uint32_t)'N' ? 1u << 29 : 0u);
        int k = 0;
        while (k < ncls && codes[k] != code) k++;
        if (k == ncls) {
          if (ncls < kPdTabClasses) codes[ncls] = code;
          ncls++;
        }
      }
      hap_ncls[h] = ncls <= kPdTabClasses ? (uint8_t)ncls : 0;
    }
    // order: table haplotypes, then the other clean ones, then those with odd bases (each group longest first)
    std::stable_partition(hap_order.begin(), hap_order.end(), [&](int32_t h) { return hap_odd[(size_t)h] == 0; });
    n_clean_haps = 0;
    for (size_t h = 0; h < nh; h++) n_clean_haps += hap_odd[h] == 0;
    std::stable_partition(hap_order.begin(), hap_order.begin() + (ptrdiff_t)n_clean_haps, [&](int32_t h) { return hap_ncls[(size_t)h] != 0; });
    for (size_t h = 0; h < nh; h++) n_tab_haps += hap_ncls[h] != 0;
    // The haplotypes of one region share their variant sites, hence their column classes: when the union of the table
    // haplotypes' classes still fits the table, every one of them gets the union as its list -- a wavefront that takes
    // several haplotypes with the same chunk of reads (tab_group_start below) then builds the table once.
    {
      uint32_t uni[kPdTabClasses + 1];
      int n_uni = 0;
      for (size_t h = 0; h < nh && n_uni <= kPdTabClasses; h++)
        for (int k = 0; k < (int)hap_ncls[h] && n_uni <= kPdTabClasses; k++) {
          const uint32_t code = class_codes[h * 8 + (size_t)k];
          int at = 0;
          while (at < n_uni && uni[at] != code) at++;
          if (at == n_uni) { if (n_uni < kPdTabClasses) uni[n_uni] = code; n_uni++; }
        }
      if (n_uni <= kPdTabClasses)
        for (size_t h = 0; h < nh; h++)
          if (hap_ncls[h]) { for (int k = 0; k < n_uni; k++) class_codes[h * 8 + (size_t)k] = uni[k]; hap_ncls[h] = (uint8_t)n_uni; }
    }
    c->last_routing[0] = (int32_t)n_tab_haps; c->last_routing[1] = (int32_t)(n_clean_haps - n_tab_haps); c->last_routing[2] = (int32_t)(nh - n_clean_haps);
  } else {
    c->last_routing[0] = c->last_routing[1] = c->last_routing[2] = 0;
  }
  const double ms_routing = ms_since(t_begin);
  const int n_chunks_cross = (int)chunk_steps.size();
  const int64_t n_cross_jobs64 = (int64_t)n_chunks_cross * (int64_t)(cross ? nh : 0);
  if (n_cross_jobs64 + (int64_t)job_pair.size() > 0x7fffffffLL) return pd_fail(GKLHIP_ERR_INVALID_ARG, "too many jobs");
  const int n_cross_jobs = (int)n_cross_jobs64;
  const int n_cross_tab = cross ? (int)((int64_t)n_chunks_cross * (int64_t)n_tab_haps) : 0;
  // Table launch: a unit of work is (a group of consecutive table haplotypes, a chunk of reads) -- the wavefront sets the
  // chunk's rows up once per group (a twelfth of a job's time otherwise).  Groups of up to six while there are at least
  // six units per wavefront, shrinking to single haplotypes over the last part of the list (they run last and even the load out).
  std::vector<int32_t> tab_group_start;
  if (n_cross_tab > 0) {
    const int64_t per_wave = (int64_t)n_cross_tab / (256 * 8);
    const int group_max = (int)std::max<int64_t>(1, std::min<int64_t>(6, per_wave / 6));
    for (size_t k = 0; k < n_tab_haps;) {   // sizes shrink towards the end of the list: the last units are single haplotypes
      const size_t left = n_tab_haps - k;
      const size_t g = std::max<size_t>(1, std::min<size_t>((size_t)group_max, left / 7));
      tab_group_start.push_back((int32_t)k);
      k += g;
    }
    tab_group_start.push_back((int32_t)n_tab_haps);
  }
  const int n_tab_units = tab_group_start.empty() ? 0 : (int)((int64_t)n_chunks_cross * (int64_t)(tab_group_start.size() - 1));
  const int n_cross_hot = cross ? (int)((int64_t)n_chunks_cross * (int64_t)(n_clean_haps - n_tab_haps)) : 0;
  const int n_general = (int)job_pair.size();
  const int n_jobs = n_cross_jobs + n_general;
  const int entry_stride = (q.max_hap_len + 2 * kLanes + 4 + 63) / 64 * 64;   // 64 idle, the columns, 63 skew + 4 look-ahead
  const int carry_len = entry_stride;
  const int n_blocks = std::max(1, std::min(std::max(n_jobs, (int)n_tail), 256 * 8));
  // Paired layout through the table kernel (pdhmm_fwd_tab_paired_kernel): classes, special columns and the routing of the
  // jobs are found on the device.  Needs the program's 32-bit entry offsets to reach every item's stream and the step
  // marks of a job to fit the special kernel's LDS.
  const size_t n_packed = cross ? 0 : (size_t)n_general - n_striped;
  const bool tab_paired = !cross && c->use_table && n_packed > 0 && nh * (size_t)entry_stride * 4 < ((size_t)1 << 32) && entry_stride <= 12 * 1024;
  const bool tab_paired_asm = tab_paired && c->fma_mode == 1 && GKL_PD_ASM == 2;   // (the C++ step loops ballot on the lanes' own entries)
  const int sb_stride = entry_stride / 64, ns_stride = entry_stride;
  const size_t x_nc = 0, x_cc = up(nh), x_sb = x_cc + up(nh * 32), x_nt = x_sb + up(nh * (size_t)sb_stride * 8), x_hj = x_nt + up((size_t)n_general),
               x_ns = x_hj + up((size_t)n_general * 4), x_total = x_ns + (tab_paired_asm ? up((size_t)n_general * (size_t)ns_stride * 4) : 0);
  if ((rc = c->entries.reserve(nh * (size_t)entry_stride * 4))) return rc;
  if (n_tab_haps && (rc = c->entries_tab.reserve(2 * nh * (size_t)entry_stride * 4))) return rc;   // + the next-special-column table
  if (tab_paired && ((rc = c->entries_tab.reserve(nh * (size_t)entry_stride * 4)) || (rc = c->tabx.reserve(x_total)))) return rc;
  if ((rc = c->sums.reserve(n * 8))) return rc;
  if ((rc = c->misc.reserve(256))) return rc;
  if ((rc = c->carry.reserve((size_t)n_blocks * 2 * (6 * (size_t)carry_len + 64) * 8))) return rc;
  const size_t o_jl = 0, o_jp = up((size_t)n_general * kLanes * sizeof(PlanLane)), o_jn = o_jp + up((size_t)n_general * 4),
               o_js = o_jn + up((size_t)n_general * 4), o_cl = o_js + up((size_t)n_general),
               o_ho = o_cl + up(cross_lanes.size() * sizeof(PlanLane)), o_cs = o_ho + up(hap_order.size() * 4),
               o_cr = o_cs + up(chunk_steps.size() * 4), o_tl = o_cr + up(chunk_rep.size() * 4),
               o_tp = o_tl + up(tail_lanes.size() * sizeof(PlanLane)), o_tn = o_tp + up(n_tail * 4), o_ts = o_tn + up(n_tail * 4),
               o_nc = o_ts + up(n_tail), o_cc = o_nc + up(hap_ncls.size()), o_jf = o_cc + up(class_codes.size() * 4),
               o_pc = o_jf + up((size_t)n_general), o_pl = o_pc + up(place_chunk.size() * 4), o_cu = o_pl + up(place_lane.size()),
               o_fj = o_cu + up(chunk_used.size()), o_tg = o_fj + up((size_t)n_general * 4), jobs_total = o_tg + up(tab_group_start.size() * 4);
  if (n_slices == 1) {   // (sliced call: the kernels of slice k wait for slice k's upload event, see below)
    if (up_th.th.joinable()) up_th.th.join();
    PD_HIP_TRY(up_th.err);
  }
  if ((rc = c->jobs.reserve(jobs_total + 256))) return rc;
  unsigned char* dj = c->jobs.as<unsigned char>();
  const bool staged_jobs = jobs_total <= kPdStageBytes;
  if (staged_jobs && (rc = c->stage_jobs.reserve(jobs_total + 256))) return rc;
  unsigned char* hj = c->stage_jobs.as<unsigned char>();
  size_t staged_hi = 0;  // bytes of the staging block in use
  auto put = [&](size_t off, const void* src, size_t bytes) {
    if (!bytes) return hipSuccess;
    if (staged_jobs) { memcpy(hj + off, src, bytes); staged_hi = std::max(staged_hi, off + bytes); return hipSuccess; }
    return hipMemcpyAsync(dj + off, src, bytes, hipMemcpyHostToDevice, s);
  };
  PD_HIP_TRY(put(o_jp, job_pair.data(), (size_t)n_general * 4));
  PD_HIP_TRY(put(o_jn, job_steps.data(), (size_t)n_general * 4));
  PD_HIP_TRY(put(o_js, job_striped.data(), (size_t)n_general));
  PD_HIP_TRY(put(o_cl, cross_lanes.data(), cross_lanes.size() * sizeof(PlanLane)));
  PD_HIP_TRY(put(o_ho, hap_order.data(), hap_order.size() * 4));
  PD_HIP_TRY(put(o_cs, chunk_steps.data(), chunk_steps.size() * 4));
  PD_HIP_TRY(put(o_cr, chunk_rep.data(), chunk_rep.size() * 4));
  PD_HIP_TRY(put(o_tl, tail_lanes.data(), tail_lanes.size() * sizeof(PlanLane)));
  PD_HIP_TRY(put(o_tp, tail_pair.data(), n_tail * 4));
  PD_HIP_TRY(put(o_tn, tail_steps.data(), n_tail * 4));
  PD_HIP_TRY(put(o_ts, tail_striped.data(), n_tail));
  PD_HIP_TRY(put(o_nc, hap_ncls.data(), hap_ncls.size()));
  PD_HIP_TRY(put(o_cc, class_codes.data(), class_codes.size() * 4));
  PD_HIP_TRY(put(o_pc, place_chunk.data(), place_chunk.size() * 4));
  PD_HIP_TRY(put(o_pl, place_lane.data(), place_lane.size()));
  PD_HIP_TRY(put(o_cu, chunk_used.data(), chunk_used.size()));
  PD_HIP_TRY(put(o_tg, tab_group_start.data(), tab_group_start.size() * 4));
  if (n_general > 0) {
    if (staged_jobs) { memset(hj + o_jf, 0, (size_t)n_general); staged_hi = std::max(staged_hi, o_jf + (size_t)n_general); }
    else PD_HIP_TRY(hipMemsetAsync(dj + o_jf, 0, (size_t)n_general, s));
  }
  // the full launch's list of listed jobs: the striped ones (the first n_striped in the paired layout, all of them in
  // the cross layout) from here, flagged packed jobs appended by pdhmm_collect_kernel; its length lives in misc[5]
  const int32_t n_striped_listed = cross ? n_general : (int32_t)n_striped;   // (lives, like the vectors, until the stream is drained below)
  std::vector<int32_t> full_first((size_t)n_striped_listed);
  for (int32_t k = 0; k < n_striped_listed; k++) full_first[(size_t)k] = k;
  PD_HIP_TRY(hipMemsetAsync(c->misc.p, 0, 256, s));
#ifdef GKL_PD_PROF
  PD_HIP_TRY(hipMemsetAsync(c->misc.as<char>() + 128 + 13 * 8, 0xff, 8, s));
  PD_HIP_TRY(hipMemsetAsync(c->misc.as<char>() + 128 + 15 * 8, 0xff, 8, s));
#endif
  PD_HIP_TRY(put(o_fj, full_first.data(), full_first.size() * 4));
  if (staged_jobs && staged_hi > 0) PD_HIP_TRY(hipMemcpyAsync(dj, hj, staged_hi, hipMemcpyHostToDevice, s));
  PD_HIP_TRY(hipMemcpyAsync(c->misc.as<int32_t>() + 5, &n_striped_listed, 4, hipMemcpyHostToDevice, s));

  const PdTables& t = pd_tables();
  PdArgs a;
  a.hap_bases = reinterpret_cast<const int8_t*>(d + o_hb);
  a.hap_pdbases = reinterpret_cast<const int8_t*>(d + o_hp);
  a.read_bases = reinterpret_cast<const int8_t*>(d + o_rb);
  a.read_qual = reinterpret_cast<const int8_t*>(d + o_rq);
  a.read_ins = reinterpret_cast<const int8_t*>(d + o_ri);
  a.read_del = reinterpret_cast<const int8_t*>(d + o_rd);
  a.gcp = reinterpret_cast<const int8_t*>(d + o_gc);
  a.hap_len = reinterpret_cast<const int64_t*>(d + o_hl);
  a.read_len = reinterpret_cast<const int64_t*>(d + o_rl);
  a.batch = (int32_t)n; a.max_hap = q.max_hap_len; a.max_read = q.max_read_len;
  a.cross_haps = cross; a.n_hap_items = q.n_hap_items;
  a.q2err = c->tables.as<double>();
  a.mm_prob = c->tables.as<double>() + t.q2err.size();
  a.entries = c->entries.as<uint32_t>();
  a.entry_stride = entry_stride;
  a.sums = c->sums.as<double>();
  // A region-sized call (up to 131 072 pairs = 1 MB of sums): the kernels store the sums -- write-only, one store per pair --
  // straight into the pinned host block (posted writes over PCIe) instead of a device array that a copy then fetches: one
  // copy launch (~15 us of a 0.28 ms call) less.
  const bool sums_direct = n <= 131072;
  if ((rc = c->sums_pin.reserve(n * 8 + 64))) return rc;
  if (sums_direct) {
    void* dp = nullptr;
    PD_HIP_TRY(hipHostGetDevicePointer(&dp, c->sums_pin.p, 0));
    a.sums = static_cast<double*>(dp);
  }
  a.status = c->misc.as<int32_t>();
  a.next = c->misc.as<int32_t>() + 1;
  a.carry = c->carry.as<double>();
  a.carry_len = carry_len;
  a.lanes = reinterpret_cast<const LaneSlot*>(dj + o_jl);
  a.job_pair = reinterpret_cast<const int32_t*>(dj + o_jp);
  a.job_steps = reinterpret_cast<const int32_t*>(dj + o_jn);
  a.job_striped = dj + o_js;
  a.n_jobs = n_jobs;
  a.n_cross_jobs = n_cross_jobs; a.n_chunks_cross = std::max(n_chunks_cross, 1);
  a.cross_lanes = reinterpret_cast<const LaneSlot*>(dj + o_cl);
  a.hap_order = reinterpret_cast<const int32_t*>(dj + o_ho);
  a.chunk_steps = reinterpret_cast<const int32_t*>(dj + o_cs);
  a.chunk_rep = reinterpret_cast<const int32_t*>(dj + o_cr);
  a.hap_ncls = n_tab_haps ? dj + o_nc : nullptr;
  a.class_codes = reinterpret_cast<const uint32_t*>(dj + o_cc);
  a.entries_tab = c->entries_tab.as<uint32_t>();
  a.next_special = n_tab_haps ? reinterpret_cast<int32_t*>(c->entries_tab.as<uint32_t>() + nh * (size_t)entry_stride) : nullptr;
  a.job_flags = dj + o_jf;
  a.full_jobs = reinterpret_cast<const int32_t*>(dj + o_fj);
  a.full_count = c->misc.as<int32_t>() + 5;
  a.tab_group_start = reinterpret_cast<const int32_t*>(dj + o_tg);
  unsigned char* dx = c->tabx.as<unsigned char>();
  a.hap_ncls_out = nullptr; a.class_codes_out = nullptr; a.special_bits = nullptr; a.sb_stride = sb_stride;
  a.job_notab = nullptr; a.job_ns = nullptr; a.ns_stride = ns_stride;
  if (tab_paired) {
    a.hap_ncls_out = dx + x_nc;
    a.class_codes_out = reinterpret_cast<uint32_t*>(dx + x_cc);
    a.special_bits = tab_paired_asm ? reinterpret_cast<uint64_t*>(dx + x_sb) : nullptr;
    a.job_notab = dx + x_nt;
    a.job_ns = reinterpret_cast<const int32_t*>(dx + x_ns);
    PD_HIP_TRY(hipMemsetAsync(dx + x_nt, 0, (size_t)n_general, s));
  }
#ifdef GKL_PD_PROF
  a.prof = reinterpret_cast<unsigned long long*>(c->misc.as<char>() + 128);
#endif

  a.item_base = 0; a.job_base = 0;
  const bool paired_packed = !cross && !chunk_used.empty();
  if (!paired_packed && n_slices > 1) {
    // A sliced call without one packed chunk (every read striped): nothing below waits slice by slice, and the helper
    // thread may still be sending -- meet ALL the uploads before the first kernel of the other branch reads the arrays.
    if (up_th.th.joinable()) up_th.th.join();
    PD_HIP_TRY(up_th.err);
    for (int k = 0; k < n_slices; k++) PD_HIP_TRY(hipStreamWaitEvent(s, c->up_ev[k], 0));
  }
  if (paired_packed) {
    // ---- paired layout: slice by slice (one slice unless the call is big, see above) ----
    PdExpandArgs x;
    x.place_chunk = reinterpret_cast<const int32_t*>(dj + o_pc);
    x.place_lane = dj + o_pl;
    x.chunk_used = dj + o_cu;
    x.read_len = a.read_len;
    x.entries = a.entries; x.entry_stride = entry_stride;
    x.lanes = reinterpret_cast<LaneSlot*>(dj + o_jl);
    x.job_flags = dj + o_jf;
    x.hap_ncls = tab_paired ? dx + x_nc : nullptr;
    x.job_notab = tab_paired ? dx + x_nt : nullptr;
    x.n_striped = (int32_t)n_striped; x.rpl = kPdRpl;
    for (int k = 0; k < n_slices; k++) {
      const size_t lo = slice_lo[k], hi = slice_lo[k + 1];
      const int c0 = slice_chunk0[k], c1 = slice_chunk0[k + 1];
      const int j0 = (int)n_striped + c0, j1 = (int)n_striped + c1;
      if (n_slices > 1) {
        while (up_th.recorded.load(std::memory_order_acquire) <= k) std::this_thread::yield();
        PD_HIP_TRY(up_th.err);
        PD_HIP_TRY(hipStreamWaitEvent(s, c->up_ev[k], 0));
      }
      PdArgs ae = a;
      ae.item_base = (int32_t)lo;
      hipLaunchKernelGGL(pdhmm_entries_kernel, dim3((unsigned)(hi - lo)), dim3(kLanes), 0, s, ae);   // one wavefront per haplotype item
      x.pair_base = (int32_t)lo; x.n_pairs = (int32_t)hi; x.chunk_base = c0; x.n_chunks = c1;
      hipLaunchKernelGGL(pdhmm_expand_kernel, dim3((unsigned)((std::max<size_t>(hi - lo, (size_t)(c1 - c0)) + 255) / 256)), dim3(256), 0, s, x);
      if (j1 > j0)
        hipLaunchKernelGGL(pdhmm_collect_kernel, dim3((unsigned)((j1 - j0 + 255) / 256)), dim3(256), 0, s, dj + o_jf, dj + o_js, j1,
                           reinterpret_cast<int32_t*>(dj + o_fj), c->misc.as<int32_t>() + 5, tab_paired ? dx + x_nt : nullptr,
                           reinterpret_cast<int32_t*>(dx + x_hj), c->misc.as<int32_t>() + 6, j0);
      PD_HIP_TRY(hipEventRecord(n_slices > 1 ? c->sl_ev0[k] : c->ev0, s));
      if (j1 <= j0) { if (n_slices > 1) PD_HIP_TRY(hipEventRecord(c->sl_ev1[k], s)); continue; }
      if (tab_paired) {
        // table launch: the jobs' next-special-step tables first (part of the timed region: work only this route does),
        // then every listed job of the slice that is clean and whose haplotypes all have at most kPdTabClasses classes
        if (tab_paired_asm) {
          PdJobNsArgs na;
          na.lanes = a.lanes; na.job_steps = a.job_steps; na.job_striped = a.job_striped; na.job_flags = a.job_flags; na.job_notab = a.job_notab;
          na.read_len = a.read_len; na.hap_len = a.hap_len; na.special_bits = a.special_bits; na.sb_stride = sb_stride;
          na.job_ns = reinterpret_cast<int32_t*>(dx + x_ns); na.ns_stride = ns_stride; na.rpl = kPdRpl; na.job_base = j0; na.job_end = j1;
          hipLaunchKernelGGL(pdhmm_job_special_kernel, dim3((unsigned)((j1 - j0 + kPdNsJobsPerBlock - 1) / kPdNsJobsPerBlock)), dim3(kLanes * kPdNsJobsPerBlock),
                             (size_t)ns_stride * kPdNsJobsPerBlock, s, na);
        }
        PdArgs at = a;
        at.n_cross_jobs = 0; at.job_base = j0; at.n_jobs = j1;
        at.class_codes = a.class_codes_out;
        at.next = c->misc.as<int32_t>() + 8 + k;
        if (c->fma_mode) hipLaunchKernelGGL(pdhmm_fwd_tab_paired_kernel<true>, dim3(std::min(j1 - j0, n_blocks)), dim3(64), 0, s, at, t.initial_condition);
        else             hipLaunchKernelGGL(pdhmm_fwd_tab_paired_kernel<false>, dim3(std::min(j1 - j0, n_blocks)), dim3(64), 0, s, at, t.initial_condition);
      } else {
        // predicate launch: walks the slice's listed jobs and skips the flagged ones
        PdArgs ah = a;
        ah.n_cross_jobs = 0; ah.job_base = j0; ah.n_jobs = j1;
        ah.full_jobs = nullptr;
        ah.next = c->misc.as<int32_t>() + 16 + k;
        if (c->fma_mode) hipLaunchKernelGGL((pdhmm_fwd_kernel<true, false, true>), dim3(std::min(j1 - j0, n_blocks)), dim3(64), 0, s, ah, t.initial_condition);
        else             hipLaunchKernelGGL((pdhmm_fwd_kernel<false, false, true>), dim3(std::min(j1 - j0, n_blocks)), dim3(64), 0, s, ah, t.initial_condition);
      }
      if (n_slices > 1) PD_HIP_TRY(hipEventRecord(c->sl_ev1[k], s));
    }
    if (n_slices > 1) PD_HIP_TRY(hipEventRecord(c->ev0, s));
    if (tab_paired) {
      // predicate launch: the (rare) clean packed jobs with an ineligible haplotype, from the list pdhmm_collect_kernel made
      PdArgs ah = a;
      ah.n_cross_jobs = 0; ah.n_jobs = 0;
      ah.full_jobs = reinterpret_cast<const int32_t*>(dx + x_hj);
      ah.full_count = c->misc.as<int32_t>() + 6;
      if (c->fma_mode) hipLaunchKernelGGL((pdhmm_fwd_kernel<true, false, true>), dim3(std::min((int)n_packed, n_blocks)), dim3(64), 0, s, ah, t.initial_condition);
      else             hipLaunchKernelGGL((pdhmm_fwd_kernel<false, false, true>), dim3(std::min((int)n_packed, n_blocks)), dim3(64), 0, s, ah, t.initial_condition);
    }
    PdArgs af = a;   // full launch: striped reads and haplotypes with odd bases (the list: striped jobs from the host, flagged ones from pdhmm_collect_kernel)
    af.n_cross_jobs = 0; af.n_jobs = n_general;
    af.next = c->misc.as<int32_t>() + 3;
    if (c->fma_mode) hipLaunchKernelGGL(pdhmm_fwd_kernel<true>, dim3(std::min(n_general, n_blocks)), dim3(64), 0, s, af, t.initial_condition);
    else             hipLaunchKernelGGL(pdhmm_fwd_kernel<false>, dim3(std::min(n_general, n_blocks)), dim3(64), 0, s, af, t.initial_condition);
  } else {
    // ---- cross layout (and a paired call that holds striped reads only) ----
    hipLaunchKernelGGL(pdhmm_entries_kernel, dim3((unsigned)nh), dim3(kLanes), 0, s, a);   // one wavefront per haplotype item
    PD_HIP_TRY(hipEventRecord(c->ev0, s));
    // table launch: cross jobs over the haplotypes with few column classes
    if (n_cross_tab > 0) {
      PdArgs at = a;
      at.n_cross_jobs = at.n_jobs = n_tab_units;   // (units: haplotype group x chunk)
      at.next = c->misc.as<int32_t>() + 4;
      if (c->fma_mode) hipLaunchKernelGGL(pdhmm_fwd_tab_kernel<true>, dim3(std::min(n_tab_units, n_blocks)), dim3(64), 0, s, at, t.initial_condition);
      else             hipLaunchKernelGGL(pdhmm_fwd_tab_kernel<false>, dim3(std::min(n_tab_units, n_blocks)), dim3(64), 0, s, at, t.initial_condition);
    }
    // hot launch: cross jobs over the other clean haplotypes + the listed jobs the device routes to it
    PdArgs ah = a;
    ah.hap_order = a.hap_order + n_tab_haps;
    ah.n_cross_jobs = n_cross_hot;
    ah.n_jobs = n_cross_hot + (cross ? 0 : n_general);   // (cross layout: the listed jobs are striped reads, all the full kernel's)
    ah.full_jobs = nullptr;   // walks every listed job and skips the flagged ones
    if (ah.n_jobs > 0) {
      if (c->fma_mode) hipLaunchKernelGGL((pdhmm_fwd_kernel<true, false, true>), dim3(std::min(ah.n_jobs, n_blocks)), dim3(64), 0, s, ah, t.initial_condition);
      else             hipLaunchKernelGGL((pdhmm_fwd_kernel<false, false, true>), dim3(std::min(ah.n_jobs, n_blocks)), dim3(64), 0, s, ah, t.initial_condition);
    }
    // full launch: cross jobs over the haplotypes with odd bases + the listed jobs with a striped read or an odd haplotype
    PdArgs af = a;
    af.hap_order = a.hap_order + n_clean_haps;
    af.n_cross_jobs = n_cross_jobs - n_cross_hot - n_cross_tab;
    af.n_jobs = af.n_cross_jobs + n_general;
    af.next = c->misc.as<int32_t>() + 3;
    if (af.n_jobs > 0) {
      if (c->fma_mode) hipLaunchKernelGGL(pdhmm_fwd_kernel<true>, dim3(std::min(af.n_jobs, n_blocks)), dim3(64), 0, s, af, t.initial_condition);
      else             hipLaunchKernelGGL(pdhmm_fwd_kernel<false>, dim3(std::min(af.n_jobs, n_blocks)), dim3(64), 0, s, af, t.initial_condition);
    }
  }
  if (n_tail > 0) {
    PdArgs at = a;  // the tail pairs: jobs of their own, scalar-engine arithmetic (same stream: the carry rows are free again)
    at.lanes = reinterpret_cast<const LaneSlot*>(dj + o_tl);
    at.job_pair = reinterpret_cast<const int32_t*>(dj + o_tp);
    at.job_steps = reinterpret_cast<const int32_t*>(dj + o_tn);
    at.job_striped = dj + o_ts;
    at.n_jobs = (int32_t)n_tail;
    at.n_cross_jobs = 0;
    at.job_flags = nullptr;   // every tail job is this launch's
    at.full_jobs = nullptr;
    at.next = c->misc.as<int32_t>() + 2;
    hipLaunchKernelGGL((pdhmm_fwd_kernel<false, true>), dim3((unsigned)std::min<size_t>(n_tail, (size_t)n_blocks)), dim3(64), 0, s, at, t.initial_condition);  // persistent: one carry slab per block
  }
  PD_HIP_TRY(hipEventRecord(c->ev1, s));
  PD_HIP_TRY(hipGetLastError());
  double* sums = c->sums_pin.as<double>();
  int32_t* status = reinterpret_cast<int32_t*>(sums + n);
  if (!sums_direct) PD_HIP_TRY(hipMemcpyAsync(sums, c->sums.p, n * 8, hipMemcpyDeviceToHost, s));
  PD_HIP_TRY(hipMemcpyAsync(status, c->misc.p, 32, hipMemcpyDeviceToHost, s));
  const double ms_launched = ms_since(t_begin);
  PD_HIP_TRY(hipStreamSynchronize(s));
  if (up_th.th.joinable()) up_th.th.join();   // (a sliced call: the stream has waited for every upload event by now)
  PD_HIP_TRY(up_th.err);
  PD_HIP_TRY(hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
  if (paired_packed && n_slices > 1)   // kernel time of a sliced call: its slices' launches plus the closing ones (the waits for the bus lie between them)
    for (int k = 0; k < n_slices; k++) {
      float ms = 0.f;
      PD_HIP_TRY(hipEventElapsedTime(&ms, c->sl_ev0[k], c->sl_ev1[k]));
      c->last_ms += ms;
    }
#ifdef GKL_PD_PROF
  {  // development build: where the table kernel's wavefronts spent their cycles (s_memtime)
    unsigned long long pr[16];
    PD_HIP_TRY(hipMemcpy(pr, c->misc.as<char>() + 128, sizeof pr, hipMemcpyDeviceToHost));
    const double tot = (double)(pr[0] + pr[1] + pr[2] + pr[3] + pr[4] + pr[5]);
    fprintf(stderr, "[pd prof] setup+table %.3f  asm runs %.3f (%llu steps)  plain x2 loop %.3f (%llu)  plain x1 loop %.3f (%llu)  general %.3f (%llu)  other %.3f  | jobs %llu, total %.3e ticks; first wavefront done at %.3f of the launch, last at 1\n",
            pr[0] / tot, pr[1] / tot, pr[8], pr[2] / tot, pr[9], pr[3] / tot, pr[10], pr[4] / tot, pr[11], pr[5] / tot, pr[12], tot, (double)(pr[13] - pr[15]) / (double)(pr[14] - pr[15]));
  }
#endif
  if (timing)
    fprintf(stderr, "[gklhip] pdhmm call: uploads enqueued %.2f ms, jobs built %.2f, routed %.2f, launched %.2f, synchronised %.2f (kernels %.2f ms), %zu pairs\n",
            ms_uploads, ms_jobs, ms_routing, ms_launched, ms_since(t_begin), (double)c->last_ms, n);
  if (!cross) {   // paired layout: packed jobs by kernel (status[5] = the full launch's list, the striped jobs included; [6] = the predicate launch's)
    const int32_t n_full_packed = status[5] - (int32_t)n_striped, n_hot = tab_paired ? status[6] : (int32_t)n_packed - n_full_packed;
    c->last_routing[0] = tab_paired ? (int32_t)n_packed - n_hot - n_full_packed : 0;
    c->last_routing[1] = n_hot;
    c->last_routing[2] = n_full_packed;
  }
  if (status[0] != 0)  // PDHMM_INPUT_DATA_ERROR (pdhmm-serial.cc:183-199): negative ins / del / gcp quality
    return pd_fail(GKLHIP_ERR_INVALID_ARG, "Error while calculating pdhmm. Input arrays aren't valid.");
  // log10 of the sums with the HOST libm, like the reference (pdhmm.h:846) -- a region's 13 248 pairs are 0.13 ms of it on
  // one thread, three quarters of what the call costs beyond its kernels: four threads from 4096 pairs on (persistent
  // workers: a thread per call would cost more than it saves)
  const std::function<void(int64_t, int64_t)> finalise = [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; i++) out_host[i] = std::log10(sums[i]) - t.initial_condition_log10;
  };
  static const int fin_threads = std::max(1, std::min(4, (int)std::thread::hardware_concurrency()));
  try {
    c->workers.parallel_for((int64_t)n, fin_threads, finalise, 4096);
  } catch (const std::bad_alloc&) {
    return pd_fail(GKLHIP_ERR_OOM, "out of memory in the host finalisation");
  }
  return GKLHIP_OK;
}
}  // namespace

int gklhip_pdhmm_compute(gklhip_pdhmm_ctx* c, const gklhip_pdhmm_batch* b, double* out_host) {
  if (!c) return pd_fail(GKLHIP_ERR_INVALID_ARG, "context is NULL (initNative not called)");
  if (!b) return pd_fail(GKLHIP_ERR_INVALID_ARG, "batch is NULL");
  // IntelPDHMM.java:163-173
  if (b->batch <= 0) return pd_fail(GKLHIP_ERR_INVALID_ARG, "batchSize must be greater than 0");
  PdProblem q{b->batch, b->batch, b->batch, 0, b->max_hap_len, b->max_read_len, b->hap_bases, b->hap_pdbases,
              b->read_bases, b->read_qual, b->read_ins_qual, b->read_del_qual, b->gcp, b->hap_lengths, b->read_lengths, 0};
  const int rc = pd_validate(q, out_host);
  return rc ? rc : pd_run(c, q, out_host);
}

int gklhip_pdhmm_compute_cross(gklhip_pdhmm_ctx* c, const gklhip_pdhmm_cross* x, double* out_host) {
  return gklhip_pdhmm_compute_cross_batched(c, x, 0, out_host);
}

int32_t gklhip_pdhmm_available_memory_mb(int32_t max_memory_mb) {
  // pdhmm-implementation.h:204-235 (getMaxMemoryAvailable): min(maxMemoryInMB, free RAM of the host) -- taken ONCE, by
  // initNative, like the reference does; the batch cut of every later call then depends on its arguments only
  if (max_memory_mb <= 0) return 0;
  int64_t mb = max_memory_mb;
  struct sysinfo info;
  if (sysinfo(&info) == 0) mb = std::min<int64_t>(mb, (int64_t)info.freeram * (int64_t)info.mem_unit / (1024 * 1024));
  return (int32_t)std::max<int64_t>(mb, 0);
}

int64_t gklhip_pdhmm_reference_batch_pairs(int32_t max_memory_mb, int32_t max_read_len, int32_t max_hap_len, int64_t total_pairs) {
  // JavaData.h:86-101: min(totalPairs, maxMemory / memoryPerPair), maxMemory = what initNative kept (above)
  if (max_memory_mb <= 0 || max_read_len <= 0 || max_hap_len <= 0 || total_pairs <= 0) return 0;
  const int64_t per_pair = ((int64_t)max_read_len * 5 + (int64_t)max_hap_len * 2) + 8 + 16;
  return std::min(total_pairs, (int64_t)max_memory_mb * 1024 * 1024 / per_pair);
}

int gklhip_pdhmm_compute_cross_batched(gklhip_pdhmm_ctx* c, const gklhip_pdhmm_cross* x, int64_t ref_batch_pairs, double* out_host) {
  if (!c) return pd_fail(GKLHIP_ERR_INVALID_ARG, "context is NULL (initNative not called)");
  if (ref_batch_pairs < 0) return pd_fail(GKLHIP_ERR_INVALID_ARG, "ref_batch_pairs must not be negative");
  if (!x) return pd_fail(GKLHIP_ERR_INVALID_ARG, "batch is NULL");
  if (x->n_reads <= 0 || x->n_haps <= 0) return pd_fail(GKLHIP_ERR_INVALID_ARG, "no pairs to process");
  PdProblem q{(int64_t)x->n_reads * x->n_haps, x->n_reads, x->n_haps, x->n_haps, x->max_hap_len, x->max_read_len,
              x->hap_bases, x->hap_pdbases, x->read_bases, x->read_qual, x->read_ins_qual, x->read_del_qual, x->gcp,
              x->hap_lengths, x->read_lengths, ref_batch_pairs};
  const int rc = pd_validate(q, out_host);
  return rc ? rc : pd_run(c, q, out_host);
}

}  // extern "C"
