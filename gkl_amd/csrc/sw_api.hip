// C-ABI implementation (include/gkl_hip_sw.h) of the MI355X Smith-Waterman path: validation, device
// layout of a batch of pairs, one persistent-wavefront launch, read-back of the CIGAR text.
#include <hip/hip_runtime.h>
#include <atomic>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/gkl_hip_pairhmm.h"  // status codes
#include "../../include/gkl_hip_sw.h"
#include "sw_kernel.h"

using namespace gklhip;

namespace {
thread_local std::string g_sw_err;

int sw_fail(int status, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_sw_err = buf;
  return status;
}

#define SW_HIP_TRY(expr)                                                                          \
  do {                                                                                            \
    hipError_t e__ = (expr);                                                                      \
    if (e__ != hipSuccess) {                                                                      \
      (void)hipGetLastError();                                                                    \
      return sw_fail(e__ == hipErrorOutOfMemory ? GKLHIP_ERR_OOM : GKLHIP_ERR_HIP, "%s: %s", #expr, \
                     hipGetErrorString(e__));                                                     \
    }                                                                                             \
  } while (0)

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
  bool pinned = false;
  int reserve(size_t n) {
    if (n <= cap) return GKLHIP_OK;
    release();
    const size_t want = n + n / 4 + 256;
    if (pinned) SW_HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
    else SW_HIP_TRY(hipMalloc(&p, want));
    cap = want;
    return GKLHIP_OK;
  }
  void release() {
    if (p) { if (pinned) (void)hipHostFree(p); else (void)hipFree(p); }
    p = nullptr; cap = 0;
  }
  template <typename T> T* as() const { return static_cast<T*>(p); }
};

size_t up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
}  // namespace

struct gklhip_sw_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t pad_stream = nullptr;   // never used: see gklhip_sw_init
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::mutex mu;
  Buf stage_in, stage_out;   // pinned: descriptors + sequences up, text + results down
  Buf dev_in, dev_out, bt, aux, ops, misc;
  float last_ms = 0.f;
  gklhip_sw_ctx() { stage_in.pinned = stage_out.pinned = true; }
};

extern "C" {

const char* gklhip_sw_last_error(void) { return g_sw_err.c_str(); }

int gklhip_sw_init(int device, gklhip_sw_ctx** out_ctx) {
  if (!out_ctx) return sw_fail(GKLHIP_ERR_INVALID_ARG, "out_ctx is NULL");
  *out_ctx = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return sw_fail(GKLHIP_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU compute path)");
  }
  if (device < 0) SW_HIP_TRY(hipGetDevice(&device));
  if (device >= ndev) return sw_fail(GKLHIP_ERR_INVALID_ARG, "device %d of %d", device, ndev);
  SW_HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  SW_HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return sw_fail(GKLHIP_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
  gklhip_sw_ctx* c = new (std::nothrow) gklhip_sw_ctx();
  if (!c) return sw_fail(GKLHIP_ERR_OOM, "context allocation failed");
  c->device = device;
  auto bail = [&](int st) { gklhip_sw_done(c); return st; };
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(sw_fail(GKLHIP_ERR_HIP, "hipStreamCreate failed"));
  {
    // Every stream is a hardware queue of the process, and how many each process holds decides how the device's scheduler
    // shares the chip among processes: two per process measured best, THREE worst by a factor of two (docs/NOTES.md 48).
    // libgklhip_pairhmm brings two; so that a JVM which loads this library beside it holds four and not three, the first
    // context of a process here opens a second, unused stream as well.
    static std::atomic<int> contexts_made{0};
    if (contexts_made.fetch_add(1) == 0 && hipStreamCreateWithFlags(&c->pad_stream, hipStreamNonBlocking) != hipSuccess) { c->pad_stream = nullptr; (void)hipGetLastError(); }
  }
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) return bail(sw_fail(GKLHIP_ERR_HIP, "hipEventCreate failed"));
  int rc = c->misc.reserve(256);
  if (rc) return bail(rc);
  *out_ctx = c;
  return GKLHIP_OK;
}

int gklhip_sw_done(gklhip_sw_ctx* c) {
  if (!c) return GKLHIP_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (Buf* b : {&c->stage_in, &c->stage_out, &c->dev_in, &c->dev_out, &c->bt, &c->aux, &c->ops, &c->misc}) b->release();
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->pad_stream) (void)hipStreamDestroy(c->pad_stream);
  delete c;
  return GKLHIP_OK;
}

float gklhip_sw_last_kernel_ms(gklhip_sw_ctx* c) { return c ? c->last_ms : 0.f; }

int gklhip_sw_align_batch(gklhip_sw_ctx* c, const gklhip_sw_params* prm, int32_t strategy, int32_t n,
                          const uint8_t* refs, const int64_t* ref_off, const uint8_t* alts, const int64_t* alt_off,
                          char* cigars, int32_t cigar_stride, uint32_t* counts, int32_t* offsets) {
  if (!c) return sw_fail(GKLHIP_ERR_INVALID_ARG, "context is NULL (initNative not called)");
  if (!prm) return sw_fail(GKLHIP_ERR_INVALID_ARG, "parameters are NULL");
  if (n < 0) return sw_fail(GKLHIP_ERR_INVALID_ARG, "negative pair count");
  if (n == 0) return GKLHIP_OK;
  if (!refs || !alts || !ref_off || !alt_off || !cigars || !counts || !offsets)
    return sw_fail(GKLHIP_ERR_INVALID_ARG, "NULL array");
  // the checks of IntelSmithWaterman.align (IntelSmithWaterman.java:126-141)
  if (strategy < GKLHIP_SW_SOFTCLIP || strategy > GKLHIP_SW_IGNORE) return sw_fail(GKLHIP_ERR_INVALID_ARG, "Strategy is invalid.");
  if (cigar_stride <= 0) return sw_fail(GKLHIP_ERR_INVALID_ARG, "Strategy is invalid.");  // same message there
  if (prm->match > GKLHIP_SW_MAX_MATCH_VALUE)
    return sw_fail(GKLHIP_ERR_INVALID_ARG, "Match value parameter exceed maximum value of %d", GKLHIP_SW_MAX_MATCH_VALUE);
  if (ref_off[0] != 0 || alt_off[0] != 0) return sw_fail(GKLHIP_ERR_INVALID_ARG, "offsets must start at 0");
  std::vector<SwPair> pairs((size_t)n);
  const size_t ref_bytes = (size_t)ref_off[n], alt_bytes = (size_t)alt_off[n];
  size_t bt_units = 0, aux_units = 0, ops_units = 0;
  for (int32_t k = 0; k < n; k++) {
    const int64_t rl = ref_off[k + 1] - ref_off[k], al = alt_off[k + 1] - alt_off[k];
    if (rl <= 0 || al <= 0) return sw_fail(GKLHIP_ERR_INVALID_ARG, "Cannot align empty sequences");
    if (rl > GKLHIP_SW_MAX_SEQUENCE_LENGTH || al > GKLHIP_SW_MAX_SEQUENCE_LENGTH)
      return sw_fail(GKLHIP_ERR_INVALID_ARG, "Sequences exceed maximum length of %d bytes", GKLHIP_SW_MAX_SEQUENCE_LENGTH);
    SwPair& p = pairs[(size_t)k];
    p.ref_off = ref_off[k];
    p.alt_off = (int64_t)up(ref_bytes) + alt_off[k];
    p.nrow = (int32_t)rl;
    p.ncol = (int32_t)al;
    {
      // Rows per lane.  A step costs the wavefront ~30 + 20 * rpl instructions whatever the number of lanes in use and a
      // stripe takes ncol + lanes - 1 steps, so the cheapest fill uses ALL 64 lanes with as few rows each as possible:
      // as many stripes as 8 rows per lane would need, the rows spread evenly over them.
      const int64_t n_stripes = (rl + 8 * kLanes - 1) / (8 * kLanes);
      const int64_t per_stripe = (rl + n_stripes - 1) / n_stripes;
      p.rpl = (int32_t)std::max<int64_t>(1, (per_stripe + kLanes - 1) / kLanes);
    }
    {
      const size_t stripe_rows = (size_t)kLanes * (size_t)p.rpl;
      bt_units = std::max(bt_units, (((size_t)rl + stripe_rows - 1) / stripe_rows) * ((size_t)al + kLanes) * kLanes);
    }
    aux_units = std::max(aux_units, up((size_t)al + 1 + (size_t)rl + 1 + 4 * ((size_t)al + 65), 16));
    ops_units = std::max(ops_units, up((size_t)rl + (size_t)al + 4, 16));
    p.text_off = (int64_t)k * cigar_stride;
    p.cigar_len = cigar_stride;
  }
  // longest pairs first: the persistent wavefronts finish together.  A counting sort on the cell count (4096 classes
  // up to the largest pair, stable within a class) -- the comparison sort it replaces took 2-3 ms for 32 k pairs.
  std::vector<int32_t> order((size_t)n);
  {
    int64_t max_cells = 1;
    for (int32_t k = 0; k < n; k++) max_cells = std::max(max_cells, (int64_t)pairs[(size_t)k].nrow * pairs[(size_t)k].ncol);
    constexpr int kClasses = 4096;
    const auto cls = [&](int32_t k) {
      const int64_t cells = (int64_t)pairs[(size_t)k].nrow * pairs[(size_t)k].ncol;
      return (int)(kClasses - 1 - (int64_t)((__int128)cells * (kClasses - 1) / max_cells));   // 0 = largest
    };
    std::vector<int32_t> start(kClasses + 1, 0);
    for (int32_t k = 0; k < n; k++) start[(size_t)cls(k) + 1]++;
    for (int q = 0; q < kClasses; q++) start[(size_t)q + 1] += start[(size_t)q];
    for (int32_t k = 0; k < n; k++) order[(size_t)start[(size_t)cls(k)]++] = k;
  }

  std::lock_guard<std::mutex> lock(c->mu);
  SW_HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  // ---- one staged upload: sequences, descriptors, order ----
  const size_t o_ref = 0, o_alt = up(ref_bytes), o_pairs = o_alt + up(alt_bytes),
               o_order = o_pairs + up((size_t)n * sizeof(SwPair)), in_total = o_order + up((size_t)n * 4);
  int rc;
  if ((rc = c->stage_in.reserve(in_total))) return rc;
  if ((rc = c->dev_in.reserve(in_total))) return rc;
  unsigned char* hs = c->stage_in.as<unsigned char>();
  // staged in two parts so the DMA of the references runs while the host still copies the rest
  memcpy(hs + o_ref, refs, ref_bytes);
  SW_HIP_TRY(hipMemcpyAsync(c->dev_in.p, hs, o_alt, hipMemcpyHostToDevice, s));
  memcpy(hs + o_alt, alts, alt_bytes);
  memcpy(hs + o_pairs, pairs.data(), (size_t)n * sizeof(SwPair));
  memcpy(hs + o_order, order.data(), (size_t)n * 4);
  SW_HIP_TRY(hipMemcpyAsync(c->dev_in.as<unsigned char>() + o_alt, hs + o_alt, in_total - o_alt, hipMemcpyHostToDevice, s));
  // ---- device scratch and outputs ----
  const size_t text_bytes = (size_t)n * (size_t)cigar_stride;
  const size_t o_text = 0, o_res = up(text_bytes), out_total = o_res + up((size_t)n * 16);
  if ((rc = c->dev_out.reserve(out_total))) return rc;
  if ((rc = c->stage_out.reserve(out_total))) return rc;
  // scratch slabs: one per persistent wavefront, each big enough for the largest pair of the batch; the slab
  // budget (default 8 GiB of the 288) caps the wavefront count when a batch holds very long sequences
  const size_t slab_bytes = (bt_units + aux_units + ops_units) * 4;
  const size_t budget = (size_t)8 << 30;
  static const size_t waves_per_cu = [] { const char* v = getenv("GKLHIP_SW_WAVES_PER_CU"); return v ? (size_t)atoi(v) : (size_t)kSwWavesPerSimd * 4; }();
  const int n_waves = (int)std::max<size_t>(1, std::min<size_t>({(size_t)n, (size_t)256 * waves_per_cu, budget / std::max<size_t>(slab_bytes, 1)}));
  if ((rc = c->bt.reserve(bt_units * 4 * (size_t)n_waves))) return rc;
  if ((rc = c->aux.reserve(aux_units * 4 * (size_t)n_waves))) return rc;
  if ((rc = c->ops.reserve(ops_units * 4 * (size_t)n_waves))) return rc;
  SW_HIP_TRY(hipMemsetAsync(c->dev_out.p, 0, out_total, s));   // a fresh Java byte[] is zero
  SW_HIP_TRY(hipMemsetAsync(c->misc.p, 0, 64, s));
  unsigned char* din = c->dev_in.as<unsigned char>();
  unsigned char* dout = c->dev_out.as<unsigned char>();
  SwArgs a;
  a.seq = din + o_ref;
  a.pairs = reinterpret_cast<const SwPair*>(din + o_pairs);
  a.order = reinterpret_cast<const int32_t*>(din + o_order);
  a.n_pairs = n;
  a.match = prm->match; a.mismatch = prm->mismatch; a.open = prm->open; a.extend = prm->extend;
  a.strategy = strategy;
  a.bt = c->bt.as<uint32_t>();
  a.aux = c->aux.as<int32_t>();
  a.ops = c->ops.as<int32_t>();
  a.bt_stride = (int64_t)bt_units; a.aux_stride = (int64_t)aux_units; a.ops_stride = (int64_t)ops_units;
  a.text = reinterpret_cast<char*>(dout + o_text);
  a.result = reinterpret_cast<int32_t*>(dout + o_res);
  a.next = c->misc.as<int32_t>();
  SW_HIP_TRY(hipEventRecord(c->ev0, s));
  hipLaunchKernelGGL(sw_align_kernel, dim3(n_waves), dim3(64), 0, s, a);
  SW_HIP_TRY(hipEventRecord(c->ev1, s));
  SW_HIP_TRY(hipGetLastError());
  SW_HIP_TRY(hipMemcpyAsync(c->stage_out.p, c->dev_out.p, out_total, hipMemcpyDeviceToHost, s));
  SW_HIP_TRY(hipStreamSynchronize(s));
  SW_HIP_TRY(hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
  const unsigned char* ho = c->stage_out.as<unsigned char>();
  memcpy(cigars, ho + o_text, text_bytes);
  const int32_t* res = reinterpret_cast<const int32_t*>(ho + o_res);
  for (int32_t k = 0; k < n; k++) {
    offsets[k] = res[4 * (size_t)k + 0];
    counts[k] = (uint32_t)strnlen(cigars + (size_t)k * cigar_stride, (size_t)res[4 * (size_t)k + 1]);  // PairWiseSW.h:451
  }
  return GKLHIP_OK;
}

int gklhip_sw_align(gklhip_sw_ctx* c, const gklhip_sw_params* prm, int32_t strategy, const uint8_t* ref,
                    int32_t ref_len, const uint8_t* alt, int32_t alt_len, char* cigar, int32_t cigar_len,
                    uint32_t* cigar_count, int32_t* offset) {
  if (!cigar_count || !offset) return sw_fail(GKLHIP_ERR_INVALID_ARG, "NULL output");
  const int64_t ro[2] = {0, ref_len}, ao[2] = {0, alt_len};
  return gklhip_sw_align_batch(c, prm, strategy, 1, ref, ro, alt, ao, cigar, cigar_len, cigar_count, offset);
}

}  // extern "C"
