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

#include "pairhmm_ctx.h"
#include "pairhmm_device_pass.h"
#include "pairhmm_ctx_lifecycle.h"
#include "pairhmm_host_call.h"
#include "pairhmm_multi_device.h"

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

int gklhip_release_idle(gklhip_ctx* c, int32_t* streams_released) {
  if (streams_released) *streams_released = 0;
  if (!c) return fail(GKLHIP_ERR_INVALID_ARG, "context is NULL");
  std::unique_lock<std::mutex> lock(c->mu, std::try_to_lock);
  if (!lock.owns_lock()) return GKLHIP_OK;   // a call is running: nothing is idle
  int n = 0;
  // the twin engines of big host-buffer calls and the second engines of two-stream device-resident callers go whole
  // (their own stream, scratch and tables with them); the first engine of every device keeps its stream(s)
  bool busy = false;
  for (const std::vector<DevCtx*>* list : {&c->twins, &c->dev_alt})
    for (DevCtx* d : *list) {
      (void)hipSetDevice(d->device);
      for (hipStream_t s : {d->stream, d->copy_stream, d->upload_stream, d->have_last ? d->last_stream : nullptr})
        if (s && hipStreamQuery(s) != hipSuccess) { (void)hipGetLastError(); busy = true; }
    }
  if (!busy) {
    for (DevCtx* d : c->twins) { n += 1 + (d->copy_stream != nullptr) + (d->pad_stream != nullptr) + (d->upload_stream != nullptr); dev_done(d); }
    c->twins.clear();
    c->host_dev = c->dev;
    if (!c->dev_alt.empty()) {
      for (DevCtx* d : c->dev_alt) { n += 1 + (d->copy_stream != nullptr) + (d->pad_stream != nullptr) + (d->upload_stream != nullptr); dev_done(d); }
      if (c->last == &c->dev_alt) c->last = &c->dev;
      c->dev_alt.clear();
      for (size_t d = 0; d < c->shard_done_alt.size(); d++)
        if (c->shard_done_alt[d]) { (void)hipSetDevice(c->dev[d]->device); (void)hipEventDestroy(c->shard_done_alt[d]); }
      c->shard_done_alt.clear();
      if (c->inputs_ready_alt) { (void)hipSetDevice(c->dev[0]->device); (void)hipEventDestroy(c->inputs_ready_alt); c->inputs_ready_alt = nullptr; }
      c->used_set[1] = false; c->stream_of[1] = nullptr; c->last_set = 0;
    }
  }
  for (DevCtx* d : c->dev) n += trim_streams(d);
  // the process's small-call combiner on these devices: its flight streams, when no host call is inside the library and
  // no set has been launched for a second
  if (g_host_calls_in_flight.load() == 0)
    for (DevCtx* d : c->dev) n += small_combiner(d->device)->release_streams(1000000000LL);
  if (streams_released) *streams_released = n;
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
  const bool one_engine = g_env.one_device_engine;
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

#include "pairhmm_diagnostics.h"

// ---- the guarded entry points (see guarded()) ----
int gklhip_init_devices(const gklhip_config* cfg, const int32_t* devices, int32_t n_devices, gklhip_ctx** out_ctx) { return guarded([&] { return init_devices_impl(cfg, devices, n_devices, out_ctx); }); }
int gklhip_compute_device(gklhip_ctx* c, const gklhip_batch* dev_batch, double* out_dev, void* hip_stream) { return guarded([&] { return compute_device_impl(c, dev_batch, out_dev, hip_stream); }); }
int gklhip_compute(gklhip_ctx* c, const gklhip_batch* hb, double* out_host) { return guarded([&] { return compute_impl(c, hb, out_host); }); }
int gklhip_get_raw(gklhip_ctx* ctx, float* raw32, double* raw64, uint8_t* used64) { return guarded([&] { return get_raw_impl(ctx, raw32, raw64, used64); }); }
int gklhip_get_step_times(gklhip_ctx* ctx, int32_t steps_back, float* ms_main, float* ms_fallback, float* ms_total) { return guarded([&] { return get_step_times_impl(ctx, steps_back, ms_main, ms_fallback, ms_total); }); }
int gklhip_rccl_selftest(int32_t device) { return guarded([&] { return rccl_selftest_impl(device); }); }
int gklhip_init(const gklhip_config* cfg, gklhip_ctx** out_ctx) { return guarded([&] { return init_impl(cfg, out_ctx); }); }

}  // extern "C"
