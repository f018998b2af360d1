// Several devices behind one context: per-device worker threads, the lazily loaded RCCL gather with its peer-copy fall-back, gklhip_ctx,
// read partitioning, multi_compute_host / multi_compute_device.
// Part of the ONE translation unit gkl_amd/csrc/pairhmm_api.hip (included there, in this order: pairhmm_ctx.h, pairhmm_device_pass.h,
// pairhmm_ctx_lifecycle.h, pairhmm_host_call.h, pairhmm_multi_device.h, pairhmm_diagnostics.h); not a stand-alone header.
#pragma once

namespace {

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
  const bool quiet = g_env.quiet;
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
