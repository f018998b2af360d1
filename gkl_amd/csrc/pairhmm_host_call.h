// The host-buffer call of one device (dev_compute_host: H2D, device pass, reference-exact host log10) and the small-call combiner that launches
// the GATK-sized calls of several threads together.
// Part of the ONE translation unit gkl_amd/csrc/pairhmm_api.hip (included there, in this order: pairhmm_ctx.h, pairhmm_device_pass.h,
// pairhmm_ctx_lifecycle.h, pairhmm_host_call.h, pairhmm_multi_device.h, pairhmm_diagnostics.h); not a stand-alone header.
#pragma once

namespace {

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
  const int env = g_env.finalize_threads;
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
  int64_t last_made_or_used_ns = 0;   // when the flight streams were made / last carried a set
  // The flight streams go again when nothing has been combined for `idle_ns` and nothing is in the air (an idle process
  // should not hold their hardware queues: gklhip_release_idle); the next two calls that meet make them again.
  int release_streams(int64_t idle_ns) {
    std::lock_guard<std::mutex> l(mu);
    if (!streams_made || flights > 0 || !queue.empty() || now_ns() - last_made_or_used_ns < idle_ns) return 0;
    for (auto& sl : slot) if (sl.busy) return 0;
    int prev = 0, n = 0;
    const bool have_dev = hipGetDevice(&prev) == hipSuccess;
    if (hipSetDevice(device) == hipSuccess) {
      for (auto& sl : slot) {
        if (sl.stream) { (void)hipStreamSynchronize(sl.stream); (void)hipStreamDestroy(sl.stream); sl.stream = nullptr; n++; }
        if (sl.ev) { (void)hipEventDestroy(sl.ev); sl.ev = nullptr; }
      }
      streams_made = false;
    }
    if (have_dev) (void)hipSetDevice(prev);
    return n;
  }
  void make_streams() {
    last_made_or_used_ns = now_ns();
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
  return g_env.combine;
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
  // one-pass finalisation (calls up to kOnePassPairs): a region of 400 reads x 40 haplotypes is 16 000 log10's = 0.08 ms
  // on one thread, a fifth of the call -- spread over the workers from 8192 pairs on (400 x 40: 0.416 -> 0.378 ms; at 4096
  // the hand-off costs a 150 x 30 call more than it saves: 0.236 -> 0.245; tools/mid_finalize_ab.py)
  const int64_t one_pass_min = g_env.finalize_min > 0 ? g_env.finalize_min : 8192;
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
    c->stats.n_fallback = fin.all(&c->workers, c->res_pin.as<uint64_t>(), out_host, n_pairs, threads, one_pass_min);
    k->ns_stage.fetch_add(t_staged - t_call, std::memory_order_relaxed);
    k->ns_run.fetch_add(t_done - t_staged, std::memory_order_relaxed);
    k->ns_finalize.fetch_add(SmallCombiner::now_ns() - t_done, std::memory_order_relaxed);
    return GKLHIP_OK;
  }
  if (c->cfg.use_double || n_pairs <= kOnePassPairs) {
    // all-fp64 mode, or a GATK-sized call (the fp64 stage of a region without underflowed pairs -- the usual case --
    // is two launches that find nothing to do): one pass over the words once the last kernel is done
    HIP_TRY(hipStreamSynchronize(s));
    c->stats.n_fallback = fin.all(&c->workers, c->res_pin.as<uint64_t>(), out_host, n_pairs, threads, one_pass_min);
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

}  // namespace
