// A device engine's lifecycle: dev_init (streams, self-tests, tables), dev_done, trim_streams.
// Part of the ONE translation unit gkl_amd/csrc/pairhmm_api.hip (included there, in this order: pairhmm_ctx.h, pairhmm_device_pass.h,
// pairhmm_ctx_lifecycle.h, pairhmm_host_call.h, pairhmm_multi_device.h, pairhmm_diagnostics.h); not a stand-alone header.
#pragma once

namespace {

// ------------------------------------------------------------------ one device: lifecycle + host-buffer call
std::atomic<int> g_eager_upload_holders{0};   // contexts of the process that opened upload_stream with themselves (0 or 1)

// The streams a context made for a big call -- copy_stream, the padding stream, upload_stream unless it came with the
// context -- given back when nothing is queued on any of the context's streams; the next call that needs them makes
// them again (aux_streams).  Every stream is a hardware queue the device's scheduler rotates among ALL processes'
// (docs/NOTES.md 49: an idle process that holds three or more queues next to eight busy ones starves one of them for
// seconds): an idle context should hold one -- or two, for the process's first.  Returns how many streams went.
int trim_streams(DevCtx* c) {
  if (!c->copy_stream && !c->pad_stream && (!c->upload_stream || c->upload_eager)) return 0;
  (void)hipSetDevice(c->device);
  for (hipStream_t s : {c->stream, c->copy_stream, c->upload_stream, c->have_last ? c->last_stream : nullptr})
    if (s && hipStreamQuery(s) != hipSuccess) { (void)hipGetLastError(); return 0; }
  int n = 0;
  if (c->copy_stream) { (void)hipStreamDestroy(c->copy_stream); c->copy_stream = nullptr; n++; }
  if (c->pad_stream) { (void)hipStreamDestroy(c->pad_stream); c->pad_stream = nullptr; n++; }
  if (c->upload_stream && !c->upload_eager) { (void)hipStreamDestroy(c->upload_stream); c->upload_stream = nullptr; n++; }
  return n;
}

void dev_done(DevCtx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->upload_eager) g_eager_upload_holders--;
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
      // ... and in the shapes of the biggest allocations: the super-stripe kernel's (512 threads, 78 KB: two per CU, a neighbour's
      // LDS right behind) and the wide kernels' (256 threads, 128 KB: one per CU), every CU loaded four times over
      hipLaunchKernelGGL((lds_oob_selftest_big_kernel<78 * 256, 512>), dim3(4 * 2 * (unsigned)c->n_cus), dim3(512), 0, c->stream, d_out + 2);
      hipLaunchKernelGGL((lds_oob_selftest_big_kernel<128 * 256, 256>), dim3(4 * (unsigned)c->n_cus), dim3(256), 0, c->stream, d_out + 2);
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
  // (counted over the contexts that HOLD such a stream: when that context goes, the next one made takes the role)
  if (g_eager_upload_holders.fetch_add(1) == 0) {
    if (hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking) != hipSuccess) { g_eager_upload_holders--; return bail(fail(GKLHIP_ERR_HIP, "hipStreamCreate failed")); }
    c->upload_eager = true;
  } else {
    g_eager_upload_holders--;
  }
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

}  // namespace
