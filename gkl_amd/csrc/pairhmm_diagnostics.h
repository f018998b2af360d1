// Diagnostics and self-tests behind the C ABI (inside its extern "C" block): small-call counts, the issue-ceiling measurement, the RCCL
// self-test.
// Part of the ONE translation unit gkl_amd/csrc/pairhmm_api.hip (included there, in this order: pairhmm_ctx.h, pairhmm_device_pass.h,
// pairhmm_ctx_lifecycle.h, pairhmm_host_call.h, pairhmm_multi_device.h, pairhmm_diagnostics.h); not a stand-alone header.
#pragma once

int gklhip_small_call_counts(int device, int64_t out[3], int reset) {
  if (!out || device < 0) return fail(GKLHIP_ERR_INVALID_ARG, "NULL argument or negative device");
  SmallCombiner* k = small_combiner(device);
  std::lock_guard<std::mutex> l(k->mu);
  out[0] = k->n_calls; out[1] = k->n_combined; out[2] = k->n_launch_sets;
  if (g_env.timing)
    fprintf(stderr, "[gklhip] small calls: %lld calls, %lld combined, %lld launch sets; per set: queued %.1f us (sum over its calls), launch %.1f us, sync %.1f us\n",
            (long long)k->n_calls, (long long)k->n_combined, (long long)k->n_launch_sets, k->ns_queued * 1e-3 / std::max<int64_t>(1, k->n_launch_sets),
            k->ns_launch * 1e-3 / std::max<int64_t>(1, k->n_launch_sets), k->ns_sync * 1e-3 / std::max<int64_t>(1, k->n_launch_sets));
  if (g_env.timing)
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
