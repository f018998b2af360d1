// Test infrastructure (CPU): the seven C-ABI entry points jni_shim.cpp calls (include/gkl_hip_pairhmm.h), WITHOUT a
// device and without any PairHMM arithmetic, so that the JNI layer -- block-wise marshalling, helper threads attached
// through the JavaVM, the pipelined ranges, the retry after a HIP failure, every exception path -- can be driven by the
// mock JVM (mock_jni.cpp) in the `-m "not gpu"` suite.  Linked with the product's jni_shim.o into
// tests/native/libgkl_pairhmm_stub.so by tests/mockjni.py; nothing in gkl_amd/ or bench.py's timed paths loads it.
//
// "Likelihood" of pair (r, h) = position-weighted checksum of read r's five arrays * 2^-20 + checksum of haplotype h:
// every marshalled byte and its position inside its read shows in the result (tests/mockjni.py: stub_expected).
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "../../include/gkl_hip_pairhmm.h"

namespace {
std::atomic<long> g_inits{0}, g_dones{0}, g_computes{0}, g_live{0}, g_releases{0};
std::atomic<long> g_fail_from{0}, g_fail_count{0};   // computes [from, from + count) (1-based, counted from stub_reset) fail with GKLHIP_ERR_HIP
std::atomic<long> g_delay_us{0};
std::atomic<int> g_skip{0};   // stub_skip_arithmetic: gklhip_compute only counts (timing of the JNI layer alone)
thread_local const char* t_err = "";
}  // namespace

struct gklhip_ctx { gklhip_config cfg; };

extern "C" {

int gklhip_init(const gklhip_config* cfg, gklhip_ctx** out) {
  if (!cfg || !out || cfg->abi_version != GKLHIP_ABI_VERSION) { t_err = "bad config"; return GKLHIP_ERR_INVALID_ARG; }
  *out = new gklhip_ctx{*cfg};
  g_inits++; g_live++;
  return GKLHIP_OK;
}
int gklhip_done(gklhip_ctx* c) { if (c) { delete c; g_dones++; g_live--; } return GKLHIP_OK; }
const char* gklhip_last_error(void) { return t_err; }
const char* gklhip_strerror(int st) {
  switch (st) {
    case GKLHIP_OK: return "ok";
    case GKLHIP_ERR_INVALID_ARG: return "invalid argument";
    case GKLHIP_ERR_NO_DEVICE: return "no usable gfx950 device";
    case GKLHIP_ERR_OOM: return "out of memory";
    case GKLHIP_ERR_HIP: return "HIP runtime failure";
    default: return "unsupported";
  }
}
void* gklhip_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
void gklhip_host_free(void* p) { free(p); }

int gklhip_compute(gklhip_ctx* c, const gklhip_batch* b, double* out) {
  const long nth = ++g_computes;
  if (const long us = g_delay_us.load()) std::this_thread::sleep_for(std::chrono::microseconds(us));
  const long from = g_fail_from.load();
  if (from > 0 && nth >= from && nth < from + g_fail_count.load()) { t_err = "injected fault (stub)"; return GKLHIP_ERR_HIP; }
  if (!c || !b || !out) { t_err = "null argument"; return GKLHIP_ERR_INVALID_ARG; }
  if (g_skip.load()) return GKLHIP_OK;
  for (int32_t r = 0; r < b->n_reads; r++) {
    const int64_t a = b->read_off[r], n = b->read_off[r + 1] - a;
    if (n <= 0) { t_err = "empty read"; return GKLHIP_ERR_INVALID_ARG; }
    uint64_t hr = 0;
    for (int64_t i = 0; i < n; i++)
      hr += (uint64_t)(i + 1) * (b->read_bases[a + i] + 3u * b->read_quals[a + i] + 5u * b->ins_gop[a + i] + 7u * b->del_gop[a + i] + 11u * b->gcp[a + i]);
    for (int32_t h = 0; h < b->n_haps; h++) {
      const int64_t ha = b->hap_off[h], hn = b->hap_off[h + 1] - ha;
      uint64_t hh = 0;
      for (int64_t i = 0; i < hn; i++) hh += (uint64_t)(i + 1) * b->hap_bases[ha + i];
      out[(int64_t)r * b->n_haps + h] = (double)hr * (1.0 / 1048576.0) + (double)hh;
    }
  }
  return GKLHIP_OK;
}

int gklhip_release_idle(gklhip_ctx* c, int32_t* n) { if (n) *n = c ? 1 : 0; g_releases++; return GKLHIP_OK; }

// test controls
void stub_reset(void) { g_inits = g_dones = g_computes = 0; g_releases = 0; g_fail_from = g_fail_count = 0; g_delay_us = 0; g_skip = 0; }
void stub_fail(long from, long count) { g_fail_from = from; g_fail_count = count; }
void stub_delay_us(long us) { g_delay_us = us; }
void stub_skip_arithmetic(int on) { g_skip = on; }
void stub_counts(long out[4]) { out[0] = g_inits; out[1] = g_dones; out[2] = g_computes; out[3] = g_live; }
long stub_releases(void) { return g_releases; }

}  // extern "C"
