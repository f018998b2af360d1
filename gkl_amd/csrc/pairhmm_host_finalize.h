// Host side of the reference-exact finalisation (IntelPairHmm.cc:159-165 with the host libm): persistent worker
// threads and the two-phase pass over the packed raw sums.  Included by pairhmm_api.hip only.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <exception>
#include <functional>
#include <new>
#include <mutex>
#include <thread>
#include <vector>

#include "pairhmm_tables.h"

namespace gklhip {
// one packed 8-byte word per pair: the double's bits, or 0xFFFFFFFF:float bits (same constant as
// kPackedF32Tag of pairhmm_aux_kernels.h; this header stays free of HIP so it can be read on its own)
constexpr uint64_t kPackedF32TagHost = 0xFFFFFFFF00000000ull;
}

// Persistent helper threads for the host-side log10 finalisation (spawning threads per call costs more
// than the work on GATK-sized batches).  parallel_for blocks until every slice is done.
namespace gklhip {
class WorkerPool {
 public:
  ~WorkerPool() { stop(); }
  // Exceptions (a slice's vector growing under memory pressure, a thread that cannot be created) never unwind
  // through a worker thread: the pool runs with the threads it has, a failing slice is remembered, every slice is
  // waited for, and the failure is re-thrown on the calling thread (the C ABI maps it to GKLHIP_ERR_OOM).
  // (min_n: below this many items the hand-off to the workers costs more than it saves -- 16384 log10f's; the PDHMM
  //  library's fp64 log10 of a region's 13k pairs already pays at 4096)
  void parallel_for(int64_t n, int threads, const std::function<void(int64_t, int64_t)>& fn, int64_t min_n = 16384) {
    if (threads <= 1 || n < min_n) { fn(0, n); return; }
    try { ensure(threads - 1); } catch (...) {}
    threads = std::min(threads, (int)workers_.size() + 1);
    if (threads <= 1) { fn(0, n); return; }
    const int64_t per = (n + threads - 1) / threads;
    {
      std::lock_guard<std::mutex> l(mu_);
      fn_ = &fn; n_ = n; per_ = per; slices_ = threads; next_ = 1; pending_ = threads - 1;
      failed_ = false;
      gen_++;
    }
    cv_.notify_all();
    std::exception_ptr mine;
    try { fn(0, std::min(n, per)); } catch (...) { mine = std::current_exception(); }
    bool failed;
    {
      std::unique_lock<std::mutex> l(mu_);
      done_.wait(l, [&] { return pending_ == 0; });
      fn_ = nullptr;
      failed = failed_;
    }
    if (mine) std::rethrow_exception(mine);
    if (failed) throw std::bad_alloc();
  }
  void stop() {
    {
      std::lock_guard<std::mutex> l(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
    workers_.clear();
    quit_ = false;
  }

 private:
  void ensure(int n) {
    while ((int)workers_.size() < n) workers_.emplace_back([this] { loop(); });
  }
  void loop() {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> l(mu_);
    for (;;) {
      cv_.wait(l, [&] { return quit_ || (gen_ != seen && next_ < slices_); });
      if (quit_) return;
      while (next_ < slices_) {
        const int k = next_++;
        const int64_t lo = k * per_, hi = std::min(n_, lo + per_);
        const auto* fn = fn_;
        l.unlock();
        bool threw = false;
        try { if (lo < hi) (*fn)(lo, hi); } catch (...) { threw = true; }
        l.lock();
        failed_ |= threw;
        if (--pending_ == 0) done_.notify_all();
      }
      seen = gen_;
    }
  }
  std::mutex mu_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> workers_;
  const std::function<void(int64_t, int64_t)>* fn_ = nullptr;
  int64_t n_ = 0, per_ = 0;
  int slices_ = 0, next_ = 0, pending_ = 0;
  uint64_t gen_ = 0;
  bool quit_ = false, failed_ = false;
};
}  // namespace gklhip

namespace gklhip {
// IntelPairHmm.cc:159-165 verbatim in meaning (host libm log10f / log10) over the packed raw sums.
// phase 1 (`early`): finalise the fp32-tagged words and remember where the others are (their fp64 sums are
// still being computed); phase 2 (`late`): finalise those from the complete copy.  With pending == nullptr
// one pass does everything.  Returns the number of pairs that took the fp64 path.
struct HostFinalizer {
  float lf = host_tables_f32().log10_initial;
  double ld = host_tables_f64().log10_initial;
  std::vector<std::vector<int32_t>> pending;  // per slice

  static inline bool is_f32(uint64_t w) { return (w & kPackedF32TagHost) == kPackedF32TagHost; }
  inline double fin32(uint64_t w) const {
    const uint32_t lo32 = (uint32_t)w;
    float f;
    memcpy(&f, &lo32, 4);
    return (double)(log10f(f) - lf);
  }
  inline double fin64(uint64_t w) const {
    double d;
    memcpy(&d, &w, 8);
    return log10(d) - ld;
  }

  int64_t all(WorkerPool* pool, const uint64_t* packed, double* out, int64_t n, int threads, int64_t min_n = 16384) const {
    std::atomic<int64_t> n64{0};
    const std::function<void(int64_t, int64_t)> work = [&](int64_t lo, int64_t hi) {
      int64_t cnt = 0;
      for (int64_t i = lo; i < hi; i++) {
        const uint64_t w = packed[i];
        if (is_f32(w)) out[i] = fin32(w);
        else { out[i] = fin64(w); cnt++; }
      }
      n64 += cnt;
    };
    pool->parallel_for(n, threads, work, min_n);
    return n64.load();
  }
  void early(WorkerPool* pool, const uint64_t* packed, double* out, int64_t n, int threads) {
    const int slices = (threads <= 1 || n < 16384) ? 1 : threads;
    const int64_t per = (n + slices - 1) / slices;
    pending.assign((size_t)slices, {});
    const std::function<void(int64_t, int64_t)> work = [&](int64_t lo, int64_t hi) {
      std::vector<int32_t>& mine = pending[(size_t)(lo / per)];
      for (int64_t i = lo; i < hi; i++) {
        const uint64_t w = packed[i];
        if (is_f32(w)) out[i] = fin32(w);
        else mine.push_back((int32_t)i);  // whatever the word holds: the fp64 pass may be writing it right now
      }
    };
    pool->parallel_for(n, threads, work);
  }
  int64_t late(WorkerPool* pool, const uint64_t* packed, double* out, int threads) const {
    int64_t total = 0;
    for (const auto& v : pending) total += (int64_t)v.size();
    const std::function<void(int64_t, int64_t)> work = [&](int64_t lo, int64_t hi) {
      // slice [lo, hi) of the concatenated pending lists
      int64_t at = 0;
      for (const auto& v : pending) {
        const int64_t a = std::max<int64_t>(lo - at, 0), b = std::min<int64_t>(hi - at, (int64_t)v.size());
        for (int64_t k = a; k < b; k++) out[v[(size_t)k]] = fin64(packed[v[(size_t)k]]);
        at += (int64_t)v.size();
      }
    };
    pool->parallel_for(total, threads, work);
    return total;
  }
};
}  // namespace gklhip
