// Dev tool (not product): how many small dependent kernel chains per second the device sustains from N host threads with a
// stream each -- K kernels of `us` microseconds each per chain, one hipStreamSynchronize per chain (what a small
// host-buffer PairHMM call looks like to the command processor).
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void spin_kernel(long long cycles, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (cycles < 0) *sink = 1;
}
int main(int argc, char** argv) {
  for (int K : {1, 3}) for (int us : {0, 50}) for (int N : {1, 4, 16}) {
    std::atomic<long> calls{0};
    std::atomic<bool> stop{false};
    std::vector<std::thread> th;
    for (int i = 0; i < N; i++) th.emplace_back([&] {
      hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
      while (!stop.load()) {
        for (int k = 0; k < K; k++) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, s, (long long)us * 100, nullptr);  // wall_clock64: 100 MHz
        hipStreamSynchronize(s);
        calls++;
      }
      hipStreamDestroy(s);
    });
    std::this_thread::sleep_for(std::chrono::milliseconds(100));
    const long c0 = calls.load();
    const auto t0 = std::chrono::steady_clock::now();
    std::this_thread::sleep_for(std::chrono::milliseconds(400));
    const long c1 = calls.load();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stop = true;
    for (auto& t : th) t.join();
    printf("K=%d kernels of %2d us, %2d threads: %8.0f chains/s  (%.1f us per chain per thread, %.0f kernels/s)\n", K, us, N, (c1 - c0) / dt, N * dt / (c1 - c0) * 1e6, K * (c1 - c0) / dt);
  }
}
