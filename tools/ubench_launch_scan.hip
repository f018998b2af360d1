// Which launches of a fresh process are slow?  N times: one tiny kernel + hipStreamSynchronize on a non-blocking stream,
// every launch above 1 ms printed with its index (docs/NOTES.md 56).  hipcc --offload-arch=gfx950 -O2 -o tools/bin/ubench_launch_scan
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void tiny(int* p) { if (threadIdx.x == 0 && p) p[0] += 1; }
struct Big { int* p; char pad[2040]; };   // 2 KB of kernel arguments (the forward kernels' argument blocks are ~1 KB)
__global__ void tiny_big(Big b) { if (threadIdx.x == 0 && b.p) b.p[0] += b.pad[7]; }
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 4000, per = argc > 2 ? atoi(argv[2]) : 1, big = argc > 3 ? atoi(argv[3]) : 0;
  Big b{};
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int* d = nullptr;
  hipMalloc(&d, 64);
  for (int i = 0; i < n; i++) {
    const auto t0 = std::chrono::steady_clock::now();
    b.p = d;
    for (int k = 0; k < per; k++) { if (big) hipLaunchKernelGGL(tiny_big, dim3(1), dim3(64), 0, s, b); else hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d); }
    hipStreamSynchronize(s);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ms > 1.0) printf("iteration %d (launch %d): %.2f ms\n", i, i * per, ms);
  }
  printf("done: %d iterations of %d launches, %s kernel arguments\n", n, per, big ? "2 KB of" : "8 bytes of");
  return 0;
}
