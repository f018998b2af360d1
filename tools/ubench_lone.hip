// Dev tool: how fast does ONE wavefront per SIMD issue?  Independent v_fma_f32 streams of distance 1 (every op depends
// on the previous one) ... 16 (each op depends on the op 16 back), 1 / 2 / 4 wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/ubench_lone tools/ubench_lone.hip && tools/bin/ubench_lone
#include <hip/hip_runtime.h>
#include <cstdio>

template <int DIST>
__global__ __launch_bounds__(64) void chain_kernel(float* out, int iters, long long* ticks) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
  float x = out[0], y = out[1];
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 32; k++) {
      // op k updates accumulator k % DIST: distance DIST between dependent ops
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k % DIST]) : "v"(x), "v"(y));
    }
  }
  const long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += a[i];
  if (s == 123.456f) out[2] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int DIST>
void run(int blocks, const char* what) {
  float* out; long long* ticks;
  hipMalloc(&out, 64); hipMalloc(&ticks, 8);
  hipMemset(out, 0, 64);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(chain_kernel<DIST>, dim3(blocks), dim3(64), 0, 0, out, 100, ticks);
  hipEventRecord(e0);
  hipLaunchKernelGGL(chain_kernel<DIST>, dim3(blocks), dim3(64), 0, 0, out, iters, ticks);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // cycles per instruction per wavefront at an assumed 2.4 GHz, and from the event time
  printf("%-22s distance %2d: %.2f ns per instruction per wavefront = %.2f cycles at 2.4 GHz\n", what, DIST,
         ms * 1e6 / ((double)iters * 32), ms * 1e6 / ((double)iters * 32) * 2.4);
}

int main() {
  for (int w : {1, 2, 4}) {
    char what[64]; snprintf(what, sizeof what, "%d wavefront(s)/SIMD", w);
    const int blocks = 1024 * w;
    run<1>(blocks, what); run<2>(blocks, what); run<4>(blocks, what); run<8>(blocks, what); run<16>(blocks, what);
  }
  return 0;
}
