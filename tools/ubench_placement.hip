// Dev tool: where does the workgroup dispatcher put the single-wavefront blocks of a grid that does not fill the chip?
// Each block records its XCD / SE / CU / SIMD and spins ~100 us so that the whole grid is resident at once.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/ubench_placement tools/ubench_placement.hip
//   tools/bin/ubench_placement <blocks> <static LDS fits 16/CU; extra dynamic LDS bytes>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void where_kernel(uint32_t* out, long long spin_ticks) {
  __shared__ unsigned char lds[10240];
  extern __shared__ unsigned char dyn[];
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  lds[threadIdx.x] = (unsigned char)hw;
  dyn[threadIdx.x] = (unsigned char)xcc;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin_ticks) {}
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw + lds[1] * 0 + dyn[1] * 0;
    out[2 * blockIdx.x + 1] = xcc;
  }
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 2240;
  const int dyn = argc > 2 ? atoi(argv[2]) : 0;
  uint32_t* d;
  hipMalloc(&d, (size_t)blocks * 8);
  hipFuncSetAttribute((const void*)where_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 10240);
  hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(64), dyn, 0, d, 10000LL);  // 100 MHz wall clock: 100 us
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  std::vector<uint32_t> h((size_t)blocks * 2);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  std::map<uint32_t, int> per_cu, per_simd;
  for (int b = 0; b < blocks; b++) {
    const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    const uint32_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const uint32_t cu_key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    per_cu[cu_key]++;
    per_simd[(cu_key << 2) | simd]++;
  }
  std::map<int, int> hist_cu, hist_simd;
  for (auto& kv : per_cu) hist_cu[kv.second]++;
  for (auto& kv : per_simd) hist_simd[kv.second]++;
  printf("%d blocks (+%d B dynamic LDS): %zu CUs used, %zu SIMDs used\n  waves per CU : ", blocks, dyn, per_cu.size(), per_simd.size());
  for (auto& kv : hist_cu) printf("%d CUs x %d, ", kv.second, kv.first);
  printf("\n  waves per SIMD: ");
  for (auto& kv : hist_simd) printf("%d SIMDs x %d, ", kv.second, kv.first);
  printf("\n");
  return 0;
}
