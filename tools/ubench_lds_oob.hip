// Dev tool: what does a DS read beyond the workgroup's LDS allocation return on gfx950?  (ISA manuals since GCN3: 0.)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_oob tools/ubench_lds_oob.hip && /tmp/lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(uint32_t* out) {
  __shared__ uint32_t lds[2560];  // 10 KB, like the forward kernel's table
  for (int i = threadIdx.x; i < 2560; i += 64) lds[i] = 0xABCD0000u + i;
  __syncthreads();
  uint32_t base = (uint32_t)(uintptr_t)lds;
  uint32_t addrs[4] = {base + 10240 + threadIdx.x * 16, base + 0x100000 + threadIdx.x * 16, 0xFFFFF800u + threadIdx.x * 16, base + 65536 * 2 + threadIdx.x * 4};
  for (int a = 0; a < 4; a++) {
    uint32_t v0, v1, v2, v3;
    asm volatile("ds_read_b128 v[40:43], %4\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, v40\n\tv_mov_b32 %1, v41\n\tv_mov_b32 %2, v42\n\tv_mov_b32 %3, v43"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(addrs[a]) : "v40", "v41", "v42", "v43");
    out[(blockIdx.x * 4 + a) * 64 + threadIdx.x] = v0 | v1 | v2 | v3;
  }
}
int main() {
  const int blocks = 4096;  // many workgroups per CU: neighbours' allocations sit right behind ours
  uint32_t* d; hipMalloc(&d, blocks * 4 * 64 * 4);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d);
  hipDeviceSynchronize();
  uint32_t* h = new uint32_t[blocks * 4 * 64];
  hipMemcpy(h, d, blocks * 4 * 64 * 4, hipMemcpyDeviceToHost);
  for (int a = 0; a < 4; a++) {
    long nz = 0; uint32_t ex = 0;
    for (int b = 0; b < blocks; b++) for (int l = 0; l < 64; l++) { uint32_t v = h[(b * 4 + a) * 64 + l]; if (v) { nz++; ex = v; } }
    printf("address class %d: %ld non-zero of %d reads (example %08x)\n", a, nz, blocks * 64, ex);
  }
  return 0;
}
