// Dev tool (not product): issue-rate microbenchmarks for the instructions the PairHMM
// recurrence is made of, on gfx950.  Prints wave-instructions/s and the implied
// cycles per wave64 instruction per SIMD (assuming 256 CUs x 4 SIMDs at the measured
// clock).   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 4096;

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = a0 + 8, b1 = a0 + 9, b2 = a0 + 10, b3 = a0 + 11, b4 = a0 + 12, b5 = a0 + 13, b6 = a0 + 14, b7 = a0 + 15;
  float m = 0.999f, c = 1e-6f;
  __shared__ float lds[4096];  // 16 KiB
  lds[threadIdx.x] = a0; lds[threadIdx.x + 256] = a1;
  __syncthreads();
  unsigned addr = (threadIdx.x & 63) * 16;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {  // v_fma_f32, 8 independent chains x 8
      asm volatile(REP8(
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    } else if (MODE == 1) {  // v_pk_fma_f32 on register pairs, 4 independent chains x 16
      asm volatile(REP8(
          "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
          "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
          : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6)
          : "v"(*(double*)&b0), "v"(*(double*)&b2));
    } else if (MODE == 2) {  // v_mul_f32 with DPP wave_shr:1 source
      asm volatile(REP8(
          "v_mul_f32_dpp %0, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mul_f32_dpp %1, %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
          "v_mul_f32_dpp %2, %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mul_f32_dpp %3, %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
          "v_mul_f32_dpp %4, %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mul_f32_dpp %5, %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
          "v_mul_f32_dpp %6, %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mul_f32_dpp %7, %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n")
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
    } else if (MODE == 3) {  // v_and_b32 with DPP wave_shr:1 (the masked receive)
      asm volatile(REP8(
          "v_and_b32_dpp %0, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_and_b32_dpp %1, %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
          "v_and_b32_dpp %2, %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_and_b32_dpp %3, %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
          "v_and_b32_dpp %4, %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_and_b32_dpp %5, %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
          "v_and_b32_dpp %6, %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_and_b32_dpp %7, %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n")
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
    } else if (MODE == 4) {  // v_mov_b32 with DPP row_shr:1
      asm volatile(REP8(
          "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
          "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
          "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
          "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n")
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 5) {  // v_mul_f32 plain
      asm volatile(REP8(
          "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
          "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
    } else if (MODE == 6) {  // v_fma_f64, 4 chains
      asm volatile(REP8(
          "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
          "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n")
          : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6)
          : "v"(*(double*)&b0), "v"(*(double*)&b2));
    } else if (MODE == 7) {  // 7 v_fma_f32 + 1 ds_read_b128 per 8 (LDS co-issue)
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 r;
      asm volatile(REP8(
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n ds_read_b128 %7, %10\n")
          "s_waitcnt lgkmcnt(0)\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "=&v"(r) : "v"(m), "v"(c), "v"(addr));
      a7 += r.x;
    } else if (MODE == 8) {  // v_mov_b32 plain
      asm volatile(REP8(
          "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
          "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n")
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 9) {  // v_pk_mul_f32
      asm volatile(REP8(
          "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
          "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n")
          : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6)
          : "v"(*(double*)&b0));
    } else if (MODE == 10) {  // v_fmac_f32 with DPP wave_shr:1 source
      asm volatile(REP8(
          "v_fmac_f32_dpp %0, %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fmac_f32_dpp %1, %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
          "v_fmac_f32_dpp %2, %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fmac_f32_dpp %3, %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
          "v_fmac_f32_dpp %4, %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fmac_f32_dpp %5, %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
          "v_fmac_f32_dpp %6, %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fmac_f32_dpp %7, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n")
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    } else if (MODE == 11) {  // v_fmac_f32 e32 (VOP2, 4-byte encoding)
      asm volatile(REP8(
          "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
          "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n")
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    } else if (MODE == 12) {  // v_mul_f32 forced to the 8-byte VOP3 encoding
      asm volatile(REP8(
          "v_mul_f32_e64 %0, %0, %8\n v_mul_f32_e64 %1, %1, %8\n v_mul_f32_e64 %2, %2, %8\n v_mul_f32_e64 %3, %3, %8\n"
          "v_mul_f32_e64 %4, %4, %8\n v_mul_f32_e64 %5, %5, %8\n v_mul_f32_e64 %6, %6, %8\n v_mul_f32_e64 %7, %7, %8\n")
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
    } else if (MODE == 13) {  // v_fma_f32 with an SGPR multiplier and inline-constant addend
      asm volatile(REP8(
          "v_fma_f32 %0, %0, %8, 1.0\n v_fma_f32 %1, %1, %8, 1.0\n v_fma_f32 %2, %2, %8, 1.0\n v_fma_f32 %3, %3, %8, 1.0\n"
          "v_fma_f32 %4, %4, %8, 1.0\n v_fma_f32 %5, %5, %8, 1.0\n v_fma_f32 %6, %6, %8, 1.0\n v_fma_f32 %7, %7, %8, 1.0\n")
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(iters));
    } else if (MODE == 14) {  // v_fmac_f32 e32 with both multiplicands distinct per chain (a_i += b_i * m)
      asm volatile(REP8(
          "v_fmac_f32 %0, %8, %16\n v_fmac_f32 %1, %9, %16\n v_fmac_f32 %2, %10, %16\n v_fmac_f32 %3, %11, %16\n"
          "v_fmac_f32 %4, %12, %16\n v_fmac_f32 %5, %13, %16\n v_fmac_f32 %6, %14, %16\n v_fmac_f32 %7, %15, %16\n")
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7), "v"(m));
    } else if (MODE == 15) {  // v_add_f32 e32
      asm volatile(REP8(
          "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
          "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    } else if (MODE == 16) {  // v_mul_f64 / v_fma_f64 mix is VOP3 only; v_add_f64
      asm volatile(REP8(
          "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
          "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n")
          : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6)
          : "v"(*(double*)&b0));
    } else if (MODE == 17) {  // ds_read_b128 only (8 per 64 slots would starve: issue 64)
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 r0, r1, r2, r3;
      asm volatile(REP8(
          "ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
          "ds_read_b128 %0, %4 offset:4096\n ds_read_b128 %1, %4 offset:5120\n ds_read_b128 %2, %4 offset:6144\n ds_read_b128 %3, %4 offset:7168\n")
          "s_waitcnt lgkmcnt(0)\n"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr));
      a7 += r0.x + r1.x + r2.x + r3.x;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
}

template <int MODE>
int run(const char* name, float* d, int blocks, double clk_ghz, int n_cu, int waves_per_simd) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 64);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, ITER);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double wave_instr = (double)blocks * 4 /*waves per block*/ * ITER * 64.0;
  const double per_s = wave_instr / (ms * 1e-3);
  const double cyc = (double)n_cu * 4 * clk_ghz * 1e9 / per_s;
  printf("%-28s waves/SIMD=%d  %.3f ms  %.3e wave-instr/s  %.2f cycles/instr/SIMD (@%.2f GHz)\n", name, waves_per_simd, ms, per_s, cyc, clk_ghz);
  return 0;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const double clk = p.clockRate * 1e-6;
  printf("device %s  CUs %d  clock %.2f GHz\n", p.gcnArchName, p.multiProcessorCount, clk);
  float* d; CHECK(hipMalloc(&d, sizeof(float) * 256 * 256 * 8 * 4));
  for (int wps : {2, 4, 8}) {
    const int blocks = p.multiProcessorCount * wps;  // wps blocks of 4 waves per CU = wps waves per SIMD
    run<0>("v_fma_f32", d, blocks, clk, p.multiProcessorCount, wps);
    run<5>("v_mul_f32", d, blocks, clk, p.multiProcessorCount, wps);
    run<8>("v_mov_b32", d, blocks, clk, p.multiProcessorCount, wps);
    run<1>("v_pk_fma_f32", d, blocks, clk, p.multiProcessorCount, wps);
    run<9>("v_pk_mul_f32", d, blocks, clk, p.multiProcessorCount, wps);
    run<2>("v_mul_f32_dpp wave_shr:1", d, blocks, clk, p.multiProcessorCount, wps);
    run<10>("v_fmac_f32_dpp wave_shr:1", d, blocks, clk, p.multiProcessorCount, wps);
    run<3>("v_and_b32_dpp wave_shr:1", d, blocks, clk, p.multiProcessorCount, wps);
    run<4>("v_mov_b32_dpp row_shr:1", d, blocks, clk, p.multiProcessorCount, wps);
    run<6>("v_fma_f64", d, blocks, clk, p.multiProcessorCount, wps);
    run<7>("7 v_fma_f32 + ds_read_b128", d, blocks, clk, p.multiProcessorCount, wps);
    run<11>("v_fmac_f32 e32", d, blocks, clk, p.multiProcessorCount, wps);
    run<14>("v_fmac_f32 e32 distinct", d, blocks, clk, p.multiProcessorCount, wps);
    run<12>("v_mul_f32 e64", d, blocks, clk, p.multiProcessorCount, wps);
    run<13>("v_fma_f32 sgpr,const", d, blocks, clk, p.multiProcessorCount, wps);
    run<15>("v_add_f32 e32", d, blocks, clk, p.multiProcessorCount, wps);
    run<16>("v_mul_f64", d, blocks, clk, p.multiProcessorCount, wps);
    run<17>("ds_read_b128 x64", d, blocks, clk, p.multiProcessorCount, wps);
  }
  return 0;
}
