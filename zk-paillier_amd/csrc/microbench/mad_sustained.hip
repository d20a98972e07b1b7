// Sustained issue rate of the 32x32+64 multiply-add on gfx950 (MI355X): the roofline denominator of the modexp kernels.
// valu_rates.hip measures it with 16 ms kernels; the engine's launches run for seconds, so this one repeats the measurement for
// growing kernel lengths (up to seconds) for the unsigned and the signed instruction.  16 independent accumulators per lane,
// 8 waves per SIMD on every CU.   hipcc -O3 --offload-arch=gfx950 mad_sustained.hip -o mad_sustained
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
#define REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

template <bool SIGNED>
__global__ void __launch_bounds__(256) k_mad(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {
  uint64_t acc[16];
  uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = i + threadIdx.x;
  for (int it = 0; it < iters; it++) {
    if (SIGNED) {
#define M(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
      REP16(M)
#undef M
    } else {
#define M(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
      REP16(M)
#undef M
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * 8;          // 8 blocks of 4 waves per CU = 8 waves per SIMD
  uint32_t* d; CHECK(hipMalloc(&d, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int sgn = 0; sgn < 2; sgn++) {
    for (int iters : {1 << 16, 1 << 19, 1 << 22, 1 << 24}) {
      CHECK(hipEventRecord(e0));
      if (sgn) hipLaunchKernelGGL(k_mad<true>, dim3(blocks), dim3(256), 0, 0, d, 12345u, 6789u, iters);
      else hipLaunchKernelGGL(k_mad<false>, dim3(blocks), dim3(256), 0, 0, d, 12345u, 6789u, iters);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double mads = (double)blocks * 256 * 16.0 * iters;
      printf("{\"instr\": \"%s\", \"iters\": %d, \"ms\": %.2f, \"lane_mad_per_s\": %.4g, \"cycles_per_wave_instr_per_simd_at_2.4GHz\": %.3f}\n",
             sgn ? "v_mad_i64_i32" : "v_mad_u64_u32", iters, ms, mads / (ms * 1e-3), (ms * 1e-3 * 2.4e9) / ((double)blocks * 4 * 16.0 * iters / (p.multiProcessorCount * 4.0)));
      fflush(stdout);
    }
  }
  return 0;
}
