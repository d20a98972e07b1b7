// v_mad_u64_u32 issue-rate variants on gfx950: does the SGPR carry-out destination, operand kind or
// interleaving with plain VALU work change the ~4.9 cycles per wave-instruction measured in valu_rates.hip?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
#ifndef ITER_N
#define ITER_N 8192
#endif
constexpr int ITER = ITER_N;

#define DECL uint64_t acc[16]; uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x; uint32_t x[8]; \
  _Pragma("unroll") for (int i = 0; i < 16; i++) acc[i] = i + threadIdx.x; \
  _Pragma("unroll") for (int i = 0; i < 8; i++) x[i] = i * 77 + threadIdx.x;
#define FIN uint64_t s = 0; _Pragma("unroll") for (int i = 0; i < 16; i++) s ^= acc[i]; uint32_t y = 0; _Pragma("unroll") for (int i = 0; i < 8; i++) y ^= x[i]; \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ y;

__global__ void __launch_bounds__(256) k_vcc(uint32_t* out, uint32_t a0, uint32_t b0) {
  DECL
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
  }
  FIN
}
__global__ void __launch_bounds__(256) k_sdst_rot(uint32_t* out, uint32_t a0, uint32_t b0) {
  DECL
  for (int it = 0; it < ITER; it++) {
#define M(i, S) asm volatile("v_mad_u64_u32 %0, " S ", %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    M(0, "s[20:21]") M(1, "s[22:23]") M(2, "s[24:25]") M(3, "s[26:27]") M(4, "s[20:21]") M(5, "s[22:23]") M(6, "s[24:25]") M(7, "s[26:27]")
    M(8, "s[20:21]") M(9, "s[22:23]") M(10, "s[24:25]") M(11, "s[26:27]") M(12, "s[20:21]") M(13, "s[22:23]") M(14, "s[24:25]") M(15, "s[26:27]")
#undef M
  }
  FIN
}
// the multiplier operand from an SGPR
__global__ void __launch_bounds__(256) k_sgpr_src(uint32_t* out, uint32_t a0, uint32_t b0) {
  DECL
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "s"(b0) : "vcc");
  }
  FIN
}
// 16 MADs + 4 v_and interleaved (the kernel's mix is 36 : 8)
__global__ void __launch_bounds__(256) k_mix_and(uint32_t* out, uint32_t a0, uint32_t b0) {
  DECL
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
      if (i % 4 == 3) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i / 4]) : "v"(b));
    }
  }
  FIN
}
// 16 MADs + 4 v_lshl_add_u64 interleaved
__global__ void __launch_bounds__(256) k_mix_add64(uint32_t* out, uint32_t a0, uint32_t b0) {
  DECL
  uint64_t z[4] = {1, 2, 3, 4};
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
      if (i % 4 == 3) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(z[i / 4]) : "v"(acc[i]));
    }
  }
  acc[0] ^= z[0] ^ z[1] ^ z[2] ^ z[3];
  FIN
}
// 16 v_mul_lo_u32 + 16 v_mul_hi_u32 (alternative product form), no accumulation
__global__ void __launch_bounds__(256) k_mul_lo_hi(uint32_t* out, uint32_t a0, uint32_t b0) {
  DECL
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(b));
      asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(a));
    }
  }
  FIN
}
// MAD with 64-bit addend = constant 0 (the form the compiler emits to open a column)
__global__ void __launch_bounds__(256) k_mad_zero(uint32_t* out, uint32_t a0, uint32_t b0) {
  DECL
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(acc[i]) : "v"(a + (uint32_t)acc[i]), "v"(b) : "vcc");
  }
  FIN
}

typedef void (*kern_t)(uint32_t*, uint32_t, uint32_t);
static void run(const char* name, kern_t k, int wps, double instr, uint32_t* dout, int ncu) {
  int blocks = ncu * wps;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, dout, 12345u, 6789u); CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, dout, 12345u, 6789u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  double groups = (double)blocks * 4 * ITER / (ncu * 4.0);   // 16-MAD groups per SIMD
  printf("{\"kernel\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"cycles_per_16mad_group_per_simd_at_2.4GHz\": %.2f, \"cycles_per_instr\": %.2f}\n", name, wps, best,
         best * 1e-3 * 2.4e9 / groups, best * 1e-3 * 2.4e9 / groups / instr);
}
int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0)); int ncu = p.multiProcessorCount;
  uint32_t* dout; CHECK(hipMalloc(&dout, (size_t)ncu * 8 * 256 * 4 * 2));
  for (int w : {2, 8}) {
    run("16 mad, sdst=vcc", k_vcc, w, 16, dout, ncu);
    run("16 mad, sdst rotates over 4 sgpr pairs", k_sdst_rot, w, 16, dout, ncu);
    run("16 mad, multiplier from sgpr", k_sgpr_src, w, 16, dout, ncu);
    run("16 mad + 4 v_and", k_mix_and, w, 20, dout, ncu);
    run("16 mad + 4 v_lshl_add_u64", k_mix_add64, w, 20, dout, ncu);
    run("8 mul_lo + 8 mul_hi", k_mul_lo_hi, w, 16, dout, ncu);
    run("16 mad with zero addend", k_mad_zero, w, 16, dout, ncu);
  }
  return 0;
}
