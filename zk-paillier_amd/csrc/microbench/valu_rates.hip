// VALU instruction-rate microbenchmark for gfx950 (MI355X).
// Purpose: the guides list FP/MFMA peaks only; the big-integer kernels are bound by
// 32-bit integer multiply throughput, so the roofline denominator has to be measured.
// Each kernel runs ITER iterations of 16 independent instructions of one kind per lane,
// 8 waves per SIMD on every CU, and reports wave-instructions/s and cycles per
// wave-instruction per SIMD (at the measured shader clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

#ifndef ITER_N
#define ITER_N 4096
#endif
constexpr int ITER = ITER_N;

#define REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

// ---- v_mad_u64_u32: 16 independent accumulators
__global__ void __launch_bounds__(256) k_mad_u64_u32(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint64_t acc[16];
  uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = i + threadIdx.x;
  for (int it = 0; it < ITER; it++) {
#define M(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
    REP16(M)
#undef M
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

// ---- mad + addc pair (3-word column accumulator pattern): carry-out of the mad feeds an addc
__global__ void __launch_bounds__(256) k_mad_addc(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint64_t acc[16]; uint32_t c2[16];
  uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; i++) { acc[i] = i + threadIdx.x; c2[i] = 0; }
  for (int it = 0; it < ITER; it++) {
#define M(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[i]), "+v"(c2[i]) : "v"(a), "v"(b) : "vcc");
    REP16(M)
#undef M
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= acc[i] + c2[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

#define KERNEL_2OP(NAME, INSTR) \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t a0, uint32_t b0) { \
  uint32_t acc[16]; uint32_t b = b0 ^ threadIdx.x; \
  _Pragma("unroll") for (int i = 0; i < 16; i++) acc[i] = a0 + i + threadIdx.x; \
  for (int it = 0; it < ITER; it++) { \
    _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(INSTR " %0, %0, %1" : "+v"(acc[i]) : "v"(b)); \
  } \
  uint32_t s = 0; \
  _Pragma("unroll") for (int i = 0; i < 16; i++) s ^= acc[i]; \
  out[blockIdx.x * blockDim.x + threadIdx.x] = s; \
}

KERNEL_2OP(k_mul_lo_u32, "v_mul_lo_u32")
KERNEL_2OP(k_mul_hi_u32, "v_mul_hi_u32")
KERNEL_2OP(k_mul_u32_u24, "v_mul_u32_u24")
KERNEL_2OP(k_mul_hi_u32_u24, "v_mul_hi_u32_u24")
KERNEL_2OP(k_add_u32, "v_add_u32")
KERNEL_2OP(k_xor_b32, "v_xor_b32")

__global__ void __launch_bounds__(256) k_mad_u32_u24(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t acc[16]; uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = i + threadIdx.x;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_addc_chain(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t acc[16]; uint32_t b = b0 ^ threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = a0 + i + threadIdx.x;
  for (int it = 0; it < ITER; it++) {
    asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(acc[0]) : "v"(b) : "vcc");
#pragma unroll
    for (int i = 1; i < 16; i++) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(acc[i]) : "v"(b) : "vcc");
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_lshl_add_u64(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint64_t acc[16]; uint64_t b = ((uint64_t)b0 << 20) ^ threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = a0 + i + threadIdx.x;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(b));
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

// round 6: the bookkeeping instructions of a sub-step of the assembler engine (tools/bn_asm/gen.py) — is the 64-bit shift a full-rate op?
__global__ void __launch_bounds__(256) k_lshrrev_b64(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint64_t acc[16], src[16];
#pragma unroll
  for (int i = 0; i < 16; i++) { acc[i] = 0; src[i] = ((uint64_t)(a0 + i + threadIdx.x) << 33) ^ b0; }
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_lshrrev_b64 %0, 29, %1" : "=v"(acc[i]) : "v"(src[i]));
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
__global__ void __launch_bounds__(256) k_alignbit_pair(uint32_t* out, uint32_t a0, uint32_t b0) {
  // the same shift as two 32-bit instructions: v_alignbit_b32 (low word) + v_lshrrev_b32 (high word)
  uint32_t lo[16], hi[16], sl[16], sh[16];
#pragma unroll
  for (int i = 0; i < 16; i++) { lo[i] = hi[i] = 0; sl[i] = a0 + i + threadIdx.x; sh[i] = b0 ^ (i * 77 + threadIdx.x); }
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_alignbit_b32 %0, %3, %2, 29\n\tv_lshrrev_b32_e32 %1, 29, %3" : "=v"(lo[i]), "=v"(hi[i]) : "v"(sl[i]), "v"(sh[i]));
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= lo[i] ^ hi[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_and_dpp(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t acc[16], src[16]; uint32_t m = b0 | 0x1fffffff;
#pragma unroll
  for (int i = 0; i < 16; i++) { acc[i] = 0; src[i] = a0 + i + threadIdx.x; }
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_and_b32_dpp %0, %1, %2 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(acc[i]) : "v"(src[i]), "v"(m));
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_fma_f64(uint32_t* out, uint32_t a0, uint32_t b0) {
  double acc[16]; double a = 1.0 + 1e-9 * (a0 + threadIdx.x), b = 1e-9 * (b0 ^ threadIdx.x);
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = i + threadIdx.x;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(int64_t)s;
}

__global__ void __launch_bounds__(256) k_fma_f32(uint32_t* out, uint32_t a0, uint32_t b0) {
  float acc[16]; float a = 1.0f + 1e-9f * (a0 + threadIdx.x), b = 1e-9f * (b0 ^ threadIdx.x);
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = i + threadIdx.x;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(int)s;
}

// DPP move (row_shr:1) and wave_shr:1
__global__ void __launch_bounds__(256) k_dpp_row_shr(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t acc[16];
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = a0 + i + threadIdx.x;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]));
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s ^ b0;
}

__global__ void __launch_bounds__(256) k_add_dpp(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t acc[16]; uint32_t b = b0 ^ threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = a0 + i + threadIdx.x;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_add_u32_dpp %0, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(b));
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_ds_swizzle(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t acc[16];
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = a0 + i + threadIdx.x;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_ds_swizzle(acc[i], 0x0010) + 1; // and_mask=0x10: broadcast lane 0/16 of each 32-group
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s ^ b0;
}

__global__ void __launch_bounds__(256) k_ds_bpermute(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t acc[16]; int idx = (threadIdx.x & 48) << 2;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = a0 + i + threadIdx.x;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_ds_bpermute(idx, acc[i]) + 1;
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s ^ b0;
}

__global__ void __launch_bounds__(256) k_readlane(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t acc[16];
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = a0 + i + threadIdx.x;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] += __builtin_amdgcn_readlane(acc[(i + 8) & 15], 5);
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s ^ b0;
}

// dependent chain latency of v_mad_u64_u32 (1 wave per SIMD)
__global__ void __launch_bounds__(256) k_mad_dep(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint64_t acc = threadIdx.x; uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)acc ^ (uint32_t)(acc >> 32);
}

__global__ void k_clock(uint64_t* out) {
  uint64_t c0 = clock64(), w0 = wall_clock64();
  while (wall_clock64() - w0 < 1000000) {}  // 10 ms at 100 MHz
  uint64_t c1 = clock64(), w1 = wall_clock64();
  out[0] = c1 - c0; out[1] = w1 - w0;
}

typedef void (*kern_t)(uint32_t*, uint32_t, uint32_t);

static double run(const char* name, kern_t k, int waves_per_simd, int instr_per_iter, uint32_t* dout, int ncu, double clk_ghz, double work_per_instr = 0) {
  int blocks = ncu * waves_per_simd;  // 256 threads = 4 waves -> one per SIMD
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, dout, 12345u, 6789u);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, dout, 12345u, 6789u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  double wave_instrs = (double)blocks * 4 * ITER * instr_per_iter;
  double per_simd = wave_instrs / (ncu * 4.0);
  double cyc = best * 1e-3 * clk_ghz * 1e9 / per_simd;
  double lane_ops = wave_instrs * 64 / (best * 1e-3);
  printf("{\"kernel\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"cycles_per_wave_instr_per_simd\": %.2f, \"lane_ops_per_s\": %.4e}\n", name, waves_per_simd, best, cyc, lane_ops);
  fflush(stdout);
  return cyc;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  int ncu = p.multiProcessorCount;
  uint32_t* dout; CHECK(hipMalloc(&dout, (size_t)ncu * 8 * 256 * 4 * 2));
  uint64_t* dclk; CHECK(hipMalloc(&dclk, 16));
  // warm the clock with a busy kernel, then measure
  hipLaunchKernelGGL(k_mad_u64_u32, dim3(ncu * 8), dim3(256), 0, 0, dout, 1u, 2u);
  hipLaunchKernelGGL(k_clock, dim3(1), dim3(1), 0, 0, dclk);
  CHECK(hipDeviceSynchronize());
  uint64_t h[2]; CHECK(hipMemcpy(h, dclk, 16, hipMemcpyDeviceToHost));
  double clk_ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;  // wall_clock64 = 100 MHz
  printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clockRate_khz\": %d, \"measured_clock_ghz_idle_kernel\": %.3f}\n", p.name, p.gcnArchName, ncu, p.clockRate, clk_ghz);
  double clk = p.clockRate * 1e-6;  // GHz nominal
  for (int w : {1, 2, 4, 8}) run("v_mad_u64_u32", k_mad_u64_u32, w, 16, dout, ncu, clk);
  run("v_mad_u64_u32+v_addc", k_mad_addc, 8, 16, dout, ncu, clk);
  run("v_mad_u64_u32+v_addc", k_mad_addc, 4, 16, dout, ncu, clk);
  run("v_mul_lo_u32", k_mul_lo_u32, 8, 16, dout, ncu, clk);
  run("v_mul_hi_u32", k_mul_hi_u32, 8, 16, dout, ncu, clk);
  run("v_mul_u32_u24", k_mul_u32_u24, 8, 16, dout, ncu, clk);
  run("v_mul_hi_u32_u24", k_mul_hi_u32_u24, 8, 16, dout, ncu, clk);
  run("v_mad_u32_u24", k_mad_u32_u24, 8, 16, dout, ncu, clk);
  run("v_add_u32", k_add_u32, 8, 16, dout, ncu, clk);
  run("v_xor_b32", k_xor_b32, 8, 16, dout, ncu, clk);
  run("v_addc_co_u32 chain", k_addc_chain, 8, 16, dout, ncu, clk);
  run("v_lshl_add_u64", k_lshl_add_u64, 8, 16, dout, ncu, clk);
  for (int w : {2, 8}) run("v_lshrrev_b64", k_lshrrev_b64, w, 16, dout, ncu, clk);
  for (int w : {2, 8}) run("v_alignbit_b32 + v_lshrrev_b32 (per pair)", k_alignbit_pair, w, 16, dout, ncu, clk);
  for (int w : {2, 8}) run("v_and_b32_dpp quad_perm", k_and_dpp, w, 16, dout, ncu, clk);
  for (int w : {2}) run("v_lshl_add_u64", k_lshl_add_u64, w, 16, dout, ncu, clk);
  for (int w : {2}) run("v_add_u32", k_add_u32, w, 16, dout, ncu, clk);
  run("v_fma_f64", k_fma_f64, 8, 16, dout, ncu, clk);
  run("v_fma_f32", k_fma_f32, 8, 16, dout, ncu, clk);
  run("v_mov_b32_dpp row_shr:1 (+s_nop 1)", k_dpp_row_shr, 8, 16, dout, ncu, clk);
  run("v_add_u32_dpp row_shl:1", k_add_dpp, 8, 16, dout, ncu, clk);
  run("ds_swizzle_b32 (+v_add)", k_ds_swizzle, 8, 16, dout, ncu, clk);
  run("ds_bpermute_b32 (+v_add)", k_ds_bpermute, 8, 16, dout, ncu, clk);
  run("v_readlane_b32 (+v_add)", k_readlane, 8, 16, dout, ncu, clk);
  run("v_mad_u64_u32 dependent chain (1 wave/SIMD = latency)", k_mad_dep, 1, 16, dout, ncu, clk);
  run("v_mad_u64_u32 dependent chain (8 waves/SIMD)", k_mad_dep, 8, 16, dout, ncu, clk);
  return 0;
}
