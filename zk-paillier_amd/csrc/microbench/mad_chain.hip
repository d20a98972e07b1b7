// Dependent v_mad_u64_u32 chains on gfx950: how many independent accumulators does a wavefront need to reach
// the issue rate measured with 16 accumulators (valu_rates.hip)?  Also: cost of finishing a product-scanning
// column (v_and + 64-bit shift + LDS atomic add) next to a chain.  Feeds the design of the squaring product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
#ifndef ITER_N
#define ITER_N 8192
#endif
constexpr int ITER = ITER_N;

template <int K>
__global__ void __launch_bounds__(256) k_chain(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint64_t acc[K];
  uint32_t a[16], b = b0 ^ threadIdx.x;
#pragma unroll
  for (int i = 0; i < K; i++) acc[i] = i + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = a0 * (i + 1) + threadIdx.x;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i % K]) : "v"(a[i]), "v"(b) : "vcc");
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < K; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

// product-scanning column: 16 dependent MADs, then emit low 29 bits to LDS (atomic add) and keep the carry
template <int K, bool ATOMIC>
__global__ void __launch_bounds__(256) k_column(uint32_t* out, uint32_t a0, uint32_t b0) {
  __shared__ uint32_t T[256 * 4];
  uint32_t a[16], b = b0 ^ threadIdx.x;
  uint64_t acc[K];
#pragma unroll
  for (int i = 0; i < K; i++) acc[i] = i;
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = a0 * (i + 1) + threadIdx.x;
  for (int i = threadIdx.x; i < 1024; i += 256) T[i] = 0;
  __syncthreads();
  uint32_t* tp = T + threadIdx.x;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int k = 0; k < K; k++) {
#pragma unroll
      for (int i = 0; i < 16 / K; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a[i + k]), "v"(b) : "vcc");
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
      const uint32_t lo = (uint32_t)acc[k] & 0x1fffffffu;
      if (ATOMIC) __hip_atomic_fetch_add(tp + 256 * (k & 3), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else tp[256 * (k & 3)] = lo;
      acc[k] >>= 29;
    }
  }
  __syncthreads();
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < K; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ T[threadIdx.x] ^ T[threadIdx.x + 256];
}

typedef void (*kern_t)(uint32_t*, uint32_t, uint32_t);
static void run(const char* name, kern_t k, int wps, uint32_t* dout, int ncu) {
  int blocks = ncu * wps;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, dout, 12345u, 6789u); CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, dout, 12345u, 6789u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  double groups = (double)blocks * 4 * ITER / (ncu * 4.0);   // 16-MAD groups per SIMD
  printf("{\"kernel\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"cycles_per_16mad_group_per_simd_at_2.4GHz\": %.2f, \"cycles_per_mad\": %.2f}\n", name, wps, best,
         best * 1e-3 * 2.4e9 / groups, best * 1e-3 * 2.4e9 / groups / 16);
}
int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0)); int ncu = p.multiProcessorCount;
  uint32_t* dout; CHECK(hipMalloc(&dout, (size_t)ncu * 8 * 256 * 4 * 2));
  for (int w : {1, 2, 3, 8}) {
    run("chain, 1 accumulator", k_chain<1>, w, dout, ncu);
    run("chain, 2 accumulators", k_chain<2>, w, dout, ncu);
    run("chain, 4 accumulators", k_chain<4>, w, dout, ncu);
    run("chain, 16 accumulators", k_chain<16>, w, dout, ncu);
    run("column of 16: 1 acc + and/shift/ds_add", k_column<1, true>, w, dout, ncu);
    run("column of 16: 1 acc + and/shift/ds_write", k_column<1, false>, w, dout, ncu);
    run("2 columns of 8 + 2x and/shift/ds_add", k_column<2, true>, w, dout, ncu);
    run("4 columns of 4 + 4x and/shift/ds_add", k_column<4, true>, w, dout, ncu);
  }
  return 0;
}
