// Montgomery-product rate of bigint29.hpp's systolic montmul in isolation, as the ladders use it (in place on X, the B
// operand re-staged in LDS after every product), for one (W, G) geometry per build:
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-sched-strategy=iterative-ilp -DZKP_W=18 -DMM_G=8  montmul_rate.hip -o mm_w18_g8
//   hipcc ... -DZKP_W=36 -DMM_G=4 ...                                                                  -o mm_w36_g4
// Same 144-limb (4176-bit) integers in both; prints products/s and the executed-MAD rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../bigint29.hpp"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
using namespace zkp;
#ifndef MM_G
#define MM_G 8
#endif
#ifndef MM_WPE
#define MM_WPE 2
#endif
constexpr int G = MM_G;

__global__ void __launch_bounds__(256, MM_WPE) k_mm(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int iters) {
  extern __shared__ __align__(16) uint32_t lds[];
  const int lane = threadIdx.x & 63, gl = lane & (G - 1), gib = threadIdx.x / G;
  uint32_t* B = lds + gib * (G * BLK);
  uint32_t X[W], NT[W];
  const uint32_t* src = in + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2 * W;
#pragma unroll
  for (int k = 0; k < W; k++) { X[k] = src[k] & LMASK; NT[k] = src[W + k] & LMASK; }
  if (gl == 0) NT[0] = LMASK;                 // Orup multiple: == -1 mod 2^29
  balance<G>(X, gl);                          // the running value is kept in balanced digits
  wave_lds_fence(); lds_store_block(B + gl * BLK, X); wave_lds_fence();
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    montmul<G, true>(X, X, B, NT, 1u, gl);
    wave_lds_fence(); lds_store_block(B + gl * BLK, X); wave_lds_fence();
  }
  uint32_t* dst = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * W;
#pragma unroll
  for (int k = 0; k < W; k++) dst[k] = X[k];
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * MM_WPE;
  const size_t threads = (size_t)blocks * 256;
  std::vector<uint32_t> h(threads * 2 * W);
  uint32_t s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 3; }
  uint32_t *din, *dout;
  CHECK(hipMalloc(&din, h.size() * 4)); CHECK(hipMalloc(&dout, threads * W * 4));
  CHECK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const size_t ldsb = (size_t)(256 / G) * G * BLK * 4;
  int occ = 0; CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_mm, 256, ldsb));
  hipFuncAttributes fa; CHECK(hipFuncGetAttributes(&fa, (const void*)k_mm));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_mm, dim3(blocks), dim3(256), ldsb, 0, din, dout, iters); CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_mm, dim3(blocks), dim3(256), ldsb, 0, din, dout, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double products = (double)blocks * (256 / G) * iters;
  const double L = (double)G * W;
  printf("{\"W\": %d, \"G\": %d, \"waves_per_simd_requested\": %d, \"blocks_per_cu_possible\": %d, \"vgprs\": %d, \"scratch_bytes\": %zu, \"lds_bytes_per_block\": %zu, \"iters\": %d, \"ms\": %.3f, "
         "\"products_per_s\": %.4g, \"executed_mad_per_s\": %.4g, \"algorithmic_limb_mac_per_s (2*128^2+128 per product)\": %.4g}\n",
         W, G, MM_WPE, occ, fa.numRegs, (size_t)fa.localSizeBytes, ldsb, iters, best, products / (best * 1e-3), products * 2 * L * L / (best * 1e-3),
         products * 32896.0 / (best * 1e-3));
  return 0;
}
