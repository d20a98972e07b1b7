// Power side of the multiply-add: the engine multiplies ONE broadcast limb by a run of register limbs.  Does it matter (clock held by the
// board = power) whether the limb that stays put is src0 or src1 of v_mad_u64_u32, and how long the runs are?  Random 29-bit operands.
//   hipcc -O3 --offload-arch=gfx950 mad_operand_order.hip -o mad_operand_order      (run under tools/dev/mad_peak_with_clock.py)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// MODE 0: src0 varies, src1 = the run's fixed limb; MODE 1: src0 = the fixed limb, src1 varies; RUN = instructions per fixed limb (16 or 4)
template <int MODE, int RUN>
__global__ void __launch_bounds__(256) k_mad(uint32_t* out, uint32_t seed, int iters) {
  uint64_t acc[16];
  uint32_t op[16], fx[4];
#pragma unroll
  for (int i = 0; i < 16; i++) op[i] = mix(seed + 977u * i + 131071u * (blockIdx.x * 256u + threadIdx.x)) & 0x1FFFFFFFu;
#pragma unroll
  for (int i = 0; i < 4; i++) fx[i] = mix(seed * 3 + i + 8191u * threadIdx.x) & 0x1FFFFFFFu;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const uint32_t f = fx[(i / RUN) % 4];
      if (MODE == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(op[i]), "v"(f) : "vcc");
      else asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(f), "v"(op[i]) : "vcc");
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

template <int MODE, int RUN> static void run(const char* name, uint32_t* d, int blocks, int cus) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 1 << 23;
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_mad<MODE, RUN>), dim3(blocks), dim3(256), 0, 0, d, 12345u, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double mads = (double)blocks * 256 * 16.0 * iters;
  printf("{\"instr\": \"%s\", \"iters\": %d, \"ms\": %.2f, \"lane_mad_per_s\": %.4g, \"cycles_per_wave_instr_per_simd_at_2.4GHz\": %.3f}\n", name, iters, ms, mads / (ms * 1e-3),
         (ms * 1e-3 * 2.4e9) / ((double)blocks * 4 * 16.0 * iters / (cus * 4.0)));
  fflush(stdout);
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * 8;
  uint32_t* d; CHECK(hipMalloc(&d, (size_t)blocks * 256 * 4));
  run<0, 16>("warm-up", d, blocks, p.multiProcessorCount);
  run<0, 16>("src0 varies, src1 fixed for runs of 16", d, blocks, p.multiProcessorCount);
  run<1, 16>("src0 fixed for runs of 16, src1 varies", d, blocks, p.multiProcessorCount);
  run<0, 4>("src0 varies, src1 fixed for runs of 4", d, blocks, p.multiProcessorCount);
  run<1, 4>("src0 fixed for runs of 4, src1 varies", d, blocks, p.multiProcessorCount);
  return 0;
}
