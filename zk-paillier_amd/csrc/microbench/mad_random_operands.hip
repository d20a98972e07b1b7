// The multiply-add roofline for the operands the engine really has: mad_sustained.hip multiplies ONE pair of values (a, b) in every
// lane forever — the multiplier inputs never toggle; the engine's products multiply random 29-bit limbs, a new pair every instruction.
// Here every lane holds 32 random 29-bit operands and consecutive v_mad_u64_u32 take different pairs of them.  Same issue rate per
// clock; what changes is the POWER the pipe draws and therefore the clock the board holds (tools/dev/mad_peak_with_clock.py samples it).
//   hipcc -O3 --offload-arch=gfx950 mad_random_operands.hip -o mad_random_operands
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ void __launch_bounds__(256) k_mad(uint32_t* out, uint32_t seed, int iters) {
  uint64_t acc[16];
  uint32_t op[32];
#pragma unroll
  for (int i = 0; i < 32; i++) op[i] = mix(seed + 977u * i + 131071u * (blockIdx.x * 256u + threadIdx.x)) & 0x1FFFFFFFu;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++)      // pair (i, 16 + (5 i + 3) mod 16): every instruction a different pair of registers
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(op[i]), "v"(op[16 + (5 * i + 3) % 16]) : "vcc");
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * 8;          // 8 waves per SIMD
  uint32_t* d; CHECK(hipMalloc(&d, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int iters : {1 << 19, 1 << 22, 1 << 24}) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double mads = (double)blocks * 256 * 16.0 * iters;
    printf("{\"instr\": \"v_mad_u64_u32, random 29-bit operands\", \"iters\": %d, \"ms\": %.2f, \"lane_mad_per_s\": %.4g, \"cycles_per_wave_instr_per_simd_at_2.4GHz\": %.3f}\n",
           iters, ms, mads / (ms * 1e-3), (ms * 1e-3 * 2.4e9) / ((double)blocks * 4 * 16.0 * iters / (p.multiProcessorCount * 4.0)));
    fflush(stdout);
  }
  return 0;
}
