// What is the ceiling of v_mad_u64_u32 on gfx950 (MI355X), and why did csrc/microbench/mad_sustained.hip stop at 4.68 cycles per
// wavefront instruction per SIMD while the engine's own instruction mix averages 4.00 (profiles/r03_pmc_final_enc2048_shared_b4096.json)?
// A kernel cannot beat its roofline, so the round-3 denominator (3.36e13 lane-MAD/s) was not one.  This file measures the multiply-add in
// the shapes that could explain the gap, one JSON line per shape (run under tools/dev/mad_peak_with_clock.py, which adds the sampled clock):
//   pure loops    : carry-out to VCC (what mad_sustained did) or to an SGPR pair (what the compiler emits in the engine: s[6:7], s[8:9]);
//                   16 accumulators in a 16-instruction loop (mad_sustained) or 72 accumulators in a 288-instruction loop;
//                   8 / 4 / 2 / 1 wavefronts per SIMD (the engine: 2, all 256 VGPRs)
//   engine-like   : the sub-step of bigint29.hpp:montmul without its data dependencies on real values — a 36-column circular window of
//                   64-bit accumulators, 36 multiply-adds by one limb, 36 by another, random 29-bit operands — with and without the
//                   7 bookkeeping instructions of a sub-step (2 masks, 2 DPP moves, 64-bit shift, 64-bit add: counted in the ISA); and the 54.5-multiply-add squaring sub-step
//                   shape (18 / 19 + 36)
// The hardware figure these are held against: a SIMD retires 16 lanes of a 32-bit multiply per clock = 4 cycles per wave64 instruction.
//   hipcc -O3 --offload-arch=gfx950 mad_issue_ceiling.hip -o mad_issue_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
constexpr uint32_t MASK29 = 0x1FFFFFFFu;

template <int SDST> __device__ __forceinline__ void mad(uint64_t& acc, uint32_t a, uint32_t b) {
  if (SDST == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
  else asm volatile("v_mad_u64_u32 %0, s[6:7], %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "s6", "s7");
}

// ---------------------------------------------------------------- pure multiply-add loops
// NACC independent accumulators, the loop body holds REP * NACC instructions; operands: one constant pair (mad_sustained's shape)
template <int SDST, int NACC, int REP>
__global__ void __launch_bounds__(256) k_pure(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {
  extern __shared__ uint32_t pad[];
  uint64_t acc[NACC];
  uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = i + threadIdx.x;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < REP; r++)
#pragma unroll
      for (int i = 0; i < NACC; i++) mad<SDST>(acc[i], a, b);
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < NACC; i++) s ^= acc[i];
  if (iters < 0) pad[threadIdx.x] = (uint32_t)s;        // keeps the dynamic LDS (the occupancy limiter) allocated
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

// ---------------------------------------------------------------- the engine's sub-step shape
// BOOK = 0: multiply-adds only (limbs b and q of a sub-step are register values); BOOK = 1: + the bookkeeping of montmul's sub-step:
// q = low 29 bits of the bottom column broadcast over the quad (v_and + DPP move), bottom column finished (shift, add into the next
// column), its low limb handed to the neighbour lane (DPP row_shl:1) as the fresh top column.
// MADS_A = multiply-adds of the A half per sub-step: 36 (product) or 18 (squaring: the tournament keeps half of them)
template <int BOOK, int MADS_A>
__global__ void __launch_bounds__(256, 2) k_substep(uint32_t* out, uint32_t seed, uint32_t, int iters) {
  extern __shared__ uint32_t pad[];
  constexpr int W = 36;
  uint64_t acc[W];
  uint32_t A[W], N[W], bq[8];
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
#pragma unroll
  for (int i = 0; i < W; i++) { A[i] = mix(seed + 977u * i + 131071u * tid) & MASK29; N[i] = mix(seed * 7u + 31u * i + 8191u * tid) & MASK29; acc[i] = i; }
#pragma unroll
  for (int i = 0; i < 8; i++) bq[i] = mix(seed * 3u + i + 524287u * tid) & MASK29;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int s = 0; s < W; s++) {
      const uint32_t b = bq[s & 3];
#pragma unroll
      for (int j = 0; j < MADS_A; j++) mad<1>(acc[(s + j) % W], A[(j * (W / MADS_A) + (s & (W / MADS_A - 1))) % W], b);
      uint32_t q;
      if (BOOK) {
        q = (uint32_t)acc[s] & MASK29;
        q = (uint32_t)__builtin_amdgcn_mov_dpp((int)q, 0x00 /* quad_perm:[0,0,0,0] */, 0xF, 0xF, false);
      } else q = bq[4 + (s & 3)];
#pragma unroll
      for (int j = 0; j < W; j++) mad<1>(acc[(s + j) % W], N[j], q);
      if (BOOK) {
        const uint32_t lo = (uint32_t)acc[s] & MASK29;
        acc[(s + 1) % W] += acc[s] >> 29;
        acc[s] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x101 /* row_shl:1 */, 0xF, 0xF, false);
      }
    }
  }
  uint64_t r = 0;
#pragma unroll
  for (int i = 0; i < W; i++) r ^= acc[i];
  if (iters < 0) pad[threadIdx.x] = (uint32_t)r;
  out[tid] = (uint32_t)r ^ (uint32_t)(r >> 32);
}

static uint32_t* d_out;
static int n_cu;
static hipEvent_t e0, e1;

template <typename K> static void timed(const char* name, K kernel, int waves_per_simd, double mads_per_thread_iter, double valu_per_thread_iter, double target_s, uint32_t a, uint32_t b) {
  // waves per SIMD = blocks per CU (a block is 4 wavefronts, one per SIMD); dynamic LDS keeps the dispatcher from stacking more blocks on a CU
  const int blocks = n_cu * waves_per_simd;
  const size_t lds = waves_per_simd >= 8 ? 0 : (size_t)(160 * 1024 / waves_per_simd) - 1024;
  CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int occ = 0; CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 256, lds));
  int iters = 64;
  float ms = 0;
  for (int pass = 0; pass < 2; pass++) {          // pass 0 sizes the run (and warms the clock up), pass 1 is the record
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), lds, 0, d_out, a, (uint32_t)b, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipGetLastError());
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (pass == 0) iters = (int)(iters * target_s * 1e3 / ms) + 1;
  }
  const double mads = (double)blocks * 256 * mads_per_thread_iter * iters, valu = (double)blocks * 256 * valu_per_thread_iter * iters;
  const double wave_instr_per_simd = mads / 64.0 / (n_cu * 4.0), wave_valu_per_simd = valu / 64.0 / (n_cu * 4.0);
  printf("{\"instr\": \"%s\", \"waves_per_simd\": %d, \"occupancy_blocks_per_cu\": %d, \"iters\": %d, \"ms\": %.2f, \"lane_mad_per_s\": %.5g, "
         "\"cycles_per_mad_per_simd_at_2.4GHz\": %.3f, \"valu_instr_per_mad\": %.4f, \"cycles_per_valu_instr_per_simd_at_2.4GHz\": %.3f}\n",
         name, waves_per_simd, occ, iters, ms, mads / (ms * 1e-3), ms * 1e-3 * 2.4e9 / wave_instr_per_simd, valu / mads, ms * 1e-3 * 2.4e9 / wave_valu_per_simd);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double T = argc > 1 ? atof(argv[1]) : 1.0;        // seconds per record
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  n_cu = p.multiProcessorCount;
  CHECK(hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * 4));
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  timed("warm-up", k_pure<0, 16, 1>, 8, 16, 16, T, 12345u, 6789u);
  // mad_sustained's own shape and its variations
  timed("pure vcc 16acc loop16", k_pure<0, 16, 1>, 8, 16, 16, T, 12345u, 6789u);
  timed("pure sgpr 16acc loop16", k_pure<1, 16, 1>, 8, 16, 16, T, 12345u, 6789u);
  timed("pure vcc 16acc loop256", k_pure<0, 16, 16>, 8, 256, 256, T, 12345u, 6789u);
  timed("pure sgpr 16acc loop256", k_pure<1, 16, 16>, 8, 256, 256, T, 12345u, 6789u);
  timed("pure sgpr 16acc loop256", k_pure<1, 16, 16>, 4, 256, 256, T, 12345u, 6789u);
  timed("pure sgpr 16acc loop256", k_pure<1, 16, 16>, 2, 256, 256, T, 12345u, 6789u);
  timed("pure sgpr 16acc loop256", k_pure<1, 16, 16>, 1, 256, 256, T, 12345u, 6789u);
  timed("pure vcc 72acc loop288", k_pure<0, 72, 4>, 2, 288, 288, T, 12345u, 6789u);
  timed("pure sgpr 72acc loop288", k_pure<1, 72, 4>, 2, 288, 288, T, 12345u, 6789u);
  timed("pure sgpr 72acc loop288", k_pure<1, 72, 4>, 1, 288, 288, T, 12345u, 6789u);
  timed("pure sgpr 72acc loop2304", k_pure<1, 72, 32>, 2, 2304, 2304, T, 12345u, 6789u);
  // the engine's sub-step
  timed("substep product 72 mads, no bookkeeping", k_substep<0, 36>, 2, 36 * 72, 36 * 72, T, 12345u, 0);
  timed("substep product 72 mads + 7 bookkeeping", k_substep<1, 36>, 2, 36 * 72, 36 * 79, T, 12345u, 0);
  timed("substep squaring 54 mads, no bookkeeping", k_substep<0, 18>, 2, 36 * 54, 36 * 54, T, 12345u, 0);
  timed("substep squaring 54 mads + 7 bookkeeping", k_substep<1, 18>, 2, 36 * 54, 36 * 61, T, 12345u, 0);
  timed("substep product 72 mads + 7 bookkeeping", k_substep<1, 36>, 1, 36 * 72, 36 * 79, T, 12345u, 0);
  return 0;
}
