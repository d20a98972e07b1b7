// RECORD-ONLY feasibility probe (round-3 verdict, item 9): could the constant-operand half of a shared-key Montgomery product leave the VALU?
//
// Today a squaring sub-step issues 18.5 multiply-adds for X*X, 36 for M~*q (the reduction, 2/3 of the work) and ~9 bookkeeping
// instructions.  M~ is the same for every item of a shared-key launch, so the reduction can be written with separated operands —
//     T = X*X (VALU, tournament squaring, low limbs leave the array as they complete);   q = T_lo * N' mod R;   U = q * M;   result = T_hi + U_hi + carry
// — where q and U are products by CONSTANT operands: [items x 522 bytes] x Toeplitz(N') and x Toeplitz(M) are int8 GEMMs, 16 items of
// a wavefront as the 16-wide dimension of v_mfma_i32_16x16x64_i8.  Per wavefront-squaring that is ~275 MFMA instructions (2 x 522 x 522 / 2
// byte products x 16 items / 16384 per instruction), ~2 per sub-step, against 27.5 VALU instructions left in the sub-step, plus the
// conversions between 29-bit limbs and balanced 8-bit digits (re-chunking, carry resolution of 2 x 522 int32 column sums per item,
// a residue fix for the truncated high product): ~2000 wave-instructions per squaring by count, ~14 per sub-step.
//
// The two things a paper count cannot say, measured here with the engine's operand pattern (random 29-bit limbs, 36-column window,
// 2 wavefronts per SIMD, 256-thread workgroups), clock and board power sampled beside every kernel by tools/dev/mad_peak_with_clock.py:
//   (1) what a product-only sub-step costs once the 36 reduction multiply-adds are gone (the bookkeeping is then 1/3 of it);
//   (2) whether the MFMA pipe really runs BESIDE that VALU stream — issue slots, LDS reads of the B operand, and the 1.36 kW board
//       power limit that already holds the engine at 2.3 GHz.
// Shapes: the engine's squaring sub-step (baseline) | product-only | + the conversions' VALU ops | + 2 MFMA (+ their LDS reads) per sub-step.
// (valu_per_substep in the records = VALU instructions per sub-step counted in the compiled loop bodies; the MFMA variants carry 6-8 more
// for the synthetic B-operand addressing.)  `cycles_per_substep` of the last over the first is the ceiling of the speed-up of a squaring; nothing here is bit-exact arithmetic.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=200000 mfma_reduction_feasibility.hip -o mfma_reduction_feasibility
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
constexpr uint32_t MASK29 = 0x1FFFFFFFu;
typedef int v4i __attribute__((ext_vector_type(4)));

// (compiler-generated v_mad_u64_u32, as in the engine: inline-asm multiply-adds made the compiler pad every one of them with s_nop)
__device__ __forceinline__ void mad(uint64_t& acc, uint32_t a, uint32_t b) { acc += (uint64_t)a * b; }

// NQ: 1 = the sub-step keeps its 36 reduction multiply-adds (today's squaring), 0 = product only (the finished low limb leaves through LDS)
// EXTRA: cheap VALU instructions per sub-step standing in for the limb <-> digit conversions and carry resolution
// MFMA: v_mfma_i32_16x16x64_i8 per sub-step, each with a 16-byte B-operand read from LDS (the Toeplitz slice of the constant operand)
template <int NQ, int EXTRA, int MFMA>
__global__ void __launch_bounds__(256, 2) k_substep(uint32_t* out, uint32_t seed, int iters) {
  extern __shared__ __align__(16) uint32_t lds[];
  constexpr int W = 36;
  uint64_t acc[W];
  uint32_t A[W], N[W], x[8];
  v4i macc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  v4i amat = {(int)mix(seed + threadIdx.x), (int)mix(seed * 5 + threadIdx.x), (int)mix(seed * 9 + threadIdx.x), (int)mix(seed * 13 + threadIdx.x)};
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
#pragma unroll
  for (int i = 0; i < W; i++) { A[i] = mix(seed + 977u * i + 131071u * tid) & MASK29; N[i] = mix(seed * 7u + 31u * i + 8191u * tid) & MASK29; acc[i] = i; }
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = mix(seed * 11u + i + 127u * tid);
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = mix(seed + i) & MASK29;
  __syncthreads();
  uint32_t* mine = lds + 4096 + threadIdx.x * 4;                      // this lane's outgoing low limbs (product-only mode)
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int s = 0; s < W; s++) {
      // the limb of the staged B operand, as the engine reads it: one broadcast ds_read_b128 per 4 sub-steps
      const uint4 b4 = *reinterpret_cast<const uint4*>(lds + (((it * 9 + (s >> 2)) * 4) & 4092));
      const uint32_t b = (s & 3) == 0 ? b4.x : (s & 3) == 1 ? b4.y : (s & 3) == 2 ? b4.z : b4.w;
#pragma unroll
      for (int j = 0; j < 18; j++) mad(acc[(s + j) % W], A[(2 * j + (s & 1)) % W], b);      // the tournament's half of X * b
      if (NQ) {
        uint32_t q = (uint32_t)acc[s] & MASK29;
        q = (uint32_t)__builtin_amdgcn_mov_dpp((int)q, 0x00 /* quad_perm:[0,0,0,0] */, 0xF, 0xF, false);
#pragma unroll
        for (int j = 0; j < W; j++) mad(acc[(s + j) % W], N[j], q);
      }
      const uint32_t lo = (uint32_t)acc[s] & MASK29;
      acc[(s + 1) % W] += acc[s] >> 29;
      acc[s] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x101 /* row_shl:1 */, 0xF, 0xF, false);
      if (!NQ) { x[s & 3] ^= lo; if ((s & 3) == 3) *reinterpret_cast<uint4*>(mine) = make_uint4(x[0], x[1], x[2], x[3]); }   // T_lo leaves the array
#pragma unroll
      for (int e = 0; e < EXTRA; e++) {                                // shifts / masks / adds on digits: independent chains of 8 registers
        uint32_t& v = x[(e + s) & 7];
        v = (e & 1) ? (v >> 8) + x[(e + s + 3) & 7] : (v & 0x00FF00FFu) + (x[(e + s + 5) & 7] << 8);
      }
#pragma unroll
      for (int m = 0; m < MFMA; m++) {
        const uint4 bt = *reinterpret_cast<const uint4*>(lds + ((s * 64 + m * 16 + (threadIdx.x & 63) * 4 + it) & 4092));
        const v4i bmat = {(int)bt.x, (int)bt.y, (int)bt.z, (int)bt.w};
        macc[m & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(amat, bmat, macc[m & 1], 0, 0, 0);
      }
    }
  }
  uint64_t r = 0;
#pragma unroll
  for (int i = 0; i < W; i++) r ^= acc[i];
#pragma unroll
  for (int i = 0; i < 8; i++) r ^= x[i];
  r ^= (uint64_t)(uint32_t)(macc[0][0] + macc[0][1] + macc[0][2] + macc[0][3] + macc[1][0] + macc[1][1] + macc[1][2] + macc[1][3]);
  out[tid] = (uint32_t)r ^ (uint32_t)(r >> 32);
}

static uint32_t* d_out;
static int n_cu;
static hipEvent_t e0, e1;

template <typename K> static void timed(const char* name, K kernel, double mads_per_substep, double valu_per_substep, int mfma_per_substep, double target_s) {
  const int blocks = n_cu * 2;                      // 2 workgroups of 4 wavefronts per CU = 2 wavefronts per SIMD (the LDS size keeps it so)
  const size_t lds = 72 * 1024;
  CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int iters = 32;
  float ms = 0;
  for (int pass = 0; pass < 2; pass++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), lds, 0, d_out, 12345u, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipGetLastError());
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (pass == 0) iters = (int)(iters * target_s * 1e3 / ms) + 1;
  }
  const double substeps_per_simd = (double)blocks * 4 * 36.0 * iters / (n_cu * 4.0);        // wavefront sub-steps each SIMD ran
  const double mads = (double)blocks * 256 * 36.0 * iters * mads_per_substep;
  printf("{\"instr\": \"%s\", \"iters\": %d, \"ms\": %.2f, \"lane_mad_per_s\": %.5g, \"mads_per_substep\": %.1f, \"valu_per_substep\": %.1f, \"mfma_per_substep\": %d, "
         "\"cycles_per_substep_per_wave_at_2.4GHz\": %.1f, \"int8_mac_per_s\": %.4g}\n",
         name, iters, ms, mads / (ms * 1e-3), mads_per_substep, valu_per_substep, mfma_per_substep, ms * 1e-3 * 2.4e9 / substeps_per_simd,
         (double)blocks * 4 * 36.0 * iters * mfma_per_substep * 16384.0 / (ms * 1e-3));
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double T = argc > 1 ? atof(argv[1]) : 2.0;
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  n_cu = p.multiProcessorCount;
  CHECK(hipMalloc(&d_out, (size_t)n_cu * 2 * 256 * 4));
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  timed("warm-up", k_substep<1, 0, 0>, 54, 64, 0, T);
  timed("today: squaring sub-step, 18 + 36 multiply-adds + 10 bookkeeping", k_substep<1, 0, 0>, 54, 64, 0, T);
  timed("product only: 18 multiply-adds + 9 bookkeeping, low limb to LDS", k_substep<0, 0, 0>, 18, 27, 0, T);
  timed("product only + 14 conversion instructions", k_substep<0, 7, 0>, 18, 41, 0, T);
  timed("product only + 14 conversion instructions + 2 MFMA 16x16x64 i8 (+ 2 ds_read_b128)", k_substep<0, 7, 2>, 18, 49, 2, T);
  timed("product only + 28 conversion instructions + 2 MFMA (conversions twice the count)", k_substep<0, 14, 2>, 18, 63, 2, T);
  timed("2 MFMA per sub-step beside today's full sub-step (does the matrix pipe cost the VALU stream anything?)", k_substep<1, 0, 2>, 54, 75, 2, T);
  return 0;
}
