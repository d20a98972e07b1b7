// Does the issue rate of v_mad_u64_u32 depend on the VGPR banks of its operands?  (4.68 cycles per wavefront instruction per SIMD is
// what every loop of this repository measures; 4 would be one pass of a 64-lane wavefront over a 16-lane SIMD.)
// The whole loop is one asm block with hand-chosen physical registers: 16 accumulators v[32+4i : 33+4i] (register index = 0, 1 mod 4)
// and the two multiplicands in registers whose index mod 4 is chosen per pattern.
//   hipcc -O3 --offload-arch=gfx950 mad_banks.hip -o mad_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

#define MAD(i, A, B) "v_mad_u64_u32 v[" #i ":" #i "+1], vcc, " A ", " B ", v[" #i ":" #i "+1]\n"
#define MAD16(A, B) MAD(32, A, B) MAD(36, A, B) MAD(40, A, B) MAD(44, A, B) MAD(48, A, B) MAD(52, A, B) MAD(56, A, B) MAD(60, A, B) \
                    MAD(64, A, B) MAD(68, A, B) MAD(72, A, B) MAD(76, A, B) MAD(80, A, B) MAD(84, A, B) MAD(88, A, B) MAD(92, A, B)
// accumulators of both kinds of pair: v[32:33] (0,1 mod 4), v[38:39] (2,3 mod 4), ...
#define MAD16ALT(A, B) MAD(32, A, B) MAD(38, A, B) MAD(40, A, B) MAD(46, A, B) MAD(48, A, B) MAD(54, A, B) MAD(56, A, B) MAD(62, A, B) \
                       MAD(64, A, B) MAD(70, A, B) MAD(72, A, B) MAD(78, A, B) MAD(80, A, B) MAD(86, A, B) MAD(88, A, B) MAD(94, A, B)
#define CLOBBER "vcc", "s20", "v2", "v3", "v4", "v5", "v6", "v7", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", \
  "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", \
  "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95"

#define KERNEL(NAME, BODY)                                                                                               \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t a, uint32_t b, int iters) {                        \
    uint32_t r;                                                                                                          \
    asm volatile("v_mov_b32 v2, %1\n v_mov_b32 v3, %2\n v_mov_b32 v4, %1\n v_mov_b32 v5, %2\n v_mov_b32 v6, %1\n v_mov_b32 v7, %2\n" \
                 "s_mov_b32 s20, %3\n"                                                                                   \
                 "1:\n" BODY "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"                          \
                 "v_xor_b32 %0, v32, v36\n"                                                                               \
                 : "=v"(r) : "v"(a + threadIdx.x), "v"(b ^ threadIdx.x), "s"(iters) : CLOBBER);                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                                      \
  }
// accumulators at 0,1 mod 4:
KERNEL(k_a2_b3, MAD16("v2", "v3"))          // multiplicands at 2 and 3 mod 4: four different residues
KERNEL(k_a4_b5, MAD16("v4", "v5"))          // ... at 0 and 1 mod 4: both collide with the accumulator pair
KERNEL(k_a4_b3, MAD16("v4", "v3"))          // one collides
KERNEL(k_a2_b6, MAD16("v2", "v6"))          // the two multiplicands collide with each other (2, 2 mod 4)
KERNEL(k_alt_a2_b3, MAD16ALT("v2", "v3"))   // accumulators alternate between the two kinds of pair
KERNEL(k_a2_a2, MAD16("v2", "v2"))          // the same register twice

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * 8;
  uint32_t* d; CHECK(hipMalloc(&d, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 1 << 22;
  struct { const char* name; void (*k)(uint32_t*, uint32_t, uint32_t, int); } ks[] = {
    {"acc 0,1 | a 2 | b 3 (mod 4)", k_a2_b3}, {"acc 0,1 | a 0 | b 1", k_a4_b5}, {"acc 0,1 | a 0 | b 3", k_a4_b3}, {"acc 0,1 | a 2 | b 2", k_a2_b6},
    {"acc 0,1 and 2,3 alternating | a 2 | b 3", k_alt_a2_b3}, {"acc 0,1 | a 2 | a 2 (same register)", k_a2_a2}};
  // ... and on the MAGNITUDE of the multiplicands?  (the same kernel, values of 13 ... 32 bits, different in every lane)
  for (int bits : {13, 24, 29, 32}) {
    const uint32_t mask = bits == 32 ? 0xFFFFFFFFu : ((1u << bits) - 1);
    for (int rep = 0; rep < 2; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_a2_b3, dim3(blocks), dim3(256), 0, 0, d, 0xDEADBEEFu & mask & ~0xFFu, 0xC0FFEE11u & mask & ~0xFFu, iters);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double mads = (double)blocks * 256 * 16.0 * iters;
      if (rep) printf("{\"pattern\": \"acc 0,1 | a 2 | b 3, multiplicands of %d bits\", \"ms\": %.2f, \"lane_mad_per_s\": %.4g, \"cycles_per_wave_instr_per_simd_at_2.4GHz\": %.3f}\n", bits, ms,
                      mads / (ms * 1e-3), (ms * 1e-3 * 2.4e9) / ((double)blocks * 4 * 16.0 * iters / (p.multiProcessorCount * 4.0)));
      fflush(stdout);
    }
  }
  for (auto& e : ks) {
    for (int rep = 0; rep < 2; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 0x1234567u, 0x0FEDCBAu, iters);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double mads = (double)blocks * 256 * 16.0 * iters;
      if (rep) printf("{\"pattern\": \"%s\", \"ms\": %.2f, \"lane_mad_per_s\": %.4g, \"cycles_per_wave_instr_per_simd_at_2.4GHz\": %.3f}\n", e.name, ms, mads / (ms * 1e-3),
                      (ms * 1e-3 * 2.4e9) / ((double)blocks * 4 * 16.0 * iters / (p.multiProcessorCount * 4.0)));
      fflush(stdout);
    }
  }
  return 0;
}
