// What does a dependent hop cost a LONE wavefront on a SIMD?  (k_enc_basen_r2l5 — five wavefronts per Enc, csrc/kernels_basen_r2l.hpp — spends
// 218 cycles on the 24 instructions of a pair of sub-steps: the chain multiply-add -> v_readfirstlane -> scalar digit arithmetic ->
// multiply-add with the SGPR digit -> carries.)  Every kernel is one asm loop, one wavefront per workgroup, one workgroup per CU; the
// figure is ns per loop body (and cycles at the clock given on the command line, default 2.4 GHz).
//   hipcc -O3 --offload-arch=gfx950 lone_wave_hops.hip -o lone_wave_hops && ./lone_wave_hops [GHz [wavefronts per CU [iterations]]]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

#define CLOBBER "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23"
#define KERNEL(NAME, BODY)                                                                                               \
  __global__ void __launch_bounds__(64) NAME(uint32_t* out, uint32_t a, uint32_t b, int iters) {                         \
    uint32_t r;                                                                                                          \
    asm volatile("v_mov_b32 v2, %1\n v_mov_b32 v3, %2\n v_mov_b32 v4, 0x1fffffff\n v_mov_b32 v5, 0\n v_mov_b32 v10, %1\n v_mov_b32 v11, 0\n" \
                 "v_mov_b32 v12, %2\n v_mov_b32 v13, 0\n v_mov_b32 v14, %1\n v_mov_b32 v15, 0\n v_mov_b32 v16, %2\n v_mov_b32 v17, 0\n"       \
                 "v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n v_mov_b32 v20, %1\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"        \
                 "s_mov_b32 s20, %3\n s_mov_b32 s21, 5\n s_mov_b32 s22, 7\n s_mov_b32 s23, 0x1234567\n s_mov_b32 s24, 3\n s_mov_b32 s25, 1\n" \
                 "1:\n" BODY "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"                          \
                 "v_xor_b32 %0, v10, v12\n v_xor_b32 %0, %0, v14\n v_xor_b32 %0, %0, v16\n"                               \
                 : "=&v"(r) : "v"(a + threadIdx.x), "v"(b ^ threadIdx.x), "s"(iters) : CLOBBER);                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                                      \
  }
#define Y4(B) B B B B
#define Y8(B) Y4(B) Y4(B)
#define X4(B) Y8(Y4(B))      /* 32 copies per loop iteration: the three scalar instructions of the loop are < 3 % of it */
#define X8(B) Y8(Y8(B))      /* 64 copies */
#define MADC "v_mad_u64_u32 v[10:11], vcc, v2, v3, v[10:11]\n"
#define MADS "v_mad_u64_u32 v[10:11], vcc, v2, s21, v[10:11]\n"

KERNEL(k_mad_dep, X8(MADC))                                                     // 8 dependent multiply-adds
KERNEL(k_mad_indep, X4(MADC "v_mad_u64_u32 v[12:13], vcc, v2, v3, v[12:13]\n"))  // 8 multiply-adds, two chains
KERNEL(k_add_dep, X8("v_add_u32 v10, v10, v2\n"))                               // 8 dependent simple VALU
KERNEL(k_add_indep, X4("v_add_u32 v10, v10, v2\n v_add_u32 v12, v12, v2\n"))     // 8 simple VALU, two chains
KERNEL(k_add_indep4, X4("v_add_u32 v10, v10, v2\n v_add_u32 v12, v12, v2\n v_add_u32 v14, v14, v2\n v_add_u32 v16, v16, v2\n"))   // 16 simple VALU, four chains
KERNEL(k_salu_dep, X8("s_add_u32 s21, s21, s22\n"))                             // 8 dependent SALU
KERNEL(k_mad_add_mix, X4(MADC "v_add_u32 v12, v12, v2\n"))                      // 4 x (multiply-add, independent add)
KERNEL(k_mad_add2_mix, X4(MADC "v_add_u32 v12, v12, v2\n v_add_u32 v14, v14, v2\n"))   // 4 x (multiply-add, two independent adds)
// the q0 hop: multiply-add -> readfirstlane -> s_and -> multiply-add with the SGPR
KERNEL(k_hop_rfl_sand, X4(MADC "v_readfirstlane_b32 s21, v10\n s_and_b32 s21, s21, 0x1fffffff\n" MADS))
// ... the mask on the vector side
KERNEL(k_hop_vand_rfl, X4(MADC "v_and_b32 v12, v10, v4\n v_readfirstlane_b32 s21, v12\n" MADS))
// ... with fillers behind the readfirstlane (two independent multiply-adds on another accumulator)
KERNEL(k_hop_rfl_sand_fill, X4(MADC "v_readfirstlane_b32 s21, v10\n v_mad_u64_u32 v[14:15], vcc, v2, v3, v[14:15]\n s_and_b32 s21, s21, 0x1fffffff\n v_mad_u64_u32 v[16:17], vcc, v2, v3, v[16:17]\n" MADS))
// readfirstlane -> simple VALU that reads the SGPR
KERNEL(k_rfl_valu, X8("v_readfirstlane_b32 s21, v10\n v_add_u32 v10, s21, v10\n"))
// readfirstlane -> SALU -> simple VALU
KERNEL(k_rfl_salu_valu, X8("v_readfirstlane_b32 s21, v10\n s_add_u32 s21, s21, s22\n v_add_u32 v10, s21, v10\n"))
// readfirstlane -> 3 SALU (mul, add, and) -> simple VALU: the q1 arithmetic
KERNEL(k_rfl_salu3_valu, X8("v_readfirstlane_b32 s21, v10\n s_mul_i32 s21, s21, s24\n s_add_u32 s21, s21, s22\n s_and_b32 s21, s21, 0x1fffffff\n v_add_u32 v10, s21, v10\n"))
// the carry hop: 64-bit shift + 64-bit add
KERNEL(k_carry, X8("v_lshrrev_b64 v[12:13], 29, v[10:11]\n v_lshl_add_u64 v[10:11], v[10:11], 0, v[12:13]\n"))
// DPP move of a register the previous VALU instruction wrote (two wait states)
KERNEL(k_and_dpp, X8("v_add_u32 v10, v10, v2\n s_nop 1\n v_and_b32_dpp v10, v10, v4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"))
// the LDS crossbar hop the one-wavefront kernel pays: ds_bpermute + wait
KERNEL(k_bpermute, X8("ds_bpermute_b32 v10, v5, v10\n s_waitcnt lgkmcnt(0)\n v_add_u32 v10, v10, v2\n"))
// v_mul_lo_u32 dependent chain (the vector-side alternative for q0 * (N1 + 1))
KERNEL(k_mul_lo_dep, X8("v_mul_lo_u32 v10, v10, v3\n"))

int main(int argc, char** argv) {
  const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int per_cu = argc > 2 ? atoi(argv[2]) : 1;      // workgroups (= wavefronts) per CU: 1 = a quarter of the SIMDs busy, 4 = one wavefront per SIMD
  const int blocks = p.multiProcessorCount * per_cu;
  uint32_t* d; CHECK(hipMalloc(&d, (size_t)blocks * 64 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = argc > 3 ? atoi(argv[3]) : 1 << 16;
  struct { const char* name; void (*k)(uint32_t*, uint32_t, uint32_t, int); int units; const char* unit; } ks[] = {
    {"8 dependent v_mad_u64_u32", k_mad_dep, 64, "multiply-add"}, {"8 v_mad_u64_u32 on two chains", k_mad_indep, 64, "multiply-add"},
    {"8 dependent v_add_u32", k_add_dep, 64, "add"}, {"8 v_add_u32 on two chains", k_add_indep, 64, "add"}, {"16 v_add_u32 on four chains", k_add_indep4, 128, "add"},
    {"8 dependent s_add_u32", k_salu_dep, 64, "s_add"},
    {"4 x (multiply-add, independent add)", k_mad_add_mix, 32, "group"}, {"4 x (multiply-add, two independent adds)", k_mad_add2_mix, 32, "group"},
    {"4 x (mad -> readfirstlane -> s_and -> mad with the SGPR)", k_hop_rfl_sand, 32, "group of 2 mads + hop"},
    {"4 x (mad -> v_and -> readfirstlane -> mad with the SGPR)", k_hop_vand_rfl, 32, "group of 2 mads + hop"},
    {"4 x (mad -> readfirstlane, mad', s_and, mad'' -> mad with the SGPR)", k_hop_rfl_sand_fill, 32, "group of 4 mads + hop"},
    {"8 x (readfirstlane -> v_add reading the SGPR)", k_rfl_valu, 64, "pair"}, {"8 x (readfirstlane -> s_add -> v_add)", k_rfl_salu_valu, 64, "triple"},
    {"8 x (readfirstlane -> s_mul, s_add, s_and -> v_add)", k_rfl_salu3_valu, 64, "group of 5"},
    {"8 x (v_lshrrev_b64 -> v_lshl_add_u64)", k_carry, 64, "pair"}, {"8 x (v_add -> s_nop 1 -> v_and_b32_dpp)", k_and_dpp, 64, "pair"},
    {"8 x (ds_bpermute -> wait -> v_add)", k_bpermute, 64, "pair"}, {"8 dependent v_mul_lo_u32", k_mul_lo_dep, 64, "multiply"}};
  for (auto& k : ks) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(64), 0, 0, d, 0x0EADBEEFu, 0x00FFEE11u, iters);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best) best = ms;
    }
    const double ns = best * 1e6 / iters / k.units;
    printf("{\"pattern\": \"%s\", \"ms\": %.2f, \"ns_per_%s\": %.2f, \"cycles_at_%.2fGHz\": %.1f}\n", k.name, best, k.unit, ns, ghz, ns * ghz);
    fflush(stdout);
  }
  return 0;
}
