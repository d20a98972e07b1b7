// kernels_basen_r2l.hpp — ONE Paillier Enc per wavefront, as short a chain as the base-n form allows (latency engine, 9 limbs per lane, n of 2048 bits).
//
// A call of one proof is 192 - 256 independent Enc under one key: the chip has four SIMDs for every one of them, and what the caller waits
// for is the chain of dependent products inside a single Enc.  On the n^2-sized kernels that chain is 2048 squarings of 144 sub-steps
// (the pair ladder of kernels_modexp.hpp: 17.3 ms).  In base-n form (kernels_basen.hpp: x = a + b n) a squaring is TWO n-sized products
// of 72 sub-steps — a' = a^2 / R' with quotient digits Q, b' = (2 a b - Q n1) / R' — and they need not wait for each other: the a chain
// never reads a b part, so the b side of step k can run ONE PRODUCT BEHIND the a side of step k + 1, on other lanes of the same wavefront.
// The same holds for the accumulator of a right-to-left ladder: acc <- acc * s_k is p' = p a_k (digits Q'), q' = (p b_k - Q' n1) / R' + q a_k / R',
// again an a chain that runs ahead and two b-side products one step behind.  So the exponentiation x^n becomes a pipeline of FIVE lane groups
// of 8 lanes (72 limbs) each, all executing the same n-sized product body in lockstep, one SLOT per ladder step:
//
//     slot k:   A  a_(k+1) = a_k a_k / R'                                   digits Q_k   over the staged a_k
//               B  b_k     = (2 a_(k-1) b_(k-1) - Q_(k-1) n1) / R'                                                  (one slot behind A)
//               C  p_(k+1) = p_k s / R',  s = a_k (bit k of n set) | R' mod n  digits Q'_k  over the staged s
//               D  (p_(k-1) t - Q'_(k-1) n1) / R',  t = b_(k-1) | floor(R' / n)                                       (one slot behind C)
//               E  q_(k-1) s' / R',  s' = a_(k-1) | R' mod n;        q_k = D + E                                      (one slot behind C)
//
// (a clear bit multiplies the accumulator by the Montgomery form of 1 — the same value, the same slot, no divergence: the exponent is the
// launch's key, every wavefront walks the same bits).  2048 + 4 slots of 72 sub-steps instead of 2048 products of 144: half the chain.
// There is no window table and no schedule; everything an Enc touches between its operands and its raw pair lives in 8 KB of LDS.
// The raw pair goes to k_basen_finish like that of k_enc_basen.  The kernel carries its own lane geometry (template parameter RW, limbs per
// lane): 9 — five groups of 8 lanes, the engine's own — or 6 — five groups of 12 lanes: a lone wavefront on a SIMD issues a multiply-add only
// every 8 cycles whatever it depends on, so what its chain costs is the NUMBER of instructions, and 6 limbs per lane are 12 multiply-adds per
// sub-step where 9 are 18 (the 60 lanes of five 12-lane groups straddle the 16-lane DPP rows: neighbours by wave_shl / wave_shr, the quotient
// digit by ds_bpermute).  40 / 60 of the 64 lanes work: this is the kernel of calls that leave the chip
// idle anyway (up to 2 wavefronts per SIMD: 8 proofs); from there on k_enc_basen<8> (8 Enc per wavefront) and the throughput engine take over.
//
// Replaces, like k_enc: kzen-paillier EncryptWithChosenRandomness at range_proof.rs:165-169,179-183,280-291,330-334 — for the call shape
// of the reference's own benchmark, ONE RangeProofNi proof (benches/all.rs:55-71).
#pragma once
#include <utility>
#include "kernels_basen.hpp"

#ifndef ZKP_R2L_PRIO
#define ZKP_R2L_PRIO 3
#endif
namespace zkp {

#if ZKP_W == 9

#ifndef ZKP_R2L_UNROLL_ROWS
#define ZKP_R2L_UNROLL_ROWS 12      /* all of them: one 1024-Enc launch 10.70 (one row per iteration) -> 9.82 (3) -> 9.70 (4) -> 9.47 ms (profiles/r05/r2l5/ab_row_loop_unrolled_one_wavefront_kernel.jsonl) */
#endif

namespace r2l {
// LDS areas of a wavefront (word offsets / AW)
enum Area {
  SA0, SA1,      // A's staged a_k, then its quotient digits Q_k (by slot parity)
  SC0, SC1,      // C's staged multiplier (a_k or the Montgomery one), then Q'_k
  SE0, SE1,      // E's staged a_k (read one slot later)
  DA0, DA1,      // what B multiplies by: 2 a_k (the item's r for the step into the Montgomery domain)
  SB,            // the staged b_k; B's result goes back into it
  PC0, PC1,      // p_k (C's register operand; D reads it one slot later)
  PX,            // copy of the last p (the final product stages it)
  RD,            // D's result of the slot
  QQ,            // q_k
  UU,            // m * p / R' of the final product
  C3A, ZERO, ONEA, ONEB, INT1, MM,      // constants: C3, 0, R' mod n, floor(R' / n), the integer 1, the item's m
  DUM0, DUM1, DUM2, DUM3, DUM4, DUM5, DUM6, DUM7,      // one scratch area per group (idle roles compute into them)
  WBUF,          // 32-bit words on their way to limbs (the item's r, m)
  RL,            // the item's r as limbs: A's register operand of the step into the Montgomery domain
  NAREAS
};
constexpr int LIMBS = 72;                  // an n-sized integer (n of 2048 bits)

// lane geometry: RW limbs per lane, RG = 72 / RW lanes per integer, lane blocks of RBLK words in LDS
template <int RW> struct Geom {
  static_assert(RW == 9 || RW == 6, "five groups of 8 or of 12 lanes");
  static constexpr int RG = LIMBS / RW;
  static constexpr int RBLK = (RW + 3) & ~3;
  static constexpr int AW = RG * RBLK;                       // words per area: 96 either way
  static constexpr int LDS_WORDS = NAREAS * AW;
};

template <int RW> __device__ __forceinline__ void blk_load(uint32_t (&v)[RW], const uint32_t* p) {
#pragma unroll
  for (int i = 0; i + 4 <= RW; i += 4) {
    const uint4 a = *reinterpret_cast<const uint4*>(p + i);
    v[i] = a.x; v[i + 1] = a.y; v[i + 2] = a.z; v[i + 3] = a.w;
  }
  if constexpr (RW % 4 == 1) v[RW - 1] = p[RW - 1];
  else if constexpr (RW % 4 == 2) {
    const uint2 a = *reinterpret_cast<const uint2*>(p + RW - 2);
    v[RW - 2] = a.x; v[RW - 1] = a.y;
  }
}
template <int RW> __device__ __forceinline__ void blk_store(uint32_t* p, const uint32_t (&v)[RW]) {
#pragma unroll
  for (int i = 0; i + 4 <= RW; i += 4) *reinterpret_cast<uint4*>(p + i) = make_uint4(v[i], v[i + 1], v[i + 2], v[i + 3]);
  if constexpr (RW % 4 == 1) p[RW - 1] = v[RW - 1];
  else if constexpr (RW % 4 == 2) *reinterpret_cast<uint2*>(p + RW - 2) = make_uint2(v[RW - 2], v[RW - 1]);
}
template <int RW> __device__ __forceinline__ void limbs_global_load(uint32_t (&v)[RW], const uint32_t* p, int gl) {
#pragma unroll
  for (int k = 0; k < RW; k++) v[k] = p[gl * RW + k];
}
template <int RW> __device__ __forceinline__ void limbs_global_store(uint32_t* p, const uint32_t (&v)[RW], int gl) {
#pragma unroll
  for (int k = 0; k < RW; k++) p[gl * RW + k] = v[k];
}
template <int RW> __device__ __forceinline__ void limbs_from_words(uint32_t (&v)[RW], const uint32_t* words, int gl) {
#pragma unroll
  for (int k = 0; k < RW; k++) {
    const int bit = (gl * RW + k) * LB;
    const int w0 = bit >> 5, off = bit & 31;
    const uint64_t x = (uint64_t)words[w0] | ((uint64_t)words[w0 + 1] << 32);
    v[k] = (uint32_t)(x >> off) & LMASK;
  }
}
// lane j <- lane j + 1 (the top lane of a group receives the next group's lane 0: the low limb of a bottom column the quotient digit has
// just made zero — idle lanes compute on zeros —, as in bigint29.hpp from_next).  12-lane groups cross the 16-lane DPP rows: wave_shl.
template <int RW> __device__ __forceinline__ uint32_t next_lane(uint32_t v) {
  if constexpr (RW == 9) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101 /*row_shl:1*/, 0xf, 0xf, true);
  else return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
}
template <int RW> __device__ __forceinline__ uint32_t prev_lane(uint32_t v, int gl) {
  uint32_t t;
  if constexpr (RW == 9) t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
  else t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
  return gl == 0 ? 0u : t;
}
// lane 0 of the group -> every lane of the group.  lead4: byte address of the group's lane 0 for ds_bpermute (12-lane groups)
template <int RW> __device__ __forceinline__ uint32_t lead_bcast(uint32_t v, int lead4) {
  if constexpr (RW == 9) { (void)lead4; return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x0018); }
  else return (uint32_t)__builtin_amdgcn_ds_bpermute(lead4, (int)v);
}
template <int RW> __device__ __forceinline__ void dbl(uint32_t (&X)[RW], int gl) {
  uint32_t cy = 0;
#pragma unroll
  for (int k = 0; k < RW; k++) { const uint32_t t = (X[k] << 1) + cy; X[k] = t & LMASK; cy = t >> LB; }
  X[0] += prev_lane<RW>(cy, gl);
}
template <int RW> __device__ __forceinline__ void add(uint32_t (&R)[RW], const uint32_t (&A)[RW], const uint32_t (&B)[RW], int gl) {
  uint32_t cy = 0;
#pragma unroll
  for (int k = 0; k < RW; k++) { const uint32_t t = A[k] + B[k] + cy; R[k] = t & LMASK; cy = t >> LB; }
  R[0] += prev_lane<RW>(cy, gl);
}
// R = (X * B + c0 + q M~) / R' over the group's lanes; digits go over the consumed limbs of B in the lanes of `qmask`
// (bn_mul_impl of kernels_basen.hpp with everything that varies turned into data)
template <int RW>
__device__ __forceinline__ void product(uint32_t (&R)[RW], const uint32_t (&X)[RW], const uint32_t* ldsB, const uint32_t (&NT)[RW], uint64_t (&c)[RW],
                                        uint64_t qmask, int gl, int lead4) {
  using GM = Geom<RW>;
#pragma unroll 1
  for (int s = 0; s < GM::RG; s++) {
    uint32_t qd[4];
    const uint32_t row_addr = lds_byte_address(ldsB + s * GM::RBLK);
#pragma unroll
    for (int t = 0; t < RW; t++) {
      const uint32_t b = ldsB[s * GM::RBLK + t];
#pragma unroll
      for (int k = 0; k < RW; k++) c[(t + k) % RW] += (uint64_t)X[k] * b;
      const uint32_t q = lead_bcast<RW>((uint32_t)c[t] & LMASK, lead4);
      qd[t & 3] = q;
      if ((t & 3) == 3) q_write(qmask, row_addr + (t - 3) * 4, qd);
      else if (t == RW - 1 && (RW & 3) == 1) q_write1(qmask, row_addr + t * 4, qd[t & 3]);
      else if (t == RW - 1 && (RW & 3) == 2) q_write2(qmask, row_addr + (t - 1) * 4, qd[(t - 1) & 3], qd[t & 3]);
#pragma unroll
      for (int k = 0; k < RW; k++) c[(t + k) % RW] += (uint64_t)NT[k] * q;
      const uint64_t v = c[t];
      c[(t + 1) % RW] += v >> LB;
      c[t] = (uint64_t)next_lane<RW>((uint32_t)v & LMASK);
    }
  }
  uint64_t cy = 0;
#pragma unroll
  for (int k = 0; k < RW; k++) {
    const uint64_t t = c[k] + cy;
    R[k] = (uint32_t)t & LMASK;
    cy = t >> LB;
  }
  R[0] += prev_lane<RW>((uint32_t)cy, gl);
}
// The same product with TWO quotient digits per trip through the broadcast (even RW).  A lone wavefront waits out the LDS crossbar once per
// digit: 195 cycles per sub-step whether a lane holds 6 limbs or 9.  Lane 0 of a group can work out the NEXT digit by itself — its bottom two
// columns are complete, and with M~ == -1 (mod 2^29) the carry out of the bottom column is (c0 >> 29) + q0, so
//     q1 = low29( c1 + (c0 >> 29) + q0 (N_1 + 1) )
// — one 32-bit multiply more on lane 0's chain, one broadcast latency less per pair of sub-steps (both digits go out together).  The digits,
// and with them every value, are those of `product` (bigint29.hpp montmul2 does this under a 58-bit Orup multiple, which a 2048-bit n in
// 2088 bits of capacity has no room for; this variant needs none).
template <int RW>
__device__ __forceinline__ void product2(uint32_t (&R)[RW], const uint32_t (&X)[RW], const uint32_t* ldsB, const uint32_t (&NT)[RW], uint64_t (&c)[RW],
                                         uint64_t qmask, int gl, int lead4) {
  static_assert((RW & 1) == 0, "pairs of sub-steps");
  using GM = Geom<RW>;
  const uint32_t n1p = NT[1] + 1;
  // (the rows of the staged operand: a taken branch costs a lone wavefront ~20 ns — csrc/microbench/lone_wave_hops.hip —, twelve of them 5 % of a
  // product of 12 lanes x 6 limbs; ZKP_R2L_UNROLL_ROWS copies per loop iteration: A/B switch)
  ZKP_UNROLL(ZKP_R2L_UNROLL_ROWS)
  for (int s = 0; s < GM::RG; s++) {
    uint32_t qd[4];
    const uint32_t row_addr = lds_byte_address(ldsB + s * GM::RBLK);
#pragma unroll
    for (int t = 0; t < RW; t += 2) {
      const int i0 = t % RW, i1 = (t + 1) % RW, i2 = (t + 2) % RW;
      const uint32_t b0 = ldsB[s * GM::RBLK + t], b1 = ldsB[s * GM::RBLK + t + 1];
      c[i0] += (uint64_t)X[0] * b0;
      c[i1] += (uint64_t)X[1] * b0;
      c[i1] += (uint64_t)X[0] * b1;
      const uint32_t q0l = (uint32_t)c[i0] & LMASK;
      const uint32_t q1l = ((uint32_t)c[i1] + (uint32_t)(c[i0] >> LB) + q0l * n1p) & LMASK;
      const uint32_t q0 = lead_bcast<RW>(q0l, lead4);
      const uint32_t q1 = lead_bcast<RW>(q1l, lead4);
      qd[t & 3] = q0; qd[(t + 1) & 3] = q1;
      if (((t + 1) & 3) == 3) q_write(qmask, row_addr + (t - 2) * 4, qd);
      else if (t + 2 == RW) q_write2(qmask, row_addr + t * 4, q0, q1);
#pragma unroll
      for (int k = 2; k < RW; k++) c[(t + k) % RW] += (uint64_t)X[k] * b0;
#pragma unroll
      for (int k = 1; k < RW - 1; k++) c[(t + 1 + k) % RW] += (uint64_t)X[k] * b1;
#pragma unroll
      for (int k = 0; k < RW; k++) c[(t + k) % RW] += (uint64_t)NT[k] * q0;
#pragma unroll
      for (int k = 0; k < RW - 1; k++) c[(t + 1 + k) % RW] += (uint64_t)NT[k] * q1;
      {
        const uint64_t v = c[i0];
        c[i1] += v >> LB;
        c[i0] = (uint64_t)next_lane<RW>((uint32_t)v & LMASK);          // slot i0 is column t + RW from here on
      }
      c[i0] += (uint64_t)X[RW - 1] * b1;                                 // the top products of the second digit
      c[i0] += (uint64_t)NT[RW - 1] * q1;
      {
        const uint64_t v = c[i1];
        c[i2] += v >> LB;
        c[i1] = (uint64_t)next_lane<RW>((uint32_t)v & LMASK);
      }
    }
  }
  uint64_t cy = 0;
#pragma unroll
  for (int k = 0; k < RW; k++) {
    const uint64_t t = c[k] + cy;
    R[k] = (uint32_t)t & LMASK;
    cy = t >> LB;
  }
  R[0] += prev_lane<RW>((uint32_t)cy, gl);
}
}  // namespace r2l

#ifndef ZKP_R2L_TWO_DIGITS
#define ZKP_R2L_TWO_DIGITS 1
#endif

// One wavefront per workgroup, one Enc per claim.  bcst[OFF_OK] == 0 (a key the form does not take): return at once, the launch behind this
// one — which claims from the same counter — does the work.
// `h`: the transcript hashes of a verify call that travel with this launch (as in k_enc_basen_r2l5 below): workgroup b < h.batch hashes proof b.
// A launch of at most one wavefront per SIMD gives every wavefront a SIMD to itself; as a launch of its own on a second stream the hash
// wavefront of 2 - 4 proofs shared one with an Enc wavefront in 4 calls of 10 (verify 10.0 or 12.5 ms).  Beyond (two wavefronts per SIMD:
// 6 - 8 proofs) the hashes are the launch's first workgroups all the same — one of a SIMD's two wavefronts instead of a third (14.6 or
// 17.2 ms) — at 16 blocks per batch, so that the LDS every workgroup of the launch then carries (6.6 KB) leaves eight of them on a unit.
template <int RW>
__global__ void __launch_bounds__(64) k_enc_basen_r2l(EncArgs a, const uint32_t* __restrict__ bcst, uint32_t* __restrict__ raw, RangeHashArgs h) {
  using namespace r2l;
  using GM = Geom<RW>;
  using BC = BnConst<8>;                     // (the key's record: limb-linear arrays of 72 limbs, whatever the lanes)
  constexpr int L = LIMBS, E = 2 * L, RG = GM::RG, AW = GM::AW, RBLK = GM::RBLK;
  if (blockIdx.x < h.batch) {
    if (h.wave_blocks == 16) range_hash_wave_body<16>(h, (int)threadIdx.x, blockIdx.x);
    else range_hash_wave_body<HW_BLOCKS>(h, (int)threadIdx.x, blockIdx.x);
    return;
  }
  if (!bcst[BC::OFF_OK]) return;
  __shared__ __align__(16) uint32_t lds[GM::LDS_WORDS];
  const int lane = threadIdx.x & 63, role = lane / RG, gl = lane - role * RG;
  const int lead4 = (lane - gl) * 4;
  auto area = [&](int i) -> uint32_t* { return lds + i * AW; };
  auto blk = [&](int i) -> uint32_t* { return lds + i * AW + gl * RBLK; };
  const int kw = a.n_bits / 32;
  const uint64_t count = a.count_ptr ? (uint64_t)*a.count_ptr : a.count;
  uint32_t NT[RW];
  limbs_global_load<RW>(NT, bcst + BC::OFF_MT, gl);
  const uint32_t n1 = bcst[BC::OFF_NI];
  // constants into their areas (every group stores the same blocks: harmless)
  {
    uint32_t T[RW];
    limbs_global_load<RW>(T, bcst + BC::OFF_C3, gl);  blk_store<RW>(blk(C3A), T);
    limbs_global_load<RW>(T, bcst + BC::OFF_R1A, gl); blk_store<RW>(blk(ONEA), T);
    limbs_global_load<RW>(T, bcst + BC::OFF_R1B, gl); blk_store<RW>(blk(ONEB), T);
#pragma unroll
    for (int k = 0; k < RW; k++) T[k] = 0;
    blk_store<RW>(blk(ZERO), T);
    if (gl == 0) T[0] = 1;
    blk_store<RW>(blk(INT1), T);
  }
  // bit length of the exponent (the key): wave-uniform
  int t_bits = 0;
  for (int w = kw - 1; w >= 0; w--) {
    const uint32_t v = __builtin_amdgcn_readfirstlane(a.n[w]);
    if (v) { t_bits = w * 32 + (32 - __clz(v)); break; }
  }
  auto nbit = [&](int k) -> bool { return k >= 0 && k < t_bits && ((__builtin_amdgcn_readfirstlane(a.n[k >> 5]) >> (k & 31)) & 1u); };
  const int dummy = DUM0 + role;
  for (;;) {
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(a.work_counter, 1ull);
    base = __shfl(base, 0);
    if (base >= count) break;
    const uint64_t item = base;
    const BnItem it = bn_item(a, item, nullptr);
    // ---- the item's r and m as limb blocks: r -> DA1 (B's multiplier of slot 0) and RL (A's register operand of slot -1); m -> MM
    {
      uint32_t T[RW];
      wave_lds_fence();
      for (int w = lane; w < 72; w += 64) area(WBUF)[w] = (w < it.rw) ? it.pr[w] : 0u;
      wave_lds_fence();
      limbs_from_words<RW>(T, area(WBUF), gl);
      wave_lds_fence();
      blk_store<RW>(blk(DA1), T);
      blk_store<RW>(blk(RL), T);
      for (int w = lane; w < 72; w += 64) area(WBUF)[w] = (it.pm && w < it.mw) ? it.pm[w] : 0u;
      wave_lds_fence();
      limbs_from_words<RW>(T, area(WBUF), gl);
      wave_lds_fence();
      blk_store<RW>(blk(MM), T);
      limbs_global_load<RW>(T, bcst + BC::OFF_RRA, gl); blk_store<RW>(blk(SA1), T);      // slot -1: A multiplies r by RR's a part (parity of -1: 1)
      limbs_global_load<RW>(T, bcst + BC::OFF_RRB, gl); blk_store<RW>(blk(SB), T);       // slot  0: B multiplies r by RR's b part
      wave_lds_fence();
    }
    // ---- the slots
#pragma unroll 1
    for (int k = -1; k <= t_bits + 2; k++) {
      const int par = k & 1, prev = par ^ 1;                         // (k & 1 is 1 for k = -1)
      const bool bit_prev = nbit(k - 1);
      const bool fin1 = k == t_bits + 1, fin2 = k == t_bits + 2;
      // which roles work in this slot
      const bool actA = k <= t_bits - 2, actB = k >= 0 && k <= t_bits - 1, actC = (k >= 1 && k <= t_bits - 1) || fin1;
      const bool actD = (k >= 2 && k <= t_bits) || fin2, actE = (k >= 2 && k <= t_bits) || fin1;
      // per group: register operand, staged operand, digits to start from (or none), where the result goes
      int xa = ZERO, ba = dummy, qa = -1, d0 = dummy;
      if (role == 0 && actA) { xa = k < 0 ? (int)RL : SA0 + par; ba = SA0 + par; d0 = SA0 + prev; }
      if (role == 1 && actB) { xa = DA0 + prev; ba = SB; qa = SA0 + prev; d0 = SB; }
      if (role == 2 && actC) { xa = fin1 ? (int)INT1 : PC0 + par; ba = fin1 ? (int)PX : SC0 + par; d0 = fin1 ? dummy : PC0 + prev; }
      if (role == 3 && actD) { xa = fin2 ? (int)INT1 : PC0 + prev; ba = fin2 ? (int)QQ : (bit_prev ? (int)SB : (int)ONEB); qa = fin2 ? (int)PX : SC0 + prev; d0 = RD; }
      if (role == 4 && actE) { xa = fin1 ? (int)MM : (int)QQ; ba = fin1 ? (int)SE0 + (t_bits & 1) : (bit_prev ? SE0 + prev : (int)ONEA); d0 = fin1 ? (int)UU : dummy; }
      // (fin1: C stages p over PX itself and leaves Q'' there; E needs an untouched copy of p: SE[t & 1] is rewritten with it below, at the end of slot t)
      // lanes that write quotient digits: lane 0 of A, lane 0 of C (an SGPR pair for s_and_saveexec: made uniform explicitly)
      const uint32_t qlo = (actA ? 1u : 0u) | ((actC && 2 * RG < 32) ? 1u << ((2 * RG) & 31) : 0u);
      const uint64_t qmask = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)qlo);
      uint32_t X[RW], R[RW];
      uint64_t c[RW];
      wave_lds_fence();
      blk_load<RW>(X, blk(xa));
      {
        uint32_t Q[RW], C3[RW];
        blk_load<RW>(Q, blk(qa < 0 ? (int)ZERO : qa));
        blk_load<RW>(C3, blk(qa < 0 ? (int)ZERO : (int)C3A));
        const uint32_t n1e = qa < 0 ? 0u : n1;
#pragma unroll
        for (int i = 0; i < RW; i++) c[i] = (uint64_t)C3[i] + (uint64_t)((1u << LB) - Q[i]) * n1e;
      }
      wave_lds_fence();
      if constexpr (ZKP_R2L_TWO_DIGITS && (RW & 1) == 0) product2<RW>(R, X, area(ba), NT, c, qmask, gl, lead4);
      else product<RW>(R, X, area(ba), NT, c, qmask, gl, lead4);
      wave_lds_fence();
      blk_store<RW>(blk(d0), R);
      wave_lds_fence();
      // ---- what follows a result
      if (role == 0 && actA) {
        blk_store<RW>(blk(SE0 + prev), R);
        uint32_t T[RW];
        if (nbit(k + 1)) {
#pragma unroll
          for (int i = 0; i < RW; i++) T[i] = R[i];
        } else blk_load<RW>(T, blk(ONEA));
        blk_store<RW>(blk(SC0 + prev), T);                           // C's multiplier of the next slot: a_(k+1), or the Montgomery one
        if (k < 0) blk_store<RW>(blk(PC0 + par), R);                 // the accumulator starts as s_0: p_1 = a_0 (n is odd), read by C in slot 1
#pragma unroll
        for (int i = 0; i < RW; i++) T[i] = R[i];
        dbl<RW>(T, gl);
        blk_store<RW>(blk(DA0 + prev), T);
      }
      if (role == 1 && actB && k == 0) blk_store<RW>(blk(QQ), R);    // q_1 = b_0
      if (role == 2 && actC) {
        if (fin1) limbs_global_store<RW>(raw + item * E, R, gl);
        else blk_store<RW>(blk(PX), R);
      }
      if (role == 2 && k == t_bits) {                                // p_t once more, for E's product by m in the next slot (C stages PX itself there)
        uint32_t T[RW];
        blk_load<RW>(T, blk(PX));
        blk_store<RW>(blk(SE0 + (t_bits & 1)), T);
      }
      if (role == 4 && actE && !fin1) {
        uint32_t D[RW];
        blk_load<RW>(D, blk(RD));
        add<RW>(R, R, D, gl);
        blk_store<RW>(blk(QQ), R);
      }
      if (role == 3 && fin2) {
        uint32_t U[RW];
        blk_load<RW>(U, blk(UU));
        add<RW>(R, R, U, gl);
        limbs_global_store<RW>(raw + item * E + L, R, gl);
      }
    }
  }
}


// ---- the same ladder with FIVE wavefronts per Enc: one role per wavefront, 36 lanes x 2 limbs per n-sized integer ------------------------
// One proof is 192 - 256 Enc and the chip has 1024 SIMDs: the one-wavefront kernel above leaves three quarters of them idle, and what
// bounds it is the instruction count of a lone wavefront's chain (an issue slot every 4 cycles, a multiply-add every 8): 24 multiply-adds
// and ~17 other instructions per pair of sub-steps plus one trip through the LDS crossbar for the two quotient digits.  Here every role
// (A, B, C, D, E of the header) is a wavefront of its own in a workgroup of five:
//   * 2 limbs per lane — 8 multiply-adds per pair of sub-steps instead of 24;
//   * ONE lane group per wavefront, so the quotient digits are WAVE-UNIFORM: lane 0's columns go through v_readfirstlane, the second digit
//     is worked out on the scalar unit, and the multiply-adds take the digits as SGPR operands — no trip through the LDS crossbar;
//   * the staged operand's limbs are broadcast reads of one address (two rows per ds_read_b128, fetched one iteration ahead);
//   * the digits of a sub-step pair are parked in lane s of a register pair (v_writelane) and leave as one 8-byte store per lane after
//     the product — the layout B and D read them in.
// The roles meet at two workgroup barriers per slot: one between the products and the stores that overwrite what a neighbour's product
// read (B's result over the b_k that D staged, A's over the digits B started from ...), one between the stores and the next slot's loads.
// E's sum q_k = D + E, which needs D's result of the SAME slot, is taken at the start of the next slot (E keeps its product in registers).
// Values, digits and the raw pair are those of k_enc_basen_r2l (tests/test_basen_r2l_model.py states them): same BnConst record, same
// k_basen_finish behind it.  320 threads, 11 KB of LDS; lanes 36 - 63 of every wavefront hold zeros (the top lane's neighbour).
// (timing probes of tools/dev/r2l5_variants.sh, never on in a shipped build: a product over a fraction of the rows, a slot without its barriers —
// both compute garbage; the difference to the real kernel is what a row pair and what a barrier costs)
#ifndef ZKP_R2L5_FOLD
#define ZKP_R2L5_FOLD 1      /* v_and_b32_dpp and the shifted-in limb as a multiply-add's addend: A/B switch (profiles/r05/r2l5/) */
#endif
#ifndef ZKP_R2L5_VALU_DIGITS
#define ZKP_R2L5_VALU_DIGITS 1      /* the quotient digits' arithmetic on the vector side (no scalar instruction between v_readfirstlane and its multiply-add): A/B switch */
#endif
#ifndef ZKP_R2L5_ASM_MADS
#define ZKP_R2L5_ASM_MADS 1      /* every multiply-add as the instruction on its own accumulator (needs FOLD and VALU_DIGITS): A/B switch */
#endif
#ifndef ZKP_R2L5_ONE_DIGIT
#define ZKP_R2L5_ONE_DIGIT 1      /* one quotient digit per sub-step (18 instead of 21 instructions per pair): A/B switch */
#endif
#ifndef ZKP_R2L5_PIN_ORDER
#define ZKP_R2L5_PIN_ORDER 0      /* the instruction order of a sub-step pinned by hand (no consumer straight behind an asm producer, the DPP read third after its source): measured 4.6 % SLOWER than the scheduler's own order (profiles/r05/r2l5/ab_pinned_order.jsonl) — off */
#endif
#ifndef ZKP_R2L5_ROLE_LOOPS
#define ZKP_R2L5_ROLE_LOOPS 1      /* the slot loop instantiated per role: A/B switch */
#endif
#ifndef ZKP_R2L5_REGS
#define ZKP_R2L5_REGS 1      /* per-role constants in registers, one exponent-bit read per slot: A/B switch */
#endif
#ifndef ZKP_R2L5_DEV_ROW_DIVISOR
#define ZKP_R2L5_DEV_ROW_DIVISOR 1
#endif
#ifndef ZKP_R2L5_DEV_NO_BARRIERS
#define ZKP_R2L5_DEV_NO_BARRIERS 0
#endif
namespace r2l5 {
using namespace r2l;
__device__ __forceinline__ void slot_barrier() {
  if constexpr (ZKP_R2L5_DEV_NO_BARRIERS) wave_lds_fence(); else __syncthreads();
}
// An area is the 72 limbs of an integer, limb-linear, plus 8 words that stay ZERO: lanes 36 - 63 of a wavefront hold zeros (the top lane's
// neighbour must), and instead of masking them out of every load and store — an execution-mask region each, ~0.1 us per slot — they all
// read and write word 72 of the same area: they load zeros, compute zeros, store zeros (`lw`, the lane's word offset; carries between
// lanes take such a lane for lane 0 of a group: `gle`).
constexpr int RW = 2, RG = LIMBS / RW, AW = LIMBS + 8, WAVES = 5;
constexpr int NEXP = NAREAS;                 // one more area: the exponent's words (the key), read bit by bit
constexpr int LDS_WORDS = (NAREAS + 1) * AW + 8;

__device__ __forceinline__ void ld2(uint32_t (&v)[RW], const uint32_t* area, int lw) {
  const uint2 t = *reinterpret_cast<const uint2*>(area + lw); v[0] = t.x; v[1] = t.y;
}
__device__ __forceinline__ void st2(uint32_t* area, const uint32_t (&v)[RW], int lw) {
  *reinterpret_cast<uint2*>(area + lw) = make_uint2(v[0], v[1]);
}
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// R = (X * B + c + q M~) / R' with B staged limb-linear at ldsB; the digits of row s (sub-steps 2 s, 2 s + 1) land in lane s of Qd when CAPTURE
template <bool CAPTURE>
__device__ __forceinline__ void product(uint32_t (&R)[RW], uint32_t (&Qd)[RW], const uint32_t (&X)[RW], const uint32_t* ldsB, const uint32_t (&NT)[RW],
                                        const uint64_t (&cin)[RW], uint32_t n1p, int gl) {
  uint64_t c0 = cin[0], c1 = cin[1];
  const uint32_t X0 = X[0], X1 = X[1], N0 = NT[0], N1 = NT[1];
  uint32_t qa = 0, qb = 0;
#if ZKP_R2L5_FOLD
  uint32_t lm;
  asm("v_mov_b32 %0, 0x1fffffff" : "=v"(lm));
#endif
#if ZKP_R2L5_VALU_DIGITS
  uint32_t vmask;                                      // (a register the compiler cannot see through: it would move the mask behind the v_readfirstlane, onto the scalar unit)
  asm("v_mov_b32 %0, 0x1fffffff" : "=v"(vmask));
#endif
  uint4 nx = *reinterpret_cast<const uint4*>(ldsB);
#if ZKP_R2L5_ASM_MADS
  // Every multiply-add written as the instruction on ITS accumulator: left to itself the compiler opens side sums (v_mad ... , 0) to shorten
  // dependency chains and merges them with two more 64-bit adds per pair of sub-steps — a lone wavefront is bound by what it gets issued,
  // not by those chains (a dependent multiply-add costs it the 2.1 ns an independent one does).
  auto madv = [](uint64_t& c, uint32_t a, uint32_t b) { uint64_t sink; asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(c), "=s"(sink) : "v"(a), "v"(b)); };
  auto mads = [](uint64_t& c, uint32_t a, uint32_t q) { uint64_t sink; asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(c), "=s"(sink) : "v"(a), "s"(q)); };
  // (a column that opens — the limb shifted in from the neighbour lane, upper word zero — is the ADDEND of its first multiply-add: no copy)
  auto mad3 = [](uint32_t a, uint32_t b, uint64_t in) { uint64_t c, sink; asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(c), "=s"(sink) : "v"(a), "v"(b), "v"(in)); return c; };
  auto trip = [&](uint32_t b0, uint32_t b1, auto rowc) {
    constexpr int row = decltype(rowc)::value;
    madv(c0, X0, b0);
    c1 = mad3(X1, b0, c1);
    madv(c1, X0, b1);
    const uint32_t q0v = (uint32_t)c0 & vmask;
    const uint32_t q0 = uni(q0v);
    const uint32_t q1 = uni(((uint32_t)c1 + q0v * n1p + (uint32_t)(c0 >> LB)) & vmask);
    if constexpr (CAPTURE) {
      asm("v_writelane_b32 %0, %1, %2" : "+v"(qa) : "s"(q0), "n"(row));
      asm("v_writelane_b32 %0, %1, %2" : "+v"(qb) : "s"(q1), "n"(row));
    }
    mads(c0, N0, q0);
    mads(c1, N1, q0);
    mads(c1, N0, q1);
    {
      const uint64_t v = c0;
      c1 += v >> LB;
      c0 = mad3(X1, b1, (uint64_t)(next_lane<RW>((uint32_t)v) & lm));
    }
    mads(c0, N1, q1);
    {
      const uint64_t v = c1;
      c0 += v >> LB;
      c1 = (uint64_t)(next_lane<RW>((uint32_t)v) & lm);
    }
  };
#else
  auto trip = [&](uint32_t b0, uint32_t b1, auto rowc) {
    constexpr int row = decltype(rowc)::value;
    c0 += (uint64_t)X0 * b0;
    c1 += (uint64_t)X1 * b0;
    c1 += (uint64_t)X0 * b1;
    // M~ == -1 (mod 2^29): the first digit is lane 0's bottom limb, the second follows from its bottom two columns (product2 above)
#if ZKP_R2L5_VALU_DIGITS
    // all of the digit arithmetic on the vector side, one v_readfirstlane per digit and nothing scalar in between: a scalar instruction between
    // a v_readfirstlane and the multiply-add that takes its result costs a lone wavefront 7 - 8 ns (csrc/microbench/lone_wave_hops.hip)
    const uint32_t q0v = (uint32_t)c0 & vmask;
    const uint32_t q0 = uni(q0v);
    const uint32_t q1 = uni(((uint32_t)(c1 + (uint64_t)q0v * n1p) + (uint32_t)(c0 >> LB)) & vmask);
#else
    const uint32_t q0 = uni((uint32_t)c0) & LMASK;
    const uint32_t t = uni((uint32_t)c1 + (uint32_t)(c0 >> LB));
    const uint32_t q1 = (t + q0 * n1p) & LMASK;
#endif
    if constexpr (CAPTURE) {
      // (lane `row` of the pair takes the two digits: one SGPR operand and an inline-constant lane select each)
      asm("v_writelane_b32 %0, %1, %2" : "+v"(qa) : "s"(q0), "n"(row));
      asm("v_writelane_b32 %0, %1, %2" : "+v"(qb) : "s"(q1), "n"(row));
    }
    c0 += (uint64_t)N0 * q0;
    c1 += (uint64_t)N1 * q0;
    c1 += (uint64_t)N0 * q1;
    // a lone wavefront pays for every instruction it issues: the limb mask is applied AFTER the DPP move, from a register the compiler cannot
    // see through, so that (move, and) become one v_and_b32_dpp; the limb that arrives is the addend of the column's first multiply-add
    {
      const uint64_t v = c0;
      c1 += v >> LB;
#if ZKP_R2L5_FOLD
      const uint64_t in = (uint64_t)(next_lane<RW>((uint32_t)v) & lm);
      uint64_t sink;
      asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(c0), "=s"(sink) : "v"(X1), "v"(b1), "v"(in));
#else
      c0 = (uint64_t)next_lane<RW>((uint32_t)v & LMASK);
      c0 += (uint64_t)X1 * b1;
#endif
    }
    c0 += (uint64_t)N1 * q1;
    {
      const uint64_t v = c1;
      c0 += v >> LB;
#if ZKP_R2L5_FOLD
      c1 = (uint64_t)(next_lane<RW>((uint32_t)v) & lm);
#else
      c1 = (uint64_t)next_lane<RW>((uint32_t)v & LMASK);
#endif
    }
  };
#endif
#if ZKP_R2L5_ONE_DIGIT && ZKP_R2L5_ASM_MADS
  // One quotient digit per sub-step after all: the two-digit form exists to save a trip through the LDS crossbar, and here a digit is a
  // v_readfirstlane away.  Taken one at a time the second digit needs no arithmetic of its own — it is the low limb of the next bottom
  // column once the first digit's carry is in — 18 instead of 21 instructions per pair of sub-steps for a wavefront that is bound by
  // what it gets issued.  Same digits, same values (`product` above).  The window of a lane: `bot` (column t) and the column that opens
  // with the limb shifted in from the neighbour lane, `in` — the addend of its first multiply-add.
  uint64_t bot = c0, in = c1;
  // (ZKP_R2L5_PIN_ORDER, an experiment that stays off: a scheduling barrier behind every statement and an order in which no consumer follows
  // an asm producer directly and the DPP read comes third after its source — the hazards that cost 5.5 s_nop per pair of sub-steps.  The
  // wait states stay (asm statements do not count as such, v_readfirstlane wants one after its source, an SGPR two before a vector
  // instruction reads it) and the scheduler's own order is 4.6 % faster.)
  auto pin = [] { if constexpr (ZKP_R2L5_PIN_ORDER) __builtin_amdgcn_sched_barrier(0); };
  auto step = [&](uint32_t b, auto tc) {
    constexpr int t = decltype(tc)::value;
#if ZKP_R2L5_PIN_ORDER
    // (only the multiply-add that OPENS a column is written as the instruction — its addend is what the compiler would turn into a side
    // sum —; it also fills the slot between the mask and the v_readfirstlane, which wants one instruction between it and its source)
    bot += (uint64_t)X0 * b; pin();
    const uint32_t qv = (uint32_t)bot & vmask; pin();
    uint64_t top = mad3(X1, b, in); pin();
    const uint32_t q = uni(qv); pin();
    if constexpr (CAPTURE) {
      if constexpr ((t & 1) == 0) asm("v_writelane_b32 %0, %1, %2" : "+v"(qa) : "s"(q), "n"(t / 2));
      else asm("v_writelane_b32 %0, %1, %2" : "+v"(qb) : "s"(q), "n"(t / 2));
      pin();
    }
    bot += (uint64_t)N0 * q; pin();
    top += (uint64_t)N1 * q; pin();
    const uint64_t cy = bot >> LB; pin();
    in = (uint64_t)(next_lane<RW>((uint32_t)bot) & lm); pin();
    top += cy; pin();
    bot = top;
#else
    madv(bot, X0, b);
    uint64_t top = mad3(X1, b, in);
    const uint32_t q = uni((uint32_t)bot & vmask);
    if constexpr (CAPTURE) {
      if constexpr ((t & 1) == 0) asm("v_writelane_b32 %0, %1, %2" : "+v"(qa) : "s"(q), "n"(t / 2));
      else asm("v_writelane_b32 %0, %1, %2" : "+v"(qb) : "s"(q), "n"(t / 2));
    }
    mads(bot, N0, q);
    mads(top, N1, q);
    top += bot >> LB;
    in = (uint64_t)(next_lane<RW>((uint32_t)bot) & lm);
    bot = top;
#endif
  };
  static_for<RG / 2 / ZKP_R2L5_DEV_ROW_DIVISOR>([&](auto ic) {
    constexpr int s = 2 * decltype(ic)::value;
    const uint4 cur = nx;
    if constexpr (s + 2 < RG) nx = *reinterpret_cast<const uint4*>(ldsB + RW * (s + 2));
    step(cur.x, std::integral_constant<int, 2 * s>{});
    step(cur.y, std::integral_constant<int, 2 * s + 1>{});
    step(cur.z, std::integral_constant<int, 2 * s + 2>{});
    step(cur.w, std::integral_constant<int, 2 * s + 3>{});
  });
  c0 = bot; c1 = in;
#else
  // 18 pairs of rows, fully unrolled (the lane selects are immediates; ~1100 instructions per variant)
  static_for<RG / 2 / ZKP_R2L5_DEV_ROW_DIVISOR>([&](auto ic) {
    constexpr int s = 2 * decltype(ic)::value;
    const uint4 cur = nx;
    if constexpr (s + 2 < RG) nx = *reinterpret_cast<const uint4*>(ldsB + RW * (s + 2));
    trip(cur.x, cur.y, std::integral_constant<int, s>{});
    trip(cur.z, cur.w, std::integral_constant<int, s + 1>{});
  });
#endif
  uint64_t t0 = c0;
  R[0] = (uint32_t)t0 & LMASK;
  t0 = c1 + (t0 >> LB);
  R[1] = (uint32_t)t0 & LMASK;
  R[0] += prev_lane<RW>((uint32_t)(t0 >> LB), gl);
  Qd[0] = qa; Qd[1] = qb;
}
}  // namespace r2l5

// `h`: the transcript hashes of a verify call that travel with this launch (h.batch of them, 0: none): workgroup b < h.batch is not an Enc
// workgroup — its first wavefront hashes proof b (range_hash_wave_body; the launch then carries hw_lds_words(kw) words of dynamic LDS), the
// others leave.  Why here and not in a launch of its own on a second stream: the dispatcher gives every workgroup of a launch of at most
// one per compute unit a unit to itself, so the hash wavefront shares its SIMD with nobody; as a launch of its own it landed beside the
// role wavefronts of an Enc workgroup three times out of four, and on the SIMD that carries two of them once in four: the slow mode of the
// one-proof verify (profiles/r06/one_proof/).
__global__ void __launch_bounds__(320) k_enc_basen_r2l5(EncArgs a, const uint32_t* __restrict__ bcst, uint32_t* __restrict__ raw, RangeHashArgs h) {
  using namespace r2l5;
  using BC = BnConst<8>;
  constexpr int L = LIMBS, E = 2 * L;
  if (blockIdx.x < h.batch) {
    if (threadIdx.x < 64) range_hash_wave_body(h, (int)threadIdx.x, blockIdx.x);
    return;
  }
  if (!bcst[BC::OFF_OK]) return;
  // The five role wavefronts are ONE dependent chain in lockstep: a stranger on one of their SIMDs — the transcript-hash wavefront of a
  // one-proof verify, which runs beside this launch — holds all five up.  They issue ahead of it (ZKP_R2L_PRIO, A/B: profiles/r06/one_proof/).
  __builtin_amdgcn_s_setprio(ZKP_R2L_PRIO);
  __shared__ __align__(16) uint32_t lds[LDS_WORDS];
  __shared__ unsigned long long claim;
  const int tid = threadIdx.x, lane = tid & 63, gl = lane;
  const int role_rt = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool on = lane < RG;
  const int lw = on ? RW * gl : LIMBS, gle = on ? gl : 0;
  auto area = [&](int i) -> uint32_t* { return lds + i * AW; };
  const int kw = a.n_bits / 32;
  const uint64_t count = a.count_ptr ? (uint64_t)*a.count_ptr : a.count;
  uint32_t NT[RW];
  NT[0] = on ? bcst[BC::OFF_MT + RW * gl] : 0u;
  NT[1] = on ? bcst[BC::OFF_MT + RW * gl + 1] : 0u;
  const uint32_t n1 = bcst[BC::OFF_NI];
  const uint32_t n1p = uni(NT[1]) + 1;
  for (int w = tid; w < LDS_WORDS; w += 64 * WAVES) lds[w] = 0;
  __syncthreads();
  for (int w = tid; w < LIMBS; w += 64 * WAVES) {
    area(C3A)[w] = bcst[BC::OFF_C3 + w];
    area(ONEA)[w] = bcst[BC::OFF_R1A + w];
    area(ONEB)[w] = bcst[BC::OFF_R1B + w];
    area(ZERO)[w] = 0;
    area(INT1)[w] = w == 0 ? 1u : 0u;
    area(NEXP)[w] = w < kw ? a.n[w] : 0u;
  }
  __syncthreads();
  int t_bits = 0;
  for (int w = kw - 1; w >= 0; w--) {
    const uint32_t v = uni(area(NEXP)[w]);
    if (v) { t_bits = w * 32 + (32 - __clz(v)); break; }
  }
  // bit k of the exponent out of a 32-bit window in a scalar register (one LDS read every 32 slots)
  int ew_at = -1;
  uint32_t ew = 0;
  auto nbit = [&](int k) -> bool {
    if (k < 0 || k >= t_bits) return false;
    if ((k >> 5) != ew_at) { ew_at = k >> 5; ew = uni(area(NEXP)[ew_at]); }
    return ((ew >> (k & 31)) & 1u) != 0;
  };
  uint32_t C3r[RW], ONEr[RW];                          // C3 (B's and D's initial columns) and the Montgomery one (A hands it to C at a clear bit)
  ld2(C3r, area(C3A), lw);
  ld2(ONEr, area(ONEA), lw);
  for (;;) {
    __syncthreads();                                   // (the previous item's areas are free)
    if (tid == 0) claim = atomicAdd(a.work_counter, 1ull);
    __syncthreads();
    const unsigned long long base = claim;
    const uint64_t item = ((uint64_t)uni((uint32_t)(base >> 32)) << 32) | uni((uint32_t)base);
    if (item >= count) break;
    const BnItem it = bn_item(a, item, nullptr);
    // ---- the item's r and m as limbs: r -> DA1 (B's multiplier of slot 0) and RL (A's register operand of slot -1); m -> MM
    for (int w = tid; w < LIMBS; w += 64 * WAVES) {
      area(WBUF)[w] = (w < it.rw) ? it.pr[w] : 0u;
      area(DUM0)[w] = (it.pm && w < it.mw) ? it.pm[w] : 0u;
    }
    __syncthreads();
    if (tid < L) {
      const int bit = tid * LB, w0 = bit >> 5, off = bit & 31;
      const uint64_t xr = (uint64_t)area(WBUF)[w0] | ((uint64_t)area(WBUF)[w0 + 1] << 32);
      const uint64_t xm = (uint64_t)area(DUM0)[w0] | ((uint64_t)area(DUM0)[w0 + 1] << 32);
      const uint32_t lr = (uint32_t)(xr >> off) & LMASK, lm = (uint32_t)(xm >> off) & LMASK;
      area(DA1)[tid] = lr; area(RL)[tid] = lr; area(MM)[tid] = lm;
      area(SA1)[tid] = bcst[BC::OFF_RRA + tid];      // slot -1: A multiplies r by RR's a part (parity of -1: 1)
      area(SB)[tid] = bcst[BC::OFF_RRB + tid];       // slot  0: B multiplies r by RR's b part
    }
    __syncthreads();
    // the slots, once per role: the compiler sees which areas a role reads and writes and which product (with or without digit capture) it runs
    // (one loop for all five — the role a run-time value — spent ~0.2 us per slot on selecting them: ZKP_R2L5_ROLE_LOOPS=0, profiles/r05/r2l5/)
    auto slots = [&](auto rolec) {
#if ZKP_R2L5_ROLE_LOOPS
    constexpr int role = decltype(rolec)::value;
#else
    const int role = role_rt;
#endif
    uint32_t Es[RW] = {0, 0};                          // E's product of the previous slot, until D's result of that slot can be added
    bool pend = false;
    bool b_prev = false, b_cur = false, b_next = nbit(0);      // bits k - 1, k, k + 1 of the exponent: one LDS read per slot, issued ahead of the product
#pragma unroll 1
    for (int k = -1; k <= t_bits + 2; k++) {
      const int par = k & 1, prev = par ^ 1;
#if ZKP_R2L5_REGS
      const bool bit_prev = b_prev, bit_next = b_next;
      b_prev = b_cur; b_cur = b_next; b_next = nbit(k + 2);
#else
      const bool bit_prev = nbit(k - 1), bit_next = nbit(k + 1);
#endif
      const bool fin1 = k == t_bits + 1, fin2 = k == t_bits + 2;
      bool act = false, capture = false;
      int xa = ZERO, ba = ZERO, qa = -1, d0 = -1;
      if (role == 0) { act = k <= t_bits - 2; capture = true; xa = k < 0 ? (int)RL : SA0 + par; ba = SA0 + par; d0 = SA0 + prev; }
      else if (role == 1) { act = k >= 0 && k <= t_bits - 1; xa = DA0 + prev; ba = SB; qa = SA0 + prev; d0 = SB; }
      else if (role == 2) { act = (k >= 1 && k <= t_bits - 1) || fin1; capture = true; xa = fin1 ? (int)INT1 : PC0 + par; ba = fin1 ? (int)PX : SC0 + par; d0 = fin1 ? -1 : PC0 + prev; }
      else if (role == 3) { act = (k >= 2 && k <= t_bits) || fin2; xa = fin2 ? (int)INT1 : PC0 + prev; ba = fin2 ? (int)QQ : (bit_prev ? (int)SB : (int)ONEB); qa = fin2 ? (int)PX : SC0 + prev; d0 = RD; }
      else { act = (k >= 2 && k <= t_bits) || fin1; xa = fin1 ? (int)MM : (int)QQ; ba = fin1 ? (int)SE0 + (t_bits & 1) : (bit_prev ? SE0 + prev : (int)ONEA); d0 = fin1 ? (int)UU : -1; }
      uint32_t X[RW], R[RW] = {0, 0}, Qd[RW] = {0, 0};
      if (role == 4 && pend) {    // q_(k-1) = D + E of the previous slot (RD is D's until the stores of this slot)
        uint32_t D[RW];
        ld2(D, area(RD), lw);
        add<RW>(Es, Es, D, gle);
        st2(area(QQ), Es, lw);
        pend = false;
      }
      if (act) {
        if (role == 4 && !fin1 && k > 2) { X[0] = Es[0]; X[1] = Es[1]; }
        else ld2(X, area(xa), lw);
        uint64_t c[RW];
        {
          uint32_t Q[RW], C3[RW];
          ld2(Q, area(qa < 0 ? (int)ZERO : qa), lw);
#if ZKP_R2L5_REGS
          C3[0] = qa < 0 ? 0u : C3r[0]; C3[1] = qa < 0 ? 0u : C3r[1];
#else
          ld2(C3, area(qa < 0 ? (int)ZERO : (int)C3A), lw);
#endif
          const uint32_t n1e = (qa < 0 || !on) ? 0u : n1;
#pragma unroll
          for (int i = 0; i < RW; i++) c[i] = (uint64_t)C3[i] + (uint64_t)((1u << LB) - Q[i]) * n1e;
        }
        if (capture) product<true>(R, Qd, X, area(ba), NT, c, n1p, gle);
        else product<false>(R, Qd, X, area(ba), NT, c, n1p, gle);
      }
      slot_barrier();                                  // every product of the slot has read what it reads
      if (act) {
        if (capture) st2(area(ba), Qd, lw);        // the digits over the staged operand, where B / D look for them one slot later
        if (d0 >= 0) st2(area(d0), R, lw);
        if (role == 0) {
          st2(area(SE0 + prev), R, lw);
          uint32_t T[RW];
          if (bit_next) { T[0] = R[0]; T[1] = R[1]; }
#if ZKP_R2L5_REGS
          else { T[0] = ONEr[0]; T[1] = ONEr[1]; }
#else
          else ld2(T, area(ONEA), lw);
#endif
          st2(area(SC0 + prev), T, lw);            // C's multiplier of the next slot: a_(k+1), or the Montgomery one
          if (k < 0) st2(area(PC0 + par), R, lw);  // the accumulator starts as s_0: p_1 = a_0 (n is odd), read by C in slot 1
          T[0] = R[0]; T[1] = R[1];
          dbl<RW>(T, gle);
          st2(area(DA0 + prev), T, lw);
        }
        if (role == 1 && k == 0) st2(area(QQ), R, lw);          // q_1 = b_0
        if (role == 2) {
          if (fin1) { if (on) { raw[item * E + RW * gl] = R[0]; raw[item * E + RW * gl + 1] = R[1]; } }
          else st2(area(PX), R, lw);
        }
        if (role == 3 && fin2) {
          uint32_t U[RW];
          ld2(U, area(UU), lw);
          add<RW>(R, R, U, gle);
          if (on) { raw[item * E + L + RW * gl] = R[0]; raw[item * E + L + RW * gl + 1] = R[1]; }
        }
        if (role == 4 && !fin1) { Es[0] = R[0]; Es[1] = R[1]; pend = true; }
      }
      if (role == 2 && k == t_bits) {   // p_t once more, for E's product by m in the next slot (C stages PX itself there)
        uint32_t T[RW];
        ld2(T, area(PX), lw);
        st2(area(SE0 + (t_bits & 1)), T, lw);
      }
      slot_barrier();                                  // ... and every store of the slot is in place
    }
    };
#if ZKP_R2L5_ROLE_LOOPS
    switch (role_rt) {
      case 0: slots(std::integral_constant<int, 0>{}); break;
      case 1: slots(std::integral_constant<int, 1>{}); break;
      case 2: slots(std::integral_constant<int, 2>{}); break;
      case 3: slots(std::integral_constant<int, 3>{}); break;
      default: slots(std::integral_constant<int, 4>{}); break;
    }
#else
    slots(std::integral_constant<int, -1>{});
#endif
  }
}

#endif  // ZKP_W == 9

}  // namespace zkp
