// bigint29.hpp — wave-cooperative fixed-width big-integer arithmetic for gfx950 (MI355X).
//
// Representation (chosen from measurements, see DESIGN.md §3 and profiles/valu_rates_r01.jsonl):
//   * radix 2^29 limbs held in 32-bit registers.  On gfx950 v_mad_u64_u32 issues at ~4.9
//     cycles per wave64 but every carry-flag instruction (v_addc_co_u32) costs ~4.3 cycles
//     too, so a full-radix 2^32 multiply-accumulate with explicit carries is ~2x slower than
//     a carry-free one.  With 29-bit limbs a 64-bit column accumulator absorbs 64 products
//     before it can overflow, so the inner loops are pure v_mad_u64_u32 chains.
//   * one big integer is spread over a group of G lanes of a wavefront, W limbs per lane
//     (W = 36 by default; -DZKP_W=18 / 9 build the earlier geometries); G*W = 72 limbs = 2088 bits (moduli up to 2048 bits:
//     n, N of the DLog proof), 144 limbs = 4176 bits (n^2 for a 2048-bit n), 288 limbs =
//     8352 bits (n^2 for a 4096-bit n); a 64-lane wavefront works on 64/G independent
//     modular exponentiations.
//   * Montgomery radix R = 2^(29*G*W) exceeds the modulus by >= 40 bits, hence every
//     Montgomery product of operands < 2M is again < 2M and no conditional subtraction is
//     needed inside an exponentiation (one exact canonicalisation at the very end).
//
// Montgomery multiplication is a word-level CIOS on a systolic lane array (see montmul below):
// the partial sum for one output position travels down the lanes while it is completed.
//
// This file replaces what the reference reaches through curv::BigInt -> GMP
// (mpz_powm / mpz_mul / mpz_tdiv_r); see include/zkp_hip.h for the call sites.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// (kernels that are not templates are defined in the headers under #ifndef ZKP_TEMPLATE_KERNELS_ONLY: a translation unit that only
// instantiates template kernels — zkp_kernels_keys.hip — leaves them out, the one copy that is launched is the one of zkp_api.hip)

namespace zkp {

#ifndef ZKP_W
#define ZKP_W 36
#endif
constexpr int LB = 29;                      // bits per limb
constexpr int W = ZKP_W;                    // limbs per lane (36; 18 and 9 still build: -DZKP_W=18)
constexpr uint32_t LMASK = (1u << LB) - 1;
constexpr int BLK = (W + 3) & ~3;           // LDS words per W-limb block (16-B multiple: keeps ds_read_b128 aligned)
static_assert(W == 9 || W == 18 || W == 36, "limbs per lane");
// minimum waves per SIMD requested from the register allocator for the modexp-class kernels: the hot
// loop (montmul) needs ~70 VGPRs at W = 9, ~150 at W = 18 and all 256 at W = 36 (72 accumulator + 72 operand registers);
// values that live across an exponentiation spill around it.  Measured on MI355X (Enc/s at n = 2048): W=9 @5 waves 251 K,
// W=18 @2 waves 269 K (322 K at the end of round 1), W=36 @2 waves 337 K — every doubling of the window halves the
// bookkeeping instructions per multiply-add.
#ifndef ZKP_WPE
#define ZKP_WPE (ZKP_W == 9 ? 5 : 2)   /* W = 36 also runs 2 waves per SIMD (256 VGPRs) */
#endif

template <int G> struct Geo {
  static constexpr int L = G * W;           // 29-bit limbs per integer
  static constexpr int CAPBITS = L * LB;    // log2(R)
  static constexpr int LDS_B = G * BLK;     // words of the B-operand staging area
};

// ---------------------------------------------------------------- cross-lane primitives
// value held by lane 0 of the group -> every lane of the group (the quotient digit, once per sub-step).  Groups that fit a
// quad take ONE DPP move (quad_perm): against the LDS-crossbar ds_swizzle (8.4 issue cycles and a trip through the LDS queue)
// that is +1.2 % verifies/s on the throughput engine, whose Paillier kernels are G = 4 (A/B on one box).  Wider
// groups keep ds_swizzle: a 16-lane broadcast needs three dependent DPP moves (quad_perm, row_shr:4 and row_shr:8 under bank
// masks), and on the latency engine, where one wavefront per SIMD waits out every dependent instruction, that chain doubled
// the time of a sub-step (one proof 27 -> 48 ms) — measured, not adopted.
#ifndef ZKP_BCAST_DPP
#define ZKP_BCAST_DPP 1
#endif
#ifndef ZKP_BCAST8_DPP
#define ZKP_BCAST8_DPP 0      /* 8-lane groups (latency engine, n-sized integers): two DPP moves instead of ds_swizzle — A/B switch, see DESIGN.md section 8 */
#endif
template <int G> __device__ __forceinline__ uint32_t bcast0(uint32_t v) {
  static_assert(G == 2 || G == 4 || G == 8 || G == 16 || G == 32, "group size");
  if constexpr (ZKP_BCAST_DPP && G == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xA0 /*quad_perm:[0,0,2,2]*/, 0xf, 0xf, true);
  else if constexpr (ZKP_BCAST_DPP && G == 4) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00 /*quad_perm:[0,0,0,0]*/, 0xf, 0xf, true);
#if ZKP_BCAST8_DPP
  else if constexpr (G == 8) {
    // 8-lane groups: lane 0 of every quad over its quad, then lanes 4-7 (and 12-15) of a row take their left neighbour quad's copy
    // (row_shr:4 under bank mask 0b1010; the other lanes keep the first move's result): two dependent DPP moves, no trip through LDS
    const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x00 /*quad_perm:[0,0,0,0]*/, 0xf, 0xf, true);
    return (uint32_t)__builtin_amdgcn_update_dpp(t, t, 0x114 /*row_shr:4*/, 0xf, 0xa, false);
  }
#endif
  else if constexpr (G == 2) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x001E);
  else if constexpr (G == 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x0000);
  else if constexpr (G == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x0010);
  else if constexpr (G == 8) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x0018);
  else return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x001C);
}

// ZKP_AND_DPP=1 (A/B switch, round 4): the 29-bit limb mask is applied AFTER the DPP move and from a VGPR the compiler cannot see through,
// so that its DPP combiner folds each (move, and) pair of a sub-step into one v_and_b32_dpp: 2 of the ~8 bookkeeping instructions of a
// sub-step.  Measured: see DESIGN.md section 8 (the kernels are at the board's power cap; fewer instructions buy a lower clock).
#ifndef ZKP_AND_DPP
#define ZKP_AND_DPP 0
#endif
__device__ __forceinline__ uint32_t limb_mask_operand() {
#if ZKP_AND_DPP
  uint32_t m;
  asm("v_mov_b32 %0, 0x1fffffff" : "=v"(m));
  return m;
#else
  return LMASK;
#endif
}

// lane j receives the value of lane j+1 (one DPP move).  The top lane of a group receives the value of the NEXT group's
// lane 0 (or 0 at the end of a DPP row): inside montmul that value is the low limb of lane 0's finished bottom column,
// which the quotient digit has just made zero, so no select is needed to clear it (ZKP_SELECT_TOP=1 restores the select).
#ifndef ZKP_SELECT_TOP
#define ZKP_SELECT_TOP 0
#endif
template <int G> __device__ __forceinline__ uint32_t from_next(uint32_t v, int gl) {
  if constexpr (G == 32) {
    uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
    return (ZKP_SELECT_TOP && gl == G - 1) ? 0u : t;
  } else {
    uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101 /*row_shl:1*/, 0xf, 0xf, true);
    return (ZKP_SELECT_TOP && G != 16 && gl == G - 1) ? 0u : t;
  }
}

// lane j receives the value of lane j-1 of its group; lane 0 receives 0
template <int G> __device__ __forceinline__ uint32_t from_prev(uint32_t v, int gl) {
  if constexpr (G == 16) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
  } else if constexpr (G == 8 || G == 4 || G == 2) {
    uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    return gl == 0 ? 0u : t;
  } else {
    uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
    return gl == 0 ? 0u : t;
  }
}

// LDS traffic between lanes of ONE wavefront needs no s_barrier (the LDS queue of a wave is
// in order); this only stops the compiler from moving LDS accesses across the hand-off.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------- block load/store (W limbs)
__device__ __forceinline__ void lds_load_block(uint32_t (&v)[W], const uint32_t* p) {
#pragma unroll
  for (int i = 0; i + 4 <= W; i += 4) {
    const uint4 a = *reinterpret_cast<const uint4*>(p + i);
    v[i] = a.x; v[i + 1] = a.y; v[i + 2] = a.z; v[i + 3] = a.w;
  }
  if constexpr (W % 4 == 1) {
    v[W - 1] = p[W - 1];
  } else if constexpr (W % 4 == 2) {
    const uint2 a = *reinterpret_cast<const uint2*>(p + W - 2);
    v[W - 2] = a.x; v[W - 1] = a.y;
  }
}
__device__ __forceinline__ void lds_store_block(uint32_t* p, const uint32_t (&v)[W]) {
#pragma unroll
  for (int i = 0; i + 4 <= W; i += 4) *reinterpret_cast<uint4*>(p + i) = make_uint4(v[i], v[i + 1], v[i + 2], v[i + 3]);
  if constexpr (W % 4 == 1) {
    p[W - 1] = v[W - 1];
  } else if constexpr (W % 4 == 2) {
    *reinterpret_cast<uint2*>(p + W - 2) = make_uint2(v[W - 2], v[W - 1]);
  }
}

// ---------------------------------------------------------------- column capacity (W = 36)
// A column accumulator lives for W sub-steps and takes two products per sub-step, A_k*b and N_k*q, every k exactly once
// over its life.  With 29-bit limbs a product is < 2^58, so 64 of them fit a 64-bit register: W <= 31 is safe for any
// operands (W = 18: 36 products), W = 36 is not (72 products).  Over a column's life
//     sum  <=  max(b) * S_A  +  max(q) * S_N  + carry + digit,     S_A = sum of this lane's limbs of A (<= 36 * 2^29 + 16),
//                                                                   S_N = sum of this lane's limbs of the modulus operand,
// so the plain W = 36 loop is exact for ANY A and B as long as every lane's S_N <= COL_FAST_SN_LIMIT (mean limb <= 0.777 *
// 2^29; a random modulus has mean 0.5 +- 0.05 per lane).  k_setup measures S_N of the modulus operand of the ladders (the
// Orup multiple M~) once per key and records whether the FAST product may be used; otherwise, and for the few general
// products around the ladders (whose modulus operand is M itself), the SAFE product runs: at half of a column's life it
// moves the column's upper 32-bit word up into the next column (2^32 = 8 * 2^29: one multiply-add and one move per
// sub-step, about 7 % slower), after which no column can exceed 2^63.2 whatever the operands are.
// tests/test_lane_model.py states both bounds executably.
constexpr bool COL_NEEDS_CARE = 2 * W > 63;
constexpr uint64_t COL_FAST_SN_LIMIT = ((~0ull) - (1ull << 36) - ((1ull << LB) + 16) * (W * (1ull << LB) + 16)) >> LB;

// ---------------------------------------------------------------- Montgomery multiplication
// R = A * B / 2^(29*G*W) mod M.  A: this lane's block of A (registers); B: staged in LDS as G
// blocks of BLK words; N: this lane's block of the modulus; n1 = -M^-1 mod 2^29.
// Operand limbs may be "almost normalised" (< 2^29 + 2^8); operand values < 2M.
// Result: limbs 1.. < 2^29, limb 0 < 2^29 + 16; value < 2M (<= M when B == 1).  R may alias A (in place).
//
// Word-level CIOS over a sliding window of W 64-bit column accumulators per lane.  Sub-step
// (s,t) handles limb b = B[s*W+t]: every lane adds A_j*b to its W columns, lane 0's bottom
// column fixes the quotient digit q = (c0 * n1) mod 2^29 (one ds_swizzle broadcast), every
// lane adds N_j*q, then the bottom column is finished: its low 29 bits go to lane j-1 (one DPP
// move) where they open a fresh top column, its high bits carry into the next column.  Only
// ONE column is normalised per sub-step and no carry flag is ever used: 2W v_mad_u64_u32 per
// ~6 other VALU instructions.  The window is a circular register file: after W sub-steps
// (fully unrolled) the register assignment repeats, so the loop over blocks stays rolled.
//
// ORUP = true: the caller passes N := M~ = M * n1 (a multiple of M with M~ == -1 mod 2^29, Orup's trick):
// the quotient digit is then just the low limb of the bottom column, one multiply less per sub-step.  The
// result is correct modulo M (not reduced below M~): used inside exponentiation ladders only.
// SAFE: see "column capacity" above (no effect for W <= 31).
// SCHED_FENCE > 0: a scheduling barrier every SCHED_FENCE sub-steps.  The machine scheduler runs the quotient-digit chain several
// sub-steps ahead of the bulk of the multiply-adds; in the loop of the sliding-window ladder, next to the squaring body, that look-ahead
// cost 11 (G = 4) / 27 (G = 8) scratch accesses per block of the product.  Fenced every 12 sub-steps the block holds one, and the
// shared-key kernels gain 0.5 % (n = 2048) / 1.6 % (n = 4096) — the fixed-window ladder loses 1.4 % with it and stays unfenced.
template <int G, bool ORUP = false, bool SAFE = true, int SCHED_FENCE = 0>
__device__ __forceinline__ void montmul(uint32_t (&R)[W], const uint32_t (&A)[W], const uint32_t* ldsB,
                                        const uint32_t (&N)[W], uint32_t n1, int gl) {
  uint64_t c[W];
#pragma unroll
  for (int k = 0; k < W; k++) c[k] = 0;
  [[maybe_unused]] const uint32_t lm = limb_mask_operand();
  [[maybe_unused]] uint64_t sink;   // carry-out operand of the explicit v_mad_u64_u32 below (never set: the sums stay below 2^64)

#pragma unroll 1
  for (int s = 0; s < G; s++) {
#pragma unroll
    for (int t = 0; t < W; t++) {
      if constexpr (SCHED_FENCE > 0) { if (t % SCHED_FENCE == 0 && t) __builtin_amdgcn_sched_barrier(0); }
      const uint32_t b = ldsB[s * BLK + t];
#pragma unroll
      for (int k = 0; k < W; k++) c[(t + k) % W] += (uint64_t)A[k] * b;
#if ZKP_AND_DPP
      const uint32_t q = bcast0<G>(ORUP ? (uint32_t)c[t] : (uint32_t)c[t] * n1) & lm;
#else
      const uint32_t q = bcast0<G>((ORUP ? (uint32_t)c[t] : (uint32_t)c[t] * n1) & LMASK);
#endif
#pragma unroll
      for (int k = 0; k < W; k++) c[(t + k) % W] += (uint64_t)N[k] * q;
      const uint64_t v = c[t];
      c[(t + 1) % W] += v >> LB;
#if ZKP_AND_DPP
      c[t] = (uint64_t)(from_next<G>((uint32_t)v, gl) & lm);
#else
      c[t] = (uint64_t)from_next<G>((uint32_t)v & LMASK, gl);
#endif
      if constexpr (SAFE && COL_NEEDS_CARE) {
        constexpr int H = W / 2;
        const uint32_t hi = (uint32_t)(c[(t + H) % W] >> 32);
        // (written as the instruction: the compiler would expand hi * 8 into a 64-bit shift-and-add over a temporary register pair)
        asm("v_mad_u64_u32 %0, %1, %2, 8, %0" : "+v"(c[(t + H + 1) % W]), "=s"(sink) : "v"(hi));
        c[(t + H) % W] &= 0xFFFFFFFFull;
      }
    }
  }

  // lane-local exact chain; the (tiny) carry out of each block lands on limb 0 of the next lane
  uint64_t cy = 0;
#pragma unroll
  for (int k = 0; k < W; k++) {
    const uint64_t t = c[k] + cy;
    R[k] = (uint32_t)t & LMASK;
    cy = t >> LB;
  }
  R[0] += from_prev<G>((uint32_t)cy, gl);
}

// ---------------------------------------------------------------- Montgomery squaring: X = X * X / R on the Orup multiple
// 86 % of the products of a ladder are squarings.  A*A holds every cross product a_m a_s twice, but the systolic array pins
// the product a_m * b_s to lane(m) at sub-step s, so "compute the upper triangle only" would leave the low lanes idle in
// lockstep and save nothing.  What the array does allow is a choice PER LIMB POSITION: at sub-step s (position t = s mod W
// inside its block) every lane multiplies only the limbs k of a fixed set K_t by b_s, the same register indices in all lanes.
// The ordered pair (m, s) is then computed iff (m mod W) is in K_(s mod W), and if the sets form a TOURNAMENT on the W
// positions — for k != t exactly one of "k in K_t", "t in K_k" holds — every unordered pair of limbs at different positions
// is computed exactly once, whatever lanes the two limbs live in: those products are doubled (b_s + b_s as the multiplier).
// Limbs at the SAME position (k == t, different blocks or the true square a_s^2) keep both orders, undoubled.
// K_t = { t } + { k : (k - t) mod W in 1 .. W/2 - 1 } + { t + W/2 if t < W/2 }: 18 or 19 multiply-adds per sub-step instead of 36,
// with the sub-step's bookkeeping, quotient digit and N * q half unchanged (54.5 instead of 72 multiply-adds: -24 %).
// Every product lands in the column and at a sub-step where montmul puts a product of the same column, so the column a
// quotient digit is read from is complete exactly as before: the digits, and with them the result, are those of
// montmul(X, X) (tests/test_lane_model.py: same value, column bound).  Column capacity: over a column's life the doubled and
// single products of a lane add up to at most 2 * 18 (or 2 * 17 + 2) limb products — the FAST bound of "column capacity"
// above holds unchanged; there is no SAFE variant (keys whose M~ fails the digit-sum test keep montmul for every product).
template <int G>
__device__ __forceinline__ void montsqr(uint32_t (&X)[W], const uint32_t* ldsB /* the staged copy of X */, const uint32_t (&N)[W], int gl) {
  constexpr int H = W / 2;   // (odd W: distances 1 .. (W-1)/2 form a regular tournament by themselves)
  uint64_t c[W];
#pragma unroll
  for (int k = 0; k < W; k++) c[k] = 0;
  [[maybe_unused]] const uint32_t lm = limb_mask_operand();
#pragma unroll 1
  for (int s = 0; s < G; s++) {
#pragma unroll
    for (int t = 0; t < W; t++) {
      const uint32_t b = ldsB[s * BLK + t];
      c[(2 * t) % W] += (uint64_t)X[t] * b;
      const uint32_t b2 = b + b;      // (doubling in place by inline asm, and scheduling barriers every 4 sub-steps, were measured: +-1 %)
#pragma unroll
      for (int k = 0; k < W; k++) {
        const int d = (k - t + W) % W;
        const bool take = (W & 1) ? (d >= 1 && d <= H) : ((d >= 1 && d < H) || (d == H && t < H));
        if (take) c[(t + k) % W] += (uint64_t)X[k] * b2;
      }
#if ZKP_AND_DPP
      const uint32_t q = bcast0<G>((uint32_t)c[t]) & lm;
#else
      const uint32_t q = bcast0<G>((uint32_t)c[t] & LMASK);
#endif
#pragma unroll
      for (int k = 0; k < W; k++) c[(t + k) % W] += (uint64_t)N[k] * q;
      const uint64_t v = c[t];
      c[(t + 1) % W] += v >> LB;
#if ZKP_AND_DPP
      c[t] = (uint64_t)(from_next<G>((uint32_t)v, gl) & lm);
#else
      c[t] = (uint64_t)from_next<G>((uint32_t)v & LMASK, gl);
#endif
    }
  }
  uint64_t cy = 0;
#pragma unroll
  for (int k = 0; k < W; k++) {
    const uint64_t t = c[k] + cy;
    X[k] = (uint32_t)t & LMASK;
    cy = t >> LB;
  }
  X[0] += from_prev<G>((uint32_t)cy, gl);
}

// ---------------------------------------------------------------- two quotient digits per chain step (latency engine)
// The one dependency chain through a product is the quotient digit: bottom column -> digit -> broadcast -> N*q into the next
// column -> carry -> next digit.  A lone wavefront per SIMD (the latency engine's small calls) waits it out W*G times per
// product.  With a modulus multiple M~~ = M * n2, n2 = -M^-1 mod 2^58, the two low limbs of the operand N are 2^29 - 1 and BOTH
// digits of a pair of sub-steps follow from the bottom two columns without a multiplication and without each other:
//     q0 = low29(c[t]),    q1 = low29(c[t+1] + (c[t] >> 29))
// (c[t] + q0 (2^29 - 1) = (hi + q0) 2^29, so column t hands hi + q0 up; column t+1 then holds c[t+1] + q0 (2^29 - 1) + hi + q0 =
// c[t+1] + hi (mod 2^29), whatever q0 is).  Two broadcasts go out together and the chain advances two limbs of B per trip.  W is
// odd (9): a block of W sub-steps is (W-1)/2 pairs and one single step, for which M~~ == -1 (mod 2^29) serves as well.
// Same R = 2^(29 G W) as montmul, so values move freely between the two products; M~~ < 2^58 M needs 2 M~~ < R / 4, i.e. the
// modulus 60 bits below the capacity: Paillier's n^2 (4096 bits in 4176, 8192 in 8352) fits, a 2048-bit modulus in 2088 does
// not (k_setup records which, ConstLayout::OFF_ST + 2).
template <int G>
__device__ __forceinline__ void montmul2(uint32_t (&R)[W], const uint32_t (&A)[W], const uint32_t* ldsB, const uint32_t (&N)[W], int gl) {
  static_assert(!COL_NEEDS_CARE, "the double-digit product is built for the short column window of the latency engine");
  uint64_t c[W];
#pragma unroll
  for (int k = 0; k < W; k++) c[k] = 0;
#pragma unroll 1
  for (int s = 0; s < G; s++) {
#pragma unroll
    for (int t = 0; t + 1 < W; t += 2) {
      constexpr int dummy = 0; (void)dummy;
      const int i0 = t % W, i1 = (t + 1) % W, i2 = (t + 2) % W;
      const uint32_t b0 = ldsB[s * BLK + t], b1 = ldsB[s * BLK + t + 1];
      c[i0] += (uint64_t)A[0] * b0;
      c[i1] += (uint64_t)A[1] * b0;
      c[i1] += (uint64_t)A[0] * b1;
      const uint32_t q0 = bcast0<G>((uint32_t)c[i0] & LMASK);
      const uint32_t q1 = bcast0<G>((uint32_t)(c[i1] + (c[i0] >> LB)) & LMASK);
#pragma unroll
      for (int k = 2; k < W; k++) c[(t + k) % W] += (uint64_t)A[k] * b0;
#pragma unroll
      for (int k = 1; k < W - 1; k++) c[(t + 1 + k) % W] += (uint64_t)A[k] * b1;
#pragma unroll
      for (int k = 0; k < W; k++) c[(t + k) % W] += (uint64_t)N[k] * q0;
#pragma unroll
      for (int k = 0; k < W - 1; k++) c[(t + 1 + k) % W] += (uint64_t)N[k] * q1;
      {
        const uint64_t v = c[i0];
        c[i1] += v >> LB;
        c[i0] = (uint64_t)from_next<G>((uint32_t)v & LMASK, gl);        // slot i0 is column t + W from here on
      }
      c[i0] += (uint64_t)A[W - 1] * b1;                                 // the top products of the second digit
      c[i0] += (uint64_t)N[W - 1] * q1;
      {
        const uint64_t v = c[i1];
        c[i2] += v >> LB;
        c[i1] = (uint64_t)from_next<G>((uint32_t)v & LMASK, gl);
      }
    }
    if constexpr (W & 1) {
      constexpr int t = W - 1;
      const uint32_t b = ldsB[s * BLK + t];
#pragma unroll
      for (int k = 0; k < W; k++) c[(t + k) % W] += (uint64_t)A[k] * b;
      const uint32_t q = bcast0<G>((uint32_t)c[t] & LMASK);
#pragma unroll
      for (int k = 0; k < W; k++) c[(t + k) % W] += (uint64_t)N[k] * q;
      const uint64_t v = c[t];
      c[(t + 1) % W] += v >> LB;
      c[t] = (uint64_t)from_next<G>((uint32_t)v & LMASK, gl);
    }
  }
  uint64_t cy = 0;
#pragma unroll
  for (int k = 0; k < W; k++) {
    const uint64_t t = c[k] + cy;
    R[k] = (uint32_t)t & LMASK;
    cy = t >> LB;
  }
  R[0] += from_prev<G>((uint32_t)cy, gl);
}

// ---------------------------------------------------------------- the squaring of the double-digit product (latency engine)
// montmul2's schedule — both quotient digits of a pair of sub-steps from the bottom two columns, before the bulk of the pair's
// products — with montsqr's tournament deciding which products exist at all and which are doubled.  W is odd (9): distances
// 1 .. (W-1)/2 form a regular tournament; 5 instead of 9 multiply-adds in the A half of every sub-step.
// sqr_mult(t, k): what limb k is multiplied by at position t of a block: 0 = not at all, 1 = b (same position), 2 = 2 b
constexpr int sqr_mult(int t, int k) {
  if (k == t) return 1;
  const int d = (k - t + W) % W, H = W / 2;
  const bool take = (W & 1) ? (d >= 1 && d <= H) : ((d >= 1 && d < H) || (d == H && t < H));
  return take ? 2 : 0;
}
template <int G>
__device__ __forceinline__ void montsqr2(uint32_t (&X)[W], const uint32_t* ldsB /* the staged copy of X */, const uint32_t (&N)[W], int gl) {
  static_assert(!COL_NEEDS_CARE, "the double-digit product is built for the short column window of the latency engine");
  uint64_t c[W];
#pragma unroll
  for (int k = 0; k < W; k++) c[k] = 0;
  // c[col] += X[k] * (b | 2 b), or nothing: the position pair (tt, k) decides at compile time
  // (plain `if`: t and k are indices of fully unrolled loops, the conditions fold to constants)
#define ZKP_SQ(tt, k, col, b1x, b2x) do { const int m_ = sqr_mult((tt), (k)); if (m_ == 1) c[(col)] += (uint64_t)X[(k)] * (b1x); \
                                           else if (m_ == 2) c[(col)] += (uint64_t)X[(k)] * (b2x); } while (0)
#pragma unroll 1
  for (int s = 0; s < G; s++) {
#pragma unroll
    for (int t = 0; t + 1 < W; t += 2) {
      const int i0 = t % W, i1 = (t + 1) % W, i2 = (t + 2) % W;
      const uint32_t b0 = ldsB[s * BLK + t], b1 = ldsB[s * BLK + t + 1];
      const uint32_t b0d = b0 + b0, b1d = b1 + b1;
      ZKP_SQ(t, 0, i0, b0, b0d);                                         // the products of columns t and t + 1 first: they decide the digits
      ZKP_SQ(t, 1, i1, b0, b0d);
      ZKP_SQ(t + 1, 0, i1, b1, b1d);
      const uint32_t q0 = bcast0<G>((uint32_t)c[i0] & LMASK);
      const uint32_t q1 = bcast0<G>((uint32_t)(c[i1] + (c[i0] >> LB)) & LMASK);
#pragma unroll
      for (int k = 2; k < W; k++) ZKP_SQ(t, k, (t + k) % W, b0, b0d);
#pragma unroll
      for (int k = 1; k < W - 1; k++) ZKP_SQ(t + 1, k, (t + 1 + k) % W, b1, b1d);
#pragma unroll
      for (int k = 0; k < W; k++) c[(t + k) % W] += (uint64_t)N[k] * q0;
#pragma unroll
      for (int k = 0; k < W - 1; k++) c[(t + 1 + k) % W] += (uint64_t)N[k] * q1;
      {
        const uint64_t v = c[i0];
        c[i1] += v >> LB;
        c[i0] = (uint64_t)from_next<G>((uint32_t)v & LMASK, gl);        // slot i0 is column t + W from here on
      }
      ZKP_SQ(t + 1, W - 1, i0, b1, b1d);                                 // the top products of the second digit
      c[i0] += (uint64_t)N[W - 1] * q1;
      {
        const uint64_t v = c[i1];
        c[i2] += v >> LB;
        c[i1] = (uint64_t)from_next<G>((uint32_t)v & LMASK, gl);
      }
    }
    if constexpr (W & 1) {
      constexpr int t = W - 1;
      const uint32_t b = ldsB[s * BLK + t];
      const uint32_t bd = b + b;
#pragma unroll
      for (int k = 0; k < W; k++) ZKP_SQ(t, k, (t + k) % W, b, bd);
      const uint32_t q = bcast0<G>((uint32_t)c[t] & LMASK);
#pragma unroll
      for (int k = 0; k < W; k++) c[(t + k) % W] += (uint64_t)N[k] * q;
      const uint64_t v = c[t];
      c[(t + 1) % W] += v >> LB;
      c[t] = (uint64_t)from_next<G>((uint32_t)v & LMASK, gl);
    }
  }
#undef ZKP_SQ
  uint64_t cy = 0;
#pragma unroll
  for (int k = 0; k < W; k++) {
    const uint64_t t = c[k] + cy;
    X[k] = (uint32_t)t & LMASK;
    cy = t >> LB;
  }
  X[0] += from_prev<G>((uint32_t)cy, gl);
}

// ---------------------------------------------------------------- representation changes
// 32-bit words (LDS, `nwords` valid, zero padded up to at least nwords+2) -> this lane's W limbs
__device__ __forceinline__ void limbs_from_words(uint32_t (&v)[W], const uint32_t* words, int gl) {
#pragma unroll
  for (int k = 0; k < W; k++) {
    const int bit = (gl * W + k) * LB;
    const int w0 = bit >> 5, off = bit & 31;
    const uint64_t x = (uint64_t)words[w0] | ((uint64_t)words[w0 + 1] << 32);
    v[k] = (uint32_t)(x >> off) & LMASK;
  }
}

// Exact normalisation of a redundant value (limbs < 2^32) across the group: afterwards
// every limb is < 2^29.  Ripples are rare (limb == 2^29-1 with carry-in), the loop is
// wave-uniform (ballot) and runs at most G+1 times.
template <int G> __device__ __forceinline__ void normalize_exact(uint32_t (&v)[W], int gl) {
  for (;;) {
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < W; k++) {
      uint32_t t = v[k] + c;
      v[k] = t & LMASK;
      c = t >> LB;
    }
    uint32_t cin = from_prev<G>(c, gl);
    v[0] += cin;
    if (!__any(cin != 0)) break;
  }
  // v[0] may still equal 2^29 exactly only if cin was added in the last round, which the loop excludes
}

// exact limbs (this lane's block) -> LDS as 29-bit limb array [L] (+3 zero words of padding by the caller)
// then each lane assembles out words [gl*WPL, gl*WPL+WPL) of the 32-bit representation.
template <int G, int NWORDS>
__device__ __forceinline__ void words_from_limbs(uint32_t* out_words /*LDS, NWORDS*/, uint32_t* scratch /*LDS, L+3*/,
                                                 const uint32_t (&v)[W], int gl) {
  constexpr int L = Geo<G>::L;
#pragma unroll
  for (int k = 0; k < W; k++) scratch[gl * W + k] = v[k];
  if (gl == 0) { scratch[L] = 0; scratch[L + 1] = 0; scratch[L + 2] = 0; }
  wave_lds_fence();
  constexpr int WPL = (NWORDS + G - 1) / G;
#pragma unroll
  for (int t = 0; t < WPL; t++) {
    const int w = gl * WPL + t;
    if (w < NWORDS) {
      const int bit = w * 32;
      const int i0 = bit / LB, off = bit - i0 * LB;
      uint64_t x = (uint64_t)scratch[i0] | ((uint64_t)scratch[i0 + 1] << LB);
      uint32_t word = (uint32_t)(x >> off);
      if (off > 2 * LB - 32) word |= scratch[i0 + 2] << (2 * LB - off);   // third limb needed when off > 26
      out_words[w] = word;
    }
  }
  wave_lds_fence();
}

}  // namespace zkp
