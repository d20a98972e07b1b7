// kernels_basen.hpp — Paillier's arithmetic modulo n^2 in BASE-n FORM, for launches under ONE key (RangeProofNi prove / verify under
// the verifier's key, zkp_paillier_enc*_batch with n_stride == 0).
//
//     x (mod n^2)   is held as the pair   (a, b),   x = a + b n  (mod n^2).
//
// (a1 + b1 n)(a2 + b2 n) = a1 a2 + (a1 b2 + a2 b1) n: the b1 b2 term is a multiple of n^2, and the quotient of a1 a2 by n — which belongs
// to the b part — falls out of the Montgomery reduction of a1 a2 itself.  With M~ = n * n1 the Orup multiple of n and R' = 2^(29 L),
//
//     a1 a2 + Q M~ = a' R'          (Q: the quotient digits the systolic product produces anyway)
//  => a1 a2 = a' R' - (Q n1) n
//  => x1 x2 / R'  =  a'  +  [ (a1 b2 + a2 b1 - Q n1) / R'  mod n ] n        (mod n^2),
//
// so a Montgomery product modulo n^2 is THREE n-sized Montgomery products (a1 a2 | a1 b2 - Q n1 | a2 b1) and a squaring is a squaring plus
// one product (a^2 | 2 a b - Q n1): 6 and 3.5 units of (L x L limb products) against the 8 and 6 of the same operations on 2L-limb
// operands (bigint29.hpp montmul / montsqr on n^2) — 0.58 of the multiply-adds of a ladder.  The kernels were at the multiply-add
// issue limit of the board's power budget, so the multiply-adds are the time.
//
// The term -Q n1 enters the b side column by column without a subtraction: column i starts at (2^29 - Q_i) n1 + C3_i, where
// C3 == -n1 2^29 (R' - 1) / (2^29 - 1) (mod n) is a per-key constant: sum_i (2^29 - Q_i) n1 2^(29 i) + C3 == -Q n1 (mod n).
// An integer of L = G W limbs lives on G lanes exactly as in bigint29.hpp (G = 2: n of 2048 bits, G = 4: 4096 bits), a wavefront holds
// 64 / G pairs.  tests/basen_model.py states the arithmetic executably; tests/test_basen_model.py checks it against pow().
//
// Replaces, like k_enc: kzen-paillier EncryptWithChosenRandomness at range_proof.rs:165-169,179-183,280-291,330-334.
#pragma once
#include "kernels_modexp.hpp"

// rows of the staged operand per iteration of a product body's loop (A/B switch: a taken branch costs a LONE wavefront ~20 ns, csrc/microbench/lone_wave_hops.hip;
// the throughput engine's bodies are three to a kernel and at the instruction cache's limit: one row per iteration there)
// Latency engine (W = 9, calls of 21 ... 40 proofs at one to three wavefronts per SIMD): all rows — one 8192-Enc launch of k_enc_basen<8> (one wavefront per
// SIMD) 27.28 -> 25.29 (2 rows) -> 24.21 (4) -> 23.29 ms (8), two wavefronts per SIMD 37.5 -> 35.8 (profiles/r05/r2l5/ab_row_loops_of_the_w9_base_n_kernels.jsonl)
// (as _Pragma: a macro inside `#pragma unroll` survives -save-temps unexpanded and fails in the second pass)
#define ZKP_UNROLL_STR(x) #x
#define ZKP_UNROLL_(n) _Pragma(ZKP_UNROLL_STR(unroll n))
#define ZKP_UNROLL(n) ZKP_UNROLL_(n)
#ifndef ZKP_BN_ROW_UNROLL
#if ZKP_W == 9
#define ZKP_BN_ROW_UNROLL 8
#else
#define ZKP_BN_ROW_UNROLL 1
#endif
#endif
namespace zkp {

// ---- per-key constants (uint32 words in global memory), geometry G = lanes per n-sized integer
//   MT[L] | C3[L] | RRa[L] | RRb[L] | N[L] | R2n[L] | ONE[L] (the integer 1) | R1a[L] | R1b[L] (the Montgomery form of 1) | n1, ok, -, - | 2 L words of set-up scratch
template <int G> struct BnConst {
  static constexpr int L = Geo<G>::L;
  static constexpr int OFF_MT = 0, OFF_C3 = L, OFF_RRA = 2 * L, OFF_RRB = 3 * L, OFF_N = 4 * L, OFF_R2N = 5 * L, OFF_ONE = 6 * L, OFF_R1A = 7 * L, OFF_R1B = 8 * L, OFF_NI = 9 * L, OFF_OK = 9 * L + 1;
  static constexpr int WORDS = 9 * L + 4;
  static constexpr int STRIDE = WORDS + 2 * L;      // words between the records of a batch of keys
};

// ---- per-group LDS: the staged a | the staged b; every conversion area (32-bit words in and out, limb scratch) aliases them and is only
// used before the first and after the last product of an item.  The group stride decides the bank pattern of the 16-byte accesses of a
// quarter wavefront (64 banks x 4 bytes = 16 units of 16 bytes): the broadcast read of the staged operand (every group at the same
// offset: S g distinct mod 16) and the block accesses that stage or reload a whole integer (lane gl of group g at S g + 9 gl + i: all
// 16 lanes distinct).  G = 2: S = 38 units (6 g and 6 g + 9 over g = 0..7: the even and the odd residues); G = 4: S = 76 (12 g +
// {0, 9, 2, 11} over g = 0..3).  An odd S — the first choice here, 37 — collides in every block access: S (g - g') == 9 (mod 16) always
// has a solution (1.1 bank-conflict cycles per LDS cycle measured, profiles/r04/).
template <int G> struct BnLds {
  static constexpr int L = Geo<G>::L;
  static constexpr int OFF_A = 0, OFF_B = G * BLK;
  // W = 36: 152 / 304 words (strides of 38 / 76 units, above).  W = 9 (the latency build: 8 / 16 lanes per n-sized integer, BLK = 12 words):
  // 2 G BLK + 8 = 200 / 392 words — 50 / 98 units, 2 mod 16: the 8 / 4 groups of a wavefront read the staged operand from distinct banks
  // W = 18 (the mid engine: 4 / 8 lanes, BLK = 20): 2 G BLK + 12 = 172 / 332 words — 43 / 83 units, odd: the 16 / 8 groups of a wavefront on distinct banks
  static constexpr int WORDS = W == 36 ? (G == 2 ? 152 : 304) : W == 18 ? 2 * G * BLK + 12 : 2 * G * BLK + 8;  // >= 2 G BLK + 3 (the limb scratch of the output conversion)
  static constexpr int NW2 = (L / 72) * 128;                    // 32-bit words of a value mod n^2
  static_assert(W != 36 || G == 2 || G == 4, "group strides are chosen per geometry");
  static_assert(L == 72 || L == 144, "an n-sized integer is 72 or 144 limbs");
  static_assert(NW2 + 8 <= WORDS && 2 * L + 3 <= WORDS && 2 * G * BLK <= WORDS, "conversion areas alias the operand areas");
  static constexpr int THREADS = 256;
  static constexpr int GROUPS_PER_BLOCK = THREADS / G;
  static constexpr int BYTES_PER_BLOCK = (WORDS * GROUPS_PER_BLOCK + G * BLK) * 4; // + the workgroup's copy of C3 (G lane blocks)
};

// the FAST product of bigint29.hpp with one more 58-bit value in a column (the initial (2^29 - Q_i) n1 + C3_i of the b side)
constexpr uint64_t COL_FAST_SN_LIMIT_BN = ((~0ull) - (1ull << 36) - ((1ull << LB) + 16) * (W * (1ull << LB) + 16) - (1ull << (2 * LB)) - (1ull << LB)) >> LB;

#ifndef ZKP_BN_PEND_PRODUCTS
#define ZKP_BN_PEND_PRODUCTS 0   /* measured round 5 (profiles/r05/traffic_split.json): 40 % less global traffic, 5 % SLOWER */
#endif
#ifndef ZKP_BN_SQR_FENCE
#define ZKP_BN_SQR_FENCE 12
#endif

template <int G> struct Bn {
  uint32_t NT[W];        // this lane's block of M~ = n * n1
  uint32_t n1;           // -n^-1 mod 2^29
  int gl;
  uint32_t* lds;
  const uint32_t* c3;    // C3 in the workgroup's LDS (one copy behind the groups' areas: the b side reads it at the start of every product)
  const uint32_t* cst;   // the key's BnConst record
  __device__ __forceinline__ uint32_t* A() const { return lds + BnLds<G>::OFF_A; }
  __device__ __forceinline__ uint32_t* B() const { return lds + BnLds<G>::OFF_B; }
};

// lds_base: GROUPS x BnLds::WORDS words of group areas, then G BLK words for C3 (copied here by the whole workgroup; ends in a barrier)
template <int G> __device__ __forceinline__ void bn_init(Bn<G>& g, uint32_t* lds_base, const uint32_t* cst, int groups = BnLds<G>::GROUPS_PER_BLOCK) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  g.gl = lane & (G - 1);
  g.lds = lds_base + (wave * (64 / G) + lane / G) * BnLds<G>::WORDS;
  uint32_t* c3 = lds_base + groups * BnLds<G>::WORDS;
  for (int w = threadIdx.x; w < Geo<G>::L; w += blockDim.x) c3[(w / W) * BLK + (w % W)] = cst[BnConst<G>::OFF_C3 + w];     // (lane blocks of BLK words, as lds_load_block reads them)
  __syncthreads();
  g.c3 = c3;
  g.cst = cst;
  load_limbs_global<G>(g.NT, cst + BnConst<G>::OFF_MT, g.gl);
  g.n1 = cst[BnConst<G>::OFF_NI];
}

template <int G> __device__ __forceinline__ void bn_stage(const Bn<G>& g, uint32_t* area, const uint32_t (&v)[W]) {
  wave_lds_fence();
  lds_store_block(area + g.gl * BLK, v);
  wave_lds_fence();
}

// ---- where the quotient digits of an a side wait for the b side: IN PLACE of the limbs of the staged operand the product has already
// consumed.  Sub-step (s, t) reads limb s W + t of the staged operand for the last time; its quotient digit takes that word.  Lane 0 of
// the group writes four digits at a time, one ds_write_b128 per ds_read_b128 of the operand.  No register holds a digit beyond its four
// sub-steps, so the a side of a squaring keeps the register footprint of bigint29.hpp's montsqr — the first build of this file carried
// the 36 digits of a lane through the product bodies in registers and paid 450 scratch accesses per block for it.
// The write is ONE instruction under an execution mask that only lane 0 of every group survives, set and restored around it in a
// single asm statement.  Written as an `if`, the compiler turns the 18 writes of a product body into s_and_saveexec regions and the body
// takes 100 - 300 scratch accesses; written for all lanes (the others to a dummy unit) the address needs a multiply-add per write.
// `qmask`: the lanes that write (0: a product that keeps no digits).  Measured steps: profiles/r04/basen/README.md.
template <int G> __device__ __forceinline__ uint64_t q_write_mask(bool capture) {
  const uint64_t lanes0 = G == 2 ? 0x5555555555555555ull : G == 4 ? 0x1111111111111111ull : G == 8 ? 0x0101010101010101ull : 0x0001000100010001ull;
  return capture ? lanes0 : 0ull;
}
__device__ __forceinline__ void q_write(uint64_t qmask, uint32_t addr /* LDS byte address */, const uint32_t (&qd)[4]) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = {qd[0], qd[1], qd[2], qd[3]};
  uint64_t saved;
  // (the two instructions behind the write are its wait states: a VALU instruction must not overwrite the data registers of a DS write of
  // more than 64 bits in the two issue slots after it, and the compiler's hazard recogniser does not look inside an asm statement)
  asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b128 %2, %3\n\ts_mov_b64 exec, %0\n\ts_nop 0" : "=&s"(saved) : "s"(qmask), "v"(addr), "v"(v) : "scc");
}
__device__ __forceinline__ uint32_t lds_byte_address(const uint32_t* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)p; }
// (W = 9: eight digits go out in two 16-byte writes, the ninth on its own)
__device__ __forceinline__ void q_write1(uint64_t qmask, uint32_t addr /* LDS byte address */, uint32_t q) {
  uint64_t saved;
  asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b32 %2, %3\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "s"(qmask), "v"(addr), "v"(q) : "scc");
}
// (W = 18: sixteen digits in four 16-byte writes, the last two together)
__device__ __forceinline__ void q_write2(uint64_t qmask, uint32_t addr /* LDS byte address */, uint32_t q0, uint32_t q1) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 v = {q0, q1};
  uint64_t saved;
  asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b64 %2, %3\n\ts_mov_b64 exec, %0\n\ts_nop 0" : "=&s"(saved) : "s"(qmask), "v"(addr), "v"(v) : "scc");
}
#define ZKP_BN_QWRITE(qmask, row_addr, t, qd)                                                                            \
  do {                                                                                                                  \
    if (((t) & 3) == 3) q_write((qmask), (row_addr) + ((t) - 3) * 4, qd);                                               \
    else if ((W & 3) == 1 && (t) == W - 1) q_write1((qmask), (row_addr) + (t) * 4, qd[(t) & 3]);                        \
    else if ((W & 3) == 2 && (t) == W - 1) q_write2((qmask), (row_addr) + ((t) - 1) * 4, qd[((t) - 1) & 3], qd[(t) & 3]); \
  } while (0)
static_assert((W & 3) != 3, "the digit writes cover blocks of 4 k, 4 k + 1 or 4 k + 2 limbs");

// ---- TWO quotient digits per trip through the broadcast (W = 9, the latency engine's 8- and 16-lane groups) — a measured variant, OFF.
// Lane 0 of a group works out the next digit by itself: its bottom two columns are complete, and with M~ == -1 (mod 2^29) the carry out of
// the bottom column is (c0 >> 29) + q0, so
//     q1 = low29( c1 + (c0 >> 29) + q0 (N_1 + 1) )
// — one 32-bit multiply more on lane 0's chain, one broadcast latency less per pair of sub-steps; same digits, same values as the one-digit
// loops below (bigint29.hpp montmul2 / montsqr2 do this under a 58-bit Orup multiple, for which an n-sized integer of 2048 bits in 2088 has no
// room; this form needs none).  It is what made the one-Enc-per-wavefront ladder 10 % faster (kernels_basen_r2l.hpp: product2), where a
// sub-step is 12 multiply-adds.  Here a sub-step is 14 – 18 of them on 8 Enc at once and the broadcast is a smaller share: bit-exact
// (tests/test_gpu_basen.py green with it), 32 proofs 33.1 / 31.1 -> 31.5 / 29.5 ms (one wavefront per SIMD), 96 proofs 61.0 / 60.0 ->
// 62.7 / 62.4 (three) — profiles/r05/experiments/ab_two_quotient_digits_w9_basen.txt.  Not worth a second set of kernels.
#ifndef ZKP_BN_TWO_DIGITS
#define ZKP_BN_TWO_DIGITS 0
#endif
constexpr bool BN_TWO_DIGITS = ZKP_BN_TWO_DIGITS && W == 9;

// digits of a pair (t, t + 1) / of the odd last sub-step into LDS
#define ZKP_BN_QWRITE_PAIR(qmask, row_addr, t, qd)                                                                      \
  do {                                                                                                                  \
    if ((((t) + 1) & 3) == 3) q_write((qmask), (row_addr) + ((t) - 2) * 4, qd);                                         \
  } while (0)

template <int G, bool SQR>
__device__ __forceinline__ void bn_core2(uint64_t (&c)[W], const uint32_t (&X)[W], uint32_t* ldsB, const uint32_t (&N)[W], uint64_t writes, int gl) {
  const uint32_t n1p = N[1] + 1;
  // c[col] += X[k] * (b | 2 b), or nothing: for a squaring the position pair (tt, k) decides at compile time (bigint29.hpp sqr_mult)
#define ZKP_BN_P(tt, k, col, b1x, b2x) do { const int m_ = SQR ? sqr_mult((tt), (k)) : 1; if (m_ == 1) c[(col)] += (uint64_t)X[(k)] * (b1x); \
                                            else if (m_ == 2) c[(col)] += (uint64_t)X[(k)] * (b2x); } while (0)
ZKP_UNROLL(ZKP_BN_ROW_UNROLL)
  for (int s = 0; s < G; s++) {
    uint32_t qd[4];
    const uint32_t row_addr = lds_byte_address(ldsB + s * BLK);
#pragma unroll
    for (int t = 0; t + 1 < W; t += 2) {
      const int i0 = t % W, i1 = (t + 1) % W, i2 = (t + 2) % W;
      const uint32_t b0 = ldsB[s * BLK + t], b1 = ldsB[s * BLK + t + 1];
      const uint32_t b0d = b0 + b0, b1d = b1 + b1;
      ZKP_BN_P(t, 0, i0, b0, b0d);                                         // the products of columns t and t + 1 first: they decide the digits
      ZKP_BN_P(t, 1, i1, b0, b0d);
      ZKP_BN_P(t + 1, 0, i1, b1, b1d);
      const uint32_t q0l = (uint32_t)c[i0] & LMASK;
      const uint32_t q1l = ((uint32_t)c[i1] + (uint32_t)(c[i0] >> LB) + q0l * n1p) & LMASK;
      const uint32_t q0 = bcast0<G>(q0l);
      const uint32_t q1 = bcast0<G>(q1l);
      qd[t & 3] = q0; qd[(t + 1) & 3] = q1;
      ZKP_BN_QWRITE_PAIR(writes, row_addr, t, qd);
#pragma unroll
      for (int k = 2; k < W; k++) ZKP_BN_P(t, k, (t + k) % W, b0, b0d);
#pragma unroll
      for (int k = 1; k < W - 1; k++) ZKP_BN_P(t + 1, k, (t + 1 + k) % W, b1, b1d);
#pragma unroll
      for (int k = 0; k < W; k++) c[(t + k) % W] += (uint64_t)N[k] * q0;
#pragma unroll
      for (int k = 0; k < W - 1; k++) c[(t + 1 + k) % W] += (uint64_t)N[k] * q1;
      {
        const uint64_t v = c[i0];
        c[i1] += v >> LB;
        c[i0] = (uint64_t)from_next<G>((uint32_t)v & LMASK, gl);        // slot i0 is column t + W from here on
      }
      ZKP_BN_P(t + 1, W - 1, i0, b1, b1d);                                 // the top products of the second digit
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
      for (int k = 0; k < W; k++) ZKP_BN_P(t, k, (t + k) % W, b, bd);
      const uint32_t q = bcast0<G>((uint32_t)c[t] & LMASK);
      qd[t & 3] = q;
      ZKP_BN_QWRITE(writes, row_addr, t, qd);
#pragma unroll
      for (int k = 0; k < W; k++) c[(t + k) % W] += (uint64_t)N[k] * q;
      const uint64_t v = c[t];
      c[(t + 1) % W] += v >> LB;
      c[t] = (uint64_t)from_next<G>((uint32_t)v & LMASK, gl);
    }
  }
#undef ZKP_BN_P
}

// ---- the a side of a squaring: R = X * X / R' on M~ (bigint29.hpp montsqr, out of place), quotient digits into ldsB (see above)
template <int G>
__device__ __forceinline__ void bn_sqr_a(uint32_t (&R)[W], const uint32_t (&X)[W], uint32_t* ldsB, const uint32_t (&N)[W], const Bn<G>& g) {
  constexpr int H = W / 2;
  const int gl = g.gl;
  const uint64_t writes = q_write_mask<G>(true);
  uint64_t c[W];
#pragma unroll
  for (int k = 0; k < W; k++) c[k] = 0;
  if constexpr (BN_TWO_DIGITS) bn_core2<G, true>(c, X, ldsB, N, writes, gl);
  else
ZKP_UNROLL(ZKP_BN_ROW_UNROLL)
  for (int s = 0; s < G; s++) {
    uint32_t qd[4];
    const uint32_t row_addr = lds_byte_address(ldsB + s * BLK);
#pragma unroll
    for (int t = 0; t < W; t++) {
      const uint32_t b = ldsB[s * BLK + t];
      c[(2 * t) % W] += (uint64_t)X[t] * b;
      const uint32_t b2 = b + b;
#pragma unroll
      for (int k = 0; k < W; k++) {
        const int d = (k - t + W) % W;
        const bool take = (W & 1) ? (d >= 1 && d <= H) : ((d >= 1 && d < H) || (d == H && t < H));      // (bigint29.hpp montsqr: odd W is a regular tournament)
        if (take) c[(t + k) % W] += (uint64_t)X[k] * b2;
      }
      const uint32_t q = bcast0<G>((uint32_t)c[t] & LMASK);
      qd[t & 3] = q;
      ZKP_BN_QWRITE(writes, row_addr, t, qd);
      // (a scheduling barrier every 12 sub-steps: with the digit writes in the stream the scheduler's look-ahead cost this body 11 scratch
      // reloads of operand limbs per block; fenced it keeps 3)
      if (t % ZKP_BN_SQR_FENCE == ZKP_BN_SQR_FENCE - 1) __builtin_amdgcn_sched_barrier(0);
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

// the initial columns of a b side: c_k = C3_k + (2^29 - Q_k) n1 over this lane's block of the digits an a side left in ldsQ
// C3G: C3 from the key's record in global memory (launches with per-proof keys: there is no room for a copy per group in LDS), else from
// the workgroup's copy in LDS
template <int G, bool C3G = false> __device__ __forceinline__ void bn_b_init(uint64_t (&c)[W], const Bn<G>& g, const uint32_t* ldsQ) {
  uint32_t Q[W];
  uint32_t C3[W];
  if constexpr (C3G) load_limbs_global<G>(C3, g.cst + BnConst<G>::OFF_C3, g.gl);      // (issued before the LDS read of the digits: the longer latency first)
  wave_lds_fence();
  lds_load_block(Q, ldsQ + g.gl * BLK);
  if constexpr (!C3G) lds_load_block(C3, g.c3 + g.gl * BLK);
#pragma unroll
  for (int k = 0; k < W; k++) c[k] = (uint64_t)C3[k] + (uint64_t)((1u << LB) - Q[k]) * g.n1;
}

// ---- the n-sized product on M~ (bigint29.hpp montmul<G, true, false>): R = (A * B + c0) / R'.
//   mode 0: c0 = 0      mode 1: c0 = 0, quotient digits into ldsB (the a side of a product)
//   mode 2: c0 = the b-side columns from the digits an a side left in ldsQ (bn_b_init); `pend` (if any) — that a side's result — is staged
//           into ldsQ once the digits are read
// FENCE: a scheduling barrier every FENCE sub-steps (bigint29.hpp SCHED_FENCE)
// (a scheduling barrier every 18 sub-steps of the product body: A/B on two boxes against 12 — bigint29.hpp's choice for the n^2-sized
// ladder —: +0.8 ... +1.0 % verifies/s; 9: the same; 24: -0.7 %; none: 100+ scratch accesses per block.  profiles/r04/basen/ab_fences.txt)
#ifndef ZKP_BN_MUL_FENCE
#define ZKP_BN_MUL_FENCE 18
#endif
template <int G, bool PEND, bool C3G = false, int FENCE = ZKP_BN_MUL_FENCE>
__device__ __forceinline__ void bn_mul_impl(uint32_t (&R)[W], const uint32_t (&A)[W], uint32_t* ldsB, const Bn<G>& g, int mode, uint32_t* ldsQ,
                                            const uint32_t (&pend)[W], bool pend_now = true) {
  const int gl = g.gl;
  const uint64_t writes = q_write_mask<G>(mode == 1);
  uint64_t c[W];
  if (mode == 2) {
    bn_b_init<G, C3G>(c, g, ldsQ);
    if constexpr (PEND) { if (pend_now) bn_stage<G>(g, ldsQ, pend); }
  } else {
#pragma unroll
    for (int k = 0; k < W; k++) c[k] = 0;
  }
  if constexpr (BN_TWO_DIGITS) bn_core2<G, false>(c, A, ldsB, g.NT, writes, gl);
  else
ZKP_UNROLL(ZKP_BN_ROW_UNROLL)
  for (int s = 0; s < G; s++) {
    uint32_t qd[4];
    const uint32_t row_addr = lds_byte_address(ldsB + s * BLK);
#pragma unroll
    for (int t = 0; t < W; t++) {
      if constexpr (FENCE > 0) { if (t % FENCE == 0 && t) __builtin_amdgcn_sched_barrier(0); }
      const uint32_t b = ldsB[s * BLK + t];
#pragma unroll
      for (int k = 0; k < W; k++) c[(t + k) % W] += (uint64_t)A[k] * b;
      const uint32_t q = bcast0<G>((uint32_t)c[t] & LMASK);
      qd[t & 3] = q;
      ZKP_BN_QWRITE(writes, row_addr, t, qd);
#pragma unroll
      for (int k = 0; k < W; k++) c[(t + k) % W] += (uint64_t)g.NT[k] * q;
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
template <int G, bool C3G = false> __device__ __forceinline__ void bn_mul(uint32_t (&R)[W], const uint32_t (&A)[W], uint32_t* ldsB, const Bn<G>& g, int mode, uint32_t* ldsQ = nullptr) {
  bn_mul_impl<G, false, C3G>(R, A, ldsB, g, mode, ldsQ, A);
}
// the b side of a squaring: mode 2, and the a side's result `pend` goes into ldsQ once the digits there are read
template <int G, bool C3G = false> __device__ __forceinline__ void bn_mul_b(uint32_t (&R)[W], const uint32_t (&A)[W], uint32_t* ldsB, const Bn<G>& g, uint32_t* ldsQ, const uint32_t (&pend)[W]) {
  bn_mul_impl<G, true, C3G>(R, A, ldsB, g, 2, ldsQ, pend);
}

// X <- 2 X, limbs normalised again (the carry out of a lane's block lands on limb 0 of the next lane; 2 X < R')
template <int G> __device__ __forceinline__ void bn_double(uint32_t (&X)[W], int gl) {
  uint32_t cy = 0;
#pragma unroll
  for (int k = 0; k < W; k++) {
    const uint32_t t = (X[k] << 1) + cy;
    X[k] = t & LMASK;
    cy = t >> LB;
  }
  X[0] += from_prev<G>(cy, gl);
}
// R <- A + B, limbs normalised again
template <int G> __device__ __forceinline__ void bn_add(uint32_t (&R)[W], const uint32_t (&A)[W], const uint32_t (&Bv)[W], int gl) {
  uint32_t cy = 0;
#pragma unroll
  for (int k = 0; k < W; k++) {
    const uint32_t t = A[k] + Bv[k] + cy;
    R[k] = t & LMASK;
    cy = t >> LB;
  }
  R[0] += from_prev<G>(cy, gl);
}

// ---- whole base-n operations for the kernels that are not throughput critical (set-up, diagnostics); k_enc_basen runs the same steps
// as slots of one loop.
// squaring of the running value: X = a (registers), A() = staged a, B() = staged b.  Afterwards the same for the square.
template <int G> __device__ __forceinline__ void bn_square(const Bn<G>& g, uint32_t (&X)[W]) {
  uint32_t A2[W], R[W];
  bn_sqr_a<G>(A2, X, g.A(), g.NT, g);              // digits over the staged a
  bn_double<G>(X, g.gl);                           // 2 a, resident operand of the b side
  bn_mul_b<G>(R, X, g.B(), g, g.A(), A2);          // (2 a b + quotient term) / R'; the new a part staged once the digits are read
  bn_stage<G>(g, g.B(), R);
#pragma unroll
  for (int k = 0; k < W; k++) X[k] = A2[k];
}
// product  (running value, staged in A() | B())  x  (ra, rb) given as limb arrays in global memory.  Result in (A2, B2); the staged a
// is consumed (its area holds the quotient digits afterwards), nothing is staged.
template <int G> __device__ __forceinline__ void bn_product(const Bn<G>& g, uint32_t (&A2)[W], uint32_t (&B2)[W], const uint32_t* ra, const uint32_t* rb) {
  uint32_t T[W], U[W], R[W];
  load_limbs_global<G>(T, rb, g.gl);
  bn_mul<G>(U, T, g.A(), g, 0);                    // rb a / R'
  load_limbs_global<G>(T, ra, g.gl);
  bn_mul<G>(A2, T, g.A(), g, 1);                   // ra a / R', digits out
  bn_mul<G>(R, T, g.B(), g, 2, g.A());             // (ra b + quotient term) / R'
  bn_add<G>(B2, R, U, g.gl);
}

// ---- plain products modulo n (the two canonicalisations at the end of an item): bigint29.hpp montmul on n itself, SAFE columns
template <int G> __device__ __forceinline__ void bn_mod_n(const Bn<G>& g, uint32_t (&x)[W], const uint32_t (&N)[W], uint32_t* area) {
  // x (any value below R' / 2^11) -> its canonical residue modulo n, exact limbs
  uint32_t t[W], r[W];
  load_limbs_global<G>(t, g.cst + BnConst<G>::OFF_R2N, g.gl);
  bn_stage<G>(g, area, t);
  montmul<G, false, true>(r, x, area, N, g.n1, g.gl);          // x R' mod n  (< 2 n)
#pragma unroll
  for (int k = 0; k < W; k++) t[k] = 0;
  if (g.gl == 0) t[0] = 1;
  bn_stage<G>(g, area, t);
  montmul<G, false, true>(x, r, area, N, g.n1, g.gl);          // <= n
  normalize_exact<G>(x, g.gl);
  bool eq = true;
#pragma unroll
  for (int k = 0; k < W; k++) eq = eq && (x[k] == N[k]);
  const unsigned long long m = __ballot(eq);
  const int lane = threadIdx.x & 63;
  const unsigned long long gm = ((1ull << G) - 1) << (lane & ~(G - 1));
  if ((m & gm) == gm) {
#pragma unroll
    for (int k = 0; k < W; k++) x[k] = 0;
  }
}

// ---- HI R' + LO = A * B + init, exact limbs (A: registers, B staged, init: this lane's block of a value below R').  No reduction: the
// bottom column of lane 0 is a finished digit of the product at every sub-step.
template <int G>
__device__ __forceinline__ void bn_mul_full(uint32_t (&HI)[W], uint32_t (&LO)[W], const uint32_t (&A)[W], const uint32_t* ldsB, const uint32_t (&init)[W], int gl) {
  uint64_t c[W];
#pragma unroll
  for (int k = 0; k < W; k++) { c[k] = init[k]; LO[k] = 0; }
#pragma unroll 1
  for (int s = 0; s < G; s++) {
    const bool mine = s == gl;
#pragma unroll
    for (int t = 0; t < W; t++) {
      const uint32_t b = ldsB[s * BLK + t];
#pragma unroll
      for (int k = 0; k < W; k++) c[(t + k) % W] += (uint64_t)A[k] * b;
      const uint64_t v = c[t];
      const uint32_t lo = (uint32_t)v & LMASK;
      const uint32_t d = bcast0<G>(lo);
      LO[t] = mine ? d : LO[t];
      c[(t + 1) % W] += v >> LB;
      const uint32_t up = from_next<G>(lo, gl);
      c[t] = gl == G - 1 ? 0ull : (uint64_t)up;
    }
  }
  uint64_t cy = 0;
#pragma unroll
  for (int k = 0; k < W; k++) {
    const uint64_t t = c[k] + cy;
    HI[k] = (uint32_t)t & LMASK;
    cy = t >> LB;
  }
  HI[0] += from_prev<G>((uint32_t)cy, gl);
  normalize_exact<G>(HI, gl);
}

// limbs [base, base + W) of the value whose 32-bit words sit in LDS (zero padded two words beyond the last limb read)
__device__ __forceinline__ void limbs_from_words_at(uint32_t (&v)[W], const uint32_t* words, int base) {
#pragma unroll
  for (int k = 0; k < W; k++) {
    const int bit = (base + k) * LB;
    const int w0 = bit >> 5, off = bit & 31;
    const uint64_t x = (uint64_t)words[w0] | ((uint64_t)words[w0 + 1] << 32);
    v[k] = (uint32_t)(x >> off) & LMASK;
  }
}

// cooperative copy of `nwords` 32-bit words global -> the group's LDS from offset 0, zero padded to `upto`
template <int G> __device__ __forceinline__ void bn_fetch_words(const Bn<G>& g, const uint32_t* src, int nwords, int upto) {
  wave_lds_fence();
  for (int w = g.gl; w < upto; w += G) g.lds[w] = (src && w < nwords) ? src[w] : 0u;
  wave_lds_fence();
}
template <int G> __device__ __forceinline__ void bn_load_value(const Bn<G>& g, uint32_t (&v)[W], const uint32_t* src, int nwords) {
  constexpr int L = Geo<G>::L;
  bn_fetch_words<G>(g, src, nwords, (L * LB) / 32 + 3);
  limbs_from_words_at(v, g.lds, g.gl * W);
  wave_lds_fence();
}

// ---- the raw pair (a', b') that leaves the ladder kernel — a' <= M~ = n n1, b' < 4 M~, value a' + b' n (mod n^2) — as the canonical
// residue a0 + bf n below n^2, exact limbs LO (low L limbs) | HI.  a' = a0 + k n with k < 2^29: a0 = a' mod n and, n1 being -n^-1,
// k = (a0 - a') n1 mod 2^29; bf = (b' + k) mod n.
template <int G>
__device__ __forceinline__ void bn_canonical(const Bn<G>& g, uint32_t (&A1)[W], uint32_t (&B1)[W], uint32_t (&LO)[W], uint32_t (&HI)[W]) {
  uint32_t N[W];
  load_limbs_global<G>(N, g.cst + BnConst<G>::OFF_N, g.gl);
  normalize_exact<G>(A1, g.gl);
  const uint32_t a_low = bcast0<G>(A1[0]);
  bn_mod_n<G>(g, A1, N, g.A());
  const uint32_t a0_low = bcast0<G>(A1[0]);
  const uint32_t kq = ((a0_low - a_low) * g.n1) & LMASK;
  if (g.gl == 0) B1[0] += kq;
  normalize_exact<G>(B1, g.gl);
  bn_mod_n<G>(g, B1, N, g.A());
  bn_stage<G>(g, g.A(), B1);
  bn_mul_full<G>(HI, LO, N, g.A(), A1, g.gl);      // a0 + bf n
}

// ---- set-up of the base-n constants: one group per key, from the ConstLayout<G> records of the moduli n (k_setup<G>, square = 0).
// all_ok (preset to 1 by the host): cleared when any key of the batch does not qualify — a launch with per-proof keys runs in base-n form
// only when all of them do.  One wavefront per workgroup, BN_SETUP_LDS_WORDS of LDS per group.
constexpr int BN_SETUP_LDS_WORDS = 1024;
template <int G>
__global__ void __launch_bounds__(64) k_setup_basen(const uint32_t* __restrict__ ncst_base /* ConstLayout<G> records */, uint32_t* __restrict__ out_base /* BnConst<G> records */,
                                                    uint64_t count, uint32_t* __restrict__ all_ok,
                                                    const uint32_t* __restrict__ up_tag /* of the k_setup record this one is derived from */, uint32_t* __restrict__ tag) {
  using CL = ConstLayout<G>;
  using BC = BnConst<G>;
  constexpr int L = Geo<G>::L, CAP = Geo<G>::CAPBITS;
  // ONE key whose record is already here (kernels_modexp.hpp: the tag of a constants buffer): k_setup found its modulus unchanged and this
  // record was derived from that very record (same epoch) and QUALIFIED (a key outside the form never leaves a valid tag: all_ok has to be
  // cleared again on every call).
  if (tag && up_tag) {
    if (up_tag[0] == SETUP_TAG_MAGIC && tag[0] == SETUP_TAG_MAGIC && tag[1] == up_tag[5]) return;
  }
  extern __shared__ __align__(16) uint32_t lds_all[];
  const uint64_t gid = (uint64_t)blockIdx.x * (64 / G) + threadIdx.x / G;
  const bool live = gid < count;
  const uint64_t key = live ? gid : count - 1;      // idle groups redo the last key (uniform wave operations) and store nothing new
  const uint32_t* ncst = ncst_base + key * CL::WORDS;
  uint32_t* out = out_base + key * BC::STRIDE;
  uint32_t* lds_raw = lds_all + (threadIdx.x / G) * BN_SETUP_LDS_WORDS;
  Bn<G> g;
  g.gl = threadIdx.x & (G - 1);
  g.lds = lds_raw;
  g.c3 = lds_raw + 704;                             // (the group's areas and the word buffers of the doubling walk stay below)
  g.cst = out;
  uint32_t N[W], T[W], U[W], V[W];
  load_limbs_global<G>(N, ncst + CL::OFF_N, g.gl);
  load_limbs_global<G>(g.NT, ncst + CL::OFF_MT, g.gl);
  g.n1 = ncst[CL::OFF_NI];
  bool ok = ncst[CL::OFF_ST] == 0;
  // A modulus k_setup rejected (even: there is no Montgomery form) must not be computed WITH: the lane above a group takes the group's
  // lowest limb for zero after every sub-step (bigint29.hpp from_next), which holds for a proper reduction and for zeros, not for an even
  // modulus — its group would corrupt the constants of the key next to it.  Such a group runs on zeros (and its record says `not ok`).
  const bool sane = ok;
  if (!sane) {
#pragma unroll
    for (int k = 0; k < W; k++) { N[k] = 0; g.NT[k] = 0; }
    g.n1 = 0;
  }
  {
    uint64_t sn = 0;
#pragma unroll
    for (int k = 0; k < W; k++) sn += g.NT[k];
    const unsigned long long gmk = ((1ull << G) - 1) << ((threadIdx.x & 63) & ~(G - 1));
    ok = ok && (__ballot(sn <= COL_FAST_SN_LIMIT_BN) & gmk) == gmk;
  }
  // the modulus must leave the b parts room: bit length of n at least CAP / 2 + 64 (R' / n is the b part of the Montgomery one)
  store_limbs_global<G>(out + BC::OFF_MT, g.NT, g.gl);
  store_limbs_global<G>(out + BC::OFF_N, N, g.gl);
  load_limbs_global<G>(T, ncst + CL::OFF_R2, g.gl);
  if (!sane) {
#pragma unroll
    for (int k = 0; k < W; k++) T[k] = 0;
  }
  store_limbs_global<G>(out + BC::OFF_R2N, T, g.gl);
  if (g.gl == 0) out[BC::OFF_NI] = g.n1;
#pragma unroll
  for (int k = 0; k < W; k++) T[k] = (g.gl == 0 && k == 0) ? 1u : 0u;
  store_limbs_global<G>(out + BC::OFF_ONE, T, g.gl);
  // C3 = -(n1 2^29 S) mod n, S = the integer whose L digits are all 1:  S v / R' -> S v -> canonical -> n - .
  {
#pragma unroll
    for (int k = 0; k < W; k++) { U[k] = 1; V[k] = 0; }
    if (g.gl == 0) V[1] = g.n1;
    bn_stage<G>(g, g.A(), V);
    montmul<G, false, true>(T, U, g.A(), N, g.n1, g.gl);            // S v / R'
    load_limbs_global<G>(V, out + BC::OFF_R2N, g.gl);             // (R^2 mod n as stored above: zeros for a rejected modulus)
    bn_stage<G>(g, g.A(), V);
    montmul<G, false, true>(U, T, g.A(), N, g.n1, g.gl);            // S v mod n (< 2 n)
    bn_mod_n<G>(g, U, N, g.A());                                    // canonical (reads R2N from `out`: stored above by this lane)
#pragma unroll
    for (int k = 0; k < W; k++) T[k] = N[k] + (LMASK - U[k]);
    if (g.gl == 0) T[0] += 1;
    normalize_exact<G>(T, g.gl);                                    // n - val (the carry out of the top limb is 2^(29 L): dropped)
    if (!sane) {
#pragma unroll
      for (int k = 0; k < W; k++) T[k] = 0;
    }
    store_limbs_global<G>(out + BC::OFF_C3, T, g.gl);
    wave_lds_fence();
    lds_store_block(lds_raw + 704 + g.gl * BLK, T);
    wave_lds_fence();
  }
  __threadfence_block();
  // Montgomery form of 1: R' = rho0 + rho1 n.  rho0 = R' mod n is k_setup's R1; rho1 = floor(R' / n) by the same doubling walk, on
  // lane 0, in words: (a, b) <- (2 a, 2 b), a >= n ? (a - n, b + 1).  Starts at 2^(bl - 1) < n.
  uint32_t* aw = g.lds;                 // a words [NWH + 2]
  constexpr int NWH = (L / 72) * 64;    // words of n
  uint32_t* bw = g.lds + NWH + 4;       // b words
  uint32_t* nw = g.lds + 2 * (NWH + 4); // n words
  wave_lds_fence();
  if (g.gl == 0) for (int w = 0; w < 3 * (NWH + 4); w++) g.lds[w] = 0;
  wave_lds_fence();
  {
    // n as words from its limbs
    uint32_t* scr = g.lds + 3 * (NWH + 4);
    words_from_limbs<G, NWH>(nw, scr, N, g.gl);
  }
  int bl = 0;
  if (g.gl == 0) {
    int top = NWH - 1;
    while (top > 0 && nw[top] == 0) top--;
    bl = nw[top] ? top * 32 + (32 - __clz(nw[top])) : 0;
    if (bl >= 2) {
      aw[(bl - 1) >> 5] = 1u << ((bl - 1) & 31);
      const int nwd = top + 2;
      for (int it = 0; it < CAP - (bl - 1); it++) {
        uint32_t carry = 0;
        for (int w = 0; w < nwd; w++) { const uint32_t t = aw[w]; aw[w] = (t << 1) | carry; carry = t >> 31; }
        int ge = 1;
        for (int w = nwd - 1; w >= 0; w--) if (aw[w] != nw[w]) { ge = aw[w] > nw[w]; break; }
        if (ge) {
          uint32_t borrow = 0;
          for (int w = 0; w < nwd; w++) { const uint64_t t = (uint64_t)aw[w] - nw[w] - borrow; aw[w] = (uint32_t)t; borrow = (uint32_t)(t >> 63); }
        }
        carry = (uint32_t)ge;
        for (int w = 0; w < NWH + 2; w++) { const uint32_t t = bw[w]; bw[w] = (t << 1) | carry; carry = t >> 31; }
      }
    }
  }
  wave_lds_fence();
  bl = (int)bcast0<G>((uint32_t)bl);
  ok = ok && bl >= CAP / 2 + 64 && bl <= NWH * 32;
  uint32_t OA[W], OB[W];
  limbs_from_words_at(OA, aw, g.gl * W);
  limbs_from_words_at(OB, bw, g.gl * W);
  if (!sane) {
#pragma unroll
    for (int k = 0; k < W; k++) { OA[k] = 0; OB[k] = 0; }
  }
  wave_lds_fence();
  // RR = R'^2 mod n^2 = (Montgomery form of 2)^CAP in the Montgomery domain: square-and-multiply on pairs.  The pair is kept in a small
  // table in global memory behind the constants (entry 0: the Montgomery form of 2, the multiplier).
  uint32_t* tab = out + BC::WORDS;                  // 2 L words of scratch behind the record
  uint32_t X[W], Y[W];
  bn_add<G>(X, OA, OA, g.gl);                       // 2 rho0 (< 2 n)
  bn_add<G>(Y, OB, OB, g.gl);
  store_limbs_global<G>(tab, X, g.gl);
  store_limbs_global<G>(tab + L, Y, g.gl);
  bn_stage<G>(g, g.A(), X);
  bn_stage<G>(g, g.B(), Y);
  const int msb = 31 - __clz(CAP);
#pragma unroll 1
  for (int b = msb - 1; b >= 0; b--) {
    bn_square<G>(g, X);
    if ((CAP >> b) & 1) {
      uint32_t A2[W], B2[W];
      bn_product<G>(g, A2, B2, tab, tab + L);
      bn_stage<G>(g, g.A(), A2);
      bn_stage<G>(g, g.B(), B2);
#pragma unroll
      for (int k = 0; k < W; k++) X[k] = A2[k];
    }
  }
  lds_load_block(Y, g.B() + g.gl * BLK);
  store_limbs_global<G>(out + BC::OFF_RRA, X, g.gl);
  store_limbs_global<G>(out + BC::OFF_RRB, Y, g.gl);
  store_limbs_global<G>(out + BC::OFF_R1A, OA, g.gl);
  store_limbs_global<G>(out + BC::OFF_R1B, OB, g.gl);
  if (g.gl == 0) {
    out[BC::OFF_OK] = ok ? 1u : 0u;
    if (!ok) atomicAnd(all_ok, 0u);
  }
  if (tag && up_tag && gid == 0 && g.gl == 0) {
    tag[1] = up_tag[5];
    __threadfence();
    tag[0] = (ok && up_tag[0] == SETUP_TAG_MAGIC) ? SETUP_TAG_MAGIC : 0u;
  }
}

// ---- expected ciphertexts of the Mask rows of a verify launch (c_j * cipher_x mod n^2: one product modulo n^2 per row, on the
// n^2 geometry GS = 2 G): canonical words into `expected`, one 2 kw-word slot per work item.  k_enc's steps s6 - s9.
template <int GS>
__global__ void __launch_bounds__(256, ZKP_WPE) k_expected(EncArgs a, uint32_t* __restrict__ expected, const uint32_t* __restrict__ bn_ok) {
  using CL = ConstLayout<GS>;
  using LL = LdsLayout<GS>;
  if (!*bn_ok) return;
  extern __shared__ __align__(16) uint32_t lds_raw[];
  Grp<GS> g;
  grp_init<GS>(g, lds_raw);
  const int kw = a.n_bits / 32;
  const uint64_t count = a.count_ptr ? (uint64_t)*a.count_ptr : a.count;
  const uint64_t stride = (uint64_t)gridDim.x * LL::GROUPS_PER_BLOCK;
  const uint64_t rounds = (count + stride - 1) / stride;
  const uint32_t* cst = a.consts;
  load_modulus_consts<GS>(g, cst);
  for (uint64_t rd = 0; rd < rounds; rd++) {
    const uint64_t idx = rd * stride + (uint64_t)blockIdx.x * LL::GROUPS_PER_BLOCK + (threadIdx.x / GS);
    const bool live = idx < count;
    const uint64_t item = live ? idx : count - 1;
    uint64_t b;
    bool mask_row;
    const uint32_t* pexp;
    if (a.mode == 2) { b = item; mask_row = a.cipher_x != nullptr; pexp = a.c1 + item * 2 * kw; }
    else {
      b = a.item_proof[item];
      const uint32_t rw = a.item_row[item];
      const uint64_t row = b * a.ef + (rw >> 1);
      mask_row = a.resp_kind[row] != 0;
      const bool use_c1 = mask_row ? (a.resp_j[row] == 1) : !(rw & 1);
      pexp = (use_c1 ? a.c1 : a.c2) + row * 2 * kw;
    }
    if (a.const_stride) {                                     // per-proof keys: the n^2 record of this item's key
      const uint64_t i0 = (a.mode == 0 && a.half && item >= a.half) ? item - a.half : item;
      cst = a.consts + (a.mode == 1 ? b : i0 / a.items_per_key) * a.const_stride;
      load_modulus_consts<GS>(g, cst);
    }
    // (uniform control flow: groups of Open rows compute along on their c_j times 1 and store nothing)
    // (a key whose set-up was rejected — even —: its groups multiply zeros, so that the lane above them still receives a zero low limb)
    const bool valid = cst[CL::OFF_ST] == 0;
    uint32_t A[W], R[W], Y[W];
    load_value<GS>(g, A, pexp, valid ? 2 * kw : 0);
    stage_const<GS>(g, cst + CL::OFF_R2);
    mm<GS>(g, R, A);                                          // e R
    {
      uint32_t t[W];
      load_value<GS>(g, t, a.cipher_x + b * 2 * kw, (mask_row && valid) ? 2 * kw : 0);
      if (!mask_row && valid && g.gl == 0) t[0] = 1;
      stageB<GS>(g, t);
    }
    mm<GS>(g, Y, R);                                          // e * x   (< 2 n^2)
    stage_const<GS>(g, cst + CL::OFF_R2);
    mm<GS>(g, R, Y);
    stage_one<GS>(g);
    mm<GS>(g, Y, R);                                          // <= n^2
    canonical_words<GS>(g, Y, cst + CL::OFF_N);
    if (live && mask_row) for (int w = g.gl; w < 2 * kw; w += GS) expected[item * 2 * kw + w] = g.words()[w];
  }
}

// ---- Paillier Enc in base-n form: the work items, modes and outputs of k_enc (kernels_modexp.hpp: EncArgs), one key per launch.
//   k_enc_basen:     x~ = (r, 0) * RR / R'  ->  ladder x~^n  ->  * (1, m) / R'  =  the raw pair of (1 + m n) r^n   -> `raw`
//   k_basen_finish:  raw pair -> canonical a0 + bf n -> store (mode 0) | compare with the expected words (modes 1, 2)
// Two kernels, because everything the hot one does then goes through TWO product bodies — the a side of a squaring (bn_sqr_a) and the
// generic n-sized product (bn_mul) — each with ONE call site: every further inlined body costs the product loops registers (the
// first build of this file had 16 of them and 35 - 200 scratch accesses in each) and instruction-cache room.
//
// The item is a script of product SLOTS executed by one loop:
//   square   (a, b) staged:   S1  a^2 (quotient out), stage, T = 2 a                 S2  T * b (quotient in) -> stage b
//   product  staged x (ra, rb) from global memory:
//                             P0  rb * a -> U (parked in global scratch)            P1  ra * a (quotient out) -> dest a
//                             P2  ra * b (quotient in) + U -> dest b
// with dest = the staged pair (to-Montgomery, window multiplications), a table entry (table rounds) or `raw` (the final product by the
// plain pair (1, m), which also leaves the Montgomery domain).  The script bytes are those of k_sliding_schedule.
// Table slot of a group: TABS entries of 2 L words (a limbs | b limbs), then the entries (U | -) scratch, (r | -), the copy of X0^2, (1 | m).
struct BnItem {
  const uint32_t* pm; const uint32_t* pr; const uint32_t* pexp; uint32_t* pout;
  uint64_t b; int mw, rw; bool mask_row;
};
__device__ __forceinline__ BnItem bn_item(const EncArgs& a, uint64_t item, const uint32_t* expected) {
  BnItem it{};
  const int kw = a.n_bits / 32;
  it.mw = kw; it.rw = kw;
  if (a.mode == 0) {
    const bool second = a.half && item >= a.half;
    const uint64_t i = second ? item - a.half : item;
    it.mw = a.m_words < 0 ? 0 : (a.m_words ? a.m_words : kw);
    it.rw = a.r_words ? a.r_words : kw;
    it.pm = (second ? a.m2 : a.m) + i * it.mw;
    it.pr = (second ? a.r2 : a.r) + i * it.rw;
    it.pout = (second ? a.out2 : a.out) + i * 2 * kw;
  } else if (a.mode == 2) {
    it.b = item;
    it.pm = a.m + item * kw;
    it.pr = a.r + item * kw;
    it.mask_row = a.cipher_x != nullptr;
    it.pexp = it.mask_row ? expected + item * 2 * kw : a.c1 + item * 2 * kw;
  } else {
    it.b = a.item_proof[item];
    const uint32_t rw = a.item_row[item];
    const uint64_t row = it.b * a.ef + (rw >> 1);
    it.mask_row = a.resp_kind[row] != 0;
    const bool second = (rw & 1) != 0;
    it.pm = (second ? a.resp_w2 : a.resp_w1) + row * kw;
    it.pr = (second ? a.resp_r2 : a.resp_r1) + row * kw;
    const bool use_c1 = it.mask_row ? (a.resp_j[row] == 1) : !second;
    it.pexp = it.mask_row ? expected + item * 2 * kw : (use_c1 ? a.c1 : a.c2) + row * 2 * kw;
  }
  return it;
}

// key index of a work item (launches with per-proof keys; EncArgs as k_enc reads it)
__device__ __forceinline__ uint64_t bn_key(const EncArgs& a, uint64_t item, const BnItem& it) {
  if (a.mode == 1) return it.b;
  const uint64_t i = (a.mode == 0 && a.half && item >= a.half) ? item - a.half : item;
  return i / a.items_per_key;
}

constexpr int BN_TAB_ENTRIES = TABS + 4;      // window table | scratch (U, -) | (r, -) | copy of X0^2 | (1, m)

template <int G>
__global__ void __launch_bounds__(256, W == 9 ? 4 : 2) k_enc_basen(EncArgs a, const uint32_t* __restrict__ bcst, uint32_t* __restrict__ table, uint32_t* __restrict__ raw) {
  using BC = BnConst<G>;
  using BL = BnLds<G>;
  constexpr int L = Geo<G>::L, E = 2 * L;
  if (!bcst[BC::OFF_OK]) return;                               // this key is not for the base-n form: the k_enc launch beside this one runs
  extern __shared__ __align__(16) uint32_t lds_raw[];
  Bn<G> g;
  bn_init<G>(g, lds_raw, bcst);
  const uint64_t ggrp = (uint64_t)blockIdx.x * BL::GROUPS_PER_BLOCK + (threadIdx.x / G);
  uint32_t* tab = table + ggrp * (uint64_t)(BN_TAB_ENTRIES * E);
  uint32_t* scrU = tab + TABS * E;
  const int kw = a.n_bits / 32;
  const uint64_t count = a.count_ptr ? (uint64_t)*a.count_ptr : a.count;
  const int lane = threadIdx.x & 63;
  for (;;) {
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(a.work_counter, (unsigned long long)(64 / G));
    base = __shfl(base, 0);
    if (base >= count) break;
    const uint64_t idx = base + (uint64_t)(lane / G);
    const bool live = idx < count;
    const uint64_t item = live ? idx : count - 1;
    uint32_t T[W];
    {
      // operands as limbs into the group's operand entry (the conversions go through the LDS the staged pair will occupy)
      const BnItem it = bn_item(a, item, nullptr);
      bn_load_value<G>(g, T, it.pr, it.rw);
      store_limbs_global<G>(tab + (TABS + 1) * E, T, g.gl);
      bn_load_value<G>(g, T, it.pm, it.mw);
      store_limbs_global<G>(tab + (TABS + 3) * E + L, T, g.gl);
#pragma unroll
      for (int k = 0; k < W; k++) T[k] = (g.gl == 0 && k == 0) ? 1u : 0u;
      store_limbs_global<G>(tab + (TABS + 3) * E, T, g.gl);
      load_limbs_global<G>(T, bcst + BC::OFF_RRA, g.gl);
      bn_stage<G>(g, g.A(), T);
      load_limbs_global<G>(T, bcst + BC::OFF_RRB, g.gl);
      bn_stage<G>(g, g.B(), T);
    }
    uint32_t* rawdst = live ? raw + item * E : scrU;          // (surplus groups recompute the last item and park the result in their scratch)
    // ---- the script.  Squarings (6 of 7 steps) take a straight path of their own: a side, b side, stage.  Everything else is a base-n
    // product by a pair that sits in global memory, three slots through one product body.  Between slots nothing but scalars is live:
    // an a part on its way into A() waits in the group's scratch entry, not in registers.
    enum { D_STAGE = -1, D_RAW = -2 };
    int pc = 0, stage_id = 0;
#pragma unroll 1
    for (;;) {
      int src, dstk;
      bool has_p0 = true;
      if (stage_id == 0) {
        src = TABS + 1; dstk = D_STAGE; has_p0 = false;      // to the Montgomery domain: (r, -) x RR
      } else {
        int op = __builtin_amdgcn_readfirstlane((int)a.sched[pc]);
        if (op == 0) {                                       // ---- a squaring
          pc++;
          uint32_t A2[W], R[W];
          lds_load_block(T, g.A() + g.gl * BLK);
          bn_sqr_a<G>(A2, T, g.A(), g.NT, g);
          bn_double<G>(T, g.gl);
          bn_mul_b<G>(R, T, g.B(), g, g.A(), A2);
          bn_stage<G>(g, g.B(), R);
          continue;
        }
        const int type = op >> 5, e = op & 31;
        if (op == OP_END || op == OP_ZERO) {
          if (stage_id == 2) break;
          stage_id = 2;
          src = TABS + 3; dstk = D_RAW;                      // the final product: staged x the PLAIN pair (1, m) -> raw
        } else if (type == (OP_FIRST >> 5)) {
          pc++;
          uint32_t V[W];
          load_limbs_global<G>(V, tab + e * E, g.gl);
          bn_stage<G>(g, g.A(), V);
          load_limbs_global<G>(V, tab + e * E + L, g.gl);
          bn_stage<G>(g, g.B(), V);
          continue;
        } else if (type == (OP_SQ0 >> 5)) {
          // X0^2 = X0 * tab[0]: the multiplier of all table rounds.  Every one of them consumes its staged a part: keep a copy
          pc++;
          src = 0; dstk = TABS + 2;
        } else if (type == (OP_TAB >> 5)) {
          pc++;
          uint32_t V[W];
          load_limbs_global<G>(V, tab + (TABS + 2) * E, g.gl);
          bn_stage<G>(g, g.A(), V);
          src = e - 1; dstk = e;
        } else {
          pc++;
          src = e; dstk = D_STAGE;
        }
      }
      uint32_t* dp = dstk == D_RAW ? rawdst : dstk == D_STAGE ? scrU + L : tab + dstk * E;    // where the a part goes (staged: via the scratch entry)
#if ZKP_BN_PEND_PRODUCTS
      // A product whose result becomes the staged pair keeps its new a part IN REGISTERS from slot 1 to slot 2 (P: staged into A() inside
      // slot 2's product, once the digits there are read — what the squarings do with theirs), and slot 2 multiplies by the ra slot 1
      // loaded: 864 B of global traffic per window multiplication less than parking the a part in the scratch entry and loading ra twice
      // (round 5, profiles/r05/traffic_split.json).  Table and raw destinations store the a part where it belongs, as before.
      const bool to_stage = dstk == D_STAGE || dstk == TABS + 2;
      uint32_t P[W], U[W];
#pragma unroll 1
      for (int slot = has_p0 ? 0 : 1; slot < 3; slot++) {
        if (slot != 2) load_limbs_global<G>(T, tab + src * E + (slot == 0 ? L : 0), g.gl);
        uint32_t R[W];
        bn_mul_impl<G, true>(R, T, slot == 2 ? g.B() : g.A(), g, slot, g.A(), P, slot == 2 && to_stage);
        if (slot == 0) store_limbs_global<G>(scrU, R, g.gl);      // (kept in registers instead, U is spilled around both product bodies: the same bytes by another road)
        else if (slot == 1) {
          if (dstk != D_STAGE) store_limbs_global<G>(dp, R, g.gl);
#pragma unroll
          for (int k = 0; k < W; k++) P[k] = R[k];
        } else {
          if (has_p0) {
            load_limbs_global<G>(U, scrU, g.gl);
            bn_add<G>(R, R, U, g.gl);
          }
          if (to_stage) bn_stage<G>(g, g.B(), R);
          if (dstk != D_STAGE) store_limbs_global<G>(dp + L, R, g.gl);
        }
      }
#else      // the round-4 slots: the a part parked in the scratch entry, ra loaded by slot 1 and by slot 2 (A/B builds)
#pragma unroll 1
      for (int slot = has_p0 ? 0 : 1; slot < 3; slot++) {
        // (slot 2 reads the a side's digits inside the product; the a part follows them into A() afterwards, below)
        load_limbs_global<G>(T, tab + src * E + (slot == 0 ? L : 0), g.gl);
        uint32_t R[W];
        bn_mul<G>(R, T, slot == 2 ? g.B() : g.A(), g, slot, g.A());
        if (slot == 0) store_limbs_global<G>(scrU, R, g.gl);
        else if (slot == 1) store_limbs_global<G>(dp, R, g.gl);
        else {
          if (has_p0) {
            uint32_t U[W];
            load_limbs_global<G>(U, scrU, g.gl);
            bn_add<G>(R, R, U, g.gl);
          }
          if (dstk == D_STAGE || dstk == TABS + 2) {
            bn_stage<G>(g, g.B(), R);
            uint32_t V[W];
            load_limbs_global<G>(V, dp, g.gl);
            bn_stage<G>(g, g.A(), V);
          }
          if (dstk != D_STAGE) store_limbs_global<G>(dp + L, R, g.gl);
        }
      }
#endif
      if (stage_id == 0) {
        // the base is in the Montgomery domain and staged: it is entry 0 of the window table
        uint32_t V[W];
        lds_load_block(V, g.A() + g.gl * BLK);
        store_limbs_global<G>(tab, V, g.gl);
        lds_load_block(V, g.B() + g.gl * BLK);
        store_limbs_global<G>(tab + L, V, g.gl);
        stage_id = 1;
      } else if (stage_id == 2) break;
    }
  }
}

// ---- the same for launches with PER-PROOF KEYS (EncArgs::n_stride != 0; the exponent of an item is its own key): fixed 6-bit windows
// over the item's n as in kernels_modexp.hpp powm_fixed — table T[0] = the Montgomery form of 1, T[1] = x~, T[k] = T[k-1] x~, then from the
// top window down [6 squarings, one product by T[window]] — with uniform control flow whatever the keys are.  Constants come from the
// item's record (C3 from global memory at the start of every b side).  Items are PARTITIONED key by key (round 5): the items of keys the
// form does not take are appended to EncArgs::left_list and done by the n^2-sized k_enc<G, false> launch behind this one.
// Table slot of a group: 64 entries, then (U | -) scratch, (r | -), (1 | m).
constexpr int BN_KEYS_WIN = 6, BN_KEYS_TAB = 1 << BN_KEYS_WIN, BN_KEYS_TAB_ENTRIES = BN_KEYS_TAB + 3;
template <int G>
__global__ void __launch_bounds__(256, W == 9 ? 4 : 2) k_enc_basen_keys(EncArgs a, const uint32_t* __restrict__ bcst, const uint32_t* __restrict__ zero_rec, uint32_t* __restrict__ table,
                                                           uint32_t* __restrict__ raw) {
  using BC = BnConst<G>;
  using BL = BnLds<G>;
  constexpr int L = Geo<G>::L, E = 2 * L, WIN = BN_KEYS_WIN, TB = BN_KEYS_TAB;
  extern __shared__ __align__(16) uint32_t lds_raw[];
  Bn<G> g;
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    g.gl = lane & (G - 1);
    g.lds = lds_raw + (wave * (64 / G) + lane / G) * BL::WORDS;
    g.c3 = nullptr;
  }
  const uint64_t ggrp = (uint64_t)blockIdx.x * BL::GROUPS_PER_BLOCK + (threadIdx.x / G);
  uint32_t* tab = table + ggrp * (uint64_t)(BN_KEYS_TAB_ENTRIES * E);
  uint32_t* scrU = tab + TB * E;
  const int kw = a.n_bits / 32;
  const uint64_t count = a.count_ptr ? (uint64_t)*a.count_ptr : a.count;
  const int lane = threadIdx.x & 63;
  const int nwin = (a.n_bits + WIN - 1) / WIN;
  for (;;) {
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(a.work_counter, (unsigned long long)(64 / G));
    base = __shfl(base, 0);
    if (base >= count) break;
    const uint64_t idx = base + (uint64_t)(lane / G);
    const bool live = idx < count;
    const uint64_t item = live ? idx : count - 1;
    const BnItem it = bn_item(a, item, nullptr);
    const uint64_t key = bn_key(a, item, it);
    const uint32_t* pn = a.n + key * a.n_stride;               // the item's exponent
    // A key the form does not take (even, short, an M~ beyond the digit-sum bound): the item goes on the list of the n^2-sized launch behind
    // this one, and its group computes along on an ALL-ZERO record (`zero_rec`) and stores nothing.  Zeros, not whatever the set-up left:
    // the lane above a group takes that group's lowest limb for zero after every sub-step (bigint29.hpp from_next), which only a proper
    // Montgomery reduction — or zeros — guarantees; an even modulus has neither.  One such key in a batch of 4096 costs one Enc on the
    // other kernels, not the whole launch (round 4: the whole launch fell back).
    const bool key_ok = bcst[key * BC::STRIDE + BC::OFF_OK] != 0;
    const uint32_t* cst = key_ok ? bcst + key * BC::STRIDE : zero_rec;
    if (live && !key_ok && g.gl == 0) a.left_list[atomicAdd(a.left_count, 1ull)] = (uint32_t)item;
    g.cst = cst;
    load_limbs_global<G>(g.NT, cst + BC::OFF_MT, g.gl);
    g.n1 = cst[BC::OFF_NI];
    uint32_t T[W];
    {
      bn_load_value<G>(g, T, it.pr, it.rw);
      store_limbs_global<G>(tab + (TB + 1) * E, T, g.gl);
      bn_load_value<G>(g, T, it.pm, it.mw);
      store_limbs_global<G>(tab + (TB + 2) * E + L, T, g.gl);
#pragma unroll
      for (int k = 0; k < W; k++) T[k] = (g.gl == 0 && k == 0) ? 1u : 0u;
      store_limbs_global<G>(tab + (TB + 2) * E, T, g.gl);
      load_limbs_global<G>(T, cst + BC::OFF_RRA, g.gl);
      bn_stage<G>(g, g.A(), T);
      load_limbs_global<G>(T, cst + BC::OFF_RRB, g.gl);
      bn_stage<G>(g, g.B(), T);
      // T[0] = the Montgomery form of 1
      load_limbs_global<G>(T, cst + BC::OFF_R1A, g.gl);
      store_limbs_global<G>(tab, T, g.gl);
      load_limbs_global<G>(T, cst + BC::OFF_R1B, g.gl);
      store_limbs_global<G>(tab + L, T, g.gl);
    }
    uint32_t* rawdst = (live && key_ok) ? raw + item * E : scrU;
    auto window = [&](int wi) -> int {
      const int bit = wi * WIN;
      const int w0 = bit >> 5, off = bit & 31;
      const uint64_t x = (uint64_t)pn[w0] | ((uint64_t)(w0 + 1 < kw ? pn[w0 + 1] : 0u) << 32);
      return (int)((x >> off) & (TB - 1));
    };
    enum { D_STAGE = -1, D_RAW = -2 };
    int phase = 0, k = 2, wi = nwin - 2, sq_left = 0;
#pragma unroll 1
    for (;;) {
      int src, dstk;
      bool has_p0 = true;
      if (phase == 0) {
        src = TB + 1; dstk = D_STAGE; has_p0 = false;        // to the Montgomery domain: (r, -) x RR
      } else if (phase == 1) {
        if (k == TB) {
          // table complete: the running value starts at T[top window]
          const int e = window(nwin - 1);
          uint32_t V[W];
          load_limbs_global<G>(V, tab + e * E, g.gl);
          bn_stage<G>(g, g.A(), V);
          load_limbs_global<G>(V, tab + e * E + L, g.gl);
          bn_stage<G>(g, g.B(), V);
          phase = wi >= 0 ? 2 : 3;
          sq_left = WIN;
          continue;
        }
        // T[k] = T[k - 1] * x~ (staged; its a part was consumed by the round before: restage it from T[1])
        uint32_t V[W];
        load_limbs_global<G>(V, tab + E, g.gl);
        bn_stage<G>(g, g.A(), V);
        src = k - 1; dstk = k; k++;
      } else if (phase == 2) {
        if (sq_left) {                                       // ---- a squaring
          sq_left--;
          uint32_t A2[W], R[W];
          lds_load_block(T, g.A() + g.gl * BLK);
          bn_sqr_a<G>(A2, T, g.A(), g.NT, g);
          bn_double<G>(T, g.gl);
          bn_mul_b<G, true>(R, T, g.B(), g, g.A(), A2);
          bn_stage<G>(g, g.B(), R);
          continue;
        }
        src = window(wi); dstk = D_STAGE;
        wi--; sq_left = WIN;
        if (wi < 0) phase = 3;                               // (after this product)
      } else {
        src = TB + 2; dstk = D_RAW;                          // the final product: staged x the PLAIN pair (1, m) -> raw
        phase = 4;
      }
      uint32_t* dp = dstk == D_RAW ? rawdst : dstk == D_STAGE ? scrU + L : tab + dstk * E;
#pragma unroll 1
      for (int slot = has_p0 ? 0 : 1; slot < 3; slot++) {
        load_limbs_global<G>(T, tab + src * E + (slot == 0 ? L : 0), g.gl);
        uint32_t R[W];
        bn_mul<G, true>(R, T, slot == 2 ? g.B() : g.A(), g, slot, g.A());
        if (slot == 0) store_limbs_global<G>(scrU, R, g.gl);
        else if (slot == 1) store_limbs_global<G>(dp, R, g.gl);
        else {
          if (has_p0) {
            uint32_t U[W];
            load_limbs_global<G>(U, scrU, g.gl);
            bn_add<G>(R, R, U, g.gl);
          }
          if (dstk == D_STAGE) {
            bn_stage<G>(g, g.B(), R);
            uint32_t V[W];
            load_limbs_global<G>(V, dp, g.gl);
            bn_stage<G>(g, g.A(), V);
          } else store_limbs_global<G>(dp + L, R, g.gl);
        }
      }
      if (phase == 0) {
        // x~ is staged: it is T[1]
        uint32_t V[W];
        lds_load_block(V, g.A() + g.gl * BLK);
        store_limbs_global<G>(tab + E, V, g.gl);
        lds_load_block(V, g.B() + g.gl * BLK);
        store_limbs_global<G>(tab + E + L, V, g.gl);
        phase = 1;
      } else if (phase == 4) break;
    }
  }
}

template <int G>
__global__ void __launch_bounds__(256) k_basen_finish(EncArgs a, const uint32_t* __restrict__ bcst, const uint32_t* __restrict__ ok_word /* the key's OFF_OK, or the constant 1 */,
                                                      int per_key, const uint32_t* __restrict__ raw, const uint32_t* __restrict__ expected, const uint32_t* __restrict__ zero_rec) {
  using BC = BnConst<G>;
  using BL = BnLds<G>;
  constexpr int L = Geo<G>::L, E = 2 * L;
  if (!*ok_word) return;
  extern __shared__ __align__(16) uint32_t lds_raw[];
  Bn<G> g;
  bn_init<G>(g, lds_raw, bcst);
  const int kw = a.n_bits / 32;
  const uint64_t count = a.count_ptr ? (uint64_t)*a.count_ptr : a.count;
  const int lane = threadIdx.x & 63;
  const unsigned long long gmask = ((1ull << G) - 1) << (lane & ~(G - 1));
  const uint64_t stride = (uint64_t)gridDim.x * BL::GROUPS_PER_BLOCK;
  const uint64_t rounds = (count + stride - 1) / stride;
  for (uint64_t rd = 0; rd < rounds; rd++) {
    const uint64_t idx = rd * stride + (uint64_t)blockIdx.x * BL::GROUPS_PER_BLOCK + (threadIdx.x / G);
    const bool live = idx < count;
    const uint64_t item = live ? idx : count - 1;
    const BnItem it = bn_item(a, item, expected);
    bool mine = live;
    if (per_key) {
      const uint32_t* rec = bcst + bn_key(a, item, it) * BC::STRIDE;
      const bool key_ok = rec[BC::OFF_OK] != 0;
      mine = live && key_ok;                                     // (an item of a key outside the form belongs to the n^2-sized launch;
      g.cst = key_ok ? rec : zero_rec;                          //  its group computes along on zeros: see k_enc_basen_keys)
      g.n1 = g.cst[BC::OFF_NI];
    }
    uint32_t A1[W], B1[W], LO[W], HI[W];
    load_limbs_global<G>(A1, raw + item * E, g.gl);
    load_limbs_global<G>(B1, raw + item * E + L, g.gl);
    if (per_key && !mine) {
#pragma unroll
      for (int k = 0; k < W; k++) { A1[k] = 0; B1[k] = 0; }
    }
    bn_canonical<G>(g, A1, B1, LO, HI);
    if (a.mode == 0) {
      // exact limbs -> 32-bit words, through the group's LDS (2 L limbs + 3 words of zero padding)
      wave_lds_fence();
#pragma unroll
      for (int k = 0; k < W; k++) { g.lds[g.gl * W + k] = LO[k]; g.lds[L + g.gl * W + k] = HI[k]; }
      if (g.gl == 0) { g.lds[2 * L] = 0; g.lds[2 * L + 1] = 0; g.lds[2 * L + 2] = 0; }
      wave_lds_fence();
      if (mine) {
        for (int w = g.gl; w < 2 * kw; w += G) {
          const int bit = w * 32;
          const int i0 = bit / LB, off = bit - i0 * LB;
          const uint64_t x = (uint64_t)g.lds[i0] | ((uint64_t)g.lds[i0 + 1] << LB);
          uint32_t word = (uint32_t)(x >> off);
          if (off > 2 * LB - 32) word |= g.lds[i0 + 2] << (2 * LB - off);
          it.pout[w] = word;
        }
      }
      wave_lds_fence();
    } else {
      // Open rows compare with the raw c_j — a c_j >= n^2 can never equal a canonical value, which is the reference's comparison of
      // the raw BigInt; Mask rows with k_expected's canonical product
      uint32_t Ev[W];
      bn_fetch_words<G>(g, it.pexp, 2 * kw, BL::NW2 + 8);
      bool same = true;
      limbs_from_words_at(Ev, g.lds, g.gl * W);
#pragma unroll
      for (int k = 0; k < W; k++) same = same && (Ev[k] == LO[k]);
      limbs_from_words_at(Ev, g.lds, L + g.gl * W);
#pragma unroll
      for (int k = 0; k < W; k++) same = same && (Ev[k] == HI[k]);
      wave_lds_fence();
      const unsigned long long mk = __ballot(same);
      const bool pass = (mk & gmask) == gmask;
      if (a.mode == 2) { if (mine && g.gl == 0) a.verdict[it.b] = pass ? 1 : 0; }
      else if (mine && g.gl == 0 && !pass) a.verdict[it.b] = ZKP_VERDICT_REJECT;
    }
  }
}

// ---- diagnostics (include/zkp_hip_diag.h: zkp_diag_basen): one base-n operation on one pair of operands, raw limbs out
//   op 0: (r, 0) -> Montgomery form    1: x * y / R'    2: x * x / R'    3: the constants C3 | RRa | RRb | MT
template <int G>
__global__ void __launch_bounds__(64) k_diag_basen(const uint32_t* __restrict__ bcst, int op, const uint32_t* __restrict__ xa, const uint32_t* __restrict__ xb,
                                                   const uint32_t* __restrict__ ya, const uint32_t* __restrict__ yb, uint32_t* __restrict__ out, uint32_t* __restrict__ scratch) {
  using BC = BnConst<G>;
  constexpr int L = Geo<G>::L;
  extern __shared__ __align__(16) uint32_t lds_raw[];
  Bn<G> g;
  bn_init<G>(g, lds_raw, bcst, 1);
  if (threadIdx.x >= G) return;
  uint32_t X[W], Y[W];
  if (op == 3) {
    for (int part = 0; part < 4; part++) {
      const int off = part == 0 ? BC::OFF_C3 : part == 1 ? BC::OFF_RRA : part == 2 ? BC::OFF_RRB : BC::OFF_MT;
      load_limbs_global<G>(X, bcst + off, g.gl);
      store_limbs_global<G>(out + part * L, X, g.gl);
    }
    if (g.gl == 0) { out[4 * L] = bcst[BC::OFF_NI]; out[4 * L + 1] = bcst[BC::OFF_OK]; }
    return;
  }
  load_limbs_global<G>(X, xa, g.gl);
  load_limbs_global<G>(Y, xb, g.gl);
  if (op == 0) {
    uint32_t T[W];
#pragma unroll
    for (int k = 0; k < W; k++) T[k] = X[k];
    load_limbs_global<G>(X, bcst + BC::OFF_RRA, g.gl);
    bn_stage<G>(g, g.A(), X);
    load_limbs_global<G>(Y, bcst + BC::OFF_RRB, g.gl);
    bn_stage<G>(g, g.B(), Y);
    bn_mul<G>(X, T, g.A(), g, 1);
    bn_mul<G>(Y, T, g.B(), g, 2, g.A());
  } else if (op == 1) {
    bn_stage<G>(g, g.A(), X);
    bn_stage<G>(g, g.B(), Y);
    store_limbs_global<G>(scratch, X, g.gl);      // (operand y as a "table entry")
    load_limbs_global<G>(X, ya, g.gl);
    load_limbs_global<G>(Y, yb, g.gl);
    store_limbs_global<G>(scratch, X, g.gl);
    store_limbs_global<G>(scratch + L, Y, g.gl);
    __threadfence_block();
    uint32_t A2[W], B2[W];
    bn_product<G>(g, A2, B2, scratch, scratch + L);
#pragma unroll
    for (int k = 0; k < W; k++) { X[k] = A2[k]; Y[k] = B2[k]; }
  } else {
    bn_stage<G>(g, g.A(), X);
    bn_stage<G>(g, g.B(), Y);
    bn_square<G>(g, X);
    lds_load_block(Y, g.B() + g.gl * BLK);
  }
  store_limbs_global<G>(out, X, g.gl);
  store_limbs_global<G>(out + L, Y, g.gl);
}

}  // namespace zkp
