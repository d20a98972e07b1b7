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
// The raw pair goes to k_basen_finish like that of k_enc_basen.  40 of the 64 lanes work: this is the kernel of calls that leave the chip
// idle anyway (up to 2 wavefronts per SIMD: 8 proofs); from there on k_enc_basen<8> (8 Enc per wavefront) and the throughput engine take over.
//
// Replaces, like k_enc: kzen-paillier EncryptWithChosenRandomness at range_proof.rs:165-169,179-183,280-291,330-334 — for the call shape
// of the reference's own benchmark, ONE RangeProofNi proof (benches/all.rs:55-71).
#pragma once
#include "kernels_basen.hpp"

namespace zkp {

#if ZKP_W == 9

namespace r2l {
constexpr int G = 8;                       // lanes per n-sized integer
constexpr int AW = G * BLK;                // words per LDS area (one staged integer: 8 lane blocks of 12 words)
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
constexpr int LDS_WORDS = NAREAS * AW;
}  // namespace r2l

// R = (X * B + c0 + q M~) / R' over the group's 8 lanes; digits go, four at a time, over the consumed limbs of B in the lanes of `qmask`
// (bn_mul_impl of kernels_basen.hpp with everything that varies turned into data)
__device__ __forceinline__ void r2l_product(uint32_t (&R)[W], const uint32_t (&X)[W], const uint32_t* ldsB, const uint32_t (&NT)[W], uint64_t (&c)[W],
                                            uint64_t qmask, int gl) {
  constexpr int G = r2l::G;
#pragma unroll 1
  for (int s = 0; s < G; s++) {
    uint32_t qd[4];
    const uint32_t row_addr = lds_byte_address(ldsB + s * BLK);
#pragma unroll
    for (int t = 0; t < W; t++) {
      const uint32_t b = ldsB[s * BLK + t];
#pragma unroll
      for (int k = 0; k < W; k++) c[(t + k) % W] += (uint64_t)X[k] * b;
      const uint32_t q = bcast0<G>((uint32_t)c[t] & LMASK);
      qd[t & 3] = q;
      ZKP_BN_QWRITE(qmask, row_addr, t, qd);
#pragma unroll
      for (int k = 0; k < W; k++) c[(t + k) % W] += (uint64_t)NT[k] * q;
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

// One wavefront per workgroup, one Enc per claim.  `ok`: the key's OFF_OK word (a key the form does not take: return at once, the launch
// behind this one — which claims from the same counter — does the work).
__global__ void __launch_bounds__(64) k_enc_basen_r2l(EncArgs a, const uint32_t* __restrict__ bcst, uint32_t* __restrict__ raw) {
  using namespace r2l;
  using BC = BnConst<G>;
  constexpr int L = Geo<G>::L, E = 2 * L;
  if (!bcst[BC::OFF_OK]) return;
  __shared__ __align__(16) uint32_t lds[LDS_WORDS];
  const int lane = threadIdx.x & 63, role = lane >> 3, gl = lane & 7;
  auto area = [&](int i) -> uint32_t* { return lds + i * AW; };
  auto blk = [&](int i) -> uint32_t* { return lds + i * AW + gl * BLK; };
  const int kw = a.n_bits / 32;
  const uint64_t count = a.count_ptr ? (uint64_t)*a.count_ptr : a.count;
  uint32_t NT[W];
  load_limbs_global<G>(NT, bcst + BC::OFF_MT, gl);
  const uint32_t n1 = bcst[BC::OFF_NI];
  // constants into their areas (every group stores the same blocks: harmless)
  {
    uint32_t T[W];
    load_limbs_global<G>(T, bcst + BC::OFF_C3, gl);  lds_store_block(blk(C3A), T);
    load_limbs_global<G>(T, bcst + BC::OFF_R1A, gl); lds_store_block(blk(ONEA), T);
    load_limbs_global<G>(T, bcst + BC::OFF_R1B, gl); lds_store_block(blk(ONEB), T);
#pragma unroll
    for (int k = 0; k < W; k++) T[k] = 0;
    lds_store_block(blk(ZERO), T);
    if (gl == 0) T[0] = 1;
    lds_store_block(blk(INT1), T);
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
    // ---- the item's r and m as limb blocks: r -> DA1 (B's multiplier of slot 0) and the register operand of A's first slot; m -> MM
    {
      uint32_t T[W];
      wave_lds_fence();
      for (int w = lane; w < 72; w += 64) area(WBUF)[w] = (w < it.rw) ? it.pr[w] : 0u;
      wave_lds_fence();
      limbs_from_words_at(T, area(WBUF), gl * W);
      wave_lds_fence();
      lds_store_block(blk(DA1), T);
      lds_store_block(blk(RL), T);
      for (int w = lane; w < 72; w += 64) area(WBUF)[w] = (it.pm && w < it.mw) ? it.pm[w] : 0u;
      wave_lds_fence();
      limbs_from_words_at(T, area(WBUF), gl * W);
      wave_lds_fence();
      lds_store_block(blk(MM), T);
      load_limbs_global<G>(T, bcst + BC::OFF_RRA, gl); lds_store_block(blk(SA1), T);      // slot -1: A multiplies r by RR's a part (parity of -1: 1)
      load_limbs_global<G>(T, bcst + BC::OFF_RRB, gl); lds_store_block(blk(SB), T);       // slot  0: B multiplies r by RR's b part
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
      const uint64_t qmask = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((actA ? 1u : 0u) | ((actC ? 1u : 0u) << 16)));
      uint32_t X[W], R[W];
      uint64_t c[W];
      wave_lds_fence();
      lds_load_block(X, blk(xa));
      {
        uint32_t Q[W], C3[W];
        lds_load_block(Q, blk(qa < 0 ? (int)ZERO : qa));
        lds_load_block(C3, blk(qa < 0 ? (int)ZERO : (int)C3A));
        const uint32_t n1e = qa < 0 ? 0u : n1;
#pragma unroll
        for (int i = 0; i < W; i++) c[i] = (uint64_t)C3[i] + (uint64_t)((1u << LB) - Q[i]) * n1e;
      }
      wave_lds_fence();
      r2l_product(R, X, area(ba), NT, c, qmask, gl);
      wave_lds_fence();
      lds_store_block(blk(d0), R);
      wave_lds_fence();
      // ---- what follows a result
      if (role == 0 && actA) {
        lds_store_block(blk(SE0 + prev), R);
        uint32_t T[W];
        if (nbit(k + 1)) {
#pragma unroll
          for (int i = 0; i < W; i++) T[i] = R[i];
        } else lds_load_block(T, blk(ONEA));
        lds_store_block(blk(SC0 + prev), T);                         // C's multiplier of the next slot: a_(k+1), or the Montgomery one
        if (k < 0) lds_store_block(blk(PC0 + par), R);               // the accumulator starts as s_0: p_1 = a_0 (n is odd), read by C in slot 1
#pragma unroll
        for (int i = 0; i < W; i++) T[i] = R[i];
        bn_double<G>(T, gl);
        lds_store_block(blk(DA0 + prev), T);
      }
      if (role == 1 && actB && k == 0) lds_store_block(blk(QQ), R);  // q_1 = b_0
      if (role == 2 && actC) {
        if (fin1) { if (item < count) store_limbs_global<G>(raw + item * E, R, gl); }
        else lds_store_block(blk(PX), R);
      }
      if (role == 2 && k == t_bits) {                                // p_t once more, for E's product by m in the next slot (C stages PX itself there)
        uint32_t T[W];
        lds_load_block(T, blk(PX));
        lds_store_block(blk(SE0 + (t_bits & 1)), T);
      }
      if (role == 4 && actE && !fin1) {
        uint32_t D[W];
        lds_load_block(D, blk(RD));
        bn_add<G>(R, R, D, gl);
        lds_store_block(blk(QQ), R);
      }
      if (role == 3 && fin2) {
        uint32_t U[W];
        lds_load_block(U, blk(UU));
        bn_add<G>(R, R, U, gl);
        store_limbs_global<G>(raw + item * E + L, R, gl);
      }
    }
  }
}

#endif  // ZKP_W == 9

}  // namespace zkp
