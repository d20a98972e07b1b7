/*
 * zkp_hip.h — C ABI of libzkp_hip.so: MI355X (gfx950) batched Paillier ZK-proof engine.
 *
 * This is the drop-in boundary for the hot path of ZenGo-X/zk-paillier.  The reference has
 * no FFI of its own: its proof modules (L3) call the big-integer layer (L1: curv::BigInt
 * over GMP, kzen-paillier) through ordinary Rust calls.  Every entry point below replaces
 * one of those L3->L1 call shapes, batched; the reference call site each one replaces is
 * cited as file:line relative to the reference tree.  The Rust-side binding a maintainer
 * would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - Big integers are fixed-width little-endian arrays of 32-bit limbs (limb 0 = least
 *    significant), zero padded.  kw = n_bits/32 limbs for values of the size of n
 *    (n, r, m, w, x, sigma, masked_r ...); 2*kw limbs for values mod n^2 (ciphertexts).
 *    n_bits / mod_bits is the kernel width: 2048, 4096 or 8192 bits for moduli
 *    (n^2 of a 2048-bit n is a 4096-bit modulus).  Moduli must be odd.
 *  - Batches are structure-of-arrays, element i at ptr + i*stride (stride in limbs).  A
 *    modulus/key stride of 0 means one shared modulus/key for the whole batch.
 *  - All pointers of one call live in the same memory space: host memory by default,
 *    device (HBM) memory of the context's GPU when ZKP_F_DEVICE_PTRS is set.  The caller
 *    owns every buffer; the library keeps no pointer after a call returns.
 *  - Every function returns a zkp_status.  Proof rejection is DATA (verdict byte 0), never
 *    an error code (mirrors Result<(), IncorrectProof>, src/zkproofs/errors.rs:5-13).
 *    Nothing aborts or throws across this boundary.
 *  - One ctx = one GPU + one HIP stream.  Calls on one ctx are serialised by the caller;
 *    different ctxs are independent (one per GPU / per rank) and may be driven from different
 *    host threads at the same time (zkp_multi_* does exactly that).
 *  - There is NO CPU fallback: if no gfx950 device is present zkp_ctx_create fails with
 *    ZKP_EDEVICE.
 */
#ifndef ZKP_HIP_H
#define ZKP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  ZKP_OK = 0,
  ZKP_EINVAL = 1,        /* bad width / null pointer / count overflow */
  ZKP_ENONCANONICAL = 2, /* even modulus (Montgomery needs odd) */
  ZKP_EDEVICE = 3,       /* HIP error; text via zkp_last_error_string */
  ZKP_ENOMEM = 4
} zkp_status;

enum { ZKP_F_DEVICE_PTRS = 1u };

/* verdict bytes written by the *_verify_batch entry points */
enum {
  ZKP_VERDICT_REJECT = 0,    /* Err(IncorrectProof) */
  ZKP_VERDICT_ACCEPT = 1,    /* Ok(()) */
  ZKP_VERDICT_MALFORMED = 2  /* the reference would panic (index out of bounds / assert) */
};

/* response kinds (src/zkproofs/range_proof.rs:53-78, enum Response) */
enum { ZKP_RESP_OPEN = 0, ZKP_RESP_MASK = 1 };

#define ZKP_SECURITY_PARAMETER 128 /* src/zkproofs/range_proof_ni.rs:23 */
#define ZKP_CORRECT_KEY_M2 11      /* src/zkproofs/correct_key_ni.rs:29 */

typedef struct zkp_ctx zkp_ctx;

int32_t zkp_ctx_create(int32_t device_id, zkp_ctx** out_ctx);
/* The same, launching on a stream the caller owns (a hipStream_t, e.g. the framework's current stream); the ctx never
 * destroys it. */
int32_t zkp_ctx_create_on_stream(int32_t device_id, void* hip_stream, zkp_ctx** out_ctx);
int32_t zkp_ctx_destroy(zkp_ctx* ctx);
const char* zkp_backend_name(void);               /* "hip-gfx950" */
int32_t zkp_build_limbs_per_lane(void);           /* compile-time W of the throughput kernels (36): G = 144/W lanes per 4096-bit integer */

/* Two kernel geometries serve one ctx.  The throughput engine (libzkp_hip.so itself: W = 36 limbs per lane, 4 lanes per
 * 4096-bit integer) is the one the batch metric is quoted on.  A modular exponentiation is a chain of ~2400 dependent
 * products, so ONE proof (the reference's own bench, benches/all.rs:55-71) takes as long as ~30 of them there; the latency
 * engine (libzkp_hip_lat.so next to this library, the same sources built with W = 9: 16 lanes per integer) halves that time
 * and is chosen automatically while a call's work fits 3-5 wavefronts per SIMD of it.  Results are bit-identical.
 * zkp_ctx_set_geometry: limbs_per_lane 0 = automatic (default), 36 / 9 = always that engine (ZKP_EINVAL when it is not
 * loaded).  zkp_ctx_last_geometry: limbs per lane of the engine the most recent batch call ran on.
 * zkp_ctx_latency_limbs_per_lane: W of the loaded latency engine, 0 when there is none (small calls then run on the
 * throughput engine: slower, never wrong). */
int32_t zkp_ctx_set_geometry(zkp_ctx* ctx, int32_t limbs_per_lane);
int32_t zkp_ctx_last_geometry(zkp_ctx* ctx);
int32_t zkp_ctx_latency_limbs_per_lane(zkp_ctx* ctx);
const char* zkp_last_error_string(zkp_ctx* ctx);  /* valid until the next call on ctx */
void* zkp_ctx_stream(zkp_ctx* ctx);               /* the hipStream_t every call is ordered on (a small verify call forks part of
                                                     its work to an internal second stream and joins it back before it returns) */
int32_t zkp_ctx_synchronize(zkp_ctx* ctx);
/* Host-pointer calls stage their buffers through device blocks the ctx keeps between calls (no hipMalloc / hipFree per
 * call once warm).  This frees the cached blocks (zkp_ctx_destroy does so too). */
int32_t zkp_ctx_release_staging(zkp_ctx* ctx);

/* Kernel timing of the dominant (modexp) kernels, measured with HIP events on the ctx
 * stream.  zkp_timing_reset clears the accumulators and arms event recording;
 * zkp_timing_get synchronises and returns total milliseconds, launch count and the
 * number of modular exponentiations those launches performed. */
int32_t zkp_timing_reset(zkp_ctx* ctx, int32_t enable);
int32_t zkp_timing_get(zkp_ctx* ctx, double* out_ms, uint64_t* out_launches, uint64_t* out_modexps);
/* (Profiler calibration aids are NOT part of this boundary: include/zkp_hip_diag.h.) */

/* ------------------------------------------------------------------ L1 primitives
 * out[i] = base[i]^exp[i] mod mod[i].
 * Replaces BigInt::mod_pow (src/zkproofs/correct_key_ni.rs:92; wi_dlog_proof.rs:55,81,82).
 * mod_bits in {2048,4096,8192}; exp_bits a multiple of 32, 32..mod_bits.
 * base/out: mod_bits/32 limbs each; exp: exp_bits/32 limbs; *_stride in limbs, 0 = shared. */
int32_t zkp_modexp_batch(zkp_ctx* ctx, uint32_t mod_bits, uint32_t exp_bits, uint64_t count,
                         const uint32_t* base, const uint32_t* exp, uint64_t exp_stride,
                         const uint32_t* mod, uint64_t mod_stride, uint32_t* out, uint32_t flags);

/* out[i] = a[i]*b[i] mod mod[i].  Replaces BigInt::mod_mul (wi_dlog_proof.rs:83) and the
 * `x * y % m` forms at range_proof.rs:239,245,325,327. */
int32_t zkp_modmul_batch(zkp_ctx* ctx, uint32_t mod_bits, uint64_t count, const uint32_t* a,
                         const uint32_t* b, const uint32_t* mod, uint64_t mod_stride,
                         uint32_t* out, uint32_t flags);

/* c[i] = (1 + m[i]*n[i]) * r[i]^n[i] mod n[i]^2.
 * Replaces Paillier::encrypt_with_chosen_randomness (kzen-paillier 0.4.3; call sites
 * src/zkproofs/range_proof.rs:165-169,179-183,280-291,330-334,361).
 * n, m, r: n_bits/32 limbs; out_c: 2*n_bits/32 limbs.  n_bits in {1024,2048,4096}. */
int32_t zkp_paillier_enc_batch(zkp_ctx* ctx, uint32_t n_bits, uint64_t count, const uint32_t* n,
                               uint64_t n_stride, const uint32_t* m, const uint32_t* r,
                               uint32_t* out_c, uint32_t flags);

/* ok[i] = (Enc(m[i], r[i]) == expected[i])                       when mulc_a == mulc_b == NULL
 * ok[i] = (Enc(m[i], r[i]) == mulc_a[i] * mulc_b[i] mod n[i]^2)   when expected == NULL
 * Replaces CorrectOpening::verify_opening (src/zkproofs/correct_opening.rs:17-30) and the verifier's two equality
 * shapes: `expected_c1i != encrypted_pairs.c1[i]` (range_proof.rs:280-298: the stored value is compared as it is, so
 * an expected[i] >= n^2 never matches) and `c_j[i] * cipher_x % nn` vs Enc(masked_x, masked_r) (range_proof.rs:324-337:
 * the product is reduced, factors of any size up to 2*n_bits bits).  Exactly one of (expected) / (mulc_a, mulc_b)
 * is given.  expected, mulc_a, mulc_b: [count][2kw]; out_ok: [count] bytes 0 / 1.  An even n gives ok = 0. */
int32_t zkp_paillier_enc_check_batch(zkp_ctx* ctx, uint32_t n_bits, uint64_t count, const uint32_t* n,
                                     uint64_t n_stride, const uint32_t* m, const uint32_t* r,
                                     const uint32_t* mulc_a, const uint32_t* mulc_b, const uint32_t* expected,
                                     uint8_t* out_ok, uint32_t flags);

/* ------------------------------------------------------------------ RangeProofNi
 * Batch of B non-interactive range proofs, structure-of-arrays
 * (src/zkproofs/range_proof_ni.rs:36-44 RangeProofNi; range_proof.rs:32-81 EncryptedPairs,
 * Response, Proof).  EF = error_factor rows per proof. */
typedef struct {
  uint32_t n_bits;        /* width of n: 1024, 2048 or 4096 */
  uint32_t error_factor;  /* rows per proof; prove always writes ZKP_SECURITY_PARAMETER */
  uint64_t batch;         /* B */
  uint64_t n_stride;      /* kw, or 0 = every proof uses n[0] */
  const uint32_t* n;      /* [B or 1][kw]          ek.n                        */
  const uint32_t* range;  /* [B][kw]               q                           */
  const uint32_t* ciphertext; /* [B][2kw]          c = Enc(x, r)               */
  uint32_t* c1;           /* [B][EF][2kw]          encrypted_pairs.c1          */
  uint32_t* c2;           /* [B][EF][2kw]          encrypted_pairs.c2          */
  uint8_t* resp_kind;     /* [B][EF]               ZKP_RESP_OPEN | ZKP_RESP_MASK */
  uint8_t* resp_j;        /* [B][EF]               Mask.j (0 for Open)         */
  uint32_t* resp_w1;      /* [B][EF][kw]           Open.w1 | Mask.masked_x     */
  uint32_t* resp_r1;      /* [B][EF][kw]           Open.r1 | Mask.masked_r     */
  uint32_t* resp_w2;      /* [B][EF][kw]           Open.w2 | 0                 */
  uint32_t* resp_r2;      /* [B][EF][kw]           Open.r2 | 0                 */
} zkp_range_ni_proofs;

/* Secret prover inputs.  The reference draws (w1,w2,r1,r2) from the OS RNG inside
 * generate_encrypted_pairs (range_proof.rs:136-159); the boundary takes them as inputs
 * (already coin-flip swapped) so that proving is reproducible. */
typedef struct {
  const uint32_t* x;   /* [B][kw]      secret_x */
  const uint32_t* r;   /* [B][kw]      secret_r */
  const uint32_t* w1;  /* [B][EF][kw]  */
  const uint32_t* w2;  /* [B][EF][kw]  */
  const uint32_t* r1;  /* [B][EF][kw]  */
  const uint32_t* r2;  /* [B][EF][kw]  */
} zkp_range_ni_witness;

/* RangeProofNi::prove (src/zkproofs/range_proof_ni.rs:47-82) for B proofs:
 * generate_encrypted_pairs (range_proof.rs:161-187) -> Fiat-Shamir challenge
 * (range_proof_ni.rs:58-61, utils.rs:9-22) -> generate_proof (range_proof.rs:210-252).
 * Writes c1,c2,resp_* of `p`; out_e [B][32] receives the challenge bytes left-aligned,
 * out_e_len [B] their count (leading zero digest bytes are dropped, N2); either may be null.
 * out_status [B] (nullable): 0 ok, ZKP_VERDICT_MALFORMED if the reference would panic. */
int32_t zkp_range_ni_prove_batch(zkp_ctx* ctx, const zkp_range_ni_proofs* p,
                                 const zkp_range_ni_witness* w, uint8_t* out_e,
                                 uint8_t* out_e_len, uint8_t* out_status, uint32_t flags);

/* RangeProofNi::verify_self / verify (range_proof_ni.rs:84-128 -> range_proof.rs:254-355).
 * out_verdict [B]: ZKP_VERDICT_*.  (verify()'s two assert_eq! on ek and ciphertext are the
 * host layer's job: here the statement is whatever `p` holds.) */
int32_t zkp_range_ni_verify_batch(zkp_ctx* ctx, const zkp_range_ni_proofs* p,
                                  uint8_t* out_verdict, uint32_t flags);

/* The three functions of the interactive RangeProof, usable on their own with any error_factor <= 256 and a
 * challenge supplied by the caller (the verifier's random bits of RangeProof::verifier_commit, range_proof.rs:118-126;
 * benches/all.rs:10-53 runs them with STATISTICAL_ERROR_FACTOR = 40).  e: [B][32] challenge bytes left aligned,
 * e_len: [B] byte counts (bit i of a challenge is bit 7-(i%8) of byte i/8, range_proof.rs:221,267).
 *   generate_encrypted_pairs (range_proof.rs:128-193): c1 = Enc(w1, r1), c2 = Enc(w2, r2), EF = p->error_factor rows
 *   generate_proof           (range_proof.rs:210-252): resp_* from the witness and e
 *   verifier_output          (range_proof.rs:254-355): verdicts from (c1, c2, resp_*, e) */
int32_t zkp_range_generate_encrypted_pairs_batch(zkp_ctx* ctx, const zkp_range_ni_proofs* p, const zkp_range_ni_witness* w,
                                                 uint32_t flags);
/* The Fiat-Shamir challenge alone (utils::compute_digest, src/zkproofs/utils.rs:9-22, with the glue of
 * range_proof_ni.rs:58-61 / 89-92 / 110-113): out_e[b] = to_bytes(from_bytes(SHA256(to_bytes(n) || to_bytes(c1[0..EF)) ||
 * to_bytes(c2[0..EF))))), left aligned in 32 bytes, out_e_len[b] its length (leading zero bytes of the digest are dropped, a
 * zero digest is the one byte 00).  Reads p->n, p->c1, p->c2 only. */
int32_t zkp_range_challenge_batch(zkp_ctx* ctx, const zkp_range_ni_proofs* p, uint8_t* out_e, uint8_t* out_e_len, uint32_t flags);
int32_t zkp_range_generate_proof_batch(zkp_ctx* ctx, const zkp_range_ni_proofs* p, const zkp_range_ni_witness* w,
                                       const uint8_t* e, const uint8_t* e_len, uint8_t* out_status, uint32_t flags);
int32_t zkp_range_verifier_output_batch(zkp_ctx* ctx, const zkp_range_ni_proofs* p, const uint8_t* e, const uint8_t* e_len,
                                        uint8_t* out_verdict, uint32_t flags);

/* ------------------------------------------------------------------ NiCorrectKeyProof
 * NiCorrectKeyProof::verify (src/zkproofs/correct_key_ni.rs:73-100) for B (key, proof)
 * pairs: rho_i from the SHA-256 MGF (:77-86,105-117), sigma_i^n mod n (:90-93),
 * gcd(primorial(6370), n) == 1 (:87-88).  n: [B][kw]; sigma: [B][11][kw]. */
int32_t zkp_correct_key_ni_verify_batch(zkp_ctx* ctx, uint32_t n_bits, uint64_t batch,
                                        const uint32_t* n, const uint32_t* sigma,
                                        const uint8_t* salt, uint32_t salt_len,
                                        uint8_t* out_verdict, uint32_t flags);

/* ------------------------------------------------------------------ CompositeDLogProof
 * src/zkproofs/wi_dlog_proof.rs:46-91.  N,g,ni,x: [B][kw]; y/r: [B][y_bits/32] (y_bits a
 * multiple of 32, >= 544 for honest proofs: y = r + e*s < 2^513); secret s: [B][8].
 * prove takes the 512-bit nonce r as an input (reference: BigInt::sample_below(2^512)). */
int32_t zkp_dlog_prove_batch(zkp_ctx* ctx, uint32_t n_bits, uint32_t y_bits, uint64_t batch,
                             const uint32_t* N, const uint32_t* g, const uint32_t* ni,
                             const uint32_t* secret, const uint32_t* r, uint32_t* out_x,
                             uint32_t* out_y, uint32_t flags);
int32_t zkp_dlog_verify_batch(zkp_ctx* ctx, uint32_t n_bits, uint32_t y_bits, uint64_t batch,
                              const uint32_t* N, const uint32_t* g, const uint32_t* ni,
                              const uint32_t* x, const uint32_t* y, uint8_t* out_verdict,
                              uint32_t flags);

/* ------------------------------------------------------------------ ZeroProof / CiphertextProof
 * (single-shot sigma proofs composed from the same kernels.)
 * ZeroProof (src/zkproofs/zero_enc_proof.rs:26-95): c = r^n mod n^2 encrypts zero.
 *   prove : a = Enc(0, r'), e = H(n || c || a), z = r' * r^e mod n^2           (:44-64)
 *   verify: Enc(0, z) == c^e * a mod n^2                                        (:66-94)
 * n: [B or 1][kw]; c, z, a: [B][2kw]; r, r_prime: [B][kw] (r' is sampled by the caller:
 * BigInt::sample_below(n), :45). */
int32_t zkp_zero_proof_prove_batch(zkp_ctx* ctx, uint32_t n_bits, uint64_t batch, const uint32_t* n, uint64_t n_stride,
                                   const uint32_t* c, const uint32_t* r, const uint32_t* r_prime, uint32_t* out_z,
                                   uint32_t* out_a, uint32_t flags);
int32_t zkp_zero_proof_verify_batch(zkp_ctx* ctx, uint32_t n_bits, uint64_t batch, const uint32_t* n, uint64_t n_stride,
                                    const uint32_t* c, const uint32_t* z, const uint32_t* a, uint8_t* out_verdict,
                                    uint32_t flags);

/* CiphertextProof (src/zkproofs/correct_ciphertext.rs:23-98): knowledge of (x, r) with c = Enc(x, r).
 *   prove : c' = Enc(x', r'), e = H(n || c || c'), z1 = x' + x*e (over Z), z2 = r' * r^e mod n^2   (:42-64)
 *   verify: Enc(z1, z2) == c^e * c' mod n^2                                                          (:66-97)
 * x, r, x_prime, r_prime: [B][kw]; z1: [B][kw + ZKP_Z1_EXTRA_LIMBS] (x' + x*e < 2^(n_bits+257));
 * c, z2, c_prime: [B][2kw]. */
#define ZKP_Z1_EXTRA_LIMBS 16
int32_t zkp_ciphertext_proof_prove_batch(zkp_ctx* ctx, uint32_t n_bits, uint64_t batch, const uint32_t* n, uint64_t n_stride,
                                         const uint32_t* c, const uint32_t* x, const uint32_t* r, const uint32_t* x_prime,
                                         const uint32_t* r_prime, uint32_t* out_z1, uint32_t* out_z2, uint32_t* out_c_prime,
                                         uint32_t flags);
int32_t zkp_ciphertext_proof_verify_batch(zkp_ctx* ctx, uint32_t n_bits, uint64_t batch, const uint32_t* n, uint64_t n_stride,
                                          const uint32_t* c, const uint32_t* z1, const uint32_t* z2, const uint32_t* c_prime,
                                          uint8_t* out_verdict, uint32_t flags);

/* VerlinProof (src/zkproofs/verlin_proof.rs:35-165): phi_x = c^x * c'^x' * Enc(x'', r_x).
 *   gen_phi(c, c', y, y', y'', r_y) = c^y * c'^y' * Enc(y'', r_y) mod n^2                         (:138-165)
 *   prove : phi_a = gen_phi(c, c', a, a', a'', r_a); e = H(n || c || c' || phi_x || phi_a);
 *           z = x e + a, z' = x' e + a', z'' = x'' e + a'' (over Z); r_z = r_x^e * r_a mod n^2     (:60-99)
 *   verify: gen_phi(c, c', z, z', z'', r_z) == phi_x^e * phi_a mod n^2                             (:101-135)
 * c, c_prime, phi_x, phi_a, r_z: [B][2kw]; witness x, x_prime, x_double_prime, r_x and the nonces a, a_prime,
 * a_double_prime, r_a (sampled by the caller, :61-67): [B][kw]; z, z_prime, z_double_prime: [B][kw + ZKP_Z1_EXTRA_LIMBS]. */
int32_t zkp_verlin_proof_prove_batch(zkp_ctx* ctx, uint32_t n_bits, uint64_t batch, const uint32_t* n, uint64_t n_stride,
                                     const uint32_t* c, const uint32_t* c_prime, const uint32_t* phi_x,
                                     const uint32_t* x, const uint32_t* x_prime, const uint32_t* x_double_prime, const uint32_t* r_x,
                                     const uint32_t* a, const uint32_t* a_prime, const uint32_t* a_double_prime, const uint32_t* r_a,
                                     uint32_t* out_phi_a, uint32_t* out_z, uint32_t* out_z_prime, uint32_t* out_z_double_prime,
                                     uint32_t* out_r_z, uint32_t flags);
int32_t zkp_verlin_proof_verify_batch(zkp_ctx* ctx, uint32_t n_bits, uint64_t batch, const uint32_t* n, uint64_t n_stride,
                                      const uint32_t* c, const uint32_t* c_prime, const uint32_t* phi_x, const uint32_t* phi_a,
                                      const uint32_t* z, const uint32_t* z_prime, const uint32_t* z_double_prime, const uint32_t* r_z,
                                      uint8_t* out_verdict, uint32_t flags);

/* ------------------------------------------------------------------ modular inverse (L1)
 * out[i] = a[i]^-1 mod M[i]: curv BigInt::mod_inv (GMP mpz_invert) as used by multiplication_proof.rs:95,133 and
 * correct_message.rs:53,76,141.  mod_bits in {2048, 4096, 8192}; a, M, out: mod_bits/32 words per element.
 * out_status[i]: 0 = out[i] holds the inverse, 1 = no inverse exists (mod_inv returns None; out[i] = 0),
 * 2 = outside the domain of this entry point (a >= M, M even or M < 3; out[i] = 0). */
#define ZKP_INV_OK 0
#define ZKP_INV_NONE 1
#define ZKP_INV_DOMAIN 2
int32_t zkp_modinv_batch(zkp_ctx* ctx, uint32_t mod_bits, uint64_t count, const uint32_t* a, const uint32_t* modulus, uint64_t mod_stride,
                         uint32_t* out, uint8_t* out_status, uint32_t flags);

/* ------------------------------------------------------------------ MulProof
 * multiplication_proof.rs:60-146.  kw = n_bits/32.  Statement: e_a, e_b, e_c [B][2kw].  Witness: a, b [B][kw] (c is not read
 * by the prover), r_a, r_b, r_c [B][kw].  Nonces the reference samples (:61-62), supplied by the caller: d, r_d [B][kw].
 * Proof: f [B][kw], z1, z2, e_d, e_db [B][2kw].
 * prove: out_status[b] = 0, or ZKP_VERDICT_MALFORMED where `mod_inv(..).unwrap()` (:95) panics in the reference.
 * verify: verdict bytes as above; MALFORMED where :133 panics. */
int32_t zkp_mul_proof_prove_batch(zkp_ctx* ctx, uint32_t n_bits, uint64_t batch, const uint32_t* n, uint64_t n_stride, const uint32_t* e_a,
                                  const uint32_t* e_b, const uint32_t* e_c, const uint32_t* a, const uint32_t* b, const uint32_t* r_a,
                                  const uint32_t* r_b, const uint32_t* r_c, const uint32_t* d, const uint32_t* r_d, uint32_t* out_f,
                                  uint32_t* out_z1, uint32_t* out_z2, uint32_t* out_e_d, uint32_t* out_e_db, uint8_t* out_status, uint32_t flags);
int32_t zkp_mul_proof_verify_batch(zkp_ctx* ctx, uint32_t n_bits, uint64_t batch, const uint32_t* n, uint64_t n_stride, const uint32_t* e_a,
                                   const uint32_t* e_b, const uint32_t* e_c, const uint32_t* f, const uint32_t* z1, const uint32_t* z2,
                                   const uint32_t* e_d, const uint32_t* e_db, uint8_t* out_verdict, uint32_t flags);

/* ------------------------------------------------------------------ CorrectMessageProof
 * correct_message.rs:35-162: ring proof that a ciphertext encrypts one of K valid messages (K = num_messages >= 1, the same for
 * every proof of the batch).  valid_messages [B][K][kw]; message [B][kw].  Values the reference samples, supplied by the caller:
 * r (:43), w (:65) [B][kw]; e_sim [B][K-1][8] (:59-61, 256-bit); z_sim [B][K-1][kw] (:62-64).
 * Proof: ciphertext [B][2kw], e_vec [B][K][8], z_vec [B][K][kw], a_vec [B][K][2kw].
 * prove: out_status[b] = MALFORMED where the reference panics (no valid message equals `message`: index out of bounds :74).
 * verify: MALFORMED where `assert_eq!(chal, ei_sum)` (:132) panics; REJECT / ACCEPT from :144-161. */
int32_t zkp_correct_message_prove_batch(zkp_ctx* ctx, uint32_t n_bits, uint64_t batch, uint32_t num_messages, const uint32_t* n,
                                        uint64_t n_stride, const uint32_t* valid_messages, const uint32_t* message, const uint32_t* r,
                                        const uint32_t* e_sim, const uint32_t* z_sim, const uint32_t* w, uint32_t* out_ciphertext,
                                        uint32_t* out_e_vec, uint32_t* out_z_vec, uint32_t* out_a_vec, uint8_t* out_status, uint32_t flags);
int32_t zkp_correct_message_verify_batch(zkp_ctx* ctx, uint32_t n_bits, uint64_t batch, uint32_t num_messages, const uint32_t* n,
                                         uint64_t n_stride, const uint32_t* valid_messages, const uint32_t* ciphertext, const uint32_t* e_vec,
                                         const uint32_t* z_vec, const uint32_t* a_vec, uint8_t* out_verdict, uint32_t flags);

/* ------------------------------------------------------------------ wire format
 * The reference serialises big integers as DECIMAL strings (src/serialize.rs:1-31 `bigint`, :33-78 `vecbigint`:
 * BigInt::to_str_radix(10) / from_str_radix(s, 10) = GMP mpz_get_str / mpz_set_str).  The two L1 entry points below
 * convert between that text and the fixed-width limb arrays of this ABI on the GPU, one number per lane.
 *
 * zkp_decimal_to_limbs_batch: item i is text[text_off .. text_off+len), converted into dst[dst_off .. dst_off+words)
 * (little-endian words, zero extended).  Accepted exactly as mpz_set_str(s, 10): optional leading '-', white space
 * anywhere.  out_status[i]: ZKP_DEC_OK; ZKP_DEC_INVALID (mpz_set_str fails: serde error / `unwrap()` panic at
 * serialize.rs:66); ZKP_DEC_NEGATIVE, ZKP_DEC_OVERFLOW (a valid BigInt this fixed-width ABI cannot carry: the caller
 * keeps that proof on its CPU path).  dst words of a failed item are zero.  words <= 528. */
typedef struct zkp_dec_item { uint64_t text_off; uint64_t dst_off; uint32_t len; uint32_t words; } zkp_dec_item;
#define ZKP_DEC_OK 0
#define ZKP_DEC_INVALID 1
#define ZKP_DEC_NEGATIVE 2
#define ZKP_DEC_OVERFLOW 3
int32_t zkp_decimal_to_limbs_batch(zkp_ctx* ctx, const char* text, uint64_t text_len, const zkp_dec_item* items, uint64_t count,
                                   uint32_t* dst, uint64_t dst_words, uint8_t* out_status, uint32_t flags);
/* zkp_limbs_to_decimal_batch: src[i*src_stride .. +words) -> the decimal string of item i, right aligned in row i of
 * out_text (rows of `pitch` bytes, pitch >= zkp_decimal_pitch(words)): the string is out_text + i*pitch + pitch - out_len[i],
 * out_len[i] bytes, no terminator, no leading zeros, "0" for zero. */
uint32_t zkp_decimal_pitch(uint32_t words);
int32_t zkp_limbs_to_decimal_batch(zkp_ctx* ctx, const uint32_t* src, uint64_t src_stride, uint32_t words, uint64_t count,
                                   char* out_text, uint32_t pitch, uint32_t* out_len, uint32_t flags);

/* serde_json documents of the reference's proof types -> the SoA batch (one document per proof, documents back to back or
 * anywhere in `text`; doc_off/doc_len [B]).  Field layout = serde defaults for the derives at range_proof.rs:32-81
 * (EncryptedPairs {"c1":[..],"c2":[..]}, Proof = [{"Open":{"w1","r1","w2","r2"}} | {"Mask":{"j","masked_x","masked_r"}} ..])
 * and correct_key_ni.rs:35-39 ({"sigma_vec":[..]}).  The reader is as tolerant as serde_json with the derived Deserialize impls: white
 * space between tokens (to_string_pretty), object fields in any order, unknown fields skipped, string escapes decoded; duplicate
 * or missing fields, a Response with more than one variant key and values of the wrong JSON type are errors, as they are for serde.
 * The tokenising runs on the host (threads), every number is converted on the GPU into p->c1/c2 (pairs) or p->resp_*
 * (proof); p->error_factor rows are expected.  out_status[b]:
 *   ZKP_DOC_OK        converted;
 *   ZKP_DOC_INVALID   document b is not a value of the expected type (serde_json::from_str is Err in Rust);
 *   ZKP_DOC_HOST_PATH a well-formed document that this fixed layout cannot carry: a negative or over-wide integer (see ZKP_DEC_*), or
 *                     another number of rows.  It IS a valid value of the reference's type and has a verdict there: the caller parses
 *                     it itself (host/zkproofs.hpp: serde_json::range_proof_ni_from_str; bindings/rust: serde) and verifies it through
 *                     the host path that handles signed integers of any size (RangeProofNi::verify_batch).  Its rows here are zero.
 * ZKP_F_DEVICE_PTRS applies to the p-> arrays and out_status; text and offsets are host memory. */
#define ZKP_DOC_OK 0
#define ZKP_DOC_INVALID 2      /* == ZKP_VERDICT_MALFORMED */
#define ZKP_DOC_HOST_PATH 3
int32_t zkp_json_encrypted_pairs_batch(zkp_ctx* ctx, const char* text, const uint64_t* doc_off, const uint64_t* doc_len,
                                       const zkp_range_ni_proofs* p, uint8_t* out_status, uint32_t flags);
int32_t zkp_json_range_proof_batch(zkp_ctx* ctx, const char* text, const uint64_t* doc_off, const uint64_t* doc_len,
                                   const zkp_range_ni_proofs* p, uint8_t* out_status, uint32_t flags);
/* Whole RangeProofNi documents (range_proof_ni.rs:36-44): {"ek":{"n":..},"range":..,"ciphertext":..,"encrypted_pairs":{..},"proof":[..],
 * "error_factor":N} -> every field of the batch.  encrypted_pairs / proof as above.  ek, range and ciphertext are UN-annotated
 * in the reference: ek is kzen-paillier's EncryptionKey, range / ciphertext are bare curv BigInts; their text forms are fixed by crates
 * outside the tree and need not agree with each other, so `bigint_forms` names BOTH: ZKP_BIGINT_FORMS(key_form, bare_form) (a sample
 * written by a Rust build decides, tools/reference_vectors "serde" section).  A string that is not an integer of the named form is
 * ZKP_DOC_INVALID — an all-digit decimal read as hex would silently be another number, which is why the two forms are separate.
 * error_factor other than p->error_factor: ZKP_DOC_HOST_PATH.
 * p->n_stride = n_bits/32: one key per proof, p->n receives the documents' keys (verify_self, range_proof_ni.rs:109-128).
 * p->n_stride = 0: p->n is the VERIFIER's key, an input that is never written; a document under another key is ZKP_DOC_INVALID
 * (RangeProofNi::verify asserts equality, :86), so no received document can change the key the others are verified under.
 * Writes p->range and p->ciphertext (inputs of the other entry points, hence const in the struct).  HOST pointers only (flags 0). */
#define ZKP_BIGINT_DEC 0u     /* "1234": decimal string (serialize::bigint, serialize.rs:8-33) */
#define ZKP_BIGINT_HEX 1u     /* "04d2": hex string of the big-endian magnitude */
#define ZKP_BIGINT_BYTES 2u   /* [4,210]: array of big-endian byte values */
#define ZKP_BIGINT_FORMS(key_form, bare_form) (((key_form) << 4) | (bare_form))
int32_t zkp_json_range_proof_ni_batch(zkp_ctx* ctx, const char* text, const uint64_t* doc_off, const uint64_t* doc_len, uint32_t bigint_forms,
                                      const zkp_range_ni_proofs* p, uint8_t* out_status, uint32_t flags);
/* {"sigma_vec":["..", x11]} -> sigma [B][11][n_bits/32] */
int32_t zkp_json_correct_key_proof_batch(zkp_ctx* ctx, const char* text, const uint64_t* doc_off, const uint64_t* doc_len, uint32_t n_bits,
                                         uint64_t batch, uint32_t* out_sigma, uint8_t* out_status, uint32_t flags);

/* ------------------------------------------------------------------ several GPUs behind one caller
 * The reference spreads a proof's rows over a rayon pool (src/zkproofs/range_proof.rs:161-187,270-348); here a batch
 * is cut into contiguous blocks of PROOF indices, one block per device context, one host thread per context
 * (the only threads this library starts).  The caller hands over HOST pointers: every block reads its slice of the caller's
 * arrays and its output slab lands in the caller's output arrays — by per-GPU D2H, or, with ZKP_GATHER_RCCL, after an RCCL
 * all-gather that also leaves the whole result device-resident on every GPU (below).  (One PROCESS per GPU — torch.distributed
 * ranks — is zk-paillier_amd/shard.py + bench.py.)  device_ids may repeat in ZKP_GATHER_HOST mode: two contexts on one GPU are
 * two independent streams.  Results are identical to one call of the single-context entry point on the whole batch. */
typedef struct zkp_multi zkp_multi;
int32_t zkp_multi_create(const int32_t* device_ids, uint32_t n_devices, zkp_multi** out);
int32_t zkp_multi_destroy(zkp_multi* m);
uint32_t zkp_multi_size(zkp_multi* m);
zkp_ctx* zkp_multi_ctx(zkp_multi* m, uint32_t i);            /* context i (owned by m), e.g. for zkp_timing_* */
const char* zkp_multi_last_error_string(zkp_multi* m);
/* the most recent batch call, per device context i: the block [lo, hi) of items it was given and the wall time of its share
 * (staging, launches and the D2H of its output slab), in milliseconds */
int32_t zkp_multi_last_timing(zkp_multi* m, uint32_t i, double* out_ms, uint64_t* out_lo, uint64_t* out_hi);
/* ... and its two phases on device context i's own stream (HIP events), in milliseconds: the compute of its block (staging of its inputs
 * included) and, in the gathering modes below, the all-gather behind it — which ends when the slowest peer has delivered, so it holds the
 * wait for stragglers as well as the exchange.  ZKP_GATHER_HOST: compute = the wall time of the context's blocking call, gather = 0. */
int32_t zkp_multi_last_phases(zkp_multi* m, uint32_t i, double* out_compute_ms, double* out_gather_ms);
/* Where the outputs of the batch calls below are reassembled.
 *   ZKP_GATHER_HOST (default): every context copies its output slab into the caller's host arrays (one D2H per GPU, no collective).
 *   ZKP_GATHER_RCCL: every context works on device-resident copies of its block and writes its slab into its segment of a buffer
 *     holding the WHOLE batch; one grouped ncclAllGather per output (RCCL over xGMI, on the contexts' streams) then leaves the whole
 *     gathered output in the memory of EVERY GPU, and the caller's host arrays are filled from one GPU's copy — same bytes as
 *     ZKP_GATHER_HOST.  Gathered: the verdict bytes of the verify calls; status bytes, c1 and c2 of a prove call.  The first switch
 *     to ZKP_GATHER_RCCL creates one communicator per context (ncclCommInitAll): ZKP_EDEVICE with RCCL's text
 *     (zkp_multi_last_error_string) if that fails, e.g. for a device listed twice.
 * zkp_multi_gathered: the device-resident result of the most recent ZKP_GATHER_RCCL call on device context `device_index`:
 *   which = 0 verdict / status bytes, 1 c1, 2 c2.  Blocks of unequal size are padded to the largest: block i (the items
 *   zkp_multi_last_timing reports for context i) starts at i * *out_block_stride_bytes; *out_bytes = n_contexts * stride.  The pointer
 *   stays valid until the next batch call or zkp_multi_destroy; work that consumes it is ordered on zkp_ctx_stream(zkp_multi_ctx(m, i)). */
#define ZKP_GATHER_HOST 0u
#define ZKP_GATHER_RCCL 1u
#define ZKP_GATHER_COPY 2u   /* the device-resident gather of ZKP_GATHER_RCCL by device-to-device copies instead of the collective: for device
                                lists RCCL has no communicator for (a GPU listed several times); same layout, same zkp_multi_gathered */
int32_t zkp_multi_set_gather(zkp_multi* m, uint32_t mode);
int32_t zkp_multi_gathered(zkp_multi* m, uint32_t device_index, uint32_t which, void** out_device_ptr, uint64_t* out_block_stride_bytes,
                           uint64_t* out_bytes);
int32_t zkp_multi_range_ni_prove_batch(zkp_multi* m, const zkp_range_ni_proofs* p, const zkp_range_ni_witness* w,
                                       uint8_t* out_e, uint8_t* out_e_len, uint8_t* out_status);
int32_t zkp_multi_range_ni_verify_batch(zkp_multi* m, const zkp_range_ni_proofs* p, uint8_t* out_verdict);
int32_t zkp_multi_correct_key_ni_verify_batch(zkp_multi* m, uint32_t n_bits, uint64_t batch, const uint32_t* n,
                                              const uint32_t* sigma, const uint8_t* salt, uint32_t salt_len,
                                              uint8_t* out_verdict);

#ifdef __cplusplus
}
#endif
#endif /* ZKP_HIP_H */
