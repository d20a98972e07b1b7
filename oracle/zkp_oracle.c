/*
 * zkp_oracle.c — TEST INFRASTRUCTURE.  CPU restatement (C + libgmp) of the hot path of
 * ZenGo-X/zk-paillier.  It is the checker for the HIP engine, never the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.
 *
 * PARITY UNPINNED.  The reference is Rust (curv-kzen 0.10 + rust-gmp-kzen, kzen-paillier
 * 0.4.3 — none vendored under /root/reference, no lockfile) and there is no rustc/cargo in
 * this image, so the reference cannot be built; its own tests contain no known-answer
 * vectors (all inputs come from the OS RNG).  What this file does instead:
 *   - it calls the SAME libgmp entry points the reference's BigInt bottoms out in
 *     (mpz_powm, mpz_mul, mpz_tdiv_r/mpz_mod, mpz_gcd, mpz_import/export), GMP 6.2.1;
 *   - every function cites the reference lines it follows (paths relative to
 *     /root/reference); behaviours of the un-vendored crates are marked [upstream];
 *   - it is cross-checked against the independent pure-Python model oracle/py_model.py
 *     (tests/test_oracle.py) and against committed golden vectors (tests/golden/).
 *
 * The exported functions mirror include/zkp_hip.h one-to-one (prefix oracle_ instead of
 * zkp_, host pointers only, no ctx) so that parity tests hand both sides the same buffers.
 */
#include <gmp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/zkp_hip.h"

/* ------------------------------------------------------------------ SHA-256 (FIPS 180-4) */
typedef struct {
  uint32_t h[8];
  uint8_t buf[64];
  uint64_t len;
  uint32_t fill;
} sha256_t;

static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))

static void sha256_block(sha256_t* s, const uint8_t* p) {
  uint32_t w[64], a, b, c, d, e, f, g, h;
  for (int i = 0; i < 16; i++)
    w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
  for (int i = 16; i < 64; i++) {
    uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
    uint32_t s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  a = s->h[0]; b = s->h[1]; c = s->h[2]; d = s->h[3]; e = s->h[4]; f = s->h[5]; g = s->h[6]; h = s->h[7];
  for (int i = 0; i < 64; i++) {
    uint32_t S1 = ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25);
    uint32_t ch = (e & f) ^ (~e & g);
    uint32_t t1 = h + S1 + ch + K256[i] + w[i];
    uint32_t S0 = ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22);
    uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    uint32_t t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  s->h[0] += a; s->h[1] += b; s->h[2] += c; s->h[3] += d; s->h[4] += e; s->h[5] += f; s->h[6] += g; s->h[7] += h;
}

static void sha256_init(sha256_t* s) {
  static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  memcpy(s->h, iv, sizeof iv);
  s->len = 0;
  s->fill = 0;
}

static void sha256_update(sha256_t* s, const uint8_t* p, size_t n) {
  s->len += n;
  while (n) {
    size_t k = 64 - s->fill;
    if (k > n) k = n;
    memcpy(s->buf + s->fill, p, k);
    s->fill += (uint32_t)k; p += k; n -= k;
    if (s->fill == 64) { sha256_block(s, s->buf); s->fill = 0; }
  }
}

static void sha256_final(sha256_t* s, uint8_t out[32]) {
  uint64_t bits = s->len * 8;
  uint8_t pad = 0x80;
  sha256_update(s, &pad, 1);
  pad = 0;
  while (s->fill != 56) sha256_update(s, &pad, 1);
  uint8_t lb[8];
  for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
  sha256_update(s, lb, 8);
  for (int i = 0; i < 8; i++) {
    out[4 * i] = (uint8_t)(s->h[i] >> 24); out[4 * i + 1] = (uint8_t)(s->h[i] >> 16);
    out[4 * i + 2] = (uint8_t)(s->h[i] >> 8); out[4 * i + 3] = (uint8_t)s->h[i];
  }
}

void oracle_sha256(const uint8_t* p, uint64_t n, uint8_t out[32]) {
  sha256_t s; sha256_init(&s); sha256_update(&s, p, n); sha256_final(&s, out);
}

/* ------------------------------------------------------------------ limb <-> mpz */
static void limbs_to_mpz(mpz_t z, const uint32_t* p, size_t nlimbs) {
  mpz_import(z, nlimbs, -1, 4, 0, 0, p);
}
static void mpz_to_limbs(uint32_t* p, size_t nlimbs, const mpz_t z) {
  size_t cnt = 0;
  memset(p, 0, nlimbs * 4);
  /* caller guarantees z fits */
  mpz_export(p, &cnt, -1, 4, 0, 0, z);
}

/* [upstream curv BigInt::to_bytes / rust-gmp `impl From<&Mpz> for Vec<u8>`]:
 * (mpz_sizeinbase(x,2)+7)/8 bytes, big-endian; zero -> one 0x00 byte. */
static void hash_mpz(sha256_t* s, const mpz_t z) {
  size_t nbytes = (mpz_sizeinbase(z, 2) + 7) / 8;
  uint8_t* b = (uint8_t*)calloc(nbytes, 1);
  mpz_export(b, NULL, 1, 1, 0, 0, z);
  sha256_update(s, b, nbytes);
  free(b);
}

static int n_threads = 1;
void oracle_set_threads(int n) { n_threads = n > 0 ? n : 1; }
int oracle_get_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------ L1 primitives */
int32_t oracle_modexp_batch(uint32_t mod_bits, uint32_t exp_bits, uint64_t count, const uint32_t* base,
                            const uint32_t* exp, uint64_t exp_stride, const uint32_t* mod, uint64_t mod_stride,
                            uint32_t* out) {
  size_t L = mod_bits / 32, E = exp_bits / 32;
#pragma omp parallel num_threads(n_threads)
  {
    mpz_t b, e, m, r;
    mpz_inits(b, e, m, r, NULL);
#pragma omp for schedule(dynamic, 1)
    for (int64_t i = 0; i < (int64_t)count; i++) {
      limbs_to_mpz(b, base + i * L, L);
      limbs_to_mpz(e, exp + i * exp_stride, E);
      limbs_to_mpz(m, mod + i * mod_stride, L);
      mpz_powm(r, b, e, m); /* BigInt::mod_pow -> mpz_powm (correct_key_ni.rs:92; wi_dlog_proof.rs:55,81,82) */
      mpz_to_limbs(out + i * L, L, r);
    }
    mpz_clears(b, e, m, r, NULL);
  }
  return 0;
}

int32_t oracle_modmul_batch(uint32_t mod_bits, uint64_t count, const uint32_t* a, const uint32_t* b,
                            const uint32_t* mod, uint64_t mod_stride, uint32_t* out) {
  size_t L = mod_bits / 32;
  mpz_t x, y, m;
  mpz_inits(x, y, m, NULL);
  for (uint64_t i = 0; i < count; i++) {
    limbs_to_mpz(x, a + i * L, L);
    limbs_to_mpz(y, b + i * L, L);
    limbs_to_mpz(m, mod + i * mod_stride, L);
    mpz_mul(x, x, y);
    mpz_mod(x, x, m); /* BigInt::mod_mul (wi_dlog_proof.rs:83); `a * b % m` (range_proof.rs:239,245,325,327) */
    mpz_to_limbs(out + i * L, L, x);
  }
  mpz_clears(x, y, m, NULL);
  return 0;
}

/* [upstream kzen-paillier 0.4.3, EncryptWithChosenRandomness for (EncryptionKey, RawPlaintext, Randomness)]
 *   rn = r^n mod nn;  gm = (m*n + 1) % nn;  c = (gm*rn) % nn
 * call sites: range_proof.rs:165-169,179-183,280-291,330-334 */
static void enc_mpz(mpz_t c, const mpz_t n, const mpz_t nn, const mpz_t m, const mpz_t r, mpz_t t) {
  mpz_powm(t, r, n, nn);
  mpz_mul(c, m, n);
  mpz_add_ui(c, c, 1);
  mpz_tdiv_r(c, c, nn);
  mpz_mul(c, c, t);
  mpz_tdiv_r(c, c, nn);
}

int32_t oracle_paillier_enc_batch(uint32_t n_bits, uint64_t count, const uint32_t* n, uint64_t n_stride,
                                  const uint32_t* m, const uint32_t* r, uint32_t* out_c) {
  size_t kw = n_bits / 32;
#pragma omp parallel num_threads(n_threads)
  {
    mpz_t zn, znn, zm, zr, zc, t;
    mpz_inits(zn, znn, zm, zr, zc, t, NULL);
#pragma omp for schedule(dynamic, 1)
    for (int64_t i = 0; i < (int64_t)count; i++) {
      limbs_to_mpz(zn, n + i * n_stride, kw);
      mpz_mul(znn, zn, zn);
      limbs_to_mpz(zm, m + i * kw, kw);
      limbs_to_mpz(zr, r + i * kw, kw);
      enc_mpz(zc, zn, znn, zm, zr, t);
      mpz_to_limbs(out_c + i * 2 * kw, 2 * kw, zc);
    }
    mpz_clears(zn, znn, zm, zr, zc, t, NULL);
  }
  return 0;
}

/* CorrectOpening::verify_opening (correct_opening.rs:25-28): `c == &d` on BigInts, i.e. the stored value is compared
 * unreduced; with (mulc_a, mulc_b) the expected value is `a * b % nn` as at range_proof.rs:324-328. */
int32_t oracle_paillier_enc_check_batch(uint32_t n_bits, uint64_t count, const uint32_t* n, uint64_t n_stride, const uint32_t* m,
                                        const uint32_t* r, const uint32_t* mulc_a, const uint32_t* mulc_b, const uint32_t* expected,
                                        uint8_t* out_ok) {
  size_t kw = n_bits / 32;
#pragma omp parallel num_threads(n_threads)
  {
    mpz_t zn, znn, zm, zr, zc, t, e, f;
    mpz_inits(zn, znn, zm, zr, zc, t, e, f, NULL);
#pragma omp for schedule(dynamic, 1)
    for (int64_t i = 0; i < (int64_t)count; i++) {
      limbs_to_mpz(zn, n + i * n_stride, kw);
      mpz_mul(znn, zn, zn);
      limbs_to_mpz(zm, m + i * kw, kw);
      limbs_to_mpz(zr, r + i * kw, kw);
      enc_mpz(zc, zn, znn, zm, zr, t);
      if (mulc_a) {
        limbs_to_mpz(e, mulc_a + i * 2 * kw, 2 * kw);
        limbs_to_mpz(f, mulc_b + i * 2 * kw, 2 * kw);
        mpz_mul(e, e, f);
        mpz_tdiv_r(e, e, znn);
      } else {
        limbs_to_mpz(e, expected + i * 2 * kw, 2 * kw);
      }
      out_ok[i] = mpz_cmp(e, zc) == 0;
    }
    mpz_clears(zn, znn, zm, zr, zc, t, e, f, NULL);
  }
  return 0;
}

/* ------------------------------------------------------------------ RangeProofNi */

/* range_proof_ni.rs:58-61 / 89-92 with utils.rs:9-22:
 * e = to_bytes(from_bytes(SHA256(to_bytes(n) || to_bytes(c1[0..EF]) || to_bytes(c2[0..EF])))).
 * Leading zero bytes of the digest vanish in the BigInt round trip; an all-zero
 * digest becomes the single byte 00. */
static uint32_t fs_challenge(const mpz_t n, const uint32_t* c1, const uint32_t* c2, uint32_t ef, size_t kw, uint8_t e[32]) {
  sha256_t s;
  uint8_t d[32];
  mpz_t z;
  mpz_init(z);
  sha256_init(&s);
  hash_mpz(&s, n);
  for (uint32_t i = 0; i < ef; i++) { limbs_to_mpz(z, c1 + (size_t)i * 2 * kw, 2 * kw); hash_mpz(&s, z); }
  for (uint32_t i = 0; i < ef; i++) { limbs_to_mpz(z, c2 + (size_t)i * 2 * kw, 2 * kw); hash_mpz(&s, z); }
  sha256_final(&s, d);
  mpz_clear(z);
  uint32_t lead = 0;
  while (lead < 31 && d[lead] == 0) lead++;
  memset(e, 0, 32);
  memcpy(e, d + lead, 32 - lead);
  return 32 - lead;
}

/* bit_vec::BitVec::from_bytes indexing (range_proof.rs:221,225,267,272): MSB first. */
static int challenge_bit(const uint8_t* e, uint32_t i) { return (e[i / 8] >> (7 - (i % 8))) & 1; }

/* phases: 1 = generate_encrypted_pairs (range_proof.rs:128-193), 2 = generate_proof (:210-252).  e_in != NULL:
 * externally supplied challenge bytes (the interactive protocol, RangeProof::verifier_commit :118-126), else the
 * Fiat-Shamir challenge of RangeProofNi::prove (range_proof_ni.rs:58-61). */
static int32_t range_prove_core(const zkp_range_ni_proofs* p, const zkp_range_ni_witness* w, size_t EF, int phases, const uint8_t* e_in,
                                const uint8_t* e_len_in, uint8_t* out_e, uint8_t* out_e_len, uint8_t* out_status) {
  const size_t kw = p->n_bits / 32;
  if (phases & 1) {
  /* generate_encrypted_pairs, range_proof.rs:161-187 (randomness injected, :136-159); the pool runs over the
   * flattened (proof, c1|c2, row) list */
#pragma omp parallel num_threads(n_threads)
  {
    mpz_t zn, znn, zm, zr, zc, t;
    mpz_inits(zn, znn, zm, zr, zc, t, NULL);
    int64_t cur = -1;
#pragma omp for schedule(dynamic, 8)
    for (int64_t it = 0; it < (int64_t)(p->batch * 2 * EF); it++) {
      const int64_t b = it / (int64_t)(2 * EF), i = it % (int64_t)(2 * EF);
      if (b != cur) {
        cur = b;
        limbs_to_mpz(zn, p->n + b * p->n_stride, kw);
        mpz_mul(znn, zn, zn);
      }
      size_t row = (size_t)i % EF;
      const uint32_t* wm = (i < (int64_t)EF ? w->w1 : w->w2) + (b * EF + row) * kw;
      const uint32_t* wr = (i < (int64_t)EF ? w->r1 : w->r2) + (b * EF + row) * kw;
      limbs_to_mpz(zm, wm, kw);
      limbs_to_mpz(zr, wr, kw);
      enc_mpz(zc, zn, znn, zm, zr, t);
      mpz_to_limbs((i < (int64_t)EF ? p->c1 : p->c2) + (b * EF + row) * 2 * kw, 2 * kw, zc);
    }
    mpz_clears(zn, znn, zm, zr, zc, t, NULL);
  }
  }
  if (!(phases & 2)) return 0;
  for (uint64_t b = 0; b < p->batch; b++) {
    const uint32_t* nl = p->n + b * p->n_stride;
    uint32_t* c1 = p->c1 + b * EF * 2 * kw;
    uint32_t* c2 = p->c2 + b * EF * 2 * kw;
    mpz_t zn, x, r, third, two_thirds, t, u, wv, rv;
    mpz_inits(zn, x, r, third, two_thirds, t, u, wv, rv, NULL);
    limbs_to_mpz(zn, nl, kw);
    uint8_t e[32];
    uint32_t elen;
    if (e_in) { memcpy(e, e_in + b * 32, 32); elen = e_len_in[b]; }
    else elen = fs_challenge(zn, c1, c2, (uint32_t)EF, kw, e);
    if (out_e) memcpy(out_e + b * 32, e, 32);
    if (out_e_len) out_e_len[b] = (uint8_t)elen;
    uint8_t status = 0;
    if (elen * 8 < EF) status = ZKP_VERDICT_MALFORMED; /* bits_of_e[i] would panic (range_proof_ni.rs:63 comment) */
    /* generate_proof, range_proof.rs:210-252 */
    limbs_to_mpz(x, w->x + b * kw, kw);
    limbs_to_mpz(r, w->r + b * kw, kw);
    limbs_to_mpz(t, p->range + b * kw, kw);
    mpz_fdiv_q_ui(third, t, 3);      /* range.div_floor(3) :219 */
    mpz_mul_ui(two_thirds, third, 2); /* :220 */
    for (size_t i = 0; i < EF && !status; i++) {
      size_t o = (b * EF + i);
      const uint32_t *w1 = w->w1 + o * kw, *w2 = w->w2 + o * kw, *r1 = w->r1 + o * kw, *r2 = w->r2 + o * kw;
      memset(p->resp_w1 + o * kw, 0, kw * 4); memset(p->resp_r1 + o * kw, 0, kw * 4);
      memset(p->resp_w2 + o * kw, 0, kw * 4); memset(p->resp_r2 + o * kw, 0, kw * 4);
      if (!challenge_bit(e, (uint32_t)i)) { /* :226-232 */
        p->resp_kind[o] = ZKP_RESP_OPEN; p->resp_j[o] = 0;
        memcpy(p->resp_w1 + o * kw, w1, kw * 4); memcpy(p->resp_r1 + o * kw, r1, kw * 4);
        memcpy(p->resp_w2 + o * kw, w2, kw * 4); memcpy(p->resp_r2 + o * kw, r2, kw * 4);
        continue;
      }
      limbs_to_mpz(wv, w1, kw);
      mpz_add(t, x, wv);
      p->resp_kind[o] = ZKP_RESP_MASK;
      if (mpz_cmp(t, third) > 0 && mpz_cmp(t, two_thirds) < 0) { /* :233-234, both strict */
        p->resp_j[o] = 1;
        limbs_to_mpz(rv, r1, kw);
      } else {
        p->resp_j[o] = 2;
        limbs_to_mpz(wv, w2, kw);
        mpz_add(t, x, wv);
        limbs_to_mpz(rv, r2, kw);
      }
      if (mpz_sizeinbase(t, 2) > p->n_bits) { status = ZKP_VERDICT_MALFORMED; break; } /* does not fit the fixed-width ABI */
      mpz_to_limbs(p->resp_w1 + o * kw, kw, t);
      mpz_mul(u, r, rv);
      mpz_tdiv_r(u, u, zn); /* secret_r * r_j % n  :239,245 */
      mpz_to_limbs(p->resp_r1 + o * kw, kw, u);
    }
    if (out_status) out_status[b] = status;
    mpz_clears(zn, x, r, third, two_thirds, t, u, wv, rv, NULL);
  }
  return 0;
}

int32_t oracle_range_ni_prove_batch(const zkp_range_ni_proofs* p, const zkp_range_ni_witness* w, uint8_t* out_e,
                                    uint8_t* out_e_len, uint8_t* out_status) {
  return range_prove_core(p, w, ZKP_SECURITY_PARAMETER, 3, NULL, NULL, out_e, out_e_len, out_status);
}
int32_t oracle_range_generate_encrypted_pairs_batch(const zkp_range_ni_proofs* p, const zkp_range_ni_witness* w) {
  return range_prove_core(p, w, p->error_factor, 1, NULL, NULL, NULL, NULL, NULL);
}
int32_t oracle_range_generate_proof_batch(const zkp_range_ni_proofs* p, const zkp_range_ni_witness* w, const uint8_t* e,
                                          const uint8_t* e_len, uint8_t* out_status) {
  return range_prove_core(p, w, p->error_factor, 2, e, e_len, NULL, NULL, out_status);
}

static int32_t range_verify_core(const zkp_range_ni_proofs* p, const uint8_t* e_in, const uint8_t* e_len_in, uint8_t* out_verdict) {
  const size_t kw = p->n_bits / 32, EF = p->error_factor;
  const int64_t B = (int64_t)p->batch;
  uint8_t* e_all = (uint8_t*)calloc((size_t)B + 1, 32);
  /* phase 1: Fiat-Shamir challenge per proof (range_proof_ni.rs:89-92) */
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1)
  for (int64_t b = 0; b < B; b++) {
    mpz_t zn;
    mpz_init(zn);
    limbs_to_mpz(zn, p->n + b * p->n_stride, kw);
    uint32_t elen;
    if (e_in) { memcpy(e_all + b * 32, e_in + b * 32, 32); elen = e_len_in[b]; }
    else elen = fs_challenge(zn, p->c1 + b * EF * 2 * kw, p->c2 + b * EF * 2 * kw, (uint32_t)EF, kw, e_all + b * 32);
    /* bits_of_e[i] index panic when the challenge is shorter than error_factor bits */
    out_verdict[b] = ((size_t)elen * 8 < EF) ? ZKP_VERDICT_MALFORMED : ZKP_VERDICT_ACCEPT;
    mpz_clear(zn);
  }
  /* phase 2: range_proof.rs:270-348 — every row of every proof is evaluated (no early exit); rows are
   * independent, so the pool runs over the flattened (proof,row) list (the reference's rayon par_iter runs
   * over the rows of one proof; a caller with many proofs would parallelise over proofs as well) */
#pragma omp parallel num_threads(n_threads)
  {
    mpz_t zn, znn, cx, third, two_thirds, t, w1, r1, w2, r2, c, ex, u;
    mpz_inits(zn, znn, cx, third, two_thirds, t, w1, r1, w2, r2, c, ex, u, NULL);
    int64_t cur = -1;
#pragma omp for schedule(dynamic, 8)
    for (int64_t row = 0; row < B * (int64_t)EF; row++) {
      const int64_t b = row / (int64_t)EF;
      const size_t i = (size_t)(row % (int64_t)EF);
      if (out_verdict[b] == ZKP_VERDICT_MALFORMED) continue;
      if (b != cur) {
        cur = b;
        limbs_to_mpz(zn, p->n + b * p->n_stride, kw);
        mpz_mul(znn, zn, zn);
        limbs_to_mpz(cx, p->ciphertext + b * 2 * kw, 2 * kw);
        limbs_to_mpz(t, p->range + b * kw, kw);
        mpz_fdiv_q_ui(third, t, 3);       /* range_proof.rs:264 */
        mpz_mul_ui(two_thirds, third, 2); /* :265 */
      }
      const uint32_t* c1 = p->c1 + b * EF * 2 * kw;
      const uint32_t* c2 = p->c2 + b * EF * 2 * kw;
      const size_t o = (size_t)row;
      int ei = challenge_bit(e_all + b * 32, (uint32_t)i);
      int res = 1;
      limbs_to_mpz(w1, p->resp_w1 + o * kw, kw);
      limbs_to_mpz(r1, p->resp_r1 + o * kw, kw);
      if (!ei && p->resp_kind[o] == ZKP_RESP_OPEN) { /* :277-313 */
        limbs_to_mpz(w2, p->resp_w2 + o * kw, kw);
        limbs_to_mpz(r2, p->resp_r2 + o * kw, kw);
        enc_mpz(c, zn, znn, w1, r1, u);
        limbs_to_mpz(ex, c1 + i * 2 * kw, 2 * kw);
        if (mpz_cmp(c, ex) != 0) res = 0;
        enc_mpz(c, zn, znn, w2, r2, u);
        limbs_to_mpz(ex, c2 + i * 2 * kw, 2 * kw);
        if (mpz_cmp(c, ex) != 0) res = 0;
        int flag = (mpz_cmp(w2, third) < 0 && mpz_cmp(w1, third) > 0 && mpz_cmp(w1, two_thirds) < 0) ||
                   (mpz_cmp(w1, third) < 0 && mpz_cmp(w2, third) > 0 && mpz_cmp(w2, two_thirds) < 0); /* :300-305 */
        if (!flag) res = 0;
      } else if (ei && p->resp_kind[o] == ZKP_RESP_MASK) { /* :315-343 */
        limbs_to_mpz(ex, (p->resp_j[o] == 1 ? c1 : c2) + i * 2 * kw, 2 * kw); /* any j != 1 selects c2 :324-328 */
        mpz_mul(ex, ex, cx);
        mpz_tdiv_r(ex, ex, znn);
        enc_mpz(c, zn, znn, w1, r1, u); /* Enc(masked_x, masked_r) :330-334 */
        if (mpz_cmp(c, ex) != 0) res = 0;
        if (mpz_cmp(w1, third) < 0 || mpz_cmp(w1, two_thirds) > 0) res = 0; /* :338 */
      } else {
        res = 0; /* :345 */
      }
      if (!res) {
#pragma omp atomic write
        out_verdict[b] = ZKP_VERDICT_REJECT;
      }
    }
    mpz_clears(zn, znn, cx, third, two_thirds, t, w1, r1, w2, r2, c, ex, u, NULL);
  }
  free(e_all);
  return 0;
}

int32_t oracle_range_ni_verify_batch(const zkp_range_ni_proofs* p, uint8_t* out_verdict) {
  return range_verify_core(p, NULL, NULL, out_verdict);
}
/* RangeProof::verifier_output (range_proof.rs:254-355) with the verifier's own challenge bytes (interactive protocol) */
int32_t oracle_range_verifier_output_batch(const zkp_range_ni_proofs* p, const uint8_t* e, const uint8_t* e_len, uint8_t* out_verdict) {
  return range_verify_core(p, e, e_len, out_verdict);
}

/* ------------------------------------------------------------------ RangeProofNi::verify_self on SIGNED, ARBITRARY-PRECISION values
 * Every field of a deserialised proof is an attacker-chosen curv BigInt: any size, either sign (the decimal-string serde of
 * serialize.rs:8-31 goes through mpz_set_str, which takes a leading '-').  The fixed-width entry points above cannot even state such
 * a proof; this one restates range_proof_ni.rs:109-128 -> range_proof.rs:254-355 over mpz_t exactly as the reference's BigInt does
 * (SURVEY N4/N5), one proof per call, every value a decimal string:
 *   `a * b % m` and `(m*n + 1) % nn`  -> mpz_tdiv_r (Rust `%` on BigInt: truncated, sign of the dividend) [upstream, recalled]
 *   BigInt::mod_pow                   -> mpz_powm (result in [0, m) for a negative base)                 [upstream, recalled]
 *   range.div_floor(3)                -> mpz_fdiv_q
 *   comparisons / equality            -> mpz_cmp on the signed values
 *   to_bytes in compute_digest        -> magnitude only (mpz_export ignores the sign)                    [upstream, recalled]
 * n must be positive (ek is the verifier's own key in RangeProofNi::verify; mod_pow asserts a non-negative exponent).
 * kind[i]: ZKP_RESP_OPEN / ZKP_RESP_MASK; f1..f4[i]: w1, r1, w2, r2 of an Open row; masked_x, masked_r, -, - of a Mask row.
 * n_c1 / n_c2 / n_resp are the stored lengths: an index past them is the reference's panic (range_proof.rs:274,293,296).
 * returns ZKP_VERDICT_*; -1 when a string does not parse. */
static int set_dec(mpz_t z, const char* s) { return s && mpz_set_str(z, s, 10) == 0; }
int32_t oracle_range_ni_verify_decimal(const char* n_s, const char* range_s, const char* cipher_s, uint32_t error_factor,
                                       const char* const* c1_s, uint32_t n_c1, const char* const* c2_s, uint32_t n_c2,
                                       const uint8_t* kind, const uint8_t* j, const char* const* f1, const char* const* f2,
                                       const char* const* f3, const char* const* f4, uint32_t n_resp, uint8_t* out_e, uint8_t* out_e_len) {
  mpz_t zn, znn, cx, third, two_thirds, w1, r1, w2, r2, c, ex, u, z;
  mpz_inits(zn, znn, cx, third, two_thirds, w1, r1, w2, r2, c, ex, u, z, NULL);
  int32_t verdict = ZKP_VERDICT_ACCEPT;
  if (!set_dec(zn, n_s) || !set_dec(z, range_s) || !set_dec(cx, cipher_s) || mpz_sgn(zn) <= 0) { verdict = -1; goto done; }
  mpz_mul(znn, zn, zn);
  mpz_fdiv_q_ui(third, z, 3);       /* range.div_floor(3), range_proof.rs:264 */
  mpz_mul_ui(two_thirds, third, 2); /* :265 */
  { /* e = to_bytes(compute_digest(n, c1.., c2..)) over ALL stored elements (range_proof_ni.rs:110-113) */
    sha256_t s; uint8_t d[32];
    sha256_init(&s);
    hash_mpz(&s, zn);
    for (uint32_t i = 0; i < n_c1; i++) { if (!set_dec(z, c1_s[i])) { verdict = -1; goto done; } hash_mpz(&s, z); }
    for (uint32_t i = 0; i < n_c2; i++) { if (!set_dec(z, c2_s[i])) { verdict = -1; goto done; } hash_mpz(&s, z); }
    sha256_final(&s, d);
    uint32_t lead = 0;
    while (lead < 31 && d[lead] == 0) lead++;
    uint8_t e[32] = {0};
    uint32_t elen = 32 - lead;
    memcpy(e, d + lead, elen);
    if (out_e) memcpy(out_e, e, 32);
    if (out_e_len) *out_e_len = (uint8_t)elen;
    for (uint32_t i = 0; i < error_factor; i++) { /* no early exit: a later row may still panic (par_iter + collect, :270-348) */
      if (i >= elen * 8 || i >= n_resp) { verdict = ZKP_VERDICT_MALFORMED; break; } /* bits_of_e[i] / responses[i] */
      int ei = challenge_bit(e, i), res = 1;
      if (!set_dec(w1, f1[i]) || !set_dec(r1, f2[i])) { verdict = -1; goto done; }
      if (!ei && kind[i] == ZKP_RESP_OPEN) { /* :277-313 */
        if (!set_dec(w2, f3[i]) || !set_dec(r2, f4[i])) { verdict = -1; goto done; }
        if (i >= n_c1 || i >= n_c2) { verdict = ZKP_VERDICT_MALFORMED; break; }
        enc_mpz(c, zn, znn, w1, r1, u);
        set_dec(ex, c1_s[i]);
        if (mpz_cmp(c, ex) != 0) res = 0;
        enc_mpz(c, zn, znn, w2, r2, u);
        set_dec(ex, c2_s[i]);
        if (mpz_cmp(c, ex) != 0) res = 0;
        int flag = (mpz_cmp(w2, third) < 0 && mpz_cmp(w1, third) > 0 && mpz_cmp(w1, two_thirds) < 0) ||
                   (mpz_cmp(w1, third) < 0 && mpz_cmp(w2, third) > 0 && mpz_cmp(w2, two_thirds) < 0); /* :300-305 */
        if (!flag) res = 0;
      } else if (ei && kind[i] == ZKP_RESP_MASK) { /* :315-343 */
        if (j[i] == 1 ? i >= n_c1 : i >= n_c2) { verdict = ZKP_VERDICT_MALFORMED; break; }
        set_dec(ex, j[i] == 1 ? c1_s[i] : c2_s[i]); /* any j != 1 selects c2, :324-328 */
        mpz_mul(ex, ex, cx);
        mpz_tdiv_r(ex, ex, znn);
        enc_mpz(c, zn, znn, w1, r1, u); /* :330-334 */
        if (mpz_cmp(c, ex) != 0) res = 0;
        if (mpz_cmp(w1, third) < 0 || mpz_cmp(w1, two_thirds) > 0) res = 0; /* :338 */
      } else {
        res = 0; /* :345 */
      }
      if (!res) verdict = ZKP_VERDICT_REJECT;
    }
  }
done:
  mpz_clears(zn, znn, cx, third, two_thirds, w1, r1, w2, r2, c, ex, u, z, NULL);
  return verdict;
}

/* The signed Enc alone: c = ((m*n + 1) % nn) * (r^n mod nn) % nn with the reference's operators, decimal in and out
 * (out must hold the digits of nn plus sign and terminator). */
int32_t oracle_enc_decimal(const char* n_s, const char* m_s, const char* r_s, char* out, uint64_t out_cap) {
  mpz_t zn, znn, m, r, c, t;
  mpz_inits(zn, znn, m, r, c, t, NULL);
  int32_t st = -1;
  if (set_dec(zn, n_s) && set_dec(m, m_s) && set_dec(r, r_s) && mpz_sgn(zn) > 0) {
    mpz_mul(znn, zn, zn);
    enc_mpz(c, zn, znn, m, r, t);
    if (mpz_sizeinbase(c, 10) + 2 <= out_cap) { mpz_get_str(out, 10, c); st = 0; }
  }
  mpz_clears(zn, znn, m, r, c, t, NULL);
  return st;
}

/* ------------------------------------------------------------------ NiCorrectKeyProof */

/* utils.rs:9-22 over an array of mpz */
static void compute_digest(mpz_t out, const mpz_t* items, int n) {
  sha256_t s;
  uint8_t d[32];
  sha256_init(&s);
  for (int i = 0; i < n; i++) hash_mpz(&s, items[i]);
  sha256_final(&s, d);
  mpz_import(out, 32, 1, 1, 0, 0, d);
}

static mpz_t g_primorial;
static int g_primorial_ready = 0;
/* correct_key_ni.rs:26: P = product of all primes < 6370 (830 primes, 9095 bits) */
static void primorial_init(void) {
  if (g_primorial_ready) return;
  mpz_init_set_ui(g_primorial, 1);
  for (unsigned v = 2; v < 6370; v++) {
    int prime = 1;
    for (unsigned d = 2; d * d <= v; d++) if (v % d == 0) { prime = 0; break; }
    if (prime) mpz_mul_ui(g_primorial, g_primorial, v);
  }
  g_primorial_ready = 1;
}

/* test hook: the primorial this oracle computes, as a decimal string — tests/test_reference_constants.py compares it with the string the
 * reference parses (correct_key_ni.rs:26,87).  Returns the length, or -1 when `cap` is too small. */
int64_t oracle_primorial_decimal(char* out, uint64_t cap) {
  primorial_init();
  size_t need = mpz_sizeinbase(g_primorial, 10) + 2;
  if (need > cap) return -1;
  mpz_get_str(out, 10, g_primorial);
  return (int64_t)strlen(out);
}

/* correct_key_ni.rs:74-86 + mask_generation :105-117.  rho[i] for i < 11. */
static void correct_key_rho(mpz_t* rho, const mpz_t n, const uint8_t* salt, uint32_t salt_len) {
  size_t key_length = mpz_sizeinbase(n, 2); /* ek.n.bit_length() :74 */
  mpz_t salt_bn, seed, acc, h, items[3], iv, jv;
  mpz_inits(salt_bn, seed, acc, h, iv, jv, NULL);
  mpz_import(h, salt_len, 1, 1, 0, 0, salt); /* BigInt::from_bytes(salt) :75 */
  mpz_init_set(items[0], h);
  mpz_init(items[1]);
  mpz_init(items[2]);
  compute_digest(salt_bn, (const mpz_t*)items, 1);
  size_t msklen = key_length / 256 + 1; /* :106 */
  for (unsigned i = 0; i < ZKP_CORRECT_KEY_M2; i++) {
    mpz_set(items[0], n); mpz_set(items[1], salt_bn); mpz_set_ui(items[2], i);
    compute_digest(seed, (const mpz_t*)items, 3); /* :79-83 */
    mpz_set_ui(acc, 0);
    for (size_t j = 0; j < msklen; j++) { /* :107-116 */
      mpz_set(items[0], seed); mpz_set_ui(items[1], j);
      compute_digest(h, (const mpz_t*)items, 2);
      mpz_mul_2exp(h, h, j * 256);
      mpz_add(acc, acc, h);
    }
    mpz_tdiv_r(rho[i], acc, n); /* % &ek.n :84 */
  }
  mpz_clears(salt_bn, seed, acc, h, iv, jv, items[0], items[1], items[2], NULL);
}

int32_t oracle_correct_key_ni_verify_batch(uint32_t n_bits, uint64_t batch, const uint32_t* n, const uint32_t* sigma,
                                           const uint8_t* salt, uint32_t salt_len, uint8_t* out_verdict) {
  const size_t kw = n_bits / 32;
  primorial_init();
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1)
  for (int64_t b = 0; b < (int64_t)batch; b++) {
    mpz_t zn, g, s, d, rho[ZKP_CORRECT_KEY_M2];
    mpz_inits(zn, g, s, d, NULL);
    for (int i = 0; i < ZKP_CORRECT_KEY_M2; i++) mpz_init(rho[i]);
    limbs_to_mpz(zn, n + b * kw, kw);
    correct_key_rho(rho, zn, salt, salt_len);
    mpz_gcd(g, g_primorial, zn); /* :87-88 */
    int ok = mpz_cmp_ui(g, 1) == 0;
    for (int i = 0; i < ZKP_CORRECT_KEY_M2; i++) {
      limbs_to_mpz(s, sigma + (b * ZKP_CORRECT_KEY_M2 + i) * kw, kw);
      mpz_powm(d, s, zn, zn); /* :92 */
      if (mpz_cmp(d, rho[i]) != 0) ok = 0; /* :95 */
    }
    out_verdict[b] = ok ? ZKP_VERDICT_ACCEPT : ZKP_VERDICT_REJECT;
    for (int i = 0; i < ZKP_CORRECT_KEY_M2; i++) mpz_clear(rho[i]);
    mpz_clears(zn, g, s, d, NULL);
  }
  return 0;
}

/* NiCorrectKeyProof::proof (correct_key_ni.rs:42-71); extract_nroot [upstream kzen-paillier]:
 * sigma_i = rho_i^(n^-1 mod phi(n)) mod n (the CRT form upstream uses yields the same residue). */
int32_t oracle_correct_key_ni_prove(uint32_t n_bits, const uint32_t* p, const uint32_t* q, const uint8_t* salt,
                                    uint32_t salt_len, uint32_t* out_n, uint32_t* out_sigma) {
  const size_t kw = n_bits / 32;
  mpz_t zp, zq, zn, phi, d, t, rho[ZKP_CORRECT_KEY_M2];
  mpz_inits(zp, zq, zn, phi, d, t, NULL);
  for (int i = 0; i < ZKP_CORRECT_KEY_M2; i++) mpz_init(rho[i]);
  limbs_to_mpz(zp, p, kw / 2);
  limbs_to_mpz(zq, q, kw / 2);
  mpz_mul(zn, zp, zq);
  mpz_sub_ui(phi, zp, 1);
  mpz_sub_ui(t, zq, 1);
  mpz_mul(phi, phi, t);
  int rc = mpz_invert(d, zn, phi) ? 0 : 1;
  correct_key_rho(rho, zn, salt, salt_len);
  mpz_to_limbs(out_n, kw, zn);
  for (int i = 0; i < ZKP_CORRECT_KEY_M2 && !rc; i++) {
    mpz_powm(t, rho[i], d, zn);
    mpz_to_limbs(out_sigma + (size_t)i * kw, kw, t);
  }
  for (int i = 0; i < ZKP_CORRECT_KEY_M2; i++) mpz_clear(rho[i]);
  mpz_clears(zp, zq, zn, phi, d, t, NULL);
  return rc;
}

/* rho vector only (for tests of the MGF) */
int32_t oracle_correct_key_rho(uint32_t n_bits, const uint32_t* n, const uint8_t* salt, uint32_t salt_len, uint32_t* out_rho) {
  const size_t kw = n_bits / 32;
  mpz_t zn, rho[ZKP_CORRECT_KEY_M2];
  mpz_init(zn);
  for (int i = 0; i < ZKP_CORRECT_KEY_M2; i++) mpz_init(rho[i]);
  limbs_to_mpz(zn, n, kw);
  correct_key_rho(rho, zn, salt, salt_len);
  for (int i = 0; i < ZKP_CORRECT_KEY_M2; i++) { mpz_to_limbs(out_rho + (size_t)i * kw, kw, rho[i]); mpz_clear(rho[i]); }
  mpz_clear(zn);
  return 0;
}

/* ------------------------------------------------------------------ interactive CorrectKey (correct_key.rs:64-171)
 * The reference samples s_i, r_i < n from the OS RNG (:67-70,80-83); here they are inputs.
 * K = STATISTICAL_ERROR_FACTOR rows (40 in the reference, :26).  n: kw words; s, r, sn, z: [K][kw]; e, s_digest: 8 words
 * (a SHA-256 value). */
int32_t oracle_correct_key_challenge(uint32_t n_bits, uint32_t K, const uint32_t* n, const uint32_t* s, const uint32_t* r,
                                     uint32_t* out_sn, uint32_t* out_e, uint32_t* out_z, uint32_t* out_s_digest) {
  const size_t kw = n_bits / 32;
  mpz_t zn, e, t, dg;
  mpz_inits(zn, e, t, dg, NULL);
  limbs_to_mpz(zn, n, kw);
  mpz_t* items = (mpz_t*)malloc(sizeof(mpz_t) * (2 * K + 1));
  mpz_t* sv = (mpz_t*)malloc(sizeof(mpz_t) * K);
  mpz_init_set(items[0], zn);
  for (uint32_t i = 0; i < K; i++) {
    mpz_init(sv[i]); mpz_init(items[1 + i]); mpz_init(items[1 + K + i]);
    limbs_to_mpz(sv[i], s + i * kw, kw);
    mpz_powm(items[1 + i], sv[i], zn, zn);                 /* sn_i = s_i^n mod n  :73-76 */
    limbs_to_mpz(t, r + i * kw, kw);
    mpz_powm(items[1 + K + i], t, zn, zn);                 /* rn_i = r_i^n mod n  :86-89 */
    mpz_to_limbs(out_sn + i * kw, kw, items[1 + i]);
  }
  compute_digest(e, (const mpz_t*)items, (int)(2 * K + 1));   /* e = H(n, sn.., rn..)  :91 */
  mpz_to_limbs(out_e, 8, e);
  for (uint32_t i = 0; i < K; i++) {
    mpz_powm(t, sv[i], e, zn);                              /* z_i = r_i * s_i^e % n  :93-97 */
    limbs_to_mpz(dg, r + i * kw, kw);
    mpz_mul(t, dg, t);
    mpz_tdiv_r(t, t, zn);
    mpz_to_limbs(out_z + i * kw, kw, t);
  }
  compute_digest(dg, (const mpz_t*)sv, (int)K);               /* s_digest = H(s..)  :100 */
  mpz_to_limbs(out_s_digest, 8, dg);
  for (uint32_t i = 0; i < K; i++) { mpz_clear(sv[i]); mpz_clear(items[1 + i]); mpz_clear(items[1 + K + i]); }
  mpz_clear(items[0]);
  free(items); free(sv);
  mpz_clears(zn, e, t, dg, NULL);
  return 0;
}

/* CorrectKey::prove (correct_key.rs:104-162).  Returns 0 = Ok (out_s_digest written) or the CorrectKeyProveError:
 * 1 SniNotCoprimeWithN (:110-116), 2 ZiNotCoprimeWithN (:119-125), 3 RniNotCoprimeWithN (:143-148), 4 EWasntComputedCorrectly (:151-156).
 * e: e_words words (the challenge's e is attacker-chosen, any size); extract_nroot [upstream]: sn_i^(n^-1 mod phi) mod n. */
int32_t oracle_correct_key_prove(uint32_t n_bits, uint32_t K, const uint32_t* p, const uint32_t* q, const uint32_t* sn, const uint32_t* e,
                                 uint32_t e_words, const uint32_t* z, uint32_t* out_s_digest) {
  const size_t kw = n_bits / 32;
  int32_t rc = 0;
  mpz_t zp, zq, zn, phi, phimine, ze, t, u, g, d;
  mpz_inits(zp, zq, zn, phi, phimine, ze, t, u, g, d, NULL);
  limbs_to_mpz(zp, p, kw / 2); limbs_to_mpz(zq, q, kw / 2);
  mpz_mul(zn, zq, zp);                                        /* dk_n = q * p  :108 */
  limbs_to_mpz(ze, e, e_words);
  mpz_t* items = (mpz_t*)malloc(sizeof(mpz_t) * (2 * K + 1));
  mpz_init_set(items[0], zn);
  for (uint32_t i = 0; i < K; i++) { mpz_init(items[1 + i]); mpz_init(items[1 + K + i]); limbs_to_mpz(items[1 + i], sn + i * kw, kw); }
  for (uint32_t i = 0; i < K && !rc; i++) { mpz_gcd(g, zn, items[1 + i]); if (mpz_cmp_ui(g, 1) != 0) rc = 1; }
  for (uint32_t i = 0; i < K && !rc; i++) { limbs_to_mpz(t, z + i * kw, kw); mpz_gcd(g, zn, t); if (mpz_cmp_ui(g, 1) != 0) rc = 2; }
  if (!rc) {
    mpz_sub_ui(phi, zq, 1); mpz_sub_ui(t, zp, 1); mpz_mul(phi, phi, t);   /* :128 */
    mpz_tdiv_r(t, ze, phi); mpz_sub(phimine, phi, t);                      /* phi - (e % phi)  :130 */
    for (uint32_t i = 0; i < K; i++) {
      limbs_to_mpz(t, z + i * kw, kw);
      mpz_powm(t, t, zn, zn);                                             /* zn  :135 */
      mpz_powm(u, items[1 + i], phimine, zn);                             /* snphi  :136 */
      mpz_mul(t, t, u); mpz_tdiv_r(items[1 + K + i], t, zn);              /* :137 */
    }
    for (uint32_t i = 0; i < K && !rc; i++) { mpz_gcd(g, zn, items[1 + K + i]); if (mpz_cmp_ui(g, 1) != 0) rc = 3; }
  }
  if (!rc) {
    compute_digest(t, (const mpz_t*)items, (int)(2 * K + 1));              /* :151 */
    if (mpz_cmp(t, ze) != 0) rc = 4;
  }
  if (!rc) {
    mpz_invert(d, zn, phi);
    for (uint32_t i = 0; i < K; i++) mpz_powm(items[1 + K + i], items[1 + i], d, zn);   /* extract_nroot(dk, sn_i)  :159 */
    compute_digest(t, (const mpz_t*)(items + 1 + K), (int)K);
    mpz_to_limbs(out_s_digest, 8, t);
  }
  for (uint32_t i = 0; i < 2 * K + 1; i++) mpz_clear(items[i]);
  free(items);
  mpz_clears(zp, zq, zn, phi, phimine, ze, t, u, g, d, NULL);
  return rc;
}

/* CorrectKey::verify (correct_key.rs:164-171): proof.s_digest == va.s_digest */
int32_t oracle_correct_key_verify(const uint32_t* proof_s_digest, const uint32_t* aid_s_digest) {
  return memcmp(proof_s_digest, aid_s_digest, 32) == 0 ? ZKP_VERDICT_ACCEPT : ZKP_VERDICT_REJECT;
}

/* ------------------------------------------------------------------ CompositeDLogProof */
int32_t oracle_dlog_prove_batch(uint32_t n_bits, uint32_t y_bits, uint64_t batch, const uint32_t* N, const uint32_t* g,
                                const uint32_t* ni, const uint32_t* secret, const uint32_t* r, uint32_t* out_x,
                                uint32_t* out_y) {
  const size_t kw = n_bits / 32, yw = y_bits / 32;
  mpz_t it[4], e, zr, zs, y;
  mpz_inits(it[0], it[1], it[2], it[3], e, zr, zs, y, NULL);
  for (uint64_t b = 0; b < batch; b++) {
    limbs_to_mpz(it[1], g + b * kw, kw);
    limbs_to_mpz(it[2], N + b * kw, kw);
    limbs_to_mpz(it[3], ni + b * kw, kw);
    limbs_to_mpz(zr, r + b * 16, 16);          /* r < 2^512, wi_dlog_proof.rs:53-54 */
    limbs_to_mpz(zs, secret + b * 8, 8);       /* secret < 2^256 */
    mpz_powm(it[0], it[1], zr, it[2]);         /* x = g^r mod N :55 */
    compute_digest(e, (const mpz_t*)it, 4);    /* e = H(x,g,N,ni) :56-61 */
    mpz_mul(y, e, zs);
    mpz_add(y, y, zr);                         /* y = r + e*secret :62 */
    mpz_to_limbs(out_x + b * kw, kw, it[0]);
    mpz_to_limbs(out_y + b * yw, yw, y);
  }
  mpz_clears(it[0], it[1], it[2], it[3], e, zr, zs, y, NULL);
  return 0;
}

int32_t oracle_dlog_verify_batch(uint32_t n_bits, uint32_t y_bits, uint64_t batch, const uint32_t* N, const uint32_t* g,
                                 const uint32_t* ni, const uint32_t* x, const uint32_t* y, uint8_t* out_verdict) {
  const size_t kw = n_bits / 32, yw = y_bits / 32;
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1)
  for (int64_t b = 0; b < (int64_t)batch; b++) {
    mpz_t it[4], e, zy, t, u;
    mpz_inits(it[0], it[1], it[2], it[3], e, zy, t, u, NULL);
    limbs_to_mpz(it[0], x + b * kw, kw);
    limbs_to_mpz(it[1], g + b * kw, kw);
    limbs_to_mpz(it[2], N + b * kw, kw);
    limbs_to_mpz(it[3], ni + b * kw, kw);
    limbs_to_mpz(zy, y + b * yw, yw);
    mpz_set_ui(t, 1);
    mpz_mul_2exp(t, t, 128);
    int malformed = mpz_cmp(it[2], t) <= 0;          /* assert!(N > 2^K) :69 */
    mpz_gcd(t, it[1], it[2]);
    if (mpz_cmp_ui(t, 1) != 0) malformed = 1;        /* :72 */
    mpz_gcd(t, it[3], it[2]);
    if (mpz_cmp_ui(t, 1) != 0) malformed = 1;        /* :73 */
    if (malformed) {
      out_verdict[b] = ZKP_VERDICT_MALFORMED;
    } else {
      compute_digest(e, (const mpz_t*)it, 4);        /* :75-80 */
      mpz_powm(t, it[3], e, it[2]);                  /* ni^e :81 */
      mpz_powm(u, it[1], zy, it[2]);                 /* g^y :82 */
      mpz_mul(t, t, u);
      mpz_mod(t, t, it[2]);                          /* mod_mul :83 */
      out_verdict[b] = mpz_cmp(t, it[0]) == 0 ? ZKP_VERDICT_ACCEPT : ZKP_VERDICT_REJECT; /* :86 */
    }
    mpz_clears(it[0], it[1], it[2], it[3], e, zy, t, u, NULL);
  }
  return 0;
}

/* transcript hash only, for hashing KATs: e bytes + length for one proof's (n, c1, c2) */
int32_t oracle_fs_challenge(uint32_t n_bits, uint32_t ef, const uint32_t* n, const uint32_t* c1, const uint32_t* c2,
                            uint8_t out_e[32], uint8_t* out_e_len) {
  mpz_t zn;
  mpz_init(zn);
  limbs_to_mpz(zn, n, n_bits / 32);
  *out_e_len = (uint8_t)fs_challenge(zn, c1, c2, ef, n_bits / 32, out_e);
  mpz_clear(zn);
  return 0;
}

/* ------------------------------------------------------------------ ZeroProof / CiphertextProof
 * [upstream kzen-paillier] Paillier::mul(ek, m, c) = c^m mod nn ; Paillier::add(ek, c1, c2) = c1*c2 mod nn. */
static void sigma_challenge(mpz_t e, const mpz_t n, const mpz_t c, const mpz_t a) {
  mpz_t it[3];
  mpz_init_set(it[0], n); mpz_init_set(it[1], c); mpz_init_set(it[2], a);
  compute_digest(e, (const mpz_t*)it, 3); /* zero_enc_proof.rs:54-58,67-71 ; correct_ciphertext.rs:53-57,67-71 */
  mpz_clears(it[0], it[1], it[2], NULL);
}

/* with_x = 0: ZeroProof::prove (zero_enc_proof.rs:44-64); with_x = 1: CiphertextProof::prove (correct_ciphertext.rs:42-64) */
static int32_t sigma_prove(int with_x, uint32_t n_bits, uint64_t batch, const uint32_t* n, uint64_t n_stride, const uint32_t* c,
                           const uint32_t* x, const uint32_t* r, const uint32_t* x_prime, const uint32_t* r_prime, uint32_t* out_z1,
                           uint32_t* out_z, uint32_t* out_commit) {
  const size_t kw = n_bits / 32, z1w = kw + ZKP_Z1_EXTRA_LIMBS;
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1)
  for (int64_t b = 0; b < (int64_t)batch; b++) {
    mpz_t zn, znn, zc, zr, zrp, zx, zxp, com, e, t, u;
    mpz_inits(zn, znn, zc, zr, zrp, zx, zxp, com, e, t, u, NULL);
    limbs_to_mpz(zn, n + b * n_stride, kw);
    mpz_mul(znn, zn, zn);
    limbs_to_mpz(zc, c + b * 2 * kw, 2 * kw);
    limbs_to_mpz(zr, r + b * kw, kw);
    limbs_to_mpz(zrp, r_prime + b * kw, kw);
    if (with_x) { limbs_to_mpz(zx, x + b * kw, kw); limbs_to_mpz(zxp, x_prime + b * kw, kw); }
    enc_mpz(com, zn, znn, zxp, zrp, u);            /* a = Enc(0, r') / c' = Enc(x', r') */
    sigma_challenge(e, zn, zc, com);
    if (with_x) {
      mpz_mul(t, zx, e);
      mpz_add(t, t, zxp);                          /* z1 = x' + x*e  (correct_ciphertext.rs:59) */
      mpz_to_limbs(out_z1 + b * z1w, z1w, t);
    }
    mpz_powm(t, zr, e, znn);                       /* r^e mod nn */
    mpz_mul(t, t, zrp);
    mpz_mod(t, t, znn);                            /* mod_mul(r', r^e, nn) */
    mpz_to_limbs(out_z + b * 2 * kw, 2 * kw, t);
    mpz_to_limbs(out_commit + b * 2 * kw, 2 * kw, com);
    mpz_clears(zn, znn, zc, zr, zrp, zx, zxp, com, e, t, u, NULL);
  }
  return 0;
}

static int32_t sigma_verify(int with_z1, uint32_t n_bits, uint64_t batch, const uint32_t* n, uint64_t n_stride, const uint32_t* c,
                            const uint32_t* z1, const uint32_t* z, const uint32_t* commit, uint8_t* out_verdict) {
  const size_t kw = n_bits / 32, z1w = kw + ZKP_Z1_EXTRA_LIMBS;
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1)
  for (int64_t b = 0; b < (int64_t)batch; b++) {
    mpz_t zn, znn, zc, zz1, zz, com, e, cz, t, u;
    mpz_inits(zn, znn, zc, zz1, zz, com, e, cz, t, u, NULL);
    limbs_to_mpz(zn, n + b * n_stride, kw);
    mpz_mul(znn, zn, zn);
    limbs_to_mpz(zc, c + b * 2 * kw, 2 * kw);
    if (with_z1) limbs_to_mpz(zz1, z1 + b * z1w, z1w);
    limbs_to_mpz(zz, z + b * 2 * kw, 2 * kw);
    limbs_to_mpz(com, commit + b * 2 * kw, 2 * kw);
    sigma_challenge(e, zn, zc, com);
    enc_mpz(cz, zn, znn, zz1, zz, u);              /* c_z = Enc(z1 | 0, z) */
    mpz_powm(t, zc, e, znn);                       /* Paillier::mul: c^e */
    mpz_mul(t, t, com);
    mpz_mod(t, t, znn);                            /* Paillier::add: * a */
    out_verdict[b] = mpz_cmp(cz, t) == 0 ? ZKP_VERDICT_ACCEPT : ZKP_VERDICT_REJECT;
    mpz_clears(zn, znn, zc, zz1, zz, com, e, cz, t, u, NULL);
  }
  return 0;
}

int32_t oracle_zero_proof_prove_batch(uint32_t n_bits, uint64_t batch, const uint32_t* n, uint64_t n_stride, const uint32_t* c,
                                      const uint32_t* r, const uint32_t* r_prime, uint32_t* out_z, uint32_t* out_a) {
  return sigma_prove(0, n_bits, batch, n, n_stride, c, NULL, r, NULL, r_prime, NULL, out_z, out_a);
}
int32_t oracle_zero_proof_verify_batch(uint32_t n_bits, uint64_t batch, const uint32_t* n, uint64_t n_stride, const uint32_t* c,
                                       const uint32_t* z, const uint32_t* a, uint8_t* out_verdict) {
  return sigma_verify(0, n_bits, batch, n, n_stride, c, NULL, z, a, out_verdict);
}
int32_t oracle_ciphertext_proof_prove_batch(uint32_t n_bits, uint64_t batch, const uint32_t* n, uint64_t n_stride, const uint32_t* c,
                                            const uint32_t* x, const uint32_t* r, const uint32_t* x_prime, const uint32_t* r_prime,
                                            uint32_t* out_z1, uint32_t* out_z2, uint32_t* out_c_prime) {
  return sigma_prove(1, n_bits, batch, n, n_stride, c, x, r, x_prime, r_prime, out_z1, out_z2, out_c_prime);
}
int32_t oracle_ciphertext_proof_verify_batch(uint32_t n_bits, uint64_t batch, const uint32_t* n, uint64_t n_stride, const uint32_t* c,
                                             const uint32_t* z1, const uint32_t* z2, const uint32_t* c_prime, uint8_t* out_verdict) {
  return sigma_verify(1, n_bits, batch, n, n_stride, c, z1, z2, c_prime, out_verdict);
}

/* ------------------------------------------------------------------ VerlinProof (verlin_proof.rs:35-165) */
static void gen_phi_mpz(mpz_t out, const mpz_t n, const mpz_t nn, const mpz_t c, const mpz_t cp, const mpz_t y, const mpz_t yp,
                        const mpz_t ypp, const mpz_t ry, mpz_t t, mpz_t u) {
  mpz_powm(out, c, y, nn);          /* Paillier::mul(c, y)  :147-151 */
  mpz_powm(t, cp, yp, nn);          /* Paillier::mul(c', y') :152-156 */
  mpz_mul(out, out, t);
  mpz_mod(out, out, nn);            /* Paillier::add :162 */
  enc_mpz(t, n, nn, ypp, ry, u);    /* Enc(y'', r_y) :157-161 */
  mpz_mul(out, out, t);
  mpz_mod(out, out, nn);            /* Paillier::add :163 */
}

static void verlin_challenge(mpz_t e, const mpz_t n, const mpz_t c, const mpz_t cp, const mpz_t phi_x, const mpz_t phi_a) {
  mpz_t it[5];
  mpz_init_set(it[0], n); mpz_init_set(it[1], c); mpz_init_set(it[2], cp); mpz_init_set(it[3], phi_x); mpz_init_set(it[4], phi_a);
  compute_digest(e, (const mpz_t*)it, 5); /* :78-84, 102-108 */
  for (int i = 0; i < 5; i++) mpz_clear(it[i]);
}

int32_t oracle_verlin_proof_prove_batch(uint32_t n_bits, uint64_t batch, const uint32_t* n, uint64_t n_stride, const uint32_t* c,
                                        const uint32_t* c_prime, const uint32_t* phi_x, const uint32_t* x, const uint32_t* x_prime,
                                        const uint32_t* x_double_prime, const uint32_t* r_x, const uint32_t* a, const uint32_t* a_prime,
                                        const uint32_t* a_double_prime, const uint32_t* r_a, uint32_t* out_phi_a, uint32_t* out_z,
                                        uint32_t* out_z_prime, uint32_t* out_z_double_prime, uint32_t* out_r_z) {
  const size_t kw = n_bits / 32, zw = kw + ZKP_Z1_EXTRA_LIMBS;
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1)
  for (int64_t b = 0; b < (int64_t)batch; b++) {
    mpz_t zn, znn, zc, zcp, zphx, v[8], pa, e, t, u;
    mpz_inits(zn, znn, zc, zcp, zphx, pa, e, t, u, NULL);
    const uint32_t* src[8] = {x, x_prime, x_double_prime, r_x, a, a_prime, a_double_prime, r_a};
    for (int i = 0; i < 8; i++) { mpz_init(v[i]); limbs_to_mpz(v[i], src[i] + b * kw, kw); }
    limbs_to_mpz(zn, n + b * n_stride, kw);
    mpz_mul(znn, zn, zn);
    limbs_to_mpz(zc, c + b * 2 * kw, 2 * kw);
    limbs_to_mpz(zcp, c_prime + b * 2 * kw, 2 * kw);
    limbs_to_mpz(zphx, phi_x + b * 2 * kw, 2 * kw);
    gen_phi_mpz(pa, zn, znn, zc, zcp, v[4], v[5], v[6], v[7], t, u);   /* phi_a :69-77 */
    verlin_challenge(e, zn, zc, zcp, zphx, pa);
    uint32_t* outs[3] = {out_z, out_z_prime, out_z_double_prime};
    for (int i = 0; i < 3; i++) { mpz_mul(t, v[i], e); mpz_add(t, t, v[4 + i]); mpz_to_limbs(outs[i] + b * zw, zw, t); }   /* :85-87 */
    mpz_powm(t, v[3], e, znn);
    mpz_mul(t, t, v[7]);
    mpz_mod(t, t, znn);                                               /* r_z :88-89 */
    mpz_to_limbs(out_r_z + b * 2 * kw, 2 * kw, t);
    mpz_to_limbs(out_phi_a + b * 2 * kw, 2 * kw, pa);
    for (int i = 0; i < 8; i++) mpz_clear(v[i]);
    mpz_clears(zn, znn, zc, zcp, zphx, pa, e, t, u, NULL);
  }
  return 0;
}

int32_t oracle_verlin_proof_verify_batch(uint32_t n_bits, uint64_t batch, const uint32_t* n, uint64_t n_stride, const uint32_t* c,
                                         const uint32_t* c_prime, const uint32_t* phi_x, const uint32_t* phi_a, const uint32_t* z,
                                         const uint32_t* z_prime, const uint32_t* z_double_prime, const uint32_t* r_z, uint8_t* out_verdict) {
  const size_t kw = n_bits / 32, zw = kw + ZKP_Z1_EXTRA_LIMBS;
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1)
  for (int64_t b = 0; b < (int64_t)batch; b++) {
    mpz_t zn, znn, zc, zcp, zphx, pa, z0, z1, z2, rz, e, lhs, rhs, t, u;
    mpz_inits(zn, znn, zc, zcp, zphx, pa, z0, z1, z2, rz, e, lhs, rhs, t, u, NULL);
    limbs_to_mpz(zn, n + b * n_stride, kw);
    mpz_mul(znn, zn, zn);
    limbs_to_mpz(zc, c + b * 2 * kw, 2 * kw);
    limbs_to_mpz(zcp, c_prime + b * 2 * kw, 2 * kw);
    limbs_to_mpz(zphx, phi_x + b * 2 * kw, 2 * kw);
    limbs_to_mpz(pa, phi_a + b * 2 * kw, 2 * kw);
    limbs_to_mpz(z0, z + b * zw, zw); limbs_to_mpz(z1, z_prime + b * zw, zw); limbs_to_mpz(z2, z_double_prime + b * zw, zw);
    limbs_to_mpz(rz, r_z + b * 2 * kw, 2 * kw);
    verlin_challenge(e, zn, zc, zcp, zphx, pa);
    mpz_powm(rhs, zphx, e, znn);
    mpz_mul(rhs, rhs, pa);
    mpz_mod(rhs, rhs, znn);                                           /* phi_x^e * phi_a :109-118 */
    gen_phi_mpz(lhs, zn, znn, zc, zcp, z0, z1, z2, rz, t, u);         /* phi_z :120-128 */
    out_verdict[b] = mpz_cmp(lhs, rhs) == 0 ? ZKP_VERDICT_ACCEPT : ZKP_VERDICT_REJECT;
    mpz_clears(zn, znn, zc, zcp, zphx, pa, z0, z1, z2, rz, e, lhs, rhs, t, u, NULL);
  }
  return 0;
}

/* ------------------------------------------------------------------ mod_inv (curv BigInt::mod_inv -> mpz_invert) */
int32_t oracle_modinv_batch(uint32_t mod_bits, uint64_t count, const uint32_t* a, const uint32_t* modulus, uint64_t mod_stride,
                            uint32_t* out, uint8_t* out_status) {
  const size_t kw = mod_bits / 32;
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 4)
  for (int64_t i = 0; i < (int64_t)count; i++) {
    mpz_t za, zm, zr;
    mpz_inits(za, zm, zr, NULL);
    limbs_to_mpz(za, a + i * kw, kw);
    limbs_to_mpz(zm, modulus + i * mod_stride, kw);
    memset(out + i * kw, 0, kw * 4);
    if (mpz_cmp(za, zm) >= 0 || mpz_even_p(zm) || mpz_cmp_ui(zm, 3) < 0) out_status[i] = ZKP_INV_DOMAIN;
    else if (!mpz_invert(zr, za, zm)) out_status[i] = ZKP_INV_NONE;
    else { out_status[i] = ZKP_INV_OK; mpz_to_limbs(out + i * kw, kw, zr); }
    mpz_clears(za, zm, zr, NULL);
  }
  return 0;
}

/* ------------------------------------------------------------------ MulProof (multiplication_proof.rs:60-146) */
static void mul_challenge(mpz_t e, const mpz_t n, const mpz_t ea, const mpz_t eb, const mpz_t ec, const mpz_t ed, const mpz_t edb) {
  mpz_t it[6];
  mpz_init_set(it[0], n); mpz_init_set(it[1], ea); mpz_init_set(it[2], eb); mpz_init_set(it[3], ec); mpz_init_set(it[4], ed); mpz_init_set(it[5], edb);
  compute_digest(e, (const mpz_t*)it, 6); /* :77-84, 107-114 */
  for (int i = 0; i < 6; i++) mpz_clear(it[i]);
}

int32_t oracle_mul_proof_prove_batch(uint32_t n_bits, uint64_t batch, const uint32_t* n, uint64_t n_stride, const uint32_t* e_a,
                                     const uint32_t* e_b, const uint32_t* e_c, const uint32_t* a, const uint32_t* b, const uint32_t* r_a,
                                     const uint32_t* r_b, const uint32_t* r_c, const uint32_t* d, const uint32_t* r_d, uint32_t* out_f,
                                     uint32_t* out_z1, uint32_t* out_z2, uint32_t* out_e_d, uint32_t* out_e_db, uint8_t* out_status) {
  const size_t kw = n_bits / 32;
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1)
  for (int64_t i = 0; i < (int64_t)batch; i++) {
    mpz_t zn, znn, ea, eb, ec, za, zb, ra, rb, rc, zd, rd, ed, edb, rdb, db, e, f, z1, z2, t, u;
    mpz_inits(zn, znn, ea, eb, ec, za, zb, ra, rb, rc, zd, rd, ed, edb, rdb, db, e, f, z1, z2, t, u, NULL);
    limbs_to_mpz(zn, n + i * n_stride, kw); mpz_mul(znn, zn, zn);
    limbs_to_mpz(ea, e_a + i * 2 * kw, 2 * kw); limbs_to_mpz(eb, e_b + i * 2 * kw, 2 * kw); limbs_to_mpz(ec, e_c + i * 2 * kw, 2 * kw);
    limbs_to_mpz(za, a + i * kw, kw); limbs_to_mpz(zb, b + i * kw, kw);
    limbs_to_mpz(ra, r_a + i * kw, kw); limbs_to_mpz(rb, r_b + i * kw, kw); limbs_to_mpz(rc, r_c + i * kw, kw);
    limbs_to_mpz(zd, d + i * kw, kw); limbs_to_mpz(rd, r_d + i * kw, kw);
    enc_mpz(ed, zn, znn, zd, rd, t);                 /* :63-68 */
    mpz_mul(rdb, rd, rb);                            /* :69 */
    mpz_mul(db, zd, zb);                             /* :70 */
    enc_mpz(edb, zn, znn, db, rdb, t);               /* :71-76 */
    mul_challenge(e, zn, ea, eb, ec, ed, edb);
    mpz_mul(f, e, za); mpz_mod(f, f, zn);            /* mod_mul :87 */
    mpz_add(f, f, zd); mpz_mod(f, f, zn);            /* mod_add :88 */
    mpz_powm(z1, ra, e, znn); mpz_mul(z1, z1, rd); mpz_mod(z1, z1, znn);   /* :89-90 */
    mpz_powm(t, rc, e, znn); mpz_mul(t, rdb, t); mpz_mod(t, t, znn);       /* :92-93 */
    memset(out_f + i * kw, 0, kw * 4); memset(out_z1 + i * 2 * kw, 0, kw * 8); memset(out_z2 + i * 2 * kw, 0, kw * 8);
    mpz_to_limbs(out_e_d + i * 2 * kw, 2 * kw, ed); mpz_to_limbs(out_e_db + i * 2 * kw, 2 * kw, edb);
    if (!mpz_invert(u, t, znn)) {                    /* .unwrap() :95 */
      out_status[i] = ZKP_VERDICT_MALFORMED;
    } else {
      mpz_powm(z2, rb, f, znn); mpz_mul(z2, z2, u); mpz_mod(z2, z2, znn);  /* :91,96 */
      mpz_to_limbs(out_f + i * kw, kw, f); mpz_to_limbs(out_z1 + i * 2 * kw, 2 * kw, z1); mpz_to_limbs(out_z2 + i * 2 * kw, 2 * kw, z2);
      out_status[i] = 0;
    }
    mpz_clears(zn, znn, ea, eb, ec, za, zb, ra, rb, rc, zd, rd, ed, edb, rdb, db, e, f, z1, z2, t, u, NULL);
  }
  return 0;
}

int32_t oracle_mul_proof_verify_batch(uint32_t n_bits, uint64_t batch, const uint32_t* n, uint64_t n_stride, const uint32_t* e_a,
                                      const uint32_t* e_b, const uint32_t* e_c, const uint32_t* f, const uint32_t* z1, const uint32_t* z2,
                                      const uint32_t* e_d, const uint32_t* e_db, uint8_t* out_verdict) {
  const size_t kw = n_bits / 32;
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1)
  for (int64_t i = 0; i < (int64_t)batch; i++) {
    mpz_t zn, znn, ea, eb, ec, zf, s1, s2, ed, edb, e, c1, c2, l1, l2, t, u, zero;
    mpz_inits(zn, znn, ea, eb, ec, zf, s1, s2, ed, edb, e, c1, c2, l1, l2, t, u, zero, NULL);
    limbs_to_mpz(zn, n + i * n_stride, kw); mpz_mul(znn, zn, zn);
    limbs_to_mpz(ea, e_a + i * 2 * kw, 2 * kw); limbs_to_mpz(eb, e_b + i * 2 * kw, 2 * kw); limbs_to_mpz(ec, e_c + i * 2 * kw, 2 * kw);
    limbs_to_mpz(zf, f + i * kw, kw); limbs_to_mpz(s1, z1 + i * 2 * kw, 2 * kw); limbs_to_mpz(s2, z2 + i * 2 * kw, 2 * kw);
    limbs_to_mpz(ed, e_d + i * 2 * kw, 2 * kw); limbs_to_mpz(edb, e_db + i * 2 * kw, 2 * kw);
    mul_challenge(e, zn, ea, eb, ec, ed, edb);
    enc_mpz(c1, zn, znn, zf, s1, t);                 /* Enc(f, z1) :116-122 */
    enc_mpz(c2, zn, znn, zero, s2, t);               /* Enc(0, z2) :123-129 */
    mpz_powm(l1, ea, e, znn); mpz_mul(l1, l1, ed); mpz_mod(l1, l1, znn);    /* :131-132 */
    mpz_powm(t, ec, e, znn); mpz_mul(t, edb, t); mpz_mod(t, t, znn);        /* :133-134 */
    if (!mpz_invert(u, t, znn)) {                    /* .unwrap() :135 */
      out_verdict[i] = ZKP_VERDICT_MALFORMED;
    } else {
      mpz_powm(l2, eb, zf, znn); mpz_mul(l2, l2, u); mpz_mod(l2, l2, znn);  /* :136-137 */
      out_verdict[i] = (mpz_cmp(l1, c1) == 0 && mpz_cmp(l2, c2) == 0) ? ZKP_VERDICT_ACCEPT : ZKP_VERDICT_REJECT;
    }
    mpz_clears(zn, znn, ea, eb, ec, zf, s1, s2, ed, edb, e, c1, c2, l1, l2, t, u, zero, NULL);
  }
  return 0;
}

/* ------------------------------------------------------------------ CorrectMessageProof (correct_message.rs:35-162) */
/* u_i = ciphertext * ((m_i*n + 1) % nn)^-1 mod nn (:50-56, 136-144); returns 0 when mod_inv has no result */
static int cm_u(mpz_t u, const mpz_t ct, const mpz_t m, const mpz_t n, const mpz_t nn, mpz_t t) {
  mpz_mul(t, m, n); mpz_add_ui(t, t, 1); mpz_tdiv_r(t, t, nn);
  if (!mpz_invert(t, t, nn)) return 0;
  mpz_mul(u, ct, t); mpz_mod(u, u, nn);
  return 1;
}
static void cm_challenge(mpz_t chal, const mpz_t* a_vec, uint32_t K) {
  compute_digest(chal, a_vec, (int)K);
  mpz_fdiv_r_2exp(chal, chal, 256);                  /* .modulus(2^B) :88, :129 */
}

int32_t oracle_correct_message_prove_batch(uint32_t n_bits, uint64_t batch, uint32_t K, const uint32_t* n, uint64_t n_stride,
                                           const uint32_t* valid_messages, const uint32_t* message, const uint32_t* r, const uint32_t* e_sim,
                                           const uint32_t* z_sim, const uint32_t* w, uint32_t* out_ciphertext, uint32_t* out_e_vec,
                                           uint32_t* out_z_vec, uint32_t* out_a_vec, uint8_t* out_status) {
  const size_t kw = n_bits / 32;
  if (K < 1) return 1;
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1)
  for (int64_t b = 0; b < (int64_t)batch; b++) {
    mpz_t zn, znn, msg, zr, zw, ct, m, u, t, x, chal, sum, ei, zi;
    mpz_inits(zn, znn, msg, zr, zw, ct, m, u, t, x, chal, sum, ei, zi, NULL);
    mpz_t* av = (mpz_t*)malloc(sizeof(mpz_t) * K);
    for (uint32_t i = 0; i < K; i++) mpz_init(av[i]);
    limbs_to_mpz(zn, n + b * n_stride, kw); mpz_mul(znn, zn, zn);
    limbs_to_mpz(msg, message + b * kw, kw); limbs_to_mpz(zr, r + b * kw, kw); limbs_to_mpz(zw, w + b * kw, kw);
    enc_mpz(ct, zn, znn, msg, zr, t);                /* :44-49 */
    int panic = 0;
    uint32_t j = 0;
    uint8_t* match = (uint8_t*)calloc(K, 1);
    for (uint32_t i = 0; i < K && !panic; i++) {
      limbs_to_mpz(m, valid_messages + (b * K + i) * kw, kw);
      match[i] = mpz_cmp(m, msg) == 0;
      if (!cm_u(u, ct, m, zn, znn, t)) { panic = 1; break; }          /* :53 (never: gcd(1 + m n, nn) = 1) */
      if (match[i]) {
        mpz_powm(av[i], zw, zn, znn);                                 /* :69-70 */
      } else {
        if (j >= K - 1) { panic = 1; break; }                         /* zi_vec[j] out of bounds :72 */
        limbs_to_mpz(x, z_sim + (b * (K - 1) + j) * kw, kw);
        mpz_powm(x, x, zn, znn);                                      /* zi^n :72 */
        limbs_to_mpz(t, e_sim + (b * (K - 1) + j) * 8, 8);
        mpz_powm(u, u, t, znn);                                       /* ui^ei :73 */
        if (!mpz_invert(u, u, znn)) { panic = 1; break; }             /* :74 */
        mpz_mul(av[i], x, u); mpz_mod(av[i], av[i], znn);             /* :76 */
        j++;
      }
    }
    memset(out_e_vec + b * K * 8, 0, K * 32); memset(out_z_vec + b * K * kw, 0, K * kw * 4); memset(out_a_vec + b * K * 2 * kw, 0, K * kw * 8);
    mpz_to_limbs(out_ciphertext + b * 2 * kw, 2 * kw, ct);
    if (panic) {
      out_status[b] = ZKP_VERDICT_MALFORMED;
    } else {
      cm_challenge(chal, (const mpz_t*)av, K);
      mpz_set_ui(sum, 0);
      for (uint32_t k = 0; k + 1 < K; k++) { limbs_to_mpz(t, e_sim + (b * (K - 1) + k) * 8, 8); mpz_add(sum, sum, t); }   /* :90-91 (all K-1 of them) */
      mpz_fdiv_r_2exp(sum, sum, 256);
      mpz_sub(ei, chal, sum); mpz_fdiv_r_2exp(ei, ei, 256);            /* mod_sub :93 */
      mpz_powm(zi, zr, ei, zn); mpz_mul(zi, zw, zi); mpz_mod(zi, zi, zn);   /* :94-95 */
      j = 0;
      for (uint32_t i = 0; i < K; i++) {
        if (match[i]) {
          mpz_to_limbs(out_e_vec + (b * K + i) * 8, 8, ei); mpz_to_limbs(out_z_vec + (b * K + i) * kw, kw, zi);
        } else {
          memcpy(out_e_vec + (b * K + i) * 8, e_sim + (b * (K - 1) + j) * 8, 32);
          memcpy(out_z_vec + (b * K + i) * kw, z_sim + (b * (K - 1) + j) * kw, kw * 4);
          j++;
        }
        mpz_to_limbs(out_a_vec + (b * K + i) * 2 * kw, 2 * kw, av[i]);
      }
      out_status[b] = 0;
    }
    free(match);
    for (uint32_t i = 0; i < K; i++) mpz_clear(av[i]);
    free(av);
    mpz_clears(zn, znn, msg, zr, zw, ct, m, u, t, x, chal, sum, ei, zi, NULL);
  }
  return 0;
}

int32_t oracle_correct_message_verify_batch(uint32_t n_bits, uint64_t batch, uint32_t K, const uint32_t* n, uint64_t n_stride,
                                            const uint32_t* valid_messages, const uint32_t* ciphertext, const uint32_t* e_vec,
                                            const uint32_t* z_vec, const uint32_t* a_vec, uint8_t* out_verdict) {
  const size_t kw = n_bits / 32;
  if (K < 1) return 1;
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1)
  for (int64_t b = 0; b < (int64_t)batch; b++) {
    mpz_t zn, znn, ct, m, u, t, x, chal, sum;
    mpz_inits(zn, znn, ct, m, u, t, x, chal, sum, NULL);
    mpz_t* av = (mpz_t*)malloc(sizeof(mpz_t) * K);
    limbs_to_mpz(zn, n + b * n_stride, kw); mpz_mul(znn, zn, zn);
    limbs_to_mpz(ct, ciphertext + b * 2 * kw, 2 * kw);
    for (uint32_t i = 0; i < K; i++) {
      mpz_init(av[i]); limbs_to_mpz(av[i], a_vec + (b * K + i) * 2 * kw, 2 * kw);
      limbs_to_mpz(t, e_vec + (b * K + i) * 8, 8); mpz_add(sum, sum, t);
    }
    cm_challenge(chal, (const mpz_t*)av, K);
    mpz_fdiv_r_2exp(sum, sum, 256);
    uint8_t v = ZKP_VERDICT_ACCEPT;
    if (mpz_cmp(chal, sum) != 0) v = ZKP_VERDICT_MALFORMED;           /* assert_eq! :132 */
    for (uint32_t i = 0; i < K && v != ZKP_VERDICT_MALFORMED; i++) {
      limbs_to_mpz(m, valid_messages + (b * K + i) * kw, kw);
      if (!cm_u(u, ct, m, zn, znn, t)) { v = ZKP_VERDICT_MALFORMED; break; }
      limbs_to_mpz(x, z_vec + (b * K + i) * kw, kw);
      mpz_powm(x, x, zn, znn);                                        /* zi^n :146 */
      limbs_to_mpz(t, e_vec + (b * K + i) * 8, 8);
      mpz_powm(u, u, t, znn);                                         /* ui^ei :147 */
      mpz_mul(u, u, av[i]); mpz_mod(u, u, znn);                       /* :148 */
      if (mpz_cmp(u, x) != 0) v = ZKP_VERDICT_REJECT;                 /* :149; all() :151 (no panic can follow) */
    }
    out_verdict[b] = v;
    for (uint32_t i = 0; i < K; i++) mpz_clear(av[i]);
    free(av);
    mpz_clears(zn, znn, ct, m, u, t, x, chal, sum, NULL);
  }
  return 0;
}

/* ------------------------------------------------------------------ wire format: decimal strings (serialize.rs:1-78)
 * BigInt::from_str_radix(s, 10) -> mpz_set_str; BigInt::to_str_radix(10) -> mpz_get_str. */
int32_t oracle_decimal_to_limbs_batch(const char* text, const zkp_dec_item* items, uint64_t count, uint32_t* dst, uint8_t* out_status) {
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 16)
  for (int64_t i = 0; i < (int64_t)count; i++) {
    const zkp_dec_item it = items[i];
    char* s = (char*)malloc((size_t)it.len + 1);
    memcpy(s, text + it.text_off, it.len);
    s[it.len] = 0;
    mpz_t z;
    mpz_init(z);
    uint8_t st = ZKP_DEC_OK;
    if (memchr(s, 0, it.len) != NULL) st = ZKP_DEC_INVALID;          /* CString::new fails on an interior NUL */
    else if (mpz_set_str(z, s, 10) != 0) st = ZKP_DEC_INVALID;
    else if (mpz_sgn(z) < 0) st = ZKP_DEC_NEGATIVE;
    else if (mpz_sizeinbase(z, 2) > (size_t)it.words * 32) st = ZKP_DEC_OVERFLOW;
    memset(dst + it.dst_off, 0, (size_t)it.words * 4);
    if (st == ZKP_DEC_OK) mpz_to_limbs(dst + it.dst_off, it.words, z);
    out_status[i] = st;
    mpz_clear(z);
    free(s);
  }
  return 0;
}

int32_t oracle_limbs_to_decimal_batch(const uint32_t* src, uint64_t src_stride, uint32_t words, uint64_t count, char* out_text, uint32_t pitch,
                                      uint32_t* out_len) {
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 16)
  for (int64_t i = 0; i < (int64_t)count; i++) {
    mpz_t z;
    mpz_init(z);
    limbs_to_mpz(z, src + i * src_stride, words);
    char* s = mpz_get_str(NULL, 10, z);
    const size_t n = strlen(s);
    memcpy(out_text + i * (uint64_t)pitch + pitch - n, s, n);
    out_len[i] = (uint32_t)n;
    free(s);
    mpz_clear(z);
  }
  return 0;
}
