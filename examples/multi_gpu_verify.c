/* multi_gpu_verify.c — a plain-C caller of libzkp_hip.so that shards a RangeProofNi batch over several device contexts
 * (include/zkp_hip.h: zkp_multi_*).  This is the shape of what a Rust `extern "C"` caller does: fill the
 * structure-of-arrays buffers, one call to prove, one call to verify, read verdict bytes.
 *
 *   gcc -O2 -I../include multi_gpu_verify.c -L../zk-paillier_amd -lzkp_hip -Wl,-rpath,$PWD/../zk-paillier_amd -o multi_gpu_verify
 *   ./multi_gpu_verify 0 1 2 3        # device ids, one context each (an id may repeat: "0 0" = two contexts on GPU 0)
 *   ZKP_DEVICES=0,1,2,3,4,5,6,7 ZKP_BATCH=4096 ./multi_gpu_verify      # the same from the environment: a one-command check on an 8-GPU node
 *   ZKP_GATHER=rccl ZKP_DEVICES=0,1,2,3,4,5,6,7 ./multi_gpu_verify     # outputs reassembled by RCCL all-gathers inside the library (distinct
 *                                                                      # GPUs): the whole c1 / c2 / verdicts end up device-resident on EVERY GPU
 *
 * Prints the block, the wall time and the compute / gather phases of every device context for the prove and the verify call
 * (zkp_multi_last_timing, zkp_multi_last_phases), then runs
 * the SAME batch through ONE context (device of the first id) and compares ciphertexts, responses and verdicts byte for byte:
 * exit status 0 only if they are identical, 63/64 (B-1 of B) proofs are accepted and the tampered one is rejected.
 *
 * The statements are synthetic: range = 3 * 2^254 so that T = range / 3 = 2^254 needs no division; w1 = T + u, w2 = u with
 * u < 2^200 (range_proof.rs:133-149 draws w1 from [T, 2T) and sets w2 = w1 - T), x < 2^200 < T, randomness below 2^2000 < n.
 * Key: the fixed keypair of the reference's tests (range_proof_ni.rs:141-145), n as little-endian 32-bit limbs. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zkp_hip.h"

enum { N_BITS = 2048, KW = N_BITS / 32, EF = ZKP_SECURITY_PARAMETER };

/* n = p * q of range_proof_ni.rs:141-145, big-endian hex */
static const char* N_HEX =
    "baf57ae62d530cb97f94e1406949ac82a01272a0fc23a68b5b3829e510f34890bf4c3d4e60b089e12ce61765c7cd72689aeeb73a8344c9401f97213d9a260fb0"
    "5905d58122a5fee6b4fbee8ddd85660e5e848faecc4b9aba6634d1c720ebf561d9b8e96891b8ebbe7be5f85db63fbf42792f6713a65d17228883efbdd3141b3e"
    "358dbddeb012e044820a1996ae26a2403b0e75416f2c80cea74cdfe514e37ee25d6e449075ade778fde8fc1493ec75f92b0133eaf1465d3223941d5c9bcd32d4"
    "8519087f07a06acceaea665cd8e34861c3cdc0905063d15dd010fdeb79bd5707b3d08b2cf1baf68a78cfd0bc77a17b4dbc7528e8b329a3a1ee023b74297d758b";

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd32(void) {               /* xorshift64*: the boundary takes randomness as an input */
  rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
  return (uint32_t)((rng_state * 0x2545F4914F6CDD1Dull) >> 32);
}
static void rand_bits(uint32_t* v, int bits) {  /* uniform below 2^bits in KW limbs */
  memset(v, 0, KW * 4);
  for (int w = 0; w < (bits + 31) / 32; w++) v[w] = rnd32();
  if (bits % 32) v[bits / 32] &= (1u << (bits % 32)) - 1;
}
static void hex_to_limbs(const char* hex, uint32_t* out, int nlimbs) {
  memset(out, 0, nlimbs * 4);
  const int len = (int)strlen(hex);
  for (int i = 0; i < len; i++) {
    const char c = hex[len - 1 - i];
    const uint32_t d = c <= '9' ? c - '0' : (c | 32) - 'a' + 10;
    out[i / 8] |= d << (4 * (i % 8));
  }
}
#define CHECK(call) do { int32_t st_ = (call); if (st_ != ZKP_OK) { fprintf(stderr, "%s -> status %d: %s\n", #call, st_, zkp_multi_last_error_string(m)); return 1; } } while (0)

static void print_timing(zkp_multi* m, const int32_t* devs, const char* what) {
  for (uint32_t i = 0; i < zkp_multi_size(m); i++) {
    double ms = 0, compute_ms = 0, gather_ms = 0; uint64_t lo = 0, hi = 0;
    if (zkp_multi_last_timing(m, i, &ms, &lo, &hi) == ZKP_OK && zkp_multi_last_phases(m, i, &compute_ms, &gather_ms) == ZKP_OK)
      printf("  %-6s context %u (device %d): proofs [%llu, %llu)  %.1f ms of host time; on its stream: compute %.1f ms, gather %.2f ms\n", what, i, devs[i],
             (unsigned long long)lo, (unsigned long long)hi, ms, compute_ms, gather_ms);
  }
}

int main(int argc, char** argv) {
  int32_t devs[64]; uint32_t nd = 0;
  for (int i = 1; i < argc && nd < 64; i++) devs[nd++] = atoi(argv[i]);
  if (!nd && getenv("ZKP_DEVICES")) {                       /* "0,1,2,3" */
    char* list = strdup(getenv("ZKP_DEVICES"));
    for (char* t = strtok(list, ", "); t && nd < 64; t = strtok(NULL, ", ")) devs[nd++] = atoi(t);
    free(list);
  }
  if (!nd) devs[nd++] = 0;
  const uint64_t B = getenv("ZKP_BATCH") && atoll(getenv("ZKP_BATCH")) > 8 ? (uint64_t)atoll(getenv("ZKP_BATCH")) : 64;
  zkp_multi* m = NULL;
  if (zkp_multi_create(devs, nd, &m) != ZKP_OK) { fprintf(stderr, "zkp_multi_create failed: no gfx950 GPU (there is no CPU fallback)\n"); return 2; }
  const int rccl = getenv("ZKP_GATHER") && strcmp(getenv("ZKP_GATHER"), "rccl") == 0;
  if (rccl && zkp_multi_set_gather(m, ZKP_GATHER_RCCL) != ZKP_OK) { fprintf(stderr, "zkp_multi_set_gather(RCCL) failed: %s\n", zkp_multi_last_error_string(m)); return 3; }

  const size_t rows = B * EF;
  uint32_t* n = calloc(KW, 4);
  uint32_t *range = calloc(B * KW, 4), *ct = calloc(B * 2 * KW, 4), *c1 = calloc(rows * 2 * KW, 4), *c2 = calloc(rows * 2 * KW, 4);
  uint8_t *kind = calloc(rows, 1), *jj = calloc(rows, 1);
  uint32_t *rw1 = calloc(rows * KW, 4), *rr1 = calloc(rows * KW, 4), *rw2 = calloc(rows * KW, 4), *rr2 = calloc(rows * KW, 4);
  uint32_t *x = calloc(B * KW, 4), *r = calloc(B * KW, 4), *w1 = calloc(rows * KW, 4), *w2 = calloc(rows * KW, 4), *r1 = calloc(rows * KW, 4), *r2 = calloc(rows * KW, 4);
  hex_to_limbs(N_HEX, n, KW);
  for (uint64_t b = 0; b < B; b++) {
    range[b * KW + 7] = 0xC0000000u;                     /* 3 * 2^254 */
    rand_bits(x + b * KW, 200);
    rand_bits(r + b * KW, 2000);
    for (int i = 0; i < EF; i++) {
      uint32_t* a = w1 + (b * EF + i) * KW; uint32_t* c = w2 + (b * EF + i) * KW;
      rand_bits(c, 200);                                 /* u */
      memcpy(a, c, KW * 4); a[7] |= 0x40000000u;         /* T + u, T = 2^254 */
      if (rnd32() & 1) { uint32_t t[KW]; memcpy(t, a, sizeof t); memcpy(a, c, sizeof t); memcpy(c, t, sizeof t); }   /* the fair-coin swap, range_proof.rs:144-149 */
      rand_bits(r1 + (b * EF + i) * KW, 2000);
      rand_bits(r2 + (b * EF + i) * KW, 2000);
    }
  }
  /* ciphertext = Enc(x, r): any context of the set will do for an L1 call */
  if (zkp_paillier_enc_batch(zkp_multi_ctx(m, 0), N_BITS, B, n, 0, x, r, ct, 0) != ZKP_OK) { fprintf(stderr, "Enc failed\n"); return 1; }

  zkp_range_ni_proofs p = {N_BITS, EF, B, 0, n, range, ct, c1, c2, kind, jj, rw1, rr1, rw2, rr2};
  zkp_range_ni_witness w = {x, r, w1, w2, r1, r2};
  uint8_t* status = calloc(B, 1); uint8_t* verdict = calloc(B, 1);
  CHECK(zkp_multi_range_ni_prove_batch(m, &p, &w, NULL, NULL, status));          /* RangeProofNi::prove x B, sharded by proof index */
  print_timing(m, devs, "prove");
  rr1[(5 * EF + 0) * KW] ^= 1;                                                      /* tamper proof 5 */
  CHECK(zkp_multi_range_ni_verify_batch(m, &p, verdict));                          /* RangeProofNi::verify_self x B */
  print_timing(m, devs, "verify");
  if (rccl) {       /* the gathered verdicts are device-resident on every GPU: block i at i * stride */
    for (uint32_t i = 0; i < zkp_multi_size(m); i++) {
      void* dp = NULL; uint64_t stride = 0, bytes = 0;
      CHECK(zkp_multi_gathered(m, i, 0, &dp, &stride, &bytes));
      printf("  gather=rccl context %u (device %d): verdicts of all %llu proofs at %p, %llu bytes, block stride %llu\n", i, devs[i], (unsigned long long)B, dp,
             (unsigned long long)bytes, (unsigned long long)stride);
    }
  }
  unsigned accepted = 0, bad_status = 0;
  for (uint64_t b = 0; b < B; b++) { accepted += verdict[b] == ZKP_VERDICT_ACCEPT; bad_status += status[b] != 0; }
  printf("contexts=%u proofs=%llu accepted=%u rejected=%llu prove_status_errors=%u verdict[5]=%u\n", zkp_multi_size(m), (unsigned long long)B, accepted,
         (unsigned long long)(B - accepted), bad_status, verdict[5]);

  /* the same batch on ONE context: every byte the sharded run produced must come out again */
  uint32_t *c1s = calloc(rows * 2 * KW, 4), *c2s = calloc(rows * 2 * KW, 4), *sw1 = calloc(rows * KW, 4), *sr1 = calloc(rows * KW, 4), *sw2 = calloc(rows * KW, 4), *sr2 = calloc(rows * KW, 4);
  uint8_t *skind = calloc(rows, 1), *sjj = calloc(rows, 1), *sverdict = calloc(B, 1), *sstatus = calloc(B, 1);
  zkp_range_ni_proofs ps = {N_BITS, EF, B, 0, n, range, ct, c1s, c2s, skind, sjj, sw1, sr1, sw2, sr2};
  zkp_ctx* one = zkp_multi_ctx(m, 0);
  int mismatches = 0;
  if (zkp_range_ni_prove_batch(one, &ps, &w, NULL, NULL, sstatus, 0) != ZKP_OK) { fprintf(stderr, "single-context prove failed: %s\n", zkp_last_error_string(one)); return 1; }
  sr1[(5 * EF + 0) * KW] ^= 1;
  if (zkp_range_ni_verify_batch(one, &ps, sverdict, 0) != ZKP_OK) { fprintf(stderr, "single-context verify failed: %s\n", zkp_last_error_string(one)); return 1; }
  mismatches += memcmp(c1, c1s, rows * 2 * KW * 4) != 0; mismatches += memcmp(c2, c2s, rows * 2 * KW * 4) != 0;
  mismatches += memcmp(kind, skind, rows) != 0; mismatches += memcmp(jj, sjj, rows) != 0;
  mismatches += memcmp(rw1, sw1, rows * KW * 4) != 0; mismatches += memcmp(rr1, sr1, rows * KW * 4) != 0;
  mismatches += memcmp(rw2, sw2, rows * KW * 4) != 0; mismatches += memcmp(rr2, sr2, rows * KW * 4) != 0;
  mismatches += memcmp(verdict, sverdict, B) != 0;
  printf("single-context cross-check: %s\n", mismatches ? "MISMATCH" : "identical (c1, c2, responses, verdicts)");
  zkp_multi_destroy(m);
  return (accepted == B - 1 && verdict[5] == ZKP_VERDICT_REJECT && !bad_status && !mismatches) ? 0 : 1;
}
