/*
 * zkp_hip_diag.h — diagnostics of libzkp_hip.so that are NOT part of the drop-in boundary.
 *
 * include/zkp_hip.h is the boundary: every entry point there replaces a call shape of the reference
 * (ZenGo-X/zk-paillier) and is what a Rust / C caller binds.  What is declared here serves this repo's own
 * measurement tooling (profiles/collect_pmc.sh, bench.py --pmc-shape) and has no counterpart in the reference;
 * bindings/rust/zkp-hip-sys does not bind it.
 */
#ifndef ZKP_HIP_DIAG_H
#define ZKP_HIP_DIAG_H

#include "zkp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Moves a KNOWN number of bytes in the access pattern of the ladders' window tables — every lane of the resident
 * grid reads (mode 0) or writes (mode 1) its block of each entry of its table slot, `passes` times — so that the HBM-side PMC
 * counters behind the roofline's `traffic` figure can be calibrated (profiles/collect_pmc.sh).  out_bytes = bytes moved. */
int32_t zkp_diag_table_traffic(zkp_ctx* ctx, int32_t mode, int32_t passes, uint64_t* out_bytes);

/* One operation of the BASE-n form of Paillier's arithmetic modulo n^2 (csrc/kernels_basen.hpp) on raw 29-bit limbs, for the tests that pin
 * the kernels' building blocks to tests/basen_model.py.  n_bits = 2048 | 4096, L = 72 | 144 limbs per operand half.
 *   op 0: (xa, 0) -> Montgomery form        op 1: (xa, xb) * (ya, yb) / R'        op 2: (xa, xb)^2 / R'      -> out[0, L) = a, out[L, 2L) = b
 *   op 3: the key's constants               -> out = C3 | RRa | RRb | M~ (L limbs each) | n1 | ok */
int32_t zkp_diag_basen(zkp_ctx* ctx, uint32_t n_bits, const uint32_t* n, int32_t op, const uint32_t* xa, const uint32_t* xb, const uint32_t* ya,
                       const uint32_t* yb, uint32_t* out);

/* Did the most recent shared-key Paillier launch of this ctx run in base-n form?  out_lanes: lanes per n-sized integer of that launch
 * (0: there was none), out_qualified: 1 when the key passed the form's set-up (else the n^2-sized kernel did the work). */
int32_t zkp_diag_basen_last(zkp_ctx* ctx, int32_t* out_lanes, uint32_t* out_qualified);

/* How this library's base-n kernels of the throughput engine were built: 1 = their squarings and products run through the fixed-register
 * assembler engine (csrc/kernels_basen_asm_g*.inc: ONE product body, the cross product of a base-n product without a reduction — 2.5
 * instead of 3 n-sized products per product modulo n^2), 0 = the compiled bodies (-DZKP_BN_ASM=0, A/B builds; the other geometries).
 * bench.py prices the EXECUTED multiply-adds of a launch by it. */
int32_t zkp_diag_basen_engine(void);

/* Which Paillier launches of this ctx run in base-n form (csrc/kernels_basen.hpp) and which on the n^2-sized kernels.  The product's
 * rule is AUTO; the other values exist for A/B measurements and for the parity tests, which pin BOTH forms against the oracle at sizes
 * where AUTO would only ever pick one.  $ZKP_BASEN (0 | shared | always) presets the value when a ctx is created — the environment is
 * read there and nowhere else.  Results are bit-identical under every value. */
#define ZKP_ENC_FORM_AUTO   0   /* launches that fill the chip take the form (under one key or under per-proof keys), smaller ones do not */
#define ZKP_ENC_FORM_N2     1   /* every launch on the n^2-sized kernels */
#define ZKP_ENC_FORM_SHARED 2   /* as AUTO, but launches under per-proof keys stay on the n^2-sized kernels */
#define ZKP_ENC_FORM_ALWAYS 3   /* every launch the form can take, however small */
int32_t zkp_diag_set_enc_form(zkp_ctx* ctx, int32_t form);
int32_t zkp_diag_enc_form(zkp_ctx* ctx);   /* the current value; -1 for a null ctx */

/* Into how many blocks of proof indices the most recent zkp_range_ni_{prove,verify}_batch call on HOST arrays was cut (1: the plain path —
 * every input copied in, the kernels, every output copied out; more: copies of block k + 1 / k - 1 under the kernels of block k).
 * $ZKP_HOST_CHUNKS at ctx create: unset or 1 = never cut (the default: on the boxes measured the copies are 2 % of a call and every extra launch
 * has a tail of its own), N = N equal blocks, 0 = a quarter | the rest (verify), a quarter | half | a quarter (prove) for calls of 2048 proofs and more. */
int32_t zkp_diag_last_host_blocks(zkp_ctx* ctx);

/* The latency engine's kernel for calls of a few proofs under one 2048-bit key: ONE Enc per wavefront, the base-n exponentiation as a
 * right-to-left ladder pipelined over five lane groups (csrc/kernels_basen_r2l.hpp).  mode 0 = never (the pair ladder on the n^2-sized
 * product serves those calls, as before round 5), 1 = the library's rule (launches of up to two wavefronts per SIMD; the default),
 * 2 = every launch the kernel can take (tests).  $ZKP_R2L presets it at ctx create.  zkp_diag_r2l_last: 1 when the most recent Paillier
 * launch of the ctx ran on it. */
int32_t zkp_diag_set_r2l(zkp_ctx* ctx, int32_t mode);
/* The lane geometry of that ladder.  0 = the library's rule (the default): FIVE wavefronts per Enc — one per role, 36 lanes x 2 limbs per
 * n-sized integer, the quotient digits wave-uniform in scalar registers (k_enc_basen_r2l5) — while the launch leaves every Enc a
 * compute unit of its own (one proof), one wavefront of five groups of 12 lanes x 6 limbs beyond; 36 / 12 pin one of the two,
 * 8 the 8-lane x 9-limb variant of the one-wavefront kernel (A/B runs).  $ZKP_R2L_LANES presets it at ctx create.
 * zkp_diag_r2l_lanes_last: the geometry of the most recent launch of the ladder (0: the most recent Paillier launch was not one). */
int32_t zkp_diag_set_r2l_lanes(zkp_ctx* ctx, int32_t lanes);
int32_t zkp_diag_r2l_lanes_last(zkp_ctx* ctx);
/* limbs per lane of the mid engine (libzkp_hip_mid.so next to the library, or $ZKP_HIP_MID_LIB: the same sources at 18 limbs per lane,
 * 16 Enc per wavefront in base-n form — Paillier calls of 41 ... 64 and 129 ... 192 proofs under one 2048-bit key); 0 when it is not loaded
 * (the other engines then take those calls, a little slower, never differently).  zkp_ctx_set_geometry(ctx, 18) pins it. */
int32_t zkp_diag_mid_limbs_per_lane(zkp_ctx* ctx);
int32_t zkp_diag_r2l_last(zkp_ctx* ctx);

/* The constants of the ONE key of a shared-key call are kept across calls: the set-up kernels compare the modulus they are handed with
 * the one their record was computed from (a tag beside each constants buffer, on the device) and return at once when it is the same —
 * 1 ms of a one-proof prove + verify.  The host side clears a tag whenever its buffer was reallocated or written by a launch of several
 * keys; a rejected modulus never leaves a valid tag.  On by default; $ZKP_KEY_CACHE=0 at ctx create or zkp_diag_set_key_cache(ctx, 0)
 * turn it off (every call computes, as before round 5).  zkp_diag_key_cache_state: out[3] = {tag valid, the last set-up launch into
 * the buffer returned early, set-ups that computed so far} for buffer `which` (0 = n^2 / modexp constants, 1 = the second set (mod n),
 * 2 = the n-sized record behind the base-n form, 3 = the base-n record) of the engine the most recent routed call ran on. */
/* A RangeProofNi prove / verify call of 65 ... 96 proofs under one 2048-bit key (items between one and one and a half wavefronts per SIMD of
 * the mid engine) runs as TWO concurrent calls: the first 64 proofs on the mid engine on the ctx's stream, the rest on a second ctx of the
 * latency engine with a stream of its own (58 -> 41 ... 53 ms; csrc/zkp_api_proofs.inc: range_split_run).  Needs both secondary engines and
 * the automatic geometry / Enc form.  On by default; $ZKP_SPLIT=0 at ctx create or zkp_diag_set_split(ctx, 0) turn it off.
 * zkp_diag_last_split: the proofs of the most recent RangeProofNi call that went to the latency engine beside the mid engine (0: not split). */
int32_t zkp_diag_set_split(zkp_ctx* ctx, int32_t on);
int32_t zkp_diag_last_split(zkp_ctx* ctx);
/* A verify call of 1 ... 8 proofs at n = 2048 under one key (the latency engine's one-Enc-per-wavefront kernels, k_enc_basen_r2l5 /
 * k_enc_basen_r2l) carries its transcript hashes as the FIRST WORKGROUPS of its Enc launch instead of a launch of their own on a second
 * stream, where a hash wavefront shared a SIMD with the wavefronts of an Enc one call in four to seven (one proof: 6.2 or 8.0 ms; four:
 * 10.0 or 12.5; eight: 14.6 or 17.2).  On by default; $ZKP_FUSE_HASH=0 at ctx create or zkp_diag_set_fuse_hash(ctx, 0) turn it off.
 * zkp_diag_last_fused_hash: 1 when the most recent verify call ran that way. */
int32_t zkp_diag_set_fuse_hash(zkp_ctx* ctx, int32_t on);
int32_t zkp_diag_last_fused_hash(zkp_ctx* ctx);
int32_t zkp_diag_set_key_cache(zkp_ctx* ctx, int32_t on);
int32_t zkp_diag_key_cache_state(zkp_ctx* ctx, int32_t which, uint32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* ZKP_HIP_DIAG_H */
