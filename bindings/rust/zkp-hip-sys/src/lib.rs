//! zkp-hip-sys — raw FFI bindings of `libzkp_hip.so`, the MI355X (gfx950) batched Paillier ZK-proof engine.
//!
//! GENERATED from `include/zkp_hip.h` by `tools/gen_rust_sys.py`; do not edit.  Every function, struct and constant of the header is
//! here under its C name (the header's comments cite, per entry point, the line of ZenGo-X/zk-paillier it replaces).  The safe layer
//! that gives the crate's `zkproofs::{RangeProofNi, NiCorrectKeyProof, CompositeDLogProof}` their GPU paths is `zk-paillier-hip`
//! (`bindings/rust/zk-paillier-hip/hip.rs`).  `tests/test_rust_bindings.py` checks this file against the header, both ways.
#![allow(non_camel_case_types, non_snake_case, non_upper_case_globals, clippy::too_many_arguments)]

use std::os::raw::{c_char, c_void};

// ---------------------------------------------------------------- constants (enums and #defines of the header)
pub const ZKP_OK: i32 = 0;
pub const ZKP_EINVAL: i32 = 1;
pub const ZKP_ENONCANONICAL: i32 = 2;
pub const ZKP_EDEVICE: i32 = 3;
pub const ZKP_ENOMEM: i32 = 4;
pub const ZKP_F_DEVICE_PTRS: u32 = 1;
pub const ZKP_VERDICT_REJECT: u8 = 0;
pub const ZKP_VERDICT_ACCEPT: u8 = 1;
pub const ZKP_VERDICT_MALFORMED: u8 = 2;
pub const ZKP_RESP_OPEN: u8 = 0;
pub const ZKP_RESP_MASK: u8 = 1;
pub const ZKP_SECURITY_PARAMETER: usize = 128;
pub const ZKP_CORRECT_KEY_M2: usize = 11;
pub const ZKP_Z1_EXTRA_LIMBS: usize = 16;
pub const ZKP_INV_OK: u8 = 0;
pub const ZKP_INV_NONE: u8 = 1;
pub const ZKP_INV_DOMAIN: u8 = 2;
pub const ZKP_DEC_OK: u8 = 0;
pub const ZKP_DEC_INVALID: u8 = 1;
pub const ZKP_DEC_NEGATIVE: u8 = 2;
pub const ZKP_DEC_OVERFLOW: u8 = 3;
pub const ZKP_DOC_OK: u8 = 0;
pub const ZKP_DOC_INVALID: u8 = 2;
pub const ZKP_DOC_HOST_PATH: u8 = 3;
pub const ZKP_BIGINT_DEC: u32 = 0;
pub const ZKP_BIGINT_HEX: u32 = 1;
pub const ZKP_BIGINT_BYTES: u32 = 2;
pub const ZKP_GATHER_HOST: u32 = 0;
pub const ZKP_GATHER_RCCL: u32 = 1;
pub const ZKP_GATHER_COPY: u32 = 2;
/// `ZKP_BIGINT_FORMS(key_form, bare_form)`: the text forms of `ek.n` and of the bare BigInts of a RangeProofNi document
pub const fn ZKP_BIGINT_FORMS(key_form: u32, bare_form: u32) -> u32 {
    (key_form << 4) | bare_form
}

// ---------------------------------------------------------------- opaque handles
#[repr(C)]
pub struct zkp_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zkp_multi {
    _private: [u8; 0],
}

// ---------------------------------------------------------------- structs (field order and types = the C layout)
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct zkp_range_ni_proofs {
    pub n_bits: u32,
    pub error_factor: u32,
    pub batch: u64,
    pub n_stride: u64,
    pub n: *const u32,
    pub range: *const u32,
    pub ciphertext: *const u32,
    pub c1: *mut u32,
    pub c2: *mut u32,
    pub resp_kind: *mut u8,
    pub resp_j: *mut u8,
    pub resp_w1: *mut u32,
    pub resp_r1: *mut u32,
    pub resp_w2: *mut u32,
    pub resp_r2: *mut u32,
}
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct zkp_range_ni_witness {
    pub x: *const u32,
    pub r: *const u32,
    pub w1: *const u32,
    pub w2: *const u32,
    pub r1: *const u32,
    pub r2: *const u32,
}
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct zkp_dec_item {
    pub text_off: u64,
    pub dst_off: u64,
    pub len: u32,
    pub words: u32,
}

// ---------------------------------------------------------------- entry points
#[link(name = "zkp_hip")]
extern "C" {
    pub fn zkp_ctx_create(device_id: i32, out_ctx: *mut *mut zkp_ctx) -> i32;
    pub fn zkp_ctx_create_on_stream(device_id: i32, hip_stream: *mut c_void, out_ctx: *mut *mut zkp_ctx) -> i32;
    pub fn zkp_ctx_destroy(ctx: *mut zkp_ctx) -> i32;
    pub fn zkp_backend_name() -> *const c_char;
    pub fn zkp_build_limbs_per_lane() -> i32;
    pub fn zkp_ctx_set_geometry(ctx: *mut zkp_ctx, limbs_per_lane: i32) -> i32;
    pub fn zkp_ctx_last_geometry(ctx: *mut zkp_ctx) -> i32;
    pub fn zkp_ctx_latency_limbs_per_lane(ctx: *mut zkp_ctx) -> i32;
    pub fn zkp_last_error_string(ctx: *mut zkp_ctx) -> *const c_char;
    pub fn zkp_ctx_stream(ctx: *mut zkp_ctx) -> *mut c_void;
    pub fn zkp_ctx_synchronize(ctx: *mut zkp_ctx) -> i32;
    pub fn zkp_ctx_release_staging(ctx: *mut zkp_ctx) -> i32;
    pub fn zkp_timing_reset(ctx: *mut zkp_ctx, enable: i32) -> i32;
    pub fn zkp_timing_get(ctx: *mut zkp_ctx, out_ms: *mut f64, out_launches: *mut u64, out_modexps: *mut u64) -> i32;
    pub fn zkp_modexp_batch(ctx: *mut zkp_ctx, mod_bits: u32, exp_bits: u32, count: u64, base: *const u32, exp: *const u32, exp_stride: u64, mod_: *const u32, mod_stride: u64, out: *mut u32, flags: u32) -> i32;
    pub fn zkp_modmul_batch(ctx: *mut zkp_ctx, mod_bits: u32, count: u64, a: *const u32, b: *const u32, mod_: *const u32, mod_stride: u64, out: *mut u32, flags: u32) -> i32;
    pub fn zkp_paillier_enc_batch(ctx: *mut zkp_ctx, n_bits: u32, count: u64, n: *const u32, n_stride: u64, m: *const u32, r: *const u32, out_c: *mut u32, flags: u32) -> i32;
    pub fn zkp_paillier_enc_check_batch(ctx: *mut zkp_ctx, n_bits: u32, count: u64, n: *const u32, n_stride: u64, m: *const u32, r: *const u32, mulc_a: *const u32, mulc_b: *const u32, expected: *const u32, out_ok: *mut u8, flags: u32) -> i32;
    pub fn zkp_range_ni_prove_batch(ctx: *mut zkp_ctx, p: *const zkp_range_ni_proofs, w: *const zkp_range_ni_witness, out_e: *mut u8, out_e_len: *mut u8, out_status: *mut u8, flags: u32) -> i32;
    pub fn zkp_range_ni_verify_batch(ctx: *mut zkp_ctx, p: *const zkp_range_ni_proofs, out_verdict: *mut u8, flags: u32) -> i32;
    pub fn zkp_range_generate_encrypted_pairs_batch(ctx: *mut zkp_ctx, p: *const zkp_range_ni_proofs, w: *const zkp_range_ni_witness, flags: u32) -> i32;
    pub fn zkp_range_challenge_batch(ctx: *mut zkp_ctx, p: *const zkp_range_ni_proofs, out_e: *mut u8, out_e_len: *mut u8, flags: u32) -> i32;
    pub fn zkp_range_generate_proof_batch(ctx: *mut zkp_ctx, p: *const zkp_range_ni_proofs, w: *const zkp_range_ni_witness, e: *const u8, e_len: *const u8, out_status: *mut u8, flags: u32) -> i32;
    pub fn zkp_range_verifier_output_batch(ctx: *mut zkp_ctx, p: *const zkp_range_ni_proofs, e: *const u8, e_len: *const u8, out_verdict: *mut u8, flags: u32) -> i32;
    pub fn zkp_correct_key_ni_verify_batch(ctx: *mut zkp_ctx, n_bits: u32, batch: u64, n: *const u32, sigma: *const u32, salt: *const u8, salt_len: u32, out_verdict: *mut u8, flags: u32) -> i32;
    pub fn zkp_dlog_prove_batch(ctx: *mut zkp_ctx, n_bits: u32, y_bits: u32, batch: u64, N: *const u32, g: *const u32, ni: *const u32, secret: *const u32, r: *const u32, out_x: *mut u32, out_y: *mut u32, flags: u32) -> i32;
    pub fn zkp_dlog_verify_batch(ctx: *mut zkp_ctx, n_bits: u32, y_bits: u32, batch: u64, N: *const u32, g: *const u32, ni: *const u32, x: *const u32, y: *const u32, out_verdict: *mut u8, flags: u32) -> i32;
    pub fn zkp_zero_proof_prove_batch(ctx: *mut zkp_ctx, n_bits: u32, batch: u64, n: *const u32, n_stride: u64, c: *const u32, r: *const u32, r_prime: *const u32, out_z: *mut u32, out_a: *mut u32, flags: u32) -> i32;
    pub fn zkp_zero_proof_verify_batch(ctx: *mut zkp_ctx, n_bits: u32, batch: u64, n: *const u32, n_stride: u64, c: *const u32, z: *const u32, a: *const u32, out_verdict: *mut u8, flags: u32) -> i32;
    pub fn zkp_ciphertext_proof_prove_batch(ctx: *mut zkp_ctx, n_bits: u32, batch: u64, n: *const u32, n_stride: u64, c: *const u32, x: *const u32, r: *const u32, x_prime: *const u32, r_prime: *const u32, out_z1: *mut u32, out_z2: *mut u32, out_c_prime: *mut u32, flags: u32) -> i32;
    pub fn zkp_ciphertext_proof_verify_batch(ctx: *mut zkp_ctx, n_bits: u32, batch: u64, n: *const u32, n_stride: u64, c: *const u32, z1: *const u32, z2: *const u32, c_prime: *const u32, out_verdict: *mut u8, flags: u32) -> i32;
    pub fn zkp_verlin_proof_prove_batch(ctx: *mut zkp_ctx, n_bits: u32, batch: u64, n: *const u32, n_stride: u64, c: *const u32, c_prime: *const u32, phi_x: *const u32, x: *const u32, x_prime: *const u32, x_double_prime: *const u32, r_x: *const u32, a: *const u32, a_prime: *const u32, a_double_prime: *const u32, r_a: *const u32, out_phi_a: *mut u32, out_z: *mut u32, out_z_prime: *mut u32, out_z_double_prime: *mut u32, out_r_z: *mut u32, flags: u32) -> i32;
    pub fn zkp_verlin_proof_verify_batch(ctx: *mut zkp_ctx, n_bits: u32, batch: u64, n: *const u32, n_stride: u64, c: *const u32, c_prime: *const u32, phi_x: *const u32, phi_a: *const u32, z: *const u32, z_prime: *const u32, z_double_prime: *const u32, r_z: *const u32, out_verdict: *mut u8, flags: u32) -> i32;
    pub fn zkp_modinv_batch(ctx: *mut zkp_ctx, mod_bits: u32, count: u64, a: *const u32, modulus: *const u32, mod_stride: u64, out: *mut u32, out_status: *mut u8, flags: u32) -> i32;
    pub fn zkp_mul_proof_prove_batch(ctx: *mut zkp_ctx, n_bits: u32, batch: u64, n: *const u32, n_stride: u64, e_a: *const u32, e_b: *const u32, e_c: *const u32, a: *const u32, b: *const u32, r_a: *const u32, r_b: *const u32, r_c: *const u32, d: *const u32, r_d: *const u32, out_f: *mut u32, out_z1: *mut u32, out_z2: *mut u32, out_e_d: *mut u32, out_e_db: *mut u32, out_status: *mut u8, flags: u32) -> i32;
    pub fn zkp_mul_proof_verify_batch(ctx: *mut zkp_ctx, n_bits: u32, batch: u64, n: *const u32, n_stride: u64, e_a: *const u32, e_b: *const u32, e_c: *const u32, f: *const u32, z1: *const u32, z2: *const u32, e_d: *const u32, e_db: *const u32, out_verdict: *mut u8, flags: u32) -> i32;
    pub fn zkp_correct_message_prove_batch(ctx: *mut zkp_ctx, n_bits: u32, batch: u64, num_messages: u32, n: *const u32, n_stride: u64, valid_messages: *const u32, message: *const u32, r: *const u32, e_sim: *const u32, z_sim: *const u32, w: *const u32, out_ciphertext: *mut u32, out_e_vec: *mut u32, out_z_vec: *mut u32, out_a_vec: *mut u32, out_status: *mut u8, flags: u32) -> i32;
    pub fn zkp_correct_message_verify_batch(ctx: *mut zkp_ctx, n_bits: u32, batch: u64, num_messages: u32, n: *const u32, n_stride: u64, valid_messages: *const u32, ciphertext: *const u32, e_vec: *const u32, z_vec: *const u32, a_vec: *const u32, out_verdict: *mut u8, flags: u32) -> i32;
    pub fn zkp_decimal_to_limbs_batch(ctx: *mut zkp_ctx, text: *const c_char, text_len: u64, items: *const zkp_dec_item, count: u64, dst: *mut u32, dst_words: u64, out_status: *mut u8, flags: u32) -> i32;
    pub fn zkp_decimal_pitch(words: u32) -> u32;
    pub fn zkp_limbs_to_decimal_batch(ctx: *mut zkp_ctx, src: *const u32, src_stride: u64, words: u32, count: u64, out_text: *mut c_char, pitch: u32, out_len: *mut u32, flags: u32) -> i32;
    pub fn zkp_json_encrypted_pairs_batch(ctx: *mut zkp_ctx, text: *const c_char, doc_off: *const u64, doc_len: *const u64, p: *const zkp_range_ni_proofs, out_status: *mut u8, flags: u32) -> i32;
    pub fn zkp_json_range_proof_batch(ctx: *mut zkp_ctx, text: *const c_char, doc_off: *const u64, doc_len: *const u64, p: *const zkp_range_ni_proofs, out_status: *mut u8, flags: u32) -> i32;
    pub fn zkp_json_range_proof_ni_batch(ctx: *mut zkp_ctx, text: *const c_char, doc_off: *const u64, doc_len: *const u64, bigint_forms: u32, p: *const zkp_range_ni_proofs, out_status: *mut u8, flags: u32) -> i32;
    pub fn zkp_json_correct_key_proof_batch(ctx: *mut zkp_ctx, text: *const c_char, doc_off: *const u64, doc_len: *const u64, n_bits: u32, batch: u64, out_sigma: *mut u32, out_status: *mut u8, flags: u32) -> i32;
    pub fn zkp_multi_create(device_ids: *const i32, n_devices: u32, out: *mut *mut zkp_multi) -> i32;
    pub fn zkp_multi_destroy(m: *mut zkp_multi) -> i32;
    pub fn zkp_multi_size(m: *mut zkp_multi) -> u32;
    pub fn zkp_multi_ctx(m: *mut zkp_multi, i: u32) -> *mut zkp_ctx;
    pub fn zkp_multi_last_error_string(m: *mut zkp_multi) -> *const c_char;
    pub fn zkp_multi_last_timing(m: *mut zkp_multi, i: u32, out_ms: *mut f64, out_lo: *mut u64, out_hi: *mut u64) -> i32;
    pub fn zkp_multi_last_phases(m: *mut zkp_multi, i: u32, out_compute_ms: *mut f64, out_gather_ms: *mut f64) -> i32;
    pub fn zkp_multi_set_gather(m: *mut zkp_multi, mode: u32) -> i32;
    pub fn zkp_multi_gathered(m: *mut zkp_multi, device_index: u32, which: u32, out_device_ptr: *mut *mut c_void, out_block_stride_bytes: *mut u64, out_bytes: *mut u64) -> i32;
    pub fn zkp_multi_range_ni_prove_batch(m: *mut zkp_multi, p: *const zkp_range_ni_proofs, w: *const zkp_range_ni_witness, out_e: *mut u8, out_e_len: *mut u8, out_status: *mut u8) -> i32;
    pub fn zkp_multi_range_ni_verify_batch(m: *mut zkp_multi, p: *const zkp_range_ni_proofs, out_verdict: *mut u8) -> i32;
    pub fn zkp_multi_correct_key_ni_verify_batch(m: *mut zkp_multi, n_bits: u32, batch: u64, n: *const u32, sigma: *const u32, salt: *const u8, salt_len: u32, out_verdict: *mut u8) -> i32;
}
