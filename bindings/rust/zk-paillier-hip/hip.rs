//! `zkproofs::hip` — the GPU paths of the hot proofs of zk-paillier, behind cargo feature `hip`.
//!
//! Drop this file in as `src/zkproofs/hip.rs` of ZenGo-X/zk-paillier 0.4.4 and apply `bindings/rust/zk-paillier-hip.patch`
//! (Cargo feature + dependency, `mod hip`, field visibilities, and the five three-line dispatch blocks).  The public API of the
//! crate does not change: `RangeProofNi::{prove, verify, verify_self}`, `NiCorrectKeyProof::verify` and
//! `CompositeDLogProof::{prove, verify}` keep their signatures, results and panics; with the feature on, their inner loops — the
//! 256 / ~192 / 11 / 1-2 modular exponentiations per proof — run in `libzkp_hip.so` (hand-written HIP kernels for MI355X) through
//! the C ABI of `include/zkp_hip.h`, bound by the `zkp-hip-sys` crate.  New here are only the BATCH entry points
//! (`RangeProofNi::prove_batch / verify_batch`, `NiCorrectKeyProof::verify_batch`), which is where a GPU pays off.
//!
//! What goes to the GPU and what stays on GMP
//!   * A proof is CANONICAL when every field is a non-negative integer within its fixed width (n-sized values in n_bits bits,
//!     ciphertexts in 2 n_bits) and it stores exactly `error_factor` rows.  Every honest proof is.  Canonical proofs are flattened
//!     into the structure-of-arrays batch of `zkp_range_ni_proofs` and verified by one `zkp_range_ni_verify_batch`.
//!   * A deserialised proof may hold ANY BigInt (negative, over-wide) and any row count; the reference has a verdict — or a panic —
//!     for each of those.  Such a proof is answered by the crate's own unchanged GMP code (`RangeProof::verifier_output`): the
//!     functions below return `None` and the patched method falls through to the original body.  Same for keys the kernels do not
//!     carry (even, or wider than 4096 bits) and when no gfx950 GPU is present.  So the result is the reference's in every case.
//!   * Verdict byte `ZKP_VERDICT_MALFORMED` = "the reference would panic here" (index out of bounds): the single-proof methods
//!     re-run that proof on the GMP path so that the panic is the reference's own; the batch methods report `Verdict::WouldPanic`.
//!
//! This file is source that has NOT been compiled in the repository it ships from (no rustc there).  `tests/test_rust_bindings.py`
//! checks every `sys::` call below against `include/zkp_hip.h` (name, argument count) and the struct literals against the header.
#![cfg(feature = "hip")]

use std::ptr;
use std::sync::{Mutex, Once};

use curv::arithmetic::traits::*;
use curv::BigInt;
use paillier::EncryptionKey;
use rand::random;
use zkp_hip_sys as sys;

use super::correct_key_ni::NiCorrectKeyProof;
use super::errors::IncorrectProof;
use super::range_proof::{EncryptedPairs, Proof, Response};
use super::range_proof_ni::RangeProofNi;
use super::wi_dlog_proof::{CompositeDLogProof, DLogStatement};

const SECURITY_PARAMETER: usize = sys::ZKP_SECURITY_PARAMETER; // src/zkproofs/range_proof_ni.rs:23
const M2: usize = sys::ZKP_CORRECT_KEY_M2; // src/zkproofs/correct_key_ni.rs:29
const DLOG_Y_BITS: u32 = 768; // an honest y = r + e * s is < 2^513 (wi_dlog_proof.rs:53-62); wider responses take the GMP path

/// Outcome of one proof of a batch call.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum Verdict {
    /// `Ok(())`
    Accept,
    /// `Err(IncorrectProof)`
    Reject,
    /// the reference panics on this proof (index out of bounds / assert): call the single-proof method to get that panic
    WouldPanic,
}

impl Verdict {
    pub fn into_result(self) -> Result<(), IncorrectProof> {
        match self {
            Verdict::Accept => Ok(()),
            Verdict::Reject => Err(IncorrectProof),
            Verdict::WouldPanic => panic!("malformed proof: the reference implementation panics on it"),
        }
    }
}

// ------------------------------------------------------------------------------------------------ context
struct Ctx(*mut sys::zkp_ctx);
// one ctx = one GPU + one stream; calls on it are serialised by the Mutex below (include/zkp_hip.h, "Conventions")
unsafe impl Send for Ctx {}

static INIT: Once = Once::new();
static mut CTX: Option<Mutex<Ctx>> = None;

/// The process-wide engine context on GPU `ZKP_HIP_DEVICE` (default 0); `None` when there is no gfx950 device — the crate then
/// keeps computing on GMP (libzkp_hip.so itself has no CPU fallback: zkp_ctx_create fails with ZKP_EDEVICE).
fn with_ctx<T>(f: impl FnOnce(*mut sys::zkp_ctx) -> Option<T>) -> Option<T> {
    INIT.call_once(|| {
        let device = std::env::var("ZKP_HIP_DEVICE").ok().and_then(|s| s.parse::<i32>().ok()).unwrap_or(0);
        let mut raw: *mut sys::zkp_ctx = ptr::null_mut();
        let st = unsafe { sys::zkp_ctx_create(device, &mut raw) };
        if st == sys::ZKP_OK && !raw.is_null() {
            unsafe { CTX = Some(Mutex::new(Ctx(raw))) };
        }
    });
    let cell = unsafe { CTX.as_ref() }?;
    let guard = cell.lock().ok()?;
    f(guard.0)
}

fn ok(status: i32) -> Option<()> {
    if status == sys::ZKP_OK {
        Some(())
    } else {
        None // ZKP_ENONCANONICAL (even modulus), ZKP_EDEVICE, ...: the caller falls back to GMP
    }
}

// ------------------------------------------------------------------------------------------------ BigInt <-> limbs
/// kernel width for an n of this size (include/zkp_hip.h: n_bits in {1024, 2048, 4096}); `None` for keys the kernels do not carry
fn width_for(n: &BigInt) -> Option<u32> {
    if BigInt::is_negative(n) || n.is_even() {
        return None;
    }
    match n.bit_length() {
        0..=1 => None,
        2..=1024 => Some(1024),
        1025..=2048 => Some(2048),
        2049..=4096 => Some(4096),
        _ => None,
    }
}

/// fixed-width little-endian 32-bit limbs (the ABI's integer form); false if `x` is negative or does not fit: not canonical
fn put_limbs(dst: &mut [u32], x: &BigInt) -> bool {
    if BigInt::is_negative(x) {
        return false;
    }
    let be = BigInt::to_bytes(x); // minimal big-endian magnitude, zero -> [0]
    if be.len() > 4 * dst.len() {
        return false;
    }
    for w in dst.iter_mut() {
        *w = 0;
    }
    for (i, b) in be.iter().rev().enumerate() {
        dst[i / 4] |= (*b as u32) << (8 * (i % 4));
    }
    true
}

fn get_limbs(src: &[u32]) -> BigInt {
    let mut be = Vec::with_capacity(4 * src.len());
    for w in src.iter().rev() {
        be.extend_from_slice(&w.to_be_bytes());
    }
    BigInt::from_bytes(&be)
}

// ------------------------------------------------------------------------------------------------ RangeProofNi
/// The structure-of-arrays batch of include/zkp_hip.h (`zkp_range_ni_proofs`), owned.
struct RangeBatch {
    n_bits: u32,
    ef: usize,
    b: usize,
    n: Vec<u32>,
    range: Vec<u32>,
    ciphertext: Vec<u32>,
    c1: Vec<u32>,
    c2: Vec<u32>,
    kind: Vec<u8>,
    j: Vec<u8>,
    w1: Vec<u32>,
    r1: Vec<u32>,
    w2: Vec<u32>,
    r2: Vec<u32>,
}

impl RangeBatch {
    fn new(n_bits: u32, b: usize, ef: usize) -> RangeBatch {
        let kw = (n_bits / 32) as usize;
        RangeBatch {
            n_bits,
            ef,
            b,
            n: vec![0; kw],
            range: vec![0; b * kw],
            ciphertext: vec![0; b * 2 * kw],
            c1: vec![0; b * ef * 2 * kw],
            c2: vec![0; b * ef * 2 * kw],
            kind: vec![0; b * ef],
            j: vec![0; b * ef],
            w1: vec![0; b * ef * kw],
            r1: vec![0; b * ef * kw],
            w2: vec![0; b * ef * kw],
            r2: vec![0; b * ef * kw],
        }
    }

    /// one shared key for the whole batch: n_stride = 0
    fn raw(&mut self) -> sys::zkp_range_ni_proofs {
        sys::zkp_range_ni_proofs {
            n_bits: self.n_bits,
            error_factor: self.ef as u32,
            batch: self.b as u64,
            n_stride: 0,
            n: self.n.as_ptr(),
            range: self.range.as_ptr(),
            ciphertext: self.ciphertext.as_ptr(),
            c1: self.c1.as_mut_ptr(),
            c2: self.c2.as_mut_ptr(),
            resp_kind: self.kind.as_mut_ptr(),
            resp_j: self.j.as_mut_ptr(),
            resp_w1: self.w1.as_mut_ptr(),
            resp_r1: self.r1.as_mut_ptr(),
            resp_w2: self.w2.as_mut_ptr(),
            resp_r2: self.r2.as_mut_ptr(),
        }
    }

    /// proof `b` of the batch as the crate's types (after a prove call)
    fn proof(&self, b: usize) -> (EncryptedPairs, Proof) {
        let kw = (self.n_bits / 32) as usize;
        let mut pairs = EncryptedPairs { c1: Vec::with_capacity(self.ef), c2: Vec::with_capacity(self.ef) };
        let mut responses = Vec::with_capacity(self.ef);
        for i in 0..self.ef {
            let t = b * self.ef + i;
            pairs.c1.push(get_limbs(&self.c1[t * 2 * kw..(t + 1) * 2 * kw]));
            pairs.c2.push(get_limbs(&self.c2[t * 2 * kw..(t + 1) * 2 * kw]));
            let lo = t * kw;
            let hi = lo + kw;
            if self.kind[t] == sys::ZKP_RESP_OPEN {
                responses.push(Response::Open {
                    w1: get_limbs(&self.w1[lo..hi]),
                    r1: get_limbs(&self.r1[lo..hi]),
                    w2: get_limbs(&self.w2[lo..hi]),
                    r2: get_limbs(&self.r2[lo..hi]),
                });
            } else {
                responses.push(Response::Mask {
                    j: self.j[t],
                    masked_x: get_limbs(&self.w1[lo..hi]),
                    masked_r: get_limbs(&self.r1[lo..hi]),
                });
            }
        }
        (pairs, Proof(responses))
    }

    /// flattens `p` into slot `b`; false when the proof is not canonical (then nothing of the batch may be used for it)
    fn fill(&mut self, b: usize, p: &RangeProofNi) -> bool {
        let kw = (self.n_bits / 32) as usize;
        let ef = self.ef;
        if p.error_factor != ef || p.proof.0.len() != ef || p.encrypted_pairs.c1.len() != ef || p.encrypted_pairs.c2.len() != ef {
            return false;
        }
        if get_limbs(&self.n) != p.ek.n {
            return false; // RangeProofNi::verify asserts the verifier's key is the proof's (range_proof_ni.rs:86): GMP path, which panics there
        }
        if !put_limbs(&mut self.range[b * kw..(b + 1) * kw], &p.range) || !put_limbs(&mut self.ciphertext[b * 2 * kw..(b + 1) * 2 * kw], &p.ciphertext) {
            return false;
        }
        for i in 0..ef {
            let t = b * ef + i;
            if !put_limbs(&mut self.c1[t * 2 * kw..(t + 1) * 2 * kw], &p.encrypted_pairs.c1[i]) || !put_limbs(&mut self.c2[t * 2 * kw..(t + 1) * 2 * kw], &p.encrypted_pairs.c2[i]) {
                return false;
            }
            let lo = t * kw;
            let hi = lo + kw;
            let fits = match &p.proof.0[i] {
                Response::Open { w1, r1, w2, r2 } => {
                    self.kind[t] = sys::ZKP_RESP_OPEN;
                    self.j[t] = 0;
                    put_limbs(&mut self.w1[lo..hi], w1) && put_limbs(&mut self.r1[lo..hi], r1) && put_limbs(&mut self.w2[lo..hi], w2) && put_limbs(&mut self.r2[lo..hi], r2)
                }
                Response::Mask { j, masked_x, masked_r } => {
                    self.kind[t] = sys::ZKP_RESP_MASK;
                    self.j[t] = *j;
                    put_limbs(&mut self.w1[lo..hi], masked_x) && put_limbs(&mut self.r1[lo..hi], masked_r)
                }
            };
            if !fits {
                return false;
            }
        }
        true
    }
}

/// One prover's statement and witness for `RangeProofNi::prove_batch`.
pub struct RangeStatement<'a> {
    pub range: &'a BigInt,
    pub ciphertext: &'a BigInt,
    pub secret_x: &'a BigInt,
    pub secret_r: &'a BigInt,
}

/// `RangeProofNi::prove` (src/zkproofs/range_proof_ni.rs:47-82) for many provers under one key: the randomness is drawn exactly as
/// `RangeProof::generate_encrypted_pairs` draws it (range_proof.rs:133-159), then ONE `zkp_range_ni_prove_batch` computes the
/// 256 encryptions per proof (range_proof.rs:161-187), the Fiat-Shamir challenge (range_proof_ni.rs:58-61, utils.rs:9-22) and the
/// responses (range_proof.rs:210-252).  `None`: not representable / no GPU — prove on GMP instead.
pub fn range_ni_prove_batch(ek: &EncryptionKey, statements: &[RangeStatement]) -> Option<Vec<RangeProofNi>> {
    let n_bits = width_for(&ek.n)?;
    let kw = (n_bits / 32) as usize;
    let b = statements.len();
    let ef = SECURITY_PARAMETER;
    if b == 0 {
        return Some(Vec::new());
    }
    let mut batch = RangeBatch::new(n_bits, b, ef);
    if !put_limbs(&mut batch.n, &ek.n) {
        return None;
    }
    let (mut x, mut r) = (vec![0u32; b * kw], vec![0u32; b * kw]);
    let (mut w1, mut w2, mut r1, mut r2) = (vec![0u32; b * ef * kw], vec![0u32; b * ef * kw], vec![0u32; b * ef * kw], vec![0u32; b * ef * kw]);
    for (k, st) in statements.iter().enumerate() {
        if !put_limbs(&mut batch.range[k * kw..(k + 1) * kw], st.range)
            || !put_limbs(&mut batch.ciphertext[k * 2 * kw..(k + 1) * 2 * kw], st.ciphertext)
            || !put_limbs(&mut x[k * kw..(k + 1) * kw], st.secret_x)
            || !put_limbs(&mut r[k * kw..(k + 1) * kw], st.secret_r)
        {
            return None;
        }
        let range_scaled_third = st.range.div_floor(&BigInt::from(3)); // range_proof.rs:133
        let range_scaled_two_thirds = BigInt::from(2) * &range_scaled_third; // :134
        for i in 0..ef {
            let t = k * ef + i;
            let mut a = BigInt::sample_range(&range_scaled_third, &range_scaled_two_thirds); // :136-139
            let mut c = &a - &range_scaled_third; // :141
            if random() {
                std::mem::swap(&mut a, &mut c); // :144-149
            }
            if !put_limbs(&mut w1[t * kw..(t + 1) * kw], &a)
                || !put_limbs(&mut w2[t * kw..(t + 1) * kw], &c)
                || !put_limbs(&mut r1[t * kw..(t + 1) * kw], &BigInt::sample_below(&ek.n)) // :151-154
                || !put_limbs(&mut r2[t * kw..(t + 1) * kw], &BigInt::sample_below(&ek.n)) // :156-159
            {
                return None;
            }
        }
    }
    let witness = sys::zkp_range_ni_witness { x: x.as_ptr(), r: r.as_ptr(), w1: w1.as_ptr(), w2: w2.as_ptr(), r1: r1.as_ptr(), r2: r2.as_ptr() };
    let mut status = vec![0u8; b];
    let raw = batch.raw();
    with_ctx(|ctx| ok(unsafe { sys::zkp_range_ni_prove_batch(ctx, &raw, &witness, ptr::null_mut(), ptr::null_mut(), status.as_mut_ptr(), 0) }))?;
    if status.iter().any(|s| *s != 0) {
        return None; // a row the fixed width cannot carry (x + w wider than the key): GMP decides
    }
    let mut out = Vec::with_capacity(b);
    for (k, st) in statements.iter().enumerate() {
        let (encrypted_pairs, proof) = batch.proof(k);
        out.push(RangeProofNi {
            ek: ek.clone(),
            range: st.range.clone(),
            ciphertext: st.ciphertext.clone(),
            encrypted_pairs,
            proof,
            error_factor: SECURITY_PARAMETER,
        });
    }
    Some(out)
}

/// `RangeProofNi::verify_self` (range_proof_ni.rs:109-128 -> range_proof.rs:254-355) for many proofs under one key.
/// Entry k is `None` when proof k has to be decided by the GMP path (not canonical, see the module docs); the whole result is
/// `None` when the key is not one the kernels carry or there is no GPU.
pub fn range_ni_verify_batch(ek: &EncryptionKey, proofs: &[&RangeProofNi]) -> Option<Vec<Option<Verdict>>> {
    let n_bits = width_for(&ek.n)?;
    if proofs.is_empty() {
        return Some(Vec::new());
    }
    let ef = proofs[0].error_factor;
    if ef == 0 || ef > 256 {
        return None;
    }
    // canonical proofs are packed densely; `slot[k]` is the position of proof k in the launch
    let mut slot: Vec<Option<usize>> = Vec::with_capacity(proofs.len());
    let mut batch = RangeBatch::new(n_bits, proofs.len(), ef);
    if !put_limbs(&mut batch.n, &ek.n) {
        return None;
    }
    let mut used = 0usize;
    for p in proofs.iter() {
        if batch.fill(used, p) {
            slot.push(Some(used));
            used += 1;
        } else {
            slot.push(None);
        }
    }
    let mut verdict = vec![9u8; used];
    if used > 0 {
        batch.b = used;
        let raw = batch.raw();
        with_ctx(|ctx| ok(unsafe { sys::zkp_range_ni_verify_batch(ctx, &raw, verdict.as_mut_ptr(), 0) }))?;
    }
    Some(
        slot.iter()
            .map(|s| {
                s.map(|i| match verdict[i] {
                    sys::ZKP_VERDICT_ACCEPT => Verdict::Accept,
                    sys::ZKP_VERDICT_REJECT => Verdict::Reject,
                    _ => Verdict::WouldPanic,
                })
            })
            .collect(),
    )
}

impl RangeProofNi {
    /// Many provers, one key, one GPU launch sequence.  Falls back to `RangeProofNi::prove` per statement (whose own dispatch ends
    /// in the original GMP body) when the GPU cannot take the batch.
    pub fn prove_batch(ek: &EncryptionKey, statements: &[RangeStatement]) -> Vec<RangeProofNi> {
        match range_ni_prove_batch(ek, statements) {
            Some(v) => v,
            None => statements.iter().map(|s| RangeProofNi::prove(ek, s.range, s.ciphertext, s.secret_x, s.secret_r)).collect(),
        }
    }

    /// `proof.verify(ek, &proof.ciphertext)` for many proofs under the verifier's key `ek`; proofs the GPU path does not take are
    /// verified by the crate's GMP code (a proof under another key panics there, as range_proof_ni.rs:86 does), so every entry is the
    /// reference's outcome.
    pub fn verify_batch(ek: &EncryptionKey, proofs: &[&RangeProofNi]) -> Vec<Verdict> {
        let gpu = range_ni_verify_batch(ek, proofs);
        proofs
            .iter()
            .enumerate()
            .map(|(k, p)| match gpu.as_ref().and_then(|v| v[k]) {
                Some(v) => v,
                None => match std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| p.verify(ek, &p.ciphertext))) {
                    Ok(Ok(())) => Verdict::Accept,
                    Ok(Err(IncorrectProof)) => Verdict::Reject,
                    Err(_) => Verdict::WouldPanic,
                },
            })
            .collect()
    }
}

/// Single-proof dispatch used by the patched `RangeProofNi::{verify, verify_self}`: `Some(result)` when the GPU decided,
/// `None` -> run the original body (non-canonical proof, unsupported key, no GPU, or a proof the reference panics on: the GMP
/// path then produces that very panic).
pub fn range_ni_verify_one(ek: &EncryptionKey, proof: &RangeProofNi) -> Option<Result<(), IncorrectProof>> {
    match range_ni_verify_batch(ek, &[proof])?.pop()?? {
        Verdict::Accept => Some(Ok(())),
        Verdict::Reject => Some(Err(IncorrectProof)),
        Verdict::WouldPanic => None,
    }
}

/// Single-proof dispatch used by the patched `RangeProofNi::prove`.
pub fn range_ni_prove_one(ek: &EncryptionKey, range: &BigInt, ciphertext: &BigInt, secret_x: &BigInt, secret_r: &BigInt) -> Option<RangeProofNi> {
    range_ni_prove_batch(ek, &[RangeStatement { range, ciphertext, secret_x, secret_r }])?.pop()
}

// ------------------------------------------------------------------------------------------------ NiCorrectKeyProof
/// `NiCorrectKeyProof::verify` (src/zkproofs/correct_key_ni.rs:73-100) for many (key, proof) pairs: the SHA-256 mask generation
/// (:77-86, 105-117), the 11 sigma_i^n mod n (:90-93) and gcd(primorial, n) (:87-88) all run in `zkp_correct_key_ni_verify_batch`.
/// Keys must share one width; `None` entries / result as for the range proof.
pub fn correct_key_ni_verify_batch(items: &[(&EncryptionKey, &NiCorrectKeyProof)], salt: &[u8]) -> Option<Vec<Option<Verdict>>> {
    if items.is_empty() {
        return Some(Vec::new());
    }
    let n_bits = width_for(&items[0].0.n)?;
    let kw = (n_bits / 32) as usize;
    let mut slot: Vec<Option<usize>> = Vec::with_capacity(items.len());
    let (mut n, mut sigma) = (Vec::<u32>::new(), Vec::<u32>::new());
    let mut used = 0usize;
    for (ek, proof) in items.iter() {
        // sigma_vec[i] for i < 11: a shorter vector is an index panic in the reference (:92), a longer one is read up to 11
        let mut row_n = vec![0u32; kw];
        let mut row_s = vec![0u32; M2 * kw];
        let fits = width_for(&ek.n) == Some(n_bits)
            && proof.sigma_vec.len() >= M2
            && put_limbs(&mut row_n, &ek.n)
            && (0..M2).all(|i| put_limbs(&mut row_s[i * kw..(i + 1) * kw], &proof.sigma_vec[i]));
        if fits {
            n.extend_from_slice(&row_n);
            sigma.extend_from_slice(&row_s);
            slot.push(Some(used));
            used += 1;
        } else {
            slot.push(None);
        }
    }
    let mut verdict = vec![9u8; used];
    if used > 0 {
        with_ctx(|ctx| {
            ok(unsafe {
                sys::zkp_correct_key_ni_verify_batch(ctx, n_bits, used as u64, n.as_ptr(), sigma.as_ptr(), salt.as_ptr(), salt.len() as u32, verdict.as_mut_ptr(), 0)
            })
        })?;
    }
    Some(
        slot.iter()
            .map(|s| s.map(|i| if verdict[i] == sys::ZKP_VERDICT_ACCEPT { Verdict::Accept } else if verdict[i] == sys::ZKP_VERDICT_REJECT { Verdict::Reject } else { Verdict::WouldPanic }))
            .collect(),
    )
}

/// Single-proof dispatch used by the patched `NiCorrectKeyProof::verify`.
pub fn correct_key_ni_verify_one(proof: &NiCorrectKeyProof, ek: &EncryptionKey, salt: &[u8]) -> Option<Result<(), IncorrectProof>> {
    match correct_key_ni_verify_batch(&[(ek, proof)], salt)?.pop()?? {
        Verdict::Accept => Some(Ok(())),
        Verdict::Reject => Some(Err(IncorrectProof)),
        Verdict::WouldPanic => None,
    }
}

impl NiCorrectKeyProof {
    /// Many keys at once (e.g. the key-generation phase of a threshold-signing ceremony with many parties).
    pub fn verify_batch(items: &[(&EncryptionKey, &NiCorrectKeyProof)], salt: &[u8]) -> Vec<Verdict> {
        let gpu = correct_key_ni_verify_batch(items, salt);
        items
            .iter()
            .enumerate()
            .map(|(k, (ek, proof))| match gpu.as_ref().and_then(|v| v[k]) {
                Some(v) => v,
                None => match std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| proof.verify(ek, salt))) {
                    Ok(Ok(())) => Verdict::Accept,
                    Ok(Err(IncorrectProof)) => Verdict::Reject,
                    Err(_) => Verdict::WouldPanic,
                },
            })
            .collect()
    }
}

// ------------------------------------------------------------------------------------------------ CompositeDLogProof
fn dlog_statement_limbs(statement: &DLogStatement) -> Option<(u32, Vec<u32>, Vec<u32>, Vec<u32>)> {
    let n_bits = width_for(&statement.N)?;
    let kw = (n_bits / 32) as usize;
    let (mut nn, mut g, mut ni) = (vec![0u32; kw], vec![0u32; kw], vec![0u32; kw]);
    if put_limbs(&mut nn, &statement.N) && put_limbs(&mut g, &statement.g) && put_limbs(&mut ni, &statement.ni) {
        Some((n_bits, nn, g, ni))
    } else {
        None
    }
}

/// `CompositeDLogProof::prove` (src/zkproofs/wi_dlog_proof.rs:46-65): x = g^r mod N, e = H(x, g, N, ni), y = r + e s, with the
/// 512-bit nonce sampled here as the reference samples it (:53-54).
pub fn dlog_prove_one(statement: &DLogStatement, secret: &BigInt) -> Option<CompositeDLogProof> {
    let (n_bits, nn, g, ni) = dlog_statement_limbs(statement)?;
    let kw = (n_bits / 32) as usize;
    let yw = (DLOG_Y_BITS / 32) as usize;
    let big_r = BigInt::from(2).pow(512); // K + K_PRIME + SAMPLE_S, :53
    let r = BigInt::sample_below(&big_r); // :54
    let (mut s, mut rr) = (vec![0u32; 8], vec![0u32; yw]);
    if !put_limbs(&mut s, secret) || !put_limbs(&mut rr, &r) {
        return None; // a secret wider than 256 bits: GMP
    }
    let (mut x, mut y) = (vec![0u32; kw], vec![0u32; yw]);
    with_ctx(|ctx| {
        ok(unsafe { sys::zkp_dlog_prove_batch(ctx, n_bits, DLOG_Y_BITS, 1, nn.as_ptr(), g.as_ptr(), ni.as_ptr(), s.as_ptr(), rr.as_ptr(), x.as_mut_ptr(), y.as_mut_ptr(), 0) })
    })?;
    Some(CompositeDLogProof { x: get_limbs(&x), y: get_limbs(&y) })
}

/// `CompositeDLogProof::verify` (wi_dlog_proof.rs:67-91).  `None` also when the reference's pre-checks would panic (:69,72,73):
/// the GMP path then raises that panic.
pub fn dlog_verify_one(proof: &CompositeDLogProof, statement: &DLogStatement) -> Option<Result<(), IncorrectProof>> {
    let (n_bits, nn, g, ni) = dlog_statement_limbs(statement)?;
    let kw = (n_bits / 32) as usize;
    let yw = (DLOG_Y_BITS / 32) as usize;
    let (mut x, mut y) = (vec![0u32; kw], vec![0u32; yw]);
    if !put_limbs(&mut x, &proof.x) || !put_limbs(&mut y, &proof.y) {
        return None;
    }
    let mut verdict = [9u8; 1];
    with_ctx(|ctx| {
        ok(unsafe { sys::zkp_dlog_verify_batch(ctx, n_bits, DLOG_Y_BITS, 1, nn.as_ptr(), g.as_ptr(), ni.as_ptr(), x.as_ptr(), y.as_ptr(), verdict.as_mut_ptr(), 0) })
    })?;
    match verdict[0] {
        sys::ZKP_VERDICT_ACCEPT => Some(Ok(())),
        sys::ZKP_VERDICT_REJECT => Some(Err(IncorrectProof)),
        _ => None,
    }
}
