//! Runs the REAL zk-paillier crate (and the curv / kzen-paillier versions it pins) and writes everything this
//! repository's oracle can only restate: byte conventions, the Enc formula, Fiat-Shamir challenges, full
//! RangeProofNi transcripts, NiCorrectKeyProof sigma vectors, CompositeDLogProof transcripts.
//!
//! What can and cannot be pinned from OUTSIDE the crate:
//!  * `RangeProof::generate_encrypted_pairs` draws (w1, w2, r1, r2) from the OS RNG and `DataRandomnessPairs`
//!    has private fields and is not exported, so a prover run cannot be replayed from a seed.  Instead whole proofs
//!    are dumped: every Open row reveals (w1, r1, w2, r2) next to (c1, c2) — a known-answer test of Enc —, the
//!    Open / Mask pattern IS the bit string of the Fiat-Shamir challenge — a known-answer test of compute_digest,
//!    BigInt::to_bytes and the MSB-first bit order —, and the verdicts pin the verifier.
//!  * `NiCorrectKeyProof::proof(dk, None)` is deterministic: its sigma vector is a direct known answer.
//!
//! Output schema (all big integers as decimal strings, the crate's own wire format):
//! { "generator": ..., "to_bytes": [{"x","hex"}], "compute_digest": [{"items":[..],"digest"}],
//!   "enc": {"n", "items":[{"m","r","c"}]},
//!   "range_ni": [{"n","range","ciphertext","x","r","honest", "encrypted_pairs": <serde>, "proof": <serde>,
//!                 "error_factor", "verify_self": "ok"|"err", "raw": <serde_json of the whole RangeProofNi>}],
//!   "correct_key_ni": [{"p","q","n","salt_hex","sigma_vec":[..],"verify":"ok"|"err"}],
//!   "dlog": [{"N","g","ni","secret","x","y","verify":"ok"|"err"}],
//!   "serde": {"bigint_samples":[{"x","json":<serde of a bare BigInt>}], "encryption_key":{"n","json":<serde of EncryptionKey>},
//!             "dlog_statement":{"N","g","ni","json"}, "dlog_proof":{"x","y","json"}},
//!   "signed": {"rem":[{"a","m","rem"}], "mod_pow":[{"base","exp","modulus","out"}], "enc":[{"m","r","c"}],
//!              "range_ni":[{"name","raw":<a RangeProofNi document with one mutated decimal field>,"verify_self":"ok"|"err"|"panic"}]} }
//! The "signed" section pins what the reference does with NEGATIVE and over-wide integers, which a deserialised
//! proof may hold: Rust's `%` on BigInt, mod_pow on a negative base, Enc on negative operands, and the verdicts of verify_self
//! on documents whose decimal-string fields were edited after an honest prover made them.
use std::env;
use std::fs;

use curv::arithmetic::traits::*;
use curv::BigInt;
use paillier::{
    DecryptionKey, EncryptWithChosenRandomness, EncryptionKey, KeyGeneration, Keypair, Paillier,
    Randomness, RawCiphertext, RawPlaintext,
};
use serde_json::{json, Value};
use zk_paillier::zkproofs::{compute_digest, CompositeDLogProof, DLogStatement, NiCorrectKeyProof, RangeProofNi};

const P: &str = "148677972634832330983979593310074301486537017973460461278300587514468301043894574906886127642530475786889672304776052879927627556769456140664043088700743909632312483413393134504352834240399191134336344285483935856491230340093391784574980688823380828143810804684752914935441384845195613674104960646037368551517";
const Q: &str = "158741574437007245654463598139927898730476924736461654463975966787719309357536545869203069369466212089132653564188443272208127277664424448947476335413293018778018615899291704693105620242763173357203898195318179150836424196645745308205164116144020613415407736216097185962171301808761138424668335445923774195463";

fn dec(x: &BigInt) -> String {
    x.to_str_radix(10)
}
fn hex(b: &[u8]) -> String {
    b.iter().map(|v| format!("{:02x}", v)).collect()
}
fn big(s: &str) -> BigInt {
    BigInt::from_str_radix(s, 10).unwrap()
}

/// the fixed keypair of the crate's own tests (src/zkproofs/range_proof_ni.rs:141-145)
fn test_keypair() -> Keypair {
    Keypair { p: big(P), q: big(Q) }
}

fn enc(ek: &EncryptionKey, m: &BigInt, r: &BigInt) -> BigInt {
    let c: RawCiphertext = Paillier::encrypt_with_chosen_randomness(ek, RawPlaintext::from(m), &Randomness::from(r));
    c.0.into_owned()
}

fn range_case(ek: &EncryptionKey, honest: bool) -> Value {
    // same shapes as the crate's tests (range_proof_ni.rs:163-199)
    let range = BigInt::sample(256);
    let secret_r = BigInt::sample_below(&ek.n);
    let secret_x = if honest {
        BigInt::sample_below(&range.div_floor(&BigInt::from(3)))
    } else {
        BigInt::sample_range(&(BigInt::from(100) * &range), &(BigInt::from(10000) * &range))
    };
    let cipher_x = enc(ek, &secret_x, &secret_r);
    let proof = RangeProofNi::prove(ek, &range, &cipher_x, &secret_x, &secret_r);
    let verdict = if proof.verify_self().is_ok() { "ok" } else { "err" };
    let raw = serde_json::to_value(&proof).unwrap();
    json!({
        "n": dec(&ek.n), "range": dec(&range), "ciphertext": dec(&cipher_x), "x": dec(&secret_x), "r": dec(&secret_r),
        "honest": honest, "encrypted_pairs": raw["encrypted_pairs"].clone(), "proof": raw["proof"].clone(),
        "error_factor": raw["error_factor"].clone(), "verify_self": verdict, "raw": raw,
    })
}

/// first row of `proof` of the given kind ("Open" / "Mask")
fn first_row(doc: &Value, kind: &str) -> usize {
    doc["proof"].as_array().unwrap().iter().position(|r| r.get(kind).is_some()).expect("no such row")
}

/// field `f` of row `i` (a decimal string, serialize.rs:8-31) -> `edit(value)`
fn edit_field(doc: &mut Value, i: usize, kind: &str, f: &str, edit: &dyn Fn(&BigInt) -> BigInt) {
    let cur = big(doc["proof"][i][kind][f].as_str().unwrap());
    doc["proof"][i][kind][f] = Value::String(dec(&edit(&cur)));
}

/// verify_self of a document as the crate deserialises it; a panic of the reference (index out of bounds) is an outcome
fn verdict_of(doc: &Value) -> &'static str {
    let parsed: Result<RangeProofNi, _> = serde_json::from_value(doc.clone());
    match parsed {
        Err(_) => "serde",
        Ok(p) => match std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| p.verify_self())) {
            Ok(Ok(())) => "ok",
            Ok(Err(_)) => "err",
            Err(_) => "panic",
        },
    }
}

/// documents with one edited field each (no re-answering of the challenge: only fields that are not hashed are touched)
fn signed_range_cases(ek: &EncryptionKey) -> Vec<Value> {
    let range = BigInt::sample(256);
    let secret_r = BigInt::sample_below(&ek.n);
    let secret_x = BigInt::sample_below(&range.div_floor(&BigInt::from(3)));
    let cipher_x = enc(ek, &secret_x, &secret_r);
    let honest = serde_json::to_value(&RangeProofNi::prove(ek, &range, &cipher_x, &secret_x, &secret_r)).unwrap();
    let n = ek.n.clone();
    let wide = BigInt::from(2).pow(2100) * &n;
    let mut out: Vec<Value> = Vec::new();
    let mut push = |name: &str, doc: Value| {
        let v = verdict_of(&doc);
        out.push(json!({"name": name, "raw": doc, "verify_self": v}));
    };
    push("none", honest.clone());
    {
        let mut d = honest.clone();
        let i = first_row(&d, "Open");
        edit_field(&mut d, i, "Open", "r1", &|v| v - &n);
        push("neg_r1_open", d);
    }
    {
        let mut d = honest.clone();
        let i = first_row(&d, "Mask");
        edit_field(&mut d, i, "Mask", "masked_r", &|v| v - &n);
        push("neg_masked_r", d);
    }
    {
        let mut d = honest.clone();
        let i = first_row(&d, "Mask");
        edit_field(&mut d, i, "Mask", "masked_r", &|v| v + &wide);
        push("wide_masked_r", d);
    }
    {
        let mut d = honest.clone();
        let i = first_row(&d, "Mask");
        edit_field(&mut d, i, "Mask", "masked_x", &|v| v - &n);
        push("neg_masked_x", d);
    }
    {
        let mut d = honest.clone();
        let i = first_row(&d, "Open");
        edit_field(&mut d, i, "Open", "w2", &|v| v - &n);
        push("neg_w2_only", d);
    }
    {
        let mut d = honest.clone();
        let i = first_row(&d, "Open");
        edit_field(&mut d, i, "Open", "w1", &|v| v - &n);
        edit_field(&mut d, i, "Open", "w2", &|v| v - &n);
        push("neg_w1_and_w2", d);
    }
    {
        let mut d = honest.clone();
        d["proof"].as_array_mut().unwrap().truncate(100);
        push("short_responses", d);
    }
    out
}

fn main() {
    let out_path = env::args().nth(1).unwrap_or_else(|| "reference_vectors.json".to_string());
    let (ek, dk): (EncryptionKey, DecryptionKey) = test_keypair().keys();

    // BigInt::to_bytes conventions: zero, one byte, a value with a zero low byte
    let to_bytes: Vec<Value> = [BigInt::zero(), BigInt::from(255), BigInt::from(256), BigInt::from(65536)]
        .iter()
        .map(|x| json!({"x": dec(x), "hex": hex(&x.to_bytes())}))
        .collect();

    // compute_digest (src/zkproofs/utils.rs:9-22) incl. zero elements
    let lists: Vec<Vec<BigInt>> = vec![
        vec![BigInt::zero()],
        vec![BigInt::zero(), BigInt::zero(), BigInt::one()],
        vec![ek.n.clone(), BigInt::from(255), ek.nn.clone()],
        vec![BigInt::from(256), BigInt::from(255), BigInt::from(65536)],
    ];
    let digests: Vec<Value> = lists
        .iter()
        .map(|l| json!({"items": l.iter().map(dec).collect::<Vec<_>>(), "digest": dec(&compute_digest(l.iter()))}))
        .collect();

    // Paillier::encrypt_with_chosen_randomness incl. m >= n and r >= n (arbitrary-precision inputs are legal)
    let full = BigInt::from(2).pow(2048) - BigInt::one();
    let ms = vec![BigInt::zero(), BigInt::one(), BigInt::sample(256), &ek.n - BigInt::one(), full.clone()];
    let rs = vec![BigInt::sample_below(&ek.n), BigInt::sample_below(&ek.n), BigInt::sample_below(&ek.n), BigInt::one(), full];
    let enc_items: Vec<Value> = ms
        .iter()
        .zip(rs.iter())
        .map(|(m, r)| json!({"m": dec(m), "r": dec(r), "c": dec(&enc(&ek, m, r))}))
        .collect();

    // RangeProofNi transcripts: two honest proofs, one with an out-of-range witness (rejected)
    let range_ni: Vec<Value> = vec![range_case(&ek, true), range_case(&ek, true), range_case(&ek, false)];

    // NiCorrectKeyProof (deterministic): the fixture key and a fresh 2048-bit key
    let mut ck: Vec<Value> = Vec::new();
    let fresh = Paillier::keypair_with_modulus_size(2048);
    for kp in vec![test_keypair(), fresh] {
        let (ek2, dk2) = kp.keys();
        let proof = NiCorrectKeyProof::proof(&dk2, None);
        let v = if proof.verify(&ek2, zk_paillier::zkproofs::SALT_STRING).is_ok() { "ok" } else { "err" };
        ck.push(json!({"p": dec(&dk2.p), "q": dec(&dk2.q), "n": dec(&ek2.n), "salt_hex": hex(zk_paillier::zkproofs::SALT_STRING),
                       "sigma_vec": proof.sigma_vec.iter().map(dec).collect::<Vec<_>>(), "verify": v}));
    }
    let _ = &dk;

    // CompositeDLogProof (src/zkproofs/wi_dlog_proof.rs:117-137): honest statement ni = g^-s
    let mut dlog: Vec<Value> = Vec::new();
    {
        let n_tilde = ek.n.clone();
        let one = BigInt::one();
        let h1 = BigInt::sample_below(&(&n_tilde - &one));
        let s = BigInt::sample(256);
        let h1_inv = BigInt::mod_inv(&h1, &n_tilde).unwrap();
        let ni = BigInt::mod_pow(&h1_inv, &s, &n_tilde);
        let st = DLogStatement { N: n_tilde, g: h1, ni };
        let proof = CompositeDLogProof::prove(&st, &s);
        let v = if proof.verify(&st).is_ok() { "ok" } else { "err" };
        dlog.push(json!({"N": dec(&st.N), "g": dec(&st.g), "ni": dec(&st.ni), "secret": dec(&s), "x": dec(&proof.x), "y": dec(&proof.y), "verify": v}));
    }

    // Text forms of the UN-annotated types (no #[serde(with = ...)] in the crate: range_proof_ni.rs:36-44 `ek`, `range`,
    // `ciphertext`; wi_dlog_proof.rs:32-43): whatever curv's BigInt and kzen-paillier's EncryptionKey serialize as, with the value
    // in decimal next to every sample so that a reader can tell the encodings apart.
    let samples: Vec<Value> = [BigInt::zero(), BigInt::from(255), BigInt::from(256), ek.n.clone()]
        .iter()
        .map(|x| json!({"x": dec(x), "json": serde_json::to_value(x).unwrap()}))
        .collect();
    let serde_section = {
        let n_tilde = ek.n.clone();
        let h1 = BigInt::sample_below(&(&n_tilde - &BigInt::one()));
        let s = BigInt::sample(256);
        let ni = BigInt::mod_pow(&BigInt::mod_inv(&h1, &n_tilde).unwrap(), &s, &n_tilde);
        let st = DLogStatement { N: n_tilde, g: h1, ni };
        let pr = CompositeDLogProof::prove(&st, &s);
        json!({
            "bigint_samples": samples,
            "encryption_key": {"n": dec(&ek.n), "json": serde_json::to_value(&ek).unwrap()},
            "dlog_statement": {"N": dec(&st.N), "g": dec(&st.g), "ni": dec(&st.ni), "json": serde_json::to_value(&st).unwrap()},
            "dlog_proof": {"x": dec(&pr.x), "y": dec(&pr.y), "json": serde_json::to_value(&pr).unwrap()},
        })
    };

    // negative / over-wide operands: the operators a deserialised proof reaches with them
    let signed_section = {
        let a = BigInt::sample(300);
        let m = BigInt::sample(200) + BigInt::one();
        let neg = |x: &BigInt| BigInt::zero() - x;
        let rem: Vec<Value> = vec![(a.clone(), m.clone()), (neg(&a), m.clone()), (a.clone(), neg(&m)), (neg(&a), neg(&m)), (neg(&m), m.clone())]
            .iter()
            .map(|(x, y)| json!({"a": dec(x), "m": dec(y), "rem": dec(&(x.clone() % y))}))
            .collect();
        let base = BigInt::sample_below(&ek.n);
        let mod_pow: Vec<Value> = vec![neg(&base), neg(&(&base + &ek.nn)), &base + &ek.nn]
            .iter()
            .map(|b| json!({"base": dec(b), "exp": dec(&ek.n), "modulus": dec(&ek.nn), "out": dec(&BigInt::mod_pow(b, &ek.n, &ek.nn))}))
            .collect();
        let r = BigInt::sample_below(&ek.n);
        let x = BigInt::sample(256);
        let pairs = vec![
            (neg(&BigInt::one()), BigInt::one()),
            (neg(&x), r.clone()),
            (x.clone(), neg(&r)),
            (neg(&x), neg(&r)),
            (neg(&ek.n), r.clone()),
            (BigInt::from(7), BigInt::zero()),
            (neg(&BigInt::from(7)), BigInt::zero()),
        ];
        let enc_signed: Vec<Value> = pairs.iter().map(|(m, r)| json!({"m": dec(m), "r": dec(r), "c": dec(&enc(&ek, m, r))})).collect();
        json!({"rem": rem, "mod_pow": mod_pow, "enc": enc_signed, "range_ni": signed_range_cases(&ek)})
    };

    let doc = json!({
        "generator": "tools/reference_vectors (zk-paillier 0.4.4 + curv-kzen 0.10 / rust-gmp-kzen + kzen-paillier 0.4.3)",
        "to_bytes": to_bytes, "compute_digest": digests, "enc": {"n": dec(&ek.n), "items": enc_items},
        "range_ni": range_ni, "correct_key_ni": ck, "dlog": dlog, "serde": serde_section, "signed": signed_section,
    });
    fs::write(&out_path, serde_json::to_string(&doc).unwrap()).unwrap();
    println!("wrote {}", out_path);
}
