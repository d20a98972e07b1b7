// C++ mirror of the reference's own unit tests for the hot path, written against
// zk-paillier_amd/host/zkproofs.hpp (same names and flow as the #[cfg(test)] modules of
// src/zkproofs/range_proof_ni.rs:131-199, correct_key_ni.rs:120-138, wi_dlog_proof.rs:110-196).
// Needs a gfx950 GPU (every modexp runs through libzkp_hip.so).  Exit code 0 = all passed.
#include <cstdio>
#include <functional>
#include <string>

#include "../../zk-paillier_amd/host/zkproofs.hpp"

using namespace zkproofs;

static const size_t RANGE_BITS = 256;   // range_proof_ni.rs:133

// range_proof_ni.rs:141-145 (also benches/all.rs:73-77)
static Keypair test_keypair() {
  return Keypair{
      BigInt::from_str_radix10("148677972634832330983979593310074301486537017973460461278300587514468301043894574906886127642530475786889672304776052879927627556769456140664043088700743909632312483413393134504352834240399191134336344285483935856491230340093391784574980688823380828143810804684752914935441384845195613674104960646037368551517"),
      BigInt::from_str_radix10("158741574437007245654463598139927898730476924736461654463975966787719309357536545869203069369466212089132653564188443272208127277664424448947476335413293018778018615899291704693105620242763173357203898195318179150836424196645745308205164116144020613415407736216097185962171301808761138424668335445923774195463")};
}

static int failures = 0;
static void run(const char* name, const std::function<void()>& f, bool should_panic = false) {
  bool panicked = false;
  std::string what;
  try { f(); } catch (const Panic& e) { panicked = true; what = e.what(); }
  const bool ok = panicked == should_panic;
  std::printf("%s %s%s%s\n", ok ? "PASS" : "FAIL", name, what.empty() ? "" : "  [panic: ", what.empty() ? "" : (what + "]").c_str());
  if (!ok) failures++;
}
#define ASSERT(c) do { if (!(c)) throw Panic(std::string("assertion failed: ") + #c); } while (0)

// ---- range_proof_ni.rs tests
static void test_prover() {   // :148-160
  auto [ek, dk] = test_keypair().keys();
  BigInt range = BigInt::sample(RANGE_BITS);
  BigInt secret_r = BigInt::sample_below(ek.n);
  BigInt secret_x = BigInt::sample_below(range);
  BigInt ciphertext = Paillier::encrypt_with_chosen_randomness(ek, secret_x, secret_r);
  RangeProofNi::prove(ek, range, ciphertext, secret_x, secret_r);
}
static void test_verifier_for_correct_proof() {   // :163-177
  auto [ek, dk] = test_keypair().keys();
  BigInt range = BigInt::sample(RANGE_BITS);
  BigInt secret_r = BigInt::sample_below(ek.n);
  BigInt secret_x = BigInt::sample_below(range.div_floor(BigInt(3)));
  BigInt cipher_x = Paillier::encrypt_with_chosen_randomness(ek, secret_x, secret_r);
  RangeProofNi range_proof = RangeProofNi::prove(ek, range, cipher_x, secret_x, secret_r);
  range_proof.verify(ek, cipher_x).expect("range proof error");
  ASSERT(range_proof.verify_self().is_ok());
}
static void test_verifier_for_incorrect_proof() {   // :180-199, #[should_panic]
  auto [ek, dk] = test_keypair().keys();
  BigInt range = BigInt::sample(RANGE_BITS);
  BigInt secret_r = BigInt::sample_below(ek.n);
  BigInt secret_x = BigInt::sample_range(BigInt(100) * range, BigInt(10000) * range);
  BigInt cipher_x = Paillier::encrypt_with_chosen_randomness(ek, secret_x, secret_r);
  RangeProofNi range_proof = RangeProofNi::prove(ek, range, cipher_x, secret_x, secret_r);
  range_proof.verify(ek, cipher_x).expect("range proof error");
}
static void test_verify_asserts_statement() {   // the two assert_eq! of verify (:86,88), #[should_panic]
  auto [ek, dk] = test_keypair().keys();
  BigInt range = BigInt::sample(RANGE_BITS);
  BigInt r = BigInt::sample_below(ek.n), x = BigInt::sample_below(range.div_floor(BigInt(3)));
  BigInt c = Paillier::encrypt_with_chosen_randomness(ek, x, r);
  RangeProofNi p = RangeProofNi::prove(ek, range, c, x, r);
  p.verify(ek, c + BigInt::one());
}
static void test_batch_round_trip() {   // many provers, one key: what the GPU is for
  auto [ek, dk] = test_keypair().keys();
  std::vector<RangeProofNi::Statement> st;
  for (int i = 0; i < 6; i++) {
    BigInt range = BigInt::sample(RANGE_BITS);
    BigInt r = BigInt::sample_below(ek.n);
    BigInt x = i == 4 ? BigInt::sample_range(BigInt(100) * range, BigInt(10000) * range) : BigInt::sample_below(range.div_floor(BigInt(3)));
    st.push_back({range, Paillier::encrypt_with_chosen_randomness(ek, x, r), x, r});
  }
  auto proofs = RangeProofNi::prove_batch(ek, st);
  std::vector<const RangeProofNi*> ptr;
  for (auto& p : proofs) ptr.push_back(&p);
  auto res = RangeProofNi::verify_batch(ek, ptr);
  for (int i = 0; i < 6; i++) ASSERT(res[i].is_ok() == (i != 4));
}

static void test_batch_survives_crafted_proofs() {   // over-wide prover-chosen fields: per-proof verdicts, no exception for the batch
  auto [ek, dk] = test_keypair().keys();
  std::vector<RangeProofNi::Statement> st;
  for (int i = 0; i < 6; i++) {
    BigInt range = BigInt::sample(RANGE_BITS);
    BigInt r = BigInt::sample_below(ek.n), x = BigInt::sample_below(range.div_floor(BigInt(3)));
    st.push_back({range, Paillier::encrypt_with_chosen_randomness(ek, x, r), x, r});
  }
  auto proofs = RangeProofNi::prove_batch(ek, st);
  auto first = [](const RangeProofNi& p, Response::Kind k) { size_t i = 0; while (p.proof.responses[i].kind != k) i++; return i; };
  const BigInt wide = BigInt::pow2(2100), wider = BigInt::pow2(4200);
  {  // 1: masked_x wider than the key: fails the bound T <= masked_x <= 2T (range_proof.rs:338) -> Err
    Response& m = proofs[1].proof.responses[first(proofs[1], Response::Mask)];
    m.masked_x = m.masked_x + wide;
  }
  {  // 2: r1 + n * 2^80 on an Open row and masked_r + n * 2^80 on a Mask row: r^n mod n^2 only depends on r mod n -> still Ok
    Response& o = proofs[2].proof.responses[first(proofs[2], Response::Open)];
    o.r1 = o.r1 + ek.n * BigInt::pow2(80);
    Response& m = proofs[2].proof.responses[first(proofs[2], Response::Mask)];
    m.masked_r = m.masked_r + ek.n * BigInt::pow2(80);
  }
  {  // 3: ciphertext + n^2 * 2^4200: only its product mod n^2 is used (:324-328) and it is not hashed -> still Ok
    proofs[3].ciphertext = proofs[3].ciphertext + ek.nn * wider;
  }
  {  // 4: c1[i] + n^2 on an Open row: compared unreduced (:293-298) -> Err;  w1 over-wide would also fail the range flag
    const size_t i = first(proofs[4], Response::Open);
    proofs[4].encrypted_pairs.c1[i] = proofs[4].encrypted_pairs.c1[i] + ek.nn * wider;
  }
  proofs[5].proof.responses.resize(100);   // 5: responses[i] for i >= 100 is an index panic in the reference (:274)
  std::vector<const RangeProofNi*> ptr;
  for (auto& p : proofs) ptr.push_back(&p);
  auto res = RangeProofNi::verify_batch(ek, ptr);
  ASSERT(res[0].is_ok());
  ASSERT(res[1].is_err());
  ASSERT(res[2].is_ok());
  ASSERT(res[3].is_ok());
  ASSERT(res[4].is_err());
  ASSERT(res[5].would_panic());
  bool threw = false;
  try { (void)res[5].is_ok(); } catch (const Panic&) { threw = true; }
  ASSERT(threw);
}

// one odd proof at the HEAD of a batch (error_factor 127: the reference then checks rows 0..126 of the same transcript and accepts) must not
// push the honest proofs behind it off the single-call path (round-4 advisor finding: the batch-wide row count came from proofs[0])
static void test_odd_error_factor_at_the_head_of_a_batch() {
  auto [ek, dk] = test_keypair().keys();
  std::vector<RangeProofNi::Statement> st;
  for (int i = 0; i < 4; i++) {
    BigInt range = BigInt::sample(RANGE_BITS);
    BigInt r = BigInt::sample_below(ek.n), x = BigInt::sample_below(range.div_floor(BigInt(3)));
    st.push_back({range, Paillier::encrypt_with_chosen_randomness(ek, x, r), x, r});
  }
  auto proofs = RangeProofNi::prove_batch(ek, st);
  proofs[0].error_factor = 127;
  proofs[3].proof.responses[5].kind = proofs[3].proof.responses[5].kind == Response::Open ? Response::Mask : Response::Open;   // a plain reject among them
  std::vector<const RangeProofNi*> ptr;
  for (auto& p : proofs) ptr.push_back(&p);
  auto res = RangeProofNi::verify_batch(ek, ptr);
  ASSERT(res[0].is_ok() && res[1].is_ok() && res[2].is_ok() && res[3].is_err());
  ASSERT(last_host_timing().proofs == 4 && last_host_timing().general_proofs == 1);
  // and a batch in which EVERY proof carries the other row count still takes the single call
  for (auto& p : proofs) p.error_factor = 127;
  res = RangeProofNi::verify_batch(ek, ptr);
  ASSERT(res[0].is_ok() && res[1].is_ok() && res[2].is_ok() && res[3].is_err());
  ASSERT(last_host_timing().general_proofs == 4);      // (127 rows declared, 128 stored: not the canonical shape, so the general path — correct, and not the common case)
}

// c_j[i] + k n^2 on a Mask row.  The row equation only sees the product mod n^2 (range_proof.rs:324-328), but the Fiat-Shamir
// challenge is hashed over the RAW pairs (range_proof_ni.rs:110-113, utils.rs:9-22): (a) added AFTER the proof was made the
// challenge changes and the reference rejects; (b) a prover who hashes the raw value itself gets a proof the reference accepts.
static void test_overwide_pair_on_a_mask_row_follows_the_raw_transcript() {
  auto [ek, dk] = test_keypair().keys();
  const BigInt range = BigInt::sample(RANGE_BITS);
  const BigInt r = BigInt::sample_below(ek.n), x = BigInt::sample_below(range.div_floor(BigInt(3)));
  const BigInt cx = Paillier::encrypt_with_chosen_randomness(ek, x, r);
  const BigInt bump = ek.nn * BigInt::pow2(100);
  {  // (a)
    RangeProofNi p = RangeProofNi::prove(ek, range, cx, x, r);
    size_t i = 0;
    while (p.proof.responses[i].kind != Response::Mask) i++;
    auto& cj = p.proof.responses[i].j == 1 ? p.encrypted_pairs.c1[i] : p.encrypted_pairs.c2[i];
    cj = cj + bump;
    ASSERT(RangeProofNi::verify_batch(ek, {&p})[0].is_err());
  }
  bool built = false;
  for (size_t i = 0; i < 16 && !built; i++) {   // (b): find a row that the challenge over the bumped transcript makes a Mask row
    auto [pairs, data] = RangeProof::generate_encrypted_pairs(ek, range, RangeProofNi::SECURITY_PARAMETER);
    pairs.c1[i] = pairs.c1[i] + bump; pairs.c2[i] = pairs.c2[i] + bump;
    detail::Sha256 sh;
    sh.update(ek.n);
    for (auto& c : pairs.c1) sh.update(c);
    for (auto& c : pairs.c2) sh.update(c);
    ChallengeBits e; e.bytes = sh.finish().to_bytes();
    if (e.bytes.size() != 32 || !((e.bytes[i / 8] >> (7 - i % 8)) & 1)) continue;   // bit i clear: row i would be an Open row
    RangeProofNi p;
    p.ek = ek; p.range = range; p.ciphertext = cx; p.encrypted_pairs = pairs; p.error_factor = RangeProofNi::SECURITY_PARAMETER;
    p.proof = RangeProof::generate_proof(ek, x, r, e, range, data, RangeProofNi::SECURITY_PARAMETER);
    ASSERT(p.proof.responses[i].kind == Response::Mask);
    ASSERT(RangeProofNi::verify_batch(ek, {&p})[0].is_ok());
    // ... and next to ordinary proofs in one batch
    RangeProofNi q = RangeProofNi::prove(ek, range, cx, x, r);
    auto res = RangeProofNi::verify_batch(ek, {&q, &p, &q});
    ASSERT(res[0].is_ok() && res[1].is_ok() && res[2].is_ok());
    built = true;
  }
  ASSERT(built);
}

// ---- correct_key_ni.rs tests (the reference draws a fresh key with Paillier::keypair(); key generation is
//      not on the hot path, the fixture key is used instead)
static void test_correct_zk_proof_no_salt_str() {   // :126-130
  auto [ek, dk] = test_keypair().keys();
  NiCorrectKeyProof proof = NiCorrectKeyProof::proof(dk);
  ASSERT(proof.verify(ek, SALT_STRING, 4).is_ok());
}
static void test_correct_zk_proof_with_salt_str() {   // :133-138
  const uint8_t salt_str[8] = {90, 101, 110, 32, 71, 111, 32, 88};
  auto [ek, dk] = test_keypair().keys();
  NiCorrectKeyProof proof = NiCorrectKeyProof::proof(dk, salt_str, 8);
  ASSERT(proof.verify(ek, salt_str, 8).is_ok());
  ASSERT(proof.verify(ek, SALT_STRING, 4).is_err());          // wrong salt
  proof.sigma_vec[3] = proof.sigma_vec[3] + BigInt::one();
  ASSERT(proof.verify(ek, salt_str, 8).is_err());             // tampered root
}

static void test_correct_key_verify_batch_and_noncanonical_roots() {   // many keys in one launch; sigma + k n and sigma - n verify like sigma
  auto [ek, dk] = test_keypair().keys();
  NiCorrectKeyProof good = NiCorrectKeyProof::proof(dk), wide = good, neg = good, bad = good, few = good;
  wide.sigma_vec[2] = wide.sigma_vec[2] + ek.n * BigInt::pow2(300);        // mod_pow(sigma, n, n) only sees sigma mod n (correct_key_ni.rs:92)
  neg.sigma_vec[5] = neg.sigma_vec[5] - ek.n;                              // negative: mpz_powm reduces it into [0, n)
  bad.sigma_vec[7] = bad.sigma_vec[7] + BigInt::one();
  few.sigma_vec.resize(10);                                                // sigma_vec[10]: index panic
  {  // the wire format of such a proof: the GPU reader answers "host path" for the negative root, the host parser takes over
    const std::string doc = serde_json::to_string(good, ek);
    const size_t q = doc.find('"', doc.find('[')) + 1;                   // first root: make it negative by writing sigma_0 - n
    const size_t qe = doc.find('"', q);
    const BigInt s0 = BigInt::from_str_radix10(doc.substr(q, qe - q));
    const std::string neg_doc = doc.substr(0, q) + (s0 - ek.n).to_str_radix10() + doc.substr(qe);
    ASSERT(neg_doc.find("\"-") != std::string::npos);
    NiCorrectKeyProof back = serde_json::correct_key_from_str(ek, neg_doc);
    ASSERT(back.sigma_vec.size() == 11 && back.sigma_vec[0].is_negative() && back.sigma_vec[1] == good.sigma_vec[1]);
    ASSERT(back.verify(ek).is_ok());
  }
  EncryptionKey even{ek.n + BigInt::one(), (ek.n + BigInt::one()) * (ek.n + BigInt::one())};      // gcd(primorial, n) >= 2: Err (correct_key_ni.rs:87-88,95)
  EncryptionKey huge{BigInt::pow2(4200) + BigInt::one(), BigInt::one()};
  EncryptionKey zero{BigInt(0), BigInt(0)};                                                        // rho_i % n: the reference's division-by-zero panic (:82-85)
  auto res = NiCorrectKeyProof::verify_batch({{&ek, &good}, {&ek, &wide}, {&ek, &neg}, {&ek, &bad}, {&ek, &few}, {&even, &good}, {&huge, &good},
                                              {&even, &few}, {&zero, &good}, {&zero, &few}});
  ASSERT(res[0].is_ok() && res[1].is_ok() && res[2].is_ok());
  ASSERT(res[3].is_err());
  ASSERT(res[4].would_panic());
  ASSERT(res[5].is_err());
  ASSERT(res[6].is_unsupported());
  ASSERT(res[7].would_panic());           // the reference indexes sigma_vec[10] (:92) before it compares anything: the short vector wins over the even key
  ASSERT(res[8].would_panic() && res[9].would_panic());
}

// ---- wi_dlog_proof.rs tests
static int legendre_symbol(const BigInt& a, const BigInt& p) {   // :94-107
  BigInt e = (p - BigInt::one()).div_floor(BigInt(2));
  return mod_pow(a, e, p) == BigInt::one() ? 1 : -1;
}
static const size_t SAMPLE_S = 256;
static DLogStatement dlog_statement(int variant, BigInt* secret_out) {
  auto [ek, dk] = test_keypair().keys();
  BigInt one = BigInt::one();
  BigInt S = BigInt::pow2(SAMPLE_S);
  BigInt h1 = BigInt::sample_range(one, ek.n - one);
  while (legendre_symbol(h1, dk.p) * legendre_symbol(h1, dk.q) != -1) h1 = BigInt::sample_range(one, ek.n - one);   // Jacobi -1, :124-128
  BigInt secret = BigInt::sample_below(S);
  BigInt h2;
  if (variant == 0) h2 = mod_pow(BigInt::mod_inv(h1, ek.n), secret, ek.n);   // :130-131
  else if (variant == 1) h2 = mod_pow(h1, secret, ek.n);                      // :159 "+secret"
  else h2 = BigInt::sample_range(one, ek.n - one);                            // :187 random
  *secret_out = secret;
  return DLogStatement{ek.n, h1, h2};
}
static void test_correct_dlog_proof() {   // :117-141
  BigInt secret;
  DLogStatement st = dlog_statement(0, &secret);
  CompositeDLogProof proof = CompositeDLogProof::prove(st, secret);
  ASSERT(proof.verify(st).is_ok());
}
static void test_bad_dlog_proof() {   // :145-168 #[should_panic]
  BigInt secret;
  DLogStatement st = dlog_statement(1, &secret);
  CompositeDLogProof proof = CompositeDLogProof::prove(st, secret);
  ASSERT(proof.verify(st).is_ok());
}
static void test_bad_dlog_proof_2() {   // :172-196 #[should_panic]
  BigInt secret;
  DLogStatement st = dlog_statement(2, &secret);
  CompositeDLogProof proof = CompositeDLogProof::prove(st, secret);
  ASSERT(proof.verify(st).is_ok());
}

// ---- zero_enc_proof.rs tests (:112-155) and correct_ciphertext.rs tests (:113-162); fixture key instead of keygen
static void test_zero_proof() {
  auto [ek, dk] = test_keypair().keys();
  BigInt r = BigInt::sample_below(ek.n);
  BigInt c = Paillier::encrypt_with_chosen_randomness(ek, BigInt::zero(), r);
  ZeroProof proof = ZeroProof::prove(ZeroWitness{r}, ZeroStatement{ek, c});
  ASSERT(proof.verify(ZeroStatement{ek, c}).is_ok());
}
static void test_one_proof() {   // #[should_panic]: c encrypts 1
  auto [ek, dk] = test_keypair().keys();
  BigInt r = BigInt::sample_below(ek.n);
  BigInt c = Paillier::encrypt_with_chosen_randomness(ek, BigInt::one(), r);
  ZeroProof proof = ZeroProof::prove(ZeroWitness{r}, ZeroStatement{ek, c});
  ASSERT(proof.verify(ZeroStatement{ek, c}).is_ok());
}
static void test_ciphertext_proof() {
  auto [ek, dk] = test_keypair().keys();
  BigInt x = BigInt::sample_below(ek.n), r = BigInt::sample_below(ek.n);
  BigInt c = Paillier::encrypt_with_chosen_randomness(ek, x, r);
  CiphertextProof proof = CiphertextProof::prove(CiphertextWitness{x, r}, CiphertextStatement{ek, c});
  ASSERT(proof.verify(CiphertextStatement{ek, c}).is_ok());
}
static void test_bad_ciphertext_proof() {   // #[should_panic]: witness r + 1
  auto [ek, dk] = test_keypair().keys();
  BigInt x = BigInt::sample_below(ek.n), r = BigInt::sample_below(ek.n);
  BigInt c = Paillier::encrypt_with_chosen_randomness(ek, x, r);
  CiphertextProof proof = CiphertextProof::prove(CiphertextWitness{x, r + BigInt::one()}, CiphertextStatement{ek, c});
  ASSERT(proof.verify(CiphertextStatement{ek, c}).is_ok());
}

// ---- correct_key.rs tests (:200-236), interactive protocol
static void test_correct_zk_proof() {
  auto [ek, dk] = test_keypair().keys();
  auto [challenge, verification_aid] = CorrectKey::challenge(ek);
  auto proof_results = CorrectKey::prove(dk, challenge);
  ASSERT(proof_results.is_ok());
  ASSERT(CorrectKey::verify(proof_results.unwrap(), verification_aid).is_ok());
}
static void test_incorrect_zk_proof() {
  auto [ek, dk] = test_keypair().keys();
  auto [challenge, verification_aid] = CorrectKey::challenge(ek);
  challenge.e = challenge.e + BigInt::one();
  auto proof_results = CorrectKey::prove(dk, challenge);
  ASSERT(proof_results.is_err());   // manipulated challenge
  ASSERT(proof_results.err == CorrectKeyProveError::EWasntComputedCorrectly);
}
static void test_incorrect_zk_proof_2() {
  auto [ek, dk] = test_keypair().keys();
  auto [challenge, verification_aid] = CorrectKey::challenge(ek);
  auto proof_results = CorrectKey::prove(dk, challenge);
  ASSERT(proof_results.is_ok());
  verification_aid.s_digest = verification_aid.s_digest + BigInt::one();
  ASSERT(CorrectKey::verify(proof_results.unwrap(), verification_aid).is_err());   // manipulated aid
}

// ---- verlin_proof.rs tests (:181-262).  (Paillier::encrypt draws its own randomness; here Enc with a sampled r.)
static void verlin_case(bool bad) {
  auto [ek, dk] = test_keypair().keys();
  BigInt x = BigInt::sample_below(ek.n), x_prime = BigInt::sample_below(ek.n), x_double_prime = BigInt::sample_below(ek.n);
  BigInt r_x = BigInt::sample_below(ek.n);
  while (BigInt::gcd(r_x, ek.n) != BigInt::one()) r_x = BigInt::sample_below(ek.n);
  BigInt c = Paillier::encrypt_with_chosen_randomness(ek, x, BigInt::sample_below(ek.n));
  BigInt c_prime = Paillier::encrypt_with_chosen_randomness(ek, x_prime, BigInt::sample_below(ek.n));
  BigInt phi_x = gen_phi(ek, c, c_prime, bad ? x * BigInt(2) : x, x_prime, x_double_prime, r_x);   // bad: x_bad = 2x injected (:226-236)
  VerlinProof proof = VerlinProof::prove(VerlinWitness{x, x_prime, x_double_prime, r_x}, VerlinStatement{ek, c, c_prime, phi_x});
  ASSERT(proof.verify(VerlinStatement{ek, c, c_prime, phi_x}).is_ok());
}
static void test_verlin_proof() { verlin_case(false); }
static void test_bad_verlin_proof() { verlin_case(true); }   // #[should_panic]

// ---- wire format: prove, write as serde_json text, read back, verify
static void test_serde_round_trip() {
  auto [ek, dk] = test_keypair().keys();
  BigInt range = BigInt::sample(RANGE_BITS);
  BigInt secret_r = BigInt::sample_below(ek.n);
  BigInt secret_x = BigInt::sample_below(range.div_floor(BigInt(3)));
  BigInt cipher_x = Paillier::encrypt_with_chosen_randomness(ek, secret_x, secret_r);
  RangeProofNi proof = RangeProofNi::prove(ek, range, cipher_x, secret_x, secret_r);
  const std::string pairs = serde_json::to_string(proof.encrypted_pairs, ek), resp = serde_json::to_string(proof.proof, ek);
  ASSERT(pairs.rfind("{\"c1\":[\"", 0) == 0 && resp.rfind("[{\"", 0) == 0);
  auto back = serde_json::range_from_str(ek, RangeProofNi::SECURITY_PARAMETER, {pairs}, {resp});
  RangeProofNi copy = proof;
  copy.encrypted_pairs = back[0].first; copy.proof = back[0].second;
  ASSERT(copy.encrypted_pairs.c1 == proof.encrypted_pairs.c1 && copy.encrypted_pairs.c2 == proof.encrypted_pairs.c2);
  ASSERT(serde_json::to_string(copy.proof, ek) == resp);
  ASSERT(copy.verify(ek, cipher_x).is_ok());
  NiCorrectKeyProof ck = NiCorrectKeyProof::proof(dk);
  NiCorrectKeyProof ck2 = serde_json::correct_key_from_str(ek, serde_json::to_string(ck, ek));
  ASSERT(ck2.sigma_vec == ck.sigma_vec && ck2.verify(ek).is_ok());
}

// ---- multiplication_proof.rs tests (:172-290)
static void mul_case(bool honest) {
  auto [ek, dk] = test_keypair().keys();
  BigInt a = BigInt::sample_below(ek.n), b = BigInt::sample_below(ek.n);
  BigInt c = (a * b) % ek.n;
  if (!honest) c = c + BigInt::one();
  BigInt r_a = sample_paillier_random(ek.n), r_b = sample_paillier_random(ek.n), r_c = sample_paillier_random(ek.n);
  MulStatement st{ek, Paillier::encrypt_with_chosen_randomness(ek, a, r_a), Paillier::encrypt_with_chosen_randomness(ek, b, r_b),
                  Paillier::encrypt_with_chosen_randomness(ek, c, r_c)};
  MulWitness w{a, b, c, r_a, r_b, r_c};
  MulProof proof = MulProof::prove(w, st);
  ASSERT(proof.verify(st).is_ok());
}
static void test_mul_proof() { mul_case(true); }
static void test_bad_mul_proof() { mul_case(false); }   // #[should_panic]
static void test_mod_inv() {
  auto [ek, dk] = test_keypair().keys();
  std::vector<BigInt> v = {BigInt(3), BigInt::sample_below(ek.nn), dk.p, BigInt(0)};
  auto r = mod_inv_batch(v, ek.nn);
  ASSERT(r[0].some && (r[0].value * v[0]) % ek.nn == BigInt::one());
  ASSERT(r[1].some && (r[1].value * v[1]) % ek.nn == BigInt::one());
  ASSERT(!r[2].some && !r[3].some);
}
// ---- correct_message.rs tests (:169-200)
static void cm_case(uint64_t message) {
  std::vector<BigInt> valid = {BigInt(3), BigInt(4), BigInt(5), BigInt(6)};
  auto [ek, dk] = test_keypair().keys();
  CorrectMessageProof proof = CorrectMessageProof::prove(ek, valid, BigInt(message));
  ASSERT(proof.verify().is_ok());
}
static void test_correct_message_zk_proof() { cm_case(4); }
static void test_bad_message_zk_proof() { cm_case(7); }   // #[should_panic]

// ---- range_proof.rs tests (:385-525)
static constexpr size_t SEF = RangeProof::STATISTICAL_ERROR_FACTOR;
static void test_generate_encrypted_pairs() {   // :386-391
  auto [ek, dk] = test_keypair().keys();
  BigInt range = BigInt(0x000FFFFFFFFFFFFFull);
  auto [ep, data] = RangeProof::generate_encrypted_pairs(ek, range, SEF);
  ASSERT(ep.c1.size() == SEF && ep.c2.size() == SEF && data.w1.size() == SEF);
  for (size_t i = 0; i < SEF; i++) ASSERT(ep.c1[i] == Paillier::encrypt_with_chosen_randomness(ek, data.w1[i], data.r1[i]));
}
static void test_commit_decommit() {   // :394-408
  auto [verifier_ek, verifier_dk] = test_keypair().keys();
  auto vc = RangeProof::verifier_commit(verifier_ek);
  auto [challenge, verification_aid] = CorrectKey::challenge(verifier_ek);
  auto proof_results = CorrectKey::prove(verifier_dk, challenge);
  ASSERT(proof_results.is_ok());
  ASSERT(CorrectKey::verify(proof_results.unwrap(), verification_aid).is_ok());
  ASSERT(RangeProof::verify_commit(verifier_ek, vc.com, vc.r, vc.e).is_ok());
  ChallengeBits other = vc.e; other.bytes[0] ^= 1;
  ASSERT(!RangeProof::verify_commit(verifier_ek, vc.com, vc.r, other).is_ok());
}
static void test_generate_proof() {   // :411-428
  auto [ek, dk] = test_keypair().keys();
  auto [verifier_ek, verifier_dk] = test_keypair().keys();
  BigInt range = BigInt(0x000FFFFFFFFFFFFFull);
  auto vc = RangeProof::verifier_commit(verifier_ek);
  auto [ep, data] = RangeProof::generate_encrypted_pairs(ek, range, SEF);
  BigInt secret_r = BigInt::sample_below(ek.n);
  BigInt secret_x = BigInt(0x0FFFFFFFull);
  Proof z = RangeProof::generate_proof(ek, secret_x, secret_r, vc.e, range, data, SEF);
  ASSERT(z.responses.size() == SEF);
}
static void interactive_case(bool honest) {   // :431-525
  BigInt range = BigInt::sample(RANGE_BITS);
  auto [ek, dk] = test_keypair().keys();
  auto [verifier_ek, verifier_dk] = test_keypair().keys();
  auto vc = RangeProof::verifier_commit(verifier_ek);
  ASSERT(RangeProof::verify_commit(verifier_ek, vc.com, vc.r, vc.e).is_ok());
  auto [ep, data] = RangeProof::generate_encrypted_pairs(ek, range, SEF);
  BigInt secret_r = BigInt::sample_below(ek.n);
  BigInt secret_x = honest ? BigInt::sample_below(range.div_floor(BigInt(3))) : BigInt::sample_range(BigInt(100) * range, BigInt(10000) * range);
  BigInt cipher_x = Paillier::encrypt_with_chosen_randomness(ek, secret_x, secret_r);
  Proof z = RangeProof::generate_proof(ek, secret_x, secret_r, vc.e, range, data, SEF);
  Result result = RangeProof::verifier_output(ek, vc.e, ep, z, range, cipher_x, SEF);
  ASSERT(result.is_ok() == honest);
}
// the interactive verifier on values the fixed-width call cannot carry: a randomness field moved by a multiple of n keeps the verdict
// (r^n mod n^2 depends on r mod n), a negative masked_x / short vectors get the reference's Err / panic — through verify_general with the
// verifier's own challenge bits
static void test_interactive_verifier_on_noncanonical_values() {
  BigInt range = BigInt::sample(RANGE_BITS);
  auto [ek, dk] = test_keypair().keys();
  auto vc = RangeProof::verifier_commit(ek);
  auto [ep, data] = RangeProof::generate_encrypted_pairs(ek, range, SEF);
  BigInt secret_r = BigInt::sample_below(ek.n), secret_x = BigInt::sample_below(range.div_floor(BigInt(3)));
  BigInt cipher_x = Paillier::encrypt_with_chosen_randomness(ek, secret_x, secret_r);
  const Proof z = RangeProof::generate_proof(ek, secret_x, secret_r, vc.e, range, data, SEF);
  ASSERT(RangeProof::verifier_output(ek, vc.e, ep, z, range, cipher_x, SEF).is_ok());
  {
    Proof y = z;
    for (auto& rs : y.responses) {
      if (rs.kind == Response::Open) { rs.r1 = rs.r1 - ek.n; rs.r2 = rs.r2 + ek.n * BigInt::pow2(77); }
      else rs.masked_r = rs.masked_r - ek.n * BigInt(3);
    }
    ASSERT(RangeProof::verifier_output(ek, vc.e, ep, y, range, cipher_x, SEF).is_ok());
  }
  {
    Proof y = z;
    size_t i = 0;
    while (y.responses[i].kind != Response::Mask) i++;
    y.responses[i].masked_x = y.responses[i].masked_x - ek.n;
    ASSERT(RangeProof::verifier_output(ek, vc.e, ep, y, range, cipher_x, SEF).is_err());
  }
  {
    Proof y = z;
    y.responses.resize(SEF - 1);
    bool threw = false;
    try { (void)RangeProof::verifier_output(ek, vc.e, ep, y, range, cipher_x, SEF); } catch (const Panic&) { threw = true; }
    ASSERT(threw);
  }
  {
    // a negative ciphertext: every Mask row's product turns negative and can only equal a zero Enc -> Err as soon as there is a Mask row
    bool any_mask = false;
    for (auto& rs : z.responses) any_mask |= rs.kind == Response::Mask;
    ASSERT(RangeProof::verifier_output(ek, vc.e, ep, z, range, cipher_x - ek.nn, SEF).is_ok() == !any_mask);
  }
}
static void test_range_proof_correct_proof() { interactive_case(true); }
static void test_range_proof_incorrect_proof() { interactive_case(false); }

// CompositeDLogProof::verify on shapes the fixed-width call does not carry: a secret of 900 / 2100 bits makes y = r + e s over-wide
// (the reference does not bound it), a base moved by a multiple of N changes the hash but not the powers
static void test_dlog_verify_on_noncanonical_values() {
  auto [ek, dk] = test_keypair().keys();
  const BigInt N = ek.n;
  BigInt g = BigInt::sample_below(N - BigInt::one());
  while (BigInt::gcd(g, N) != BigInt::one()) g = BigInt::sample_below(N - BigInt::one());
  const BigInt g_inv = BigInt::mod_inv(g, N);
  for (size_t secret_bits : {256, 900, 2100}) {
    const BigInt s = BigInt::sample(secret_bits) + BigInt::pow2(secret_bits - 1);
    const BigInt ni = CompositeDLogProof::mod_pow_wide(g_inv, s, N);                    // ni = g^-s (wi_dlog_proof.rs:128-130)
    for (int shifted = 0; shifted < 2; shifted++) {
      DLogStatement st{N, shifted ? g + N * BigInt::pow2(70) : g, shifted ? ni - N : ni};     // same residues, other integers (negative ni)
      // the prover's side of :53-62 with the statement's own integers in the hash
      const BigInt r = BigInt::sample_below(BigInt::pow2(512));
      CompositeDLogProof pr;
      pr.x = mod_pow(st.g, r, N);
      const BigInt e = detail::compute_digest({&pr.x, &st.g, &st.N, &st.ni});
      pr.y = r + e * s;
      ASSERT(pr.y.bit_length() > secret_bits);
      ASSERT(pr.verify(st).is_ok());
      CompositeDLogProof bad = pr; bad.y = bad.y + BigInt::one();
      ASSERT(bad.verify(st).is_err());
      CompositeDLogProof badx = pr; badx.x = badx.x + N;                                  // x >= N can never equal a residue
      ASSERT(badx.verify(st).is_err());
    }
  }
  DLogStatement not_coprime{N, dk.p, BigInt(5)};
  CompositeDLogProof any{BigInt(3), BigInt::pow2(900)};
  bool threw = false;
  try { (void)any.verify(not_coprime); } catch (const Panic&) { threw = true; }
  ASSERT(threw);
  CompositeDLogProof negy{BigInt(3), BigInt::zero() - BigInt(7)};
  ASSERT(negy.verify(DLogStatement{N, g, BigInt(5)}).is_unsupported());
}

int main() {
  run("serde_json round trip (EncryptedPairs, Proof, NiCorrectKeyProof)", test_serde_round_trip);
  run("multiplication_proof::test_mul_proof", test_mul_proof);
  run("multiplication_proof::test_bad_mul_proof", test_bad_mul_proof, true);
  run("mod_inv batch", test_mod_inv);
  run("correct_message::test_correct_message_zk_proof", test_correct_message_zk_proof);
  run("correct_message::test_bad_message_zk_proof", test_bad_message_zk_proof, true);
  run("range_proof::test_generate_encrypted_pairs", test_generate_encrypted_pairs);
  run("range_proof::test_commit_decommit", test_commit_decommit);
  run("range_proof::test_generate_proof", test_generate_proof);
  run("range_proof::test_range_proof_correct_proof", test_range_proof_correct_proof);
  run("range_proof::test_range_proof_incorrect_proof", test_range_proof_incorrect_proof);
  run("range_proof::interactive verifier on negative / over-wide values and short vectors", test_interactive_verifier_on_noncanonical_values);
  run("range_proof_ni::test_prover", test_prover);
  run("range_proof_ni::test_verifier_for_correct_proof", test_verifier_for_correct_proof);
  run("range_proof_ni::test_verifier_for_incorrect_proof", test_verifier_for_incorrect_proof, true);
  run("range_proof_ni::verify asserts ek/ciphertext", test_verify_asserts_statement, true);
  run("range_proof_ni::batch round trip", test_batch_round_trip);
  run("range_proof_ni::batch survives crafted proofs", test_batch_survives_crafted_proofs);
  run("range_proof_ni::an odd error_factor at the head of a batch", test_odd_error_factor_at_the_head_of_a_batch);
  run("range_proof_ni::over-wide pair on a mask row follows the raw transcript", test_overwide_pair_on_a_mask_row_follows_the_raw_transcript);
  run("correct_key_ni::test_correct_zk_proof_no_salt_str", test_correct_zk_proof_no_salt_str);
  run("correct_key_ni::test_correct_zk_proof_with_salt_str", test_correct_zk_proof_with_salt_str);
  run("correct_key_ni::verify_batch, non-canonical roots, panic and unsupported key", test_correct_key_verify_batch_and_noncanonical_roots);
  run("wi_dlog_proof::verify on over-wide responses and shifted bases", test_dlog_verify_on_noncanonical_values);
  run("wi_dlog_proof::test_correct_dlog_proof", test_correct_dlog_proof);
  run("wi_dlog_proof::test_bad_dlog_proof", test_bad_dlog_proof, true);
  run("wi_dlog_proof::test_bad_dlog_proof_2", test_bad_dlog_proof_2, true);
  run("zero_enc_proof::test_zero_proof", test_zero_proof);
  run("zero_enc_proof::test_one_proof", test_one_proof, true);
  run("correct_ciphertext::test_ciphertext_proof", test_ciphertext_proof);
  run("correct_ciphertext::test_bad_ciphertext_proof", test_bad_ciphertext_proof, true);
  run("correct_key::test_correct_zk_proof", test_correct_zk_proof);
  run("correct_key::test_incorrect_zk_proof", test_incorrect_zk_proof);
  run("correct_key::test_incorrect_zk_proof_2", test_incorrect_zk_proof_2);
  run("verlin_proof::test_verlin_proof", test_verlin_proof);
  run("verlin_proof::test_bad_verlin_proof", test_bad_verlin_proof, true);
  std::printf("%d failure(s)\n", failures);
  return failures ? 1 : 0;
}
