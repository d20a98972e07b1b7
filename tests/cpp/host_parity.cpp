// Parity driver for the host compositions of zk-paillier_amd/host/zkproofs.hpp (the product's C++ mirror of the
// reference API, every modexp on the GPU through libzkp_hip.so).  tests/test_gpu_host_parity.py feeds it seeded
// inputs as hex and compares what comes back with the oracle (C/GMP and py_model).  One command per line on stdin:
//   ck_challenge <n> <K> <s_0..s_K-1> <r_0..r_K-1>     -> sn_0.. e z_0.. s_digest           (correct_key.rs:64-102)
//   ck_prove <p> <q> <e> <K> <sn_0..> <z_0..>          -> "ok <s_digest>" | "err <code 1..4>" (correct_key.rs:104-162)
//   ck_ni_proof <p> <q>                                 -> sigma_0 .. sigma_10                 (correct_key_ni.rs:42-71)
//   range_ni_verify_docs <file>                         -> one word per document: ok | err | panic | unsupported | serde
//        (file: one serde_json RangeProofNi per line, every integer a decimal string, '-' allowed: serde_json::range_proof_ni_from_str,
//         then RangeProofNi::verify_batch per run of documents under one key — canonical proofs on the fixed-width GPU path, the
//         others through verify_general)
// Needs a gfx950 GPU.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

#include "../../zk-paillier_amd/host/zkproofs.hpp"

using namespace zkproofs;

static BigInt from_hex(const std::string& h) {
  std::vector<uint8_t> b;
  std::string s = (h.size() % 2) ? "0" + h : h;
  for (size_t i = 0; i < s.size(); i += 2) b.push_back((uint8_t)std::stoul(s.substr(i, 2), nullptr, 16));
  return BigInt::from_bytes(b);
}
static std::string hex(const BigInt& v) { return v.is_zero() ? "0" : v.to_hex(); }

int main() {
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream in(line);
    std::string cmd, tok;
    if (!(in >> cmd)) continue;
    auto next = [&]() { in >> tok; return from_hex(tok); };
    try {
      if (cmd == "ck_challenge") {
        BigInt n = next();
        size_t K; in >> K;
        std::vector<BigInt> s, r;
        for (size_t i = 0; i < K; i++) s.push_back(next());
        for (size_t i = 0; i < K; i++) r.push_back(next());
        auto [ch, va] = CorrectKey::challenge_with(EncryptionKey{n, n * n}, s, r);
        for (auto& v : ch.sn) std::printf("%s ", hex(v).c_str());
        std::printf("%s ", hex(ch.e).c_str());
        for (auto& v : ch.z) std::printf("%s ", hex(v).c_str());
        std::printf("%s\n", hex(va.s_digest).c_str());
      } else if (cmd == "ck_prove") {
        BigInt p = next(), q = next(), e = next();
        size_t K; in >> K;
        Challenge ch; ch.e = e;
        for (size_t i = 0; i < K; i++) ch.sn.push_back(next());
        for (size_t i = 0; i < K; i++) ch.z.push_back(next());
        auto res = CorrectKey::prove(DecryptionKey{p, q}, ch);
        if (res.is_ok()) std::printf("ok %s\n", hex(res.unwrap().s_digest).c_str());
        else std::printf("err %d\n", 1 + (int)res.err);
      } else if (cmd == "ck_ni_proof") {
        BigInt p = next(), q = next();
        NiCorrectKeyProof pr = NiCorrectKeyProof::proof(DecryptionKey{p, q});
        for (size_t i = 0; i < pr.sigma_vec.size(); i++) std::printf("%s%s", hex(pr.sigma_vec[i]).c_str(), i + 1 < pr.sigma_vec.size() ? " " : "\n");
      } else if (cmd == "range_ni_verify_docs") {
        std::string path; in >> path;
        std::ifstream f(path);
        std::vector<RangeProofNi> proofs; std::vector<int> parsed;
        std::string doc;
        while (std::getline(f, doc)) {
          if (doc.empty()) continue;
          try { proofs.push_back(serde_json::range_proof_ni_from_str(doc)); parsed.push_back(1); }
          catch (const std::runtime_error&) { proofs.emplace_back(); parsed.push_back(0); }
        }
        std::vector<std::string> words(proofs.size(), "serde");
        for (size_t lo = 0; lo < proofs.size();) {
          if (!parsed[lo]) { lo++; continue; }
          size_t hi = lo;
          std::vector<const RangeProofNi*> run;
          while (hi < proofs.size() && parsed[hi] && proofs[hi].ek == proofs[lo].ek) run.push_back(&proofs[hi++]);
          auto res = RangeProofNi::verify_batch(proofs[lo].ek, run);
          for (size_t k = 0; k < res.size(); k++)
            words[lo + k] = res[k].is_unsupported() ? "unsupported" : res[k].would_panic() ? "panic" : res[k].is_ok() ? "ok" : "err";
          lo = hi;
        }
        for (size_t i = 0; i < words.size(); i++) std::printf("%s%s", words[i].c_str(), i + 1 < words.size() ? " " : "\n");
      } else {
        std::printf("unknown command\n");
      }
    } catch (const std::exception& ex) {
      std::printf("exception %s\n", ex.what());
    }
    std::fflush(stdout);
  }
  return 0;
}
