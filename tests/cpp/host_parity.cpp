// Parity driver for the host compositions of zk-paillier_amd/host/zkproofs.hpp (the product's C++ mirror of the
// reference API, every modexp on the GPU through libzkp_hip.so).  tests/test_gpu_host_parity.py feeds it seeded
// inputs as hex and compares what comes back with the oracle (C/GMP and py_model).  One command per line on stdin:
//   ck_challenge <n> <K> <s_0..s_K-1> <r_0..r_K-1>     -> sn_0.. e z_0.. s_digest           (correct_key.rs:64-102)
//   ck_prove <p> <q> <e> <K> <sn_0..> <z_0..>          -> "ok <s_digest>" | "err <code 1..4>" (correct_key.rs:104-162)
//   ck_ni_proof <p> <q>                                 -> sigma_0 .. sigma_10                 (correct_key_ni.rs:42-71)
// Needs a gfx950 GPU.
#include <cstdio>
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
