#!/usr/bin/env python3
"""Mints the synthetic key material of bench.py / the full-size tests (committed output:
zk-paillier_amd/bench_keys.json).  Deterministic: primes come from a SHA-256 counter stream.

  * key4096: two 2048-bit primes -> the 4096-bit Paillier modulus of BASELINE.json configs[4]
    (the reference only fixes a 2048-bit keypair, range_proof_ni.rs:141-145).
  * pool1024: 92 primes of 1024 bits; every pair (i < j) is a distinct 2048-bit RSA modulus p_i * p_j,
    4186 of them: the "4096 distinct eks" of SURVEY.md §8(d) config 3 without 8192 prime searches.

Run:  python tools/make_bench_keys.py   (about a minute of pure-Python Miller-Rabin)"""
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Stream:
    def __init__(self, seed):
        self.seed, self.ctr = seed, 0

    def bits(self, nbits):
        out = b""
        while len(out) * 8 < nbits:
            out += hashlib.sha256(self.seed + self.ctr.to_bytes(8, "big")).digest()
            self.ctr += 1
        return int.from_bytes(out, "big") >> (len(out) * 8 - nbits)


SMALL = [p for p in range(3, 2000) if all(p % d for d in range(2, int(p ** 0.5) + 1))]


def is_prime(n, s, rounds=32):
    for p in SMALL:
        if n % p == 0:
            return n == p
    d, r = n - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for _ in range(rounds):
        a = 2 + s.bits(64) % (n - 4)
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(r - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def gen_prime(s, bits):
    while True:
        c = s.bits(bits) | (3 << (bits - 2)) | 1        # top two bits set: a product of two has exactly 2*bits bits
        if is_prime(c, s):
            return c


def main():
    s = Stream(b"zk-paillier_amd bench keys v1")
    p, q = gen_prime(s, 2048), gen_prime(s, 2048)
    pool = []
    while len(pool) < 92:
        c = gen_prime(s, 1024)
        if c not in pool:
            pool.append(c)
    out = {"note": "synthetic bench key material minted by tools/make_bench_keys.py (deterministic)",
           "key4096": {"p": hex(p), "q": hex(q)}, "pool1024": [hex(v) for v in pool]}
    assert (p * q).bit_length() == 4096 and all((a * b).bit_length() == 2048 for a in pool[:3] for b in pool[3:6])
    with open(os.path.join(ROOT, "zk-paillier_amd", "bench_keys.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote bench_keys.json:", len(pool), "pool primes")


if __name__ == "__main__":
    main()
