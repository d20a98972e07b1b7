"""ctypes binding of oracle/liboracle.so (the TEST-ONLY C/GMP oracle).  Lives under tests/
because only tests (and smoke()/bench cpu_baseline) may touch the oracle."""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

capi = importlib.import_module("zk-paillier_amd.capi")
RangeNiProofs, RangeNiWitness = capi.RangeNiProofs, capi.RangeNiWitness


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self):
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(ORACLE_DIR, "zkp_oracle.c")):
            build()
        self.lib = C.CDLL(LIB)
        self.lib.oracle_get_max_threads.restype = C.c_int

    def set_threads(self, n):
        self.lib.oracle_set_threads(C.c_int(n))

    def max_threads(self):
        return self.lib.oracle_get_max_threads()

    def primorial_decimal(self) -> str:
        """the product of the primes below 6370 as this oracle computes it (correct_key_ni.rs:26)"""
        buf = C.create_string_buffer(4096)
        self.lib.oracle_primorial_decimal.restype = C.c_int64
        n = self.lib.oracle_primorial_decimal(buf, C.c_uint64(4096))
        assert n > 0
        return buf.value.decode()

    def sha256(self, data: bytes) -> bytes:
        out = (C.c_uint8 * 32)()
        self.lib.oracle_sha256(data, C.c_uint64(len(data)), out)
        return bytes(out)

    def range_ni_verify_decimal(self, proof):
        """RangeProofNi::verify_self on signed integers of any size (oracle_range_ni_verify_decimal: mpz over decimal strings).
        proof: dict(n, range, ciphertext, error_factor, c1, c2, responses) of python ints -> ("ok" | "err" | "panic", challenge bytes)"""
        def arr(vals):
            a = (C.c_char_p * max(1, len(vals)))()
            for i, v in enumerate(vals):
                a[i] = str(v).encode()
            return a
        R = proof["responses"]
        kind = (C.c_uint8 * max(1, len(R)))(*[0 if r[0] == "open" else 1 for r in R])
        j = (C.c_uint8 * max(1, len(R)))(*[0 if r[0] == "open" else r[1] for r in R])
        f1 = arr([r[1] if r[0] == "open" else r[2] for r in R]); f2 = arr([r[2] if r[0] == "open" else r[3] for r in R])
        f3 = arr([r[3] if r[0] == "open" else 0 for r in R]); f4 = arr([r[4] if r[0] == "open" else 0 for r in R])
        e = (C.c_uint8 * 32)(); el = C.c_uint8(0)
        self.lib.oracle_range_ni_verify_decimal.restype = C.c_int32
        v = self.lib.oracle_range_ni_verify_decimal(str(proof["n"]).encode(), str(proof["range"]).encode(), str(proof["ciphertext"]).encode(),
                                                    C.c_uint32(proof["error_factor"]), arr(proof["c1"]), C.c_uint32(len(proof["c1"])),
                                                    arr(proof["c2"]), C.c_uint32(len(proof["c2"])), kind, j, f1, f2, f3, f4, C.c_uint32(len(R)), e, C.byref(el))
        assert v in (0, 1, 2), v
        return {0: "err", 1: "ok", 2: "panic"}[v], bytes(e)[:el.value]

    def enc_decimal(self, n, m, r):
        out = C.create_string_buffer(4 * len(str(n)) + 8)
        self.lib.oracle_enc_decimal.restype = C.c_int32
        assert self.lib.oracle_enc_decimal(str(n).encode(), str(m).encode(), str(r).encode(), out, C.c_uint64(len(out))) == 0
        return int(out.value)

    def modexp(self, mod_bits, exp_bits, base, exp, exp_stride, mod, mod_stride):
        count = base.shape[0]
        out = np.zeros_like(base)
        self.lib.oracle_modexp_batch(C.c_uint32(mod_bits), C.c_uint32(exp_bits), C.c_uint64(count), p(base), p(exp),
                                     C.c_uint64(exp_stride), p(mod), C.c_uint64(mod_stride), p(out))
        return out

    def modmul(self, mod_bits, a, b, mod, mod_stride):
        out = np.zeros_like(a)
        self.lib.oracle_modmul_batch(C.c_uint32(mod_bits), C.c_uint64(a.shape[0]), p(a), p(b), p(mod),
                                     C.c_uint64(mod_stride), p(out))
        return out

    def paillier_enc(self, n_bits, n, n_stride, m, r):
        count = m.shape[0]
        out = np.zeros((count, 2 * n_bits // 32), dtype=np.uint32)
        self.lib.oracle_paillier_enc_batch(C.c_uint32(n_bits), C.c_uint64(count), p(n), C.c_uint64(n_stride), p(m), p(r), p(out))
        return out

    def paillier_enc_check(self, n_bits, n, n_stride, m, r, mulc_a, mulc_b, expected):
        count = m.shape[0]
        ok = np.zeros(count, dtype=np.uint8)
        self.lib.oracle_paillier_enc_check_batch(C.c_uint32(n_bits), C.c_uint64(count), p(n), C.c_uint64(n_stride), p(m), p(r),
                                                 p(mulc_a), p(mulc_b), p(expected), p(ok))
        return ok

    def range_ni_prove(self, proofs: RangeNiProofs, wit: RangeNiWitness, out_e, out_e_len, out_status):
        return self.lib.oracle_range_ni_prove_batch(C.byref(proofs), C.byref(wit), p(out_e), p(out_e_len), p(out_status))

    def range_ni_verify(self, proofs: RangeNiProofs, out_verdict):
        return self.lib.oracle_range_ni_verify_batch(C.byref(proofs), p(out_verdict))

    def correct_key_ni_verify(self, n_bits, n, sigma, salt: bytes):
        batch = n.shape[0]
        out = np.zeros(batch, dtype=np.uint8)
        self.lib.oracle_correct_key_ni_verify_batch(C.c_uint32(n_bits), C.c_uint64(batch), p(n), p(sigma), salt,
                                                    C.c_uint32(len(salt)), p(out))
        return out

    def correct_key_ni_prove(self, n_bits, pp, qq, salt: bytes):
        kw = n_bits // 32
        n = np.zeros(kw, dtype=np.uint32)
        sigma = np.zeros((11, kw), dtype=np.uint32)
        rc = self.lib.oracle_correct_key_ni_prove(C.c_uint32(n_bits), p(pp), p(qq), salt, C.c_uint32(len(salt)), p(n), p(sigma))
        assert rc == 0
        return n, sigma

    def correct_key_challenge(self, n_bits, n, s, r):
        K, kw = s.shape[0], n_bits // 32
        sn = np.zeros((K, kw), np.uint32); z = np.zeros((K, kw), np.uint32); e = np.zeros(8, np.uint32); sd = np.zeros(8, np.uint32)
        self.lib.oracle_correct_key_challenge(C.c_uint32(n_bits), C.c_uint32(K), p(n), p(s), p(r), p(sn), p(e), p(z), p(sd))
        return sn, e, z, sd

    def correct_key_prove(self, n_bits, pp, qq, sn, e, z):
        sd = np.zeros(8, np.uint32)
        rc = self.lib.oracle_correct_key_prove(C.c_uint32(n_bits), C.c_uint32(sn.shape[0]), p(pp), p(qq), p(sn), p(e), C.c_uint32(e.shape[0]), p(z), p(sd))
        return rc, sd

    def correct_key_verify(self, a, b):
        return self.lib.oracle_correct_key_verify(p(a), p(b))

    def correct_key_rho(self, n_bits, n, salt: bytes):
        rho = np.zeros((11, n_bits // 32), dtype=np.uint32)
        self.lib.oracle_correct_key_rho(C.c_uint32(n_bits), p(n), salt, C.c_uint32(len(salt)), p(rho))
        return rho

    def dlog_prove(self, n_bits, y_bits, N, g, ni, secret, r):
        batch = N.shape[0]
        x = np.zeros_like(N)
        y = np.zeros((batch, y_bits // 32), dtype=np.uint32)
        self.lib.oracle_dlog_prove_batch(C.c_uint32(n_bits), C.c_uint32(y_bits), C.c_uint64(batch), p(N), p(g), p(ni),
                                         p(secret), p(r), p(x), p(y))
        return x, y

    def dlog_verify(self, n_bits, y_bits, N, g, ni, x, y):
        batch = N.shape[0]
        out = np.zeros(batch, dtype=np.uint8)
        self.lib.oracle_dlog_verify_batch(C.c_uint32(n_bits), C.c_uint32(y_bits), C.c_uint64(batch), p(N), p(g), p(ni),
                                          p(x), p(y), p(out))
        return out

    def fs_challenge(self, n_bits, ef, n, c1, c2):
        e = np.zeros(32, dtype=np.uint8)
        elen = C.c_uint8()
        self.lib.oracle_fs_challenge(C.c_uint32(n_bits), C.c_uint32(ef), p(n), p(c1), p(c2), p(e), C.byref(elen))
        return bytes(e[:elen.value])

    # ---- ZeroProof / CiphertextProof
    def zero_proof_prove(self, n_bits, n, n_stride, c, r, r_prime):
        B = c.shape[0]
        z = np.zeros_like(c); a = np.zeros_like(c)
        self.lib.oracle_zero_proof_prove_batch(C.c_uint32(n_bits), C.c_uint64(B), p(n), C.c_uint64(n_stride), p(c), p(r), p(r_prime), p(z), p(a))
        return z, a

    def zero_proof_verify(self, n_bits, n, n_stride, c, z, a):
        B = c.shape[0]
        v = np.zeros(B, np.uint8)
        self.lib.oracle_zero_proof_verify_batch(C.c_uint32(n_bits), C.c_uint64(B), p(n), C.c_uint64(n_stride), p(c), p(z), p(a), p(v))
        return v

    def ciphertext_proof_prove(self, n_bits, n, n_stride, c, x, r, x_prime, r_prime):
        B = c.shape[0]
        z1 = np.zeros((B, n_bits // 32 + 16), np.uint32); z2 = np.zeros_like(c); cp = np.zeros_like(c)
        self.lib.oracle_ciphertext_proof_prove_batch(C.c_uint32(n_bits), C.c_uint64(B), p(n), C.c_uint64(n_stride), p(c), p(x), p(r), p(x_prime),
                                                     p(r_prime), p(z1), p(z2), p(cp))
        return z1, z2, cp

    def ciphertext_proof_verify(self, n_bits, n, n_stride, c, z1, z2, c_prime):
        B = c.shape[0]
        v = np.zeros(B, np.uint8)
        self.lib.oracle_ciphertext_proof_verify_batch(C.c_uint32(n_bits), C.c_uint64(B), p(n), C.c_uint64(n_stride), p(c), p(z1), p(z2), p(c_prime), p(v))
        return v

    # ---- VerlinProof
    def verlin_proof_prove(self, n_bits, n, n_stride, c, c_prime, phi_x, witness, nonces):
        B = c.shape[0]; zw = n_bits // 32 + 16
        phi_a = np.zeros_like(c); r_z = np.zeros_like(c)
        z = [np.zeros((B, zw), np.uint32) for _ in range(3)]
        self.lib.oracle_verlin_proof_prove_batch(C.c_uint32(n_bits), C.c_uint64(B), p(n), C.c_uint64(n_stride), p(c), p(c_prime), p(phi_x),
                                                 *[p(a) for a in witness], *[p(a) for a in nonces], p(phi_a), p(z[0]), p(z[1]), p(z[2]), p(r_z))
        return phi_a, z[0], z[1], z[2], r_z

    def verlin_proof_verify(self, n_bits, n, n_stride, c, c_prime, phi_x, phi_a, z, zp, zpp, r_z):
        B = c.shape[0]
        v = np.zeros(B, np.uint8)
        self.lib.oracle_verlin_proof_verify_batch(C.c_uint32(n_bits), C.c_uint64(B), p(n), C.c_uint64(n_stride), p(c), p(c_prime), p(phi_x), p(phi_a),
                                                  p(z), p(zp), p(zpp), p(r_z), p(v))
        return v

    # ---- interactive RangeProof building blocks
    def range_generate_encrypted_pairs(self, proofs, wit):
        return self.lib.oracle_range_generate_encrypted_pairs_batch(C.byref(proofs), C.byref(wit))

    def range_generate_proof(self, proofs, wit, e, e_len, out_status):
        return self.lib.oracle_range_generate_proof_batch(C.byref(proofs), C.byref(wit), p(e), p(e_len), p(out_status))

    def range_verifier_output(self, proofs, e, e_len, out_verdict):
        return self.lib.oracle_range_verifier_output_batch(C.byref(proofs), p(e), p(e_len), p(out_verdict))

    # ---- mod_inv, MulProof, CorrectMessageProof
    def modinv(self, mod_bits, a, mod, mod_stride):
        out = np.zeros_like(a); st = np.full(a.shape[0], 9, np.uint8)
        self.lib.oracle_modinv_batch(C.c_uint32(mod_bits), C.c_uint64(a.shape[0]), p(a), p(mod), C.c_uint64(mod_stride), p(out), p(st))
        return out, st

    def mul_proof_prove(self, n_bits, n, n_stride, e_a, e_b, e_c, a, b, r_a, r_b, r_c, d, r_d):
        B = e_a.shape[0]; kw = n_bits // 32
        f = np.zeros((B, kw), np.uint32); z1, z2, e_d, e_db = (np.zeros((B, 2 * kw), np.uint32) for _ in range(4))
        st = np.full(B, 9, np.uint8)
        self.lib.oracle_mul_proof_prove_batch(C.c_uint32(n_bits), C.c_uint64(B), p(n), C.c_uint64(n_stride), p(e_a), p(e_b), p(e_c), p(a), p(b),
                                              p(r_a), p(r_b), p(r_c), p(d), p(r_d), p(f), p(z1), p(z2), p(e_d), p(e_db), p(st))
        return f, z1, z2, e_d, e_db, st

    def mul_proof_verify(self, n_bits, n, n_stride, e_a, e_b, e_c, f, z1, z2, e_d, e_db):
        B = e_a.shape[0]
        v = np.full(B, 9, np.uint8)
        self.lib.oracle_mul_proof_verify_batch(C.c_uint32(n_bits), C.c_uint64(B), p(n), C.c_uint64(n_stride), p(e_a), p(e_b), p(e_c), p(f), p(z1),
                                               p(z2), p(e_d), p(e_db), p(v))
        return v

    def correct_message_prove(self, n_bits, K, n, n_stride, valid, message, r, e_sim, z_sim, w):
        B = message.shape[0]; kw = n_bits // 32
        ct = np.zeros((B, 2 * kw), np.uint32); e_vec = np.zeros((B, K, 8), np.uint32); z_vec = np.zeros((B, K, kw), np.uint32)
        a_vec = np.zeros((B, K, 2 * kw), np.uint32); st = np.full(B, 9, np.uint8)
        self.lib.oracle_correct_message_prove_batch(C.c_uint32(n_bits), C.c_uint64(B), C.c_uint32(K), p(n), C.c_uint64(n_stride), p(valid), p(message),
                                                    p(r), p(e_sim), p(z_sim), p(w), p(ct), p(e_vec), p(z_vec), p(a_vec), p(st))
        return ct, e_vec, z_vec, a_vec, st

    def correct_message_verify(self, n_bits, K, n, n_stride, valid, ct, e_vec, z_vec, a_vec):
        B = ct.shape[0]
        v = np.full(B, 9, np.uint8)
        self.lib.oracle_correct_message_verify_batch(C.c_uint32(n_bits), C.c_uint64(B), C.c_uint32(K), p(n), C.c_uint64(n_stride), p(valid), p(ct),
                                                     p(e_vec), p(z_vec), p(a_vec), p(v))
        return v

    # ---- wire format: decimal strings (GMP mpz_set_str / mpz_get_str)
    def decimal_to_limbs(self, text: bytes, items, dst, out_status):
        buf = (C.c_char * len(text)).from_buffer_copy(text)
        self.lib.oracle_decimal_to_limbs_batch(C.cast(buf, C.c_void_p), C.cast(items, C.c_void_p), C.c_uint64(len(items)), p(dst), p(out_status))

    def limbs_to_decimal(self, src, pitch):
        count, words = src.shape
        out = np.zeros((count, pitch), np.uint8); ln = np.zeros(count, np.uint32)
        self.lib.oracle_limbs_to_decimal_batch(p(src), C.c_uint64(words), C.c_uint32(words), C.c_uint64(count), p(out), C.c_uint32(pitch), p(ln))
        return [bytes(out[i, pitch - ln[i]:]) for i in range(count)]
