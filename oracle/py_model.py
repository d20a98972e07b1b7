"""TEST INFRASTRUCTURE — second, independent CPU oracle (pure Python: int pow + hashlib).

This file restates, in plain Python, the hot path of ZenGo-X/zk-paillier so that
  * the C/GMP oracle (oracle/zkp_oracle.c) can be cross-checked against an independent
    implementation, and
  * golden vectors can be minted (tests/golden/make_golden.py).
It is NOT part of the product: only tests/, tests/golden/make_golden.py,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/.

PARITY UNPINNED: the reference's own tests contain no known-answer vectors (every test
draws fresh OS randomness) and the reference (Rust; curv-kzen 0.10 + rust-gmp-kzen +
kzen-paillier 0.4.3, none vendored, no rustc/cargo in this image) cannot be built here.
Behaviours of those un-vendored crates are restated from their published algorithms and
are marked [upstream] below; accept/reject behaviour is anchored on the reference's tests.

Reference citations are relative to /root/reference.
"""
import hashlib

# ----------------------------------------------------------------------------- bytes / hash

def to_bytes(x: int) -> bytes:
    """[upstream curv BigInt::to_bytes over rust-gmp: mpz_export, (sizeinbase(x,2)+7)/8
    bytes, big-endian] — zero encodes as ONE 0x00 byte (sizeinbase(0,2) == 1)."""
    assert x >= 0
    if x == 0:
        return b"\x00"
    return x.to_bytes((x.bit_length() + 7) // 8, "big")


def from_bytes(b: bytes) -> int:
    """[upstream BigInt::from_bytes: big-endian unsigned]."""
    return int.from_bytes(b, "big")


def compute_digest(items) -> int:
    """src/zkproofs/utils.rs:9-22 — SHA-256 over the plain concatenation of to_bytes()."""
    h = hashlib.sha256()
    for v in items:
        h.update(to_bytes(v))
    return from_bytes(h.digest())


def challenge_bit(e_bytes: bytes, i: int) -> int:
    """bit_vec::BitVec::from_bytes (range_proof.rs:221,267): MSB-first within each byte.
    Index past the end panics in the reference -> IndexError here."""
    return (e_bytes[i // 8] >> (7 - (i % 8))) & 1


# ----------------------------------------------------------------------------- DRBG (repo-defined)

class Drbg:
    """SHA-256 counter-mode generator used by every harness in this repo (Python, C oracle,
    tests) so that {seed} reproduces identical inputs everywhere.
    block_i = SHA256(seed || be64(i)); stream = block_0 || block_1 || ...
    below(n)  = int(next ceil(bitlen(n)/8)+8 bytes, big-endian) mod n
    (sampling distribution is irrelevant to parity: the reference uses the OS RNG,
    SURVEY.md N7; the boundary takes randomness as an input)."""

    def __init__(self, seed: bytes):
        self.seed = seed
        self.ctr = 0
        self.buf = b""

    def bytes(self, n: int) -> bytes:
        while len(self.buf) < n:
            self.buf += hashlib.sha256(self.seed + self.ctr.to_bytes(8, "big")).digest()
            self.ctr += 1
        out, self.buf = self.buf[:n], self.buf[n:]
        return out

    def bits(self, nbits: int) -> int:
        """uniform in [0, 2^nbits): top bits of the first byte are cleared."""
        nb = (nbits + 7) // 8
        v = from_bytes(self.bytes(nb))
        return v & ((1 << nbits) - 1)

    def below(self, n: int) -> int:
        nb = (n.bit_length() + 7) // 8 + 8
        return from_bytes(self.bytes(nb)) % n

    def range(self, lo: int, hi: int) -> int:
        return lo + self.below(hi - lo)


# ----------------------------------------------------------------------------- Paillier

def enc(n: int, m: int, r: int) -> int:
    """[upstream kzen-paillier 0.4.3 EncryptWithChosenRandomness]
    rn = r^n mod n^2 ; gm = (m*n + 1) mod n^2 ; c = gm*rn mod n^2.
    Call sites: range_proof.rs:165-169,179-183,280-291,330-334."""
    nn = n * n
    rn = pow(r, n, nn)
    gm = (m * n + 1) % nn
    return (gm * rn) % nn


# ----------------------------------------------------------------------------- range proof

SECURITY_PARAMETER = 128  # range_proof_ni.rs:23


def generate_encrypted_pairs(n, w1, w2, r1, r2):
    """range_proof.rs:161-187 with the sampled (w1,w2,r1,r2) (after the coin-flip swap,
    :144-149) injected instead of drawn from the OS RNG."""
    c1 = [enc(n, w, r) for w, r in zip(w1, r1)]
    c2 = [enc(n, w, r) for w, r in zip(w2, r2)]
    return c1, c2


def fs_challenge(n, c1, c2) -> bytes:
    """range_proof_ni.rs:58-61 / 89-92: e = to_bytes(compute_digest(n, c1.., c2..)).
    Leading zero BYTES of the digest are dropped by the BigInt round trip (N2)."""
    return to_bytes(compute_digest([n] + list(c1) + list(c2)))


def generate_proof(n, x, r, e_bytes, rng_q, w1, w2, r1, r2, error_factor):
    """range_proof.rs:210-252.  Response = ('open', w1,r1,w2,r2) | ('mask', j, masked_x, masked_r)."""
    third = rng_q // 3          # div_floor, :219
    two_thirds = 2 * third      # :220
    out = []
    for i in range(error_factor):
        ei = challenge_bit(e_bytes, i)
        if not ei:
            out.append(("open", w1[i], r1[i], w2[i], r2[i]))          # :226-232
        elif third < x + w1[i] < two_thirds:                           # :233-234 (strict)
            out.append(("mask", 1, x + w1[i], (r * r1[i]) % n))        # :236-240
        else:
            out.append(("mask", 2, x + w2[i], (r * r2[i]) % n))        # :242-246
    return out


def verifier_output(n, e_bytes, c1, c2, responses, rng_q, cipher_x, error_factor) -> bool:
    """range_proof.rs:254-355.  No early exit: every row is evaluated (collect then all)."""
    nn = n * n
    third = rng_q // 3
    two_thirds = 2 * third
    oks = []
    for i in range(error_factor):
        ei = challenge_bit(e_bytes, i)          # may raise IndexError (= reference panic)
        resp = responses[i]                     # may raise IndexError (= reference panic)
        if (not ei) and resp[0] == "open":
            _, w1, r1, w2, r2 = resp
            res = True
            if enc(n, w1, r1) != c1[i]:
                res = False
            if enc(n, w2, r2) != c2[i]:
                res = False
            flag = (w2 < third and third < w1 < two_thirds) or (w1 < third and third < w2 < two_thirds)  # :300-305
            if not flag:
                res = False
            oks.append(res)
        elif ei and resp[0] == "mask":
            _, j, mx, mr = resp
            c = (c1[i] * cipher_x) % nn if j == 1 else (c2[i] * cipher_x) % nn   # :324-328 (any j != 1 -> c2)
            res = True
            if c != enc(n, mx, mr):
                res = False
            if mx < third or mx > two_thirds:   # :338 (inclusive bounds accepted)
                res = False
            oks.append(res)
        else:
            oks.append(False)                   # :345
    return all(oks)


def range_ni_prove(n, rng_q, ciphertext, x, r, w1, w2, r1, r2):
    """range_proof_ni.rs:47-82 with injected randomness."""
    c1, c2 = generate_encrypted_pairs(n, w1, w2, r1, r2)
    e = fs_challenge(n, c1, c2)
    responses = generate_proof(n, x, r, e, rng_q, w1, w2, r1, r2, SECURITY_PARAMETER)
    return dict(n=n, range=rng_q, ciphertext=ciphertext, c1=c1, c2=c2, e=e,
                responses=responses, error_factor=SECURITY_PARAMETER)


def range_ni_verify(proof, n, ciphertext) -> bool:
    """range_proof_ni.rs:84-107 (the two assert_eq! panics are the caller's problem here)."""
    assert n == proof["n"] and ciphertext == proof["ciphertext"]
    e = fs_challenge(n, proof["c1"], proof["c2"])
    return verifier_output(n, e, proof["c1"], proof["c2"], proof["responses"], proof["range"],
                           proof["ciphertext"], proof["error_factor"])


# ----------------------------------------------------------------------------- signed, arbitrary-precision values
# A deserialised proof holds attacker-chosen BigInts of any size and either sign (decimal strings with a leading '-',
# serialize.rs:8-31).  The functions above assume the honest domain; these restate the reference's operators on ANY integers:
#   Rust `%` on BigInt            -> truncated remainder, sign of the dividend          [upstream, recalled]  (tdiv_r)
#   BigInt::mod_pow               -> mpz_powm: result in [0, m) for a negative base    [upstream, recalled]  (Python pow agrees)
#   BigInt::to_bytes              -> the magnitude                                      [upstream, recalled]

def tdiv_r(a: int, m: int) -> int:
    """a % m as Rust's BigInt does it (mpz_tdiv_r): |a| mod |m| with the sign of a"""
    r = abs(a) % abs(m)
    return -r if a < 0 else r


def enc_signed(n: int, m: int, r: int) -> int:
    """kzen-paillier's Enc on any integers m, r (n > 0): rn = mod_pow(r, n, nn); gm = (m*n + 1) % nn; c = gm*rn % nn"""
    nn = n * n
    rn = pow(r, n, nn)
    gm = tdiv_r(m * n + 1, nn)
    return tdiv_r(gm * rn, nn)


def to_bytes_magnitude(x: int) -> bytes:
    return to_bytes(abs(x))


def fs_challenge_signed(n, c1, c2) -> bytes:
    h = hashlib.sha256()
    for v in [n] + list(c1) + list(c2):
        h.update(to_bytes_magnitude(v))
    return to_bytes(from_bytes(h.digest()))


def range_ni_verify_signed(n, rng_q, cipher_x, error_factor, c1, c2, responses):
    """RangeProofNi::verify_self (range_proof_ni.rs:109-128 -> range_proof.rs:254-355) on arbitrary integers.
    Returns True / False, or raises IndexError where the reference panics."""
    assert n > 0
    nn = n * n
    e = fs_challenge_signed(n, c1, c2)
    third = rng_q // 3              # div_floor (Python // is floored)
    two_thirds = 2 * third
    oks = []
    for i in range(error_factor):
        if i >= 8 * len(e):
            raise IndexError("bits_of_e")
        ei = challenge_bit(e, i)
        resp = responses[i]
        if (not ei) and resp[0] == "open":
            _, w1, r1, w2, r2 = resp
            res = enc_signed(n, w1, r1) == c1[i]
            if enc_signed(n, w2, r2) != c2[i]:
                res = False
            if not ((w2 < third and third < w1 < two_thirds) or (w1 < third and third < w2 < two_thirds)):
                res = False
            oks.append(res)
        elif ei and resp[0] == "mask":
            _, j, mx, mr = resp
            c = tdiv_r((c1[i] if j == 1 else c2[i]) * cipher_x, nn)
            res = c == enc_signed(n, mx, mr)
            if mx < third or mx > two_thirds:
                res = False
            oks.append(res)
        else:
            oks.append(False)
    return all(oks)


def sample_range_inputs(drbg: Drbg, n: int, rng_q: int, ef: int = SECURITY_PARAMETER):
    """Deterministic stand-in for range_proof.rs:133-159 (sample_range / swap / sample_below)."""
    third = rng_q // 3
    w1 = [drbg.range(third, 2 * third) for _ in range(ef)]
    w2 = [w - third for w in w1]
    for i in range(ef):
        if drbg.bytes(1)[0] & 1:
            w1[i], w2[i] = w2[i], w1[i]
    r1 = [drbg.below(n) for _ in range(ef)]
    r2 = [drbg.below(n) for _ in range(ef)]
    return w1, w2, r1, r2


# ----------------------------------------------------------------------------- correct key (NI)

SALT_STRING = bytes([75, 90, 101, 110])   # correct_key_ni.rs:28
M2 = 11                                   # :29
DIGEST_SIZE = 256                         # :30
ALPHA = 6370                              # primes below ALPHA make up P (:26)


def primes_below(a):
    s = bytearray([1]) * a
    s[0:2] = b"\x00\x00"
    for i in range(2, int(a ** 0.5) + 1):
        if s[i]:
            s[i * i::i] = bytearray(len(s[i * i::i]))
    return [i for i in range(a) if s[i]]


_P = None


def primorial():
    global _P
    if _P is None:
        p = 1
        for q in primes_below(ALPHA):
            p *= q
        _P = p
    return _P


def mask_generation(out_length: int, seed: int) -> int:
    """correct_key_ni.rs:105-117."""
    msklen = out_length // DIGEST_SIZE + 1
    acc = 0
    for j in range(msklen):
        acc += compute_digest([seed, j]) << (j * DIGEST_SIZE)
    return acc


def correct_key_rho(n: int, salt: bytes):
    """correct_key_ni.rs:74-86."""
    key_length = n.bit_length()
    salt_bn = compute_digest([from_bytes(salt)])
    return [mask_generation(key_length, compute_digest([n, salt_bn, i])) % n for i in range(M2)]


def correct_key_proof(p: int, q: int, salt: bytes):
    """correct_key_ni.rs:42-71; extract_nroot [upstream]: rho^(n^-1 mod phi) mod n."""
    n = p * q
    phi = (p - 1) * (q - 1)
    d = pow(n, -1, phi)
    return [pow(rho, d, n) for rho in correct_key_rho(n, salt)]


def correct_key_verify(sigma, n: int, salt: bytes) -> bool:
    """correct_key_ni.rs:73-100."""
    import math
    rho = correct_key_rho(n, salt)
    gcd_test = math.gcd(primorial(), n)
    derived = [pow(sigma[i], n, n) for i in range(M2)]   # index panic if short
    return rho == derived and gcd_test == 1


# ----------------------------------------------------------------------------- interactive CorrectKey

CK_ERRORS = {1: "SniNotCoprimeWithN", 2: "ZiNotCoprimeWithN", 3: "RniNotCoprimeWithN", 4: "EWasntComputedCorrectly"}


def correct_key_challenge(n: int, s, r):
    """correct_key.rs:64-102 with the sampled s_i, r_i (:67-70, :80-83) injected.  -> (sn, e, z, s_digest)"""
    sn = [pow(si, n, n) for si in s]
    rn = [pow(ri, n, n) for ri in r]
    e = compute_digest([n] + sn + rn)
    z = [(ri * pow(si, e, n)) % n for ri, si in zip(r, s)]
    return sn, e, z, compute_digest(s)


def correct_key_prove(p: int, q: int, sn, e: int, z):
    """correct_key.rs:104-162.  -> (0, s_digest) or (error code of CK_ERRORS, None)"""
    import math
    n = q * p
    if any(math.gcd(n, v) != 1 for v in sn):
        return 1, None
    if any(math.gcd(n, v) != 1 for v in z):
        return 2, None
    phi = (q - 1) * (p - 1)
    phimine = phi - (e % phi)
    rn = [(pow(zi, n, n) * pow(sni, phimine, n)) % n for zi, sni in zip(z, sn)]
    if any(math.gcd(n, v) != 1 for v in rn):
        return 3, None
    if e != compute_digest([n] + list(sn) + rn):
        return 4, None
    d = pow(n, -1, phi)                                   # [upstream] extract_nroot
    return 0, compute_digest([pow(v, d, n) for v in sn])


def correct_key_verify_interactive(proof_digest: int, aid_digest: int) -> bool:
    """correct_key.rs:164-171"""
    return proof_digest == aid_digest


# ----------------------------------------------------------------------------- composite dlog

def dlog_prove(N, g, ni, secret, r):
    """wi_dlog_proof.rs:46-65 with r (< 2^512, :53-54) injected."""
    x = pow(g, r, N)
    e = compute_digest([x, g, N, ni])
    y = r + e * secret
    return x, y


def dlog_verify(x, y, N, g, ni) -> bool:
    """wi_dlog_proof.rs:67-91 (the three asserts at :69,72,73 are panics in the reference)."""
    import math
    assert N > (1 << 128)
    assert math.gcd(g, N) == 1 and math.gcd(ni, N) == 1
    e = compute_digest([x, g, N, ni])
    return x == (pow(g, y, N) * pow(ni, e, N)) % N


# ----------------------------------------------------------------------------- fixtures

# The single fixed keypair of the reference's tests (range_proof_ni.rs:141-145 and 4 other places).
FIXTURE_P = 148677972634832330983979593310074301486537017973460461278300587514468301043894574906886127642530475786889672304776052879927627556769456140664043088700743909632312483413393134504352834240399191134336344285483935856491230340093391784574980688823380828143810804684752914935441384845195613674104960646037368551517
FIXTURE_Q = 158741574437007245654463598139927898730476924736461654463975966787719309357536545869203069369466212089132653564188443272208127277664424448947476335413293018778018615899291704693105620242763173357203898195318179150836424196645745308205164116144020613415407736216097185962171301808761138424668335445923774195463
FIXTURE_N = FIXTURE_P * FIXTURE_Q


# ----------------------------------------------------------------------------- ZeroProof / CiphertextProof

def zero_proof_prove(n, c, r, r_prime):
    """zero_enc_proof.rs:44-64 with r' injected (:45 samples it below n)."""
    nn = n * n
    a = enc(n, 0, r_prime)
    e = compute_digest([n, c, a])
    z = (r_prime * pow(r, e, nn)) % nn
    return z, a


def zero_proof_verify(n, c, z, a) -> bool:
    """zero_enc_proof.rs:66-94; [upstream] Paillier::mul = c^e mod nn, Paillier::add = product mod nn."""
    nn = n * n
    e = compute_digest([n, c, a])
    return enc(n, 0, z) == (pow(c, e, nn) * a) % nn


def ciphertext_proof_prove(n, c, x, r, x_prime, r_prime):
    """correct_ciphertext.rs:42-64 with (x', r') injected."""
    nn = n * n
    c_prime = enc(n, x_prime, r_prime)
    e = compute_digest([n, c, c_prime])
    return x_prime + x * e, (r_prime * pow(r, e, nn)) % nn, c_prime


def ciphertext_proof_verify(n, c, z1, z2, c_prime) -> bool:
    """correct_ciphertext.rs:66-97."""
    nn = n * n
    e = compute_digest([n, c, c_prime])
    return enc(n, z1, z2) == (pow(c, e, nn) * c_prime) % nn


# ----------------------------------------------------------------------------- VerlinProof

def gen_phi(n, c, c_prime, y, y_prime, y_double_prime, r_y):
    """verlin_proof.rs:138-165: c^y * c'^y' * Enc(y'', r_y) mod n^2."""
    nn = n * n
    return (pow(c, y, nn) * pow(c_prime, y_prime, nn) % nn) * enc(n, y_double_prime, r_y) % nn


def verlin_prove(n, c, c_prime, phi_x, x, xp, xpp, r_x, a, ap, app, r_a):
    """verlin_proof.rs:60-99 with the nonces (a, a', a'', r_a) injected."""
    nn = n * n
    phi_a = gen_phi(n, c, c_prime, a, ap, app, r_a)
    e = compute_digest([n, c, c_prime, phi_x, phi_a])
    return phi_a, x * e + a, xp * e + ap, xpp * e + app, (pow(r_x, e, nn) * r_a) % nn


def verlin_verify(n, c, c_prime, phi_x, phi_a, z, zp, zpp, r_z) -> bool:
    """verlin_proof.rs:101-135."""
    nn = n * n
    e = compute_digest([n, c, c_prime, phi_x, phi_a])
    return gen_phi(n, c, c_prime, z, zp, zpp, r_z) == (pow(phi_x, e, nn) * phi_a) % nn


# ----------------------------------------------------------------------------- mod_inv, MulProof, CorrectMessageProof

def mod_inv(a: int, m: int):
    """[upstream] curv BigInt::mod_inv -> GMP mpz_invert: None when gcd(a, m) != 1."""
    try:
        return pow(a, -1, m)
    except ValueError:
        return None


class Panic(Exception):
    """a Rust panic of the reference (unwrap on None, assert_eq!, index out of bounds)"""


def mul_proof_prove(n, e_a, e_b, e_c, a, b, r_a, r_b, r_c, d, r_d):
    """multiplication_proof.rs:60-104 with (d, r_d) injected (:61-62 samples them)."""
    nn = n * n
    e_d = enc(n, d, r_d)
    r_db = r_d * r_b                      # :69, not reduced
    db = d * b                            # :70, not reduced
    e_db = enc(n, db, r_db)
    e = compute_digest([n, e_a, e_b, e_c, e_d, e_db])
    f = (e * a % n + d) % n               # :87-88
    z1 = pow(r_a, e, nn) * r_d % nn       # :89-90
    r_b_f = pow(r_b, f, nn)
    r_db_r_c_e = r_db * pow(r_c, e, nn) % nn
    inv = mod_inv(r_db_r_c_e, nn)
    if inv is None:
        raise Panic("mod_inv unwrap :95")
    z2 = r_b_f * inv % nn
    return f, z1, z2, e_d, e_db


def mul_proof_verify(n, e_a, e_b, e_c, f, z1, z2, e_d, e_db) -> bool:
    """multiplication_proof.rs:106-146."""
    nn = n * n
    e = compute_digest([n, e_a, e_b, e_c, e_d, e_db])
    enc_f_z1 = enc(n, f, z1)
    enc_0_z2 = enc(n, 0, z2)
    lhs1 = pow(e_a, e, nn) * e_d % nn
    t = e_db * pow(e_c, e, nn) % nn
    inv = mod_inv(t, nn)
    if inv is None:
        raise Panic("mod_inv unwrap :133")
    lhs2 = pow(e_b, f, nn) * inv % nn
    return lhs1 == enc_f_z1 and lhs2 == enc_0_z2


CM_B = 256  # correct_message.rs:19


def correct_message_prove(n, valid_messages, message, r, e_sim, z_sim, w):
    """correct_message.rs:35-123 with (r, ei_vec, zi_vec, w) injected."""
    nn = n * n
    K = len(valid_messages)
    if K < 1:
        raise Panic("num_of_message - 1 underflows")
    ciphertext = enc(n, message, r)
    ui = []
    for m in valid_messages:
        gm = (m * n + 1) % nn
        gi = mod_inv(gm, nn)
        if gi is None:
            raise Panic("mod_inv unwrap :53")
        ui.append(ciphertext * gi % nn)
    a_vec = []
    j = 0
    for i in range(K):
        if valid_messages[i] == message:
            a_vec.append(pow(w, n, nn))
        else:
            if j >= len(z_sim):
                raise Panic("index out of bounds :74")
            zi_n = pow(z_sim[j], n, nn)
            ui_ei = pow(ui[i], e_sim[j], nn)
            inv = mod_inv(ui_ei, nn)
            if inv is None:
                raise Panic("mod_inv unwrap :76")
            j += 1
            a_vec.append(zi_n * inv % nn)
    two = 1 << CM_B
    chal = compute_digest(a_vec) % two
    ei = (chal - sum(e_sim) % two) % two
    zi = w * pow(r, ei, n) % n
    e_vec, z_vec = [], []
    j = 0
    for i in range(K):
        if valid_messages[i] == message:
            e_vec.append(ei); z_vec.append(zi)
        else:
            e_vec.append(e_sim[j]); z_vec.append(z_sim[j]); j += 1
    return ciphertext, e_vec, z_vec, a_vec


def correct_message_verify(n, valid_messages, ciphertext, e_vec, z_vec, a_vec) -> bool:
    """correct_message.rs:124-162."""
    nn = n * n
    two = 1 << CM_B
    chal = compute_digest(a_vec) % two
    if chal != sum(e_vec) % two:
        raise Panic("assert_eq!(chal, ei_sum) :132")
    ok = True
    for i, m in enumerate(valid_messages):
        gm = (m * n + 1) % nn
        gi = mod_inv(gm, nn)
        if gi is None:
            raise Panic("mod_inv unwrap :141")
        u = ciphertext * gi % nn
        ok = ok and (pow(u, e_vec[i], nn) * a_vec[i] % nn == pow(z_vec[i], n, nn))
    return ok
