"""GPU parity tests for the L1 boundary (modexp / modmul / Paillier Enc) through the C ABI,
bit-exact against the C/GMP oracle on the same seeded inputs."""
import math

import numpy as np
import pytest

import helpers as H
from helpers import pm, L

pytestmark = pytest.mark.gpu


def rand_limbs(d, count, nlimbs, bits=None):
    bits = bits or 32 * nlimbs
    return L.ints_to_limbs([d.bits(bits) for _ in range(count)], nlimbs)


@pytest.mark.parametrize("mod_bits,count", [(2048, 37), (4096, 21), (8192, 13)])
def test_modexp_per_item_moduli(ctx, oracle, mod_bits, count):
    d = pm.Drbg(b"gpu-modexp-%d" % mod_bits)
    nl = mod_bits // 32
    mods = [d.bits(mod_bits) | 1 | (1 << (mod_bits - 1)) for _ in range(count)]
    mods[1] = (1 << mod_bits) - 1                      # all-ones modulus
    mods[2] = (1 << (mod_bits - 1)) + 1                # top limb 0x80000000, sparse
    mods[3] = d.bits(mod_bits // 2) | 1                # half-width modulus, zero padded
    mods[4] = 3
    bases = [d.below(m) for m in mods]
    bases[0] = 0; bases[5] = 1; bases[6] = mods[6] - 1
    bases[7] = d.bits(mod_bits)                        # base >= modulus is legal for mpz_powm
    exps = [d.bits(mod_bits) for _ in range(count)]
    exps[8] = 0; exps[9] = 1; exps[10] = (1 << mod_bits) - 1; exps[11] = 1 << (mod_bits - 1)
    b, e, m = (L.ints_to_limbs(v, nl) for v in (bases, exps, mods))
    out = np.zeros_like(b)
    ctx.modexp(mod_bits, mod_bits, count, b, e, nl, m, nl, out)
    ref = oracle.modexp(mod_bits, mod_bits, b, e, nl, m, nl)
    assert np.array_equal(out, ref)


def test_modexp_shared_modulus_short_exponent(ctx, oracle):
    d = pm.Drbg(b"gpu-modexp-shared")
    nl, count = 64, 300
    mod = d.bits(2048) | 1 | (1 << 2047)
    b = L.ints_to_limbs([d.below(mod) for _ in range(count)], nl)
    e = rand_limbs(d, count, 8)                           # 256-bit exponents (DLog ni^e shape)
    m = L.ints_to_limbs([mod], nl)
    out = np.zeros_like(b)
    ctx.modexp(2048, 256, count, b, e, 8, m, 0, out)
    assert np.array_equal(out, oracle.modexp(2048, 256, b, e, 8, m, 0))
    # shared exponent too (the sigma^n mod n shape with one key)
    e1 = rand_limbs(d, 1, 64)
    ctx.modexp(2048, 2048, count, b, e1, 0, m, 0, out)
    assert np.array_equal(out, oracle.modexp(2048, 2048, b, e1, 0, m, 0))


def test_even_modulus_is_refused(ctx, zkp):
    b = np.ones((2, 64), np.uint32); e = np.ones((2, 64), np.uint32)
    m = np.zeros((2, 64), np.uint32); m[0, 0] = 7; m[1, 0] = 8
    out = np.zeros_like(b)
    with pytest.raises(zkp.ZkpError):
        ctx.modexp(2048, 2048, 2, b, e, 64, m, 64, out)


@pytest.mark.parametrize("mod_bits", [2048, 4096])
def test_modmul(ctx, oracle, mod_bits):
    d = pm.Drbg(b"gpu-modmul-%d" % mod_bits)
    nl, count = mod_bits // 32, 50
    mods = [d.bits(mod_bits) | 1 | (1 << (mod_bits - 1)) for _ in range(count)]
    a = rand_limbs(d, count, nl); b = rand_limbs(d, count, nl)     # operands may exceed the modulus
    a[0] = 0; b[1] = 0; a[2] = L.int_to_limbs(mods[2] - 1, nl); b[2] = a[2]
    m = L.ints_to_limbs(mods, nl)
    out = np.zeros_like(a)
    ctx.modmul(mod_bits, count, a, b, m, nl, out)
    assert np.array_equal(out, oracle.modmul(mod_bits, a, b, m, nl))


def test_paillier_enc_fixture_key(ctx, oracle):
    _, _, n = H.fixture_key()
    d = pm.Drbg(b"gpu-enc")
    count, kw = 70, 64
    ms = [d.bits(256) for _ in range(count)]
    rs = [d.below(n) for _ in range(count)]
    ms[0] = 0; ms[1] = 1; ms[2] = n - 1; ms[3] = (1 << 2048) - 1      # m >= n: (1+m*n) mod n^2 semantics
    rs[4] = 0; rs[5] = 1; rs[6] = n - 1; rs[7] = (1 << 2048) - 1      # r >= n is legal input to mpz_powm
    nl, m, r = L.ints_to_limbs([n], kw), L.ints_to_limbs(ms, kw), L.ints_to_limbs(rs, kw)
    out = np.zeros((count, 2 * kw), np.uint32)
    ctx.paillier_enc(2048, count, nl, 0, m, r, out)
    ref = oracle.paillier_enc(2048, nl, 0, m, r)
    assert np.array_equal(out, ref)
    assert L.limbs_to_int(out[8]) == pm.enc(n, ms[8], rs[8])


def test_paillier_enc_per_item_keys_and_widths(ctx, oracle):
    for n_bits, count in ((1024, 9), (2048, 6), (4096, 3)):
        kw = n_bits // 32
        d = pm.Drbg(b"gpu-enc-keys-%d" % n_bits)
        ns = [d.bits(n_bits) | 1 | (1 << (n_bits - 1)) for _ in range(count)]    # odd pseudo-moduli suffice for Enc parity
        ns[0] = H.test_key(512)[2]                                               # short key, zero padded
        nl = L.ints_to_limbs(ns, kw)
        m = rand_limbs(d, count, kw, 256); r = L.ints_to_limbs([d.below(v) for v in ns], kw)
        out = np.zeros((count, 2 * kw), np.uint32)
        ctx.paillier_enc(n_bits, count, nl, kw, m, r, out)
        assert np.array_equal(out, oracle.paillier_enc(n_bits, nl, kw, m, r))


@pytest.mark.parametrize("per_item_keys", [False, True])
def test_paillier_enc_check(ctx, oracle, per_item_keys):
    """zkp_paillier_enc_check_batch = CorrectOpening::verify_opening (correct_opening.rs:17-30) and the verifier's two
    equality shapes (range_proof.rs:280-298 raw compare, :324-337 compare with c_j * cipher_x % nn)."""
    n_bits, kw, count = 2048, 64, 24
    d = pm.Drbg(b"gpu-enc-check-%d" % per_item_keys)
    _, _, nfix = H.fixture_key()
    ns = [d.bits(n_bits) | 1 | (1 << (n_bits - 1)) for _ in range(count)] if per_item_keys else [nfix]
    stride = kw if per_item_keys else 0
    key = lambda i: ns[i] if per_item_keys else nfix
    ms = [d.bits(256) for _ in range(count)]
    rs = [d.below(key(i)) for i in range(count)]
    nl, m, r = L.ints_to_limbs(ns, kw), L.ints_to_limbs(ms, kw), L.ints_to_limbs(rs, kw)
    cs = [pm.enc(key(i), ms[i], rs[i]) for i in range(count)]
    # (a) expected given directly: honest, one flipped bit, c + n^2 (same residue, unreduced: the reference's == is false), zero
    exp = list(cs)
    exp[1] ^= 1 << 77
    exp[2] = cs[2] + key(2) ** 2 if (cs[2] + key(2) ** 2).bit_length() <= 4096 else cs[2] ^ 1
    exp[3] = 0
    e = L.ints_to_limbs(exp, 2 * kw)
    ok = np.full(count, 9, np.uint8)
    ctx.paillier_enc_check(n_bits, count, nl, stride, m, r, None, None, e, ok)
    ref = oracle.paillier_enc_check(n_bits, nl, stride, m, r, None, None, e)
    assert np.array_equal(ok, ref)
    assert list(ok[:4]) == [1, 0, 0, 0] and ok[4:].all()
    # (b) expected = a * b mod n^2 with unreduced factors (a >= n^2 allowed: `%` reduces the product)
    a_int, b_int = [], []
    for i in range(count):
        nn = key(i) ** 2
        bv = d.below(nn)
        while math.gcd(bv, nn) != 1:                                # invertible factor
            bv = d.below(nn)
        av = cs[i] * pow(bv, -1, nn) % nn
        a_int.append(av); b_int.append(bv)
    if a_int[5] + key(5) ** 2 < (1 << 4096):
        a_int[5] += key(5) ** 2                                     # unreduced factor, same product residue: still equal
    b_int[6] ^= 2                                                   # wrong product
    a, b = L.ints_to_limbs(a_int, 2 * kw), L.ints_to_limbs(b_int, 2 * kw)
    ok2 = np.full(count, 9, np.uint8)
    ctx.paillier_enc_check(n_bits, count, nl, stride, m, r, a, b, None, ok2)
    ref2 = oracle.paillier_enc_check(n_bits, nl, stride, m, r, a, b, None)
    assert np.array_equal(ok2, ref2)
    assert ok2[5] == 1 and ok2[6] == 0 and ok2[:5].all() and ok2[7:].all()


def test_paillier_enc_check_argument_errors(ctx, zkp):
    kw = 64
    z = np.zeros((1, kw), np.uint32); c = np.zeros((1, 2 * kw), np.uint32); ok = np.zeros(1, np.uint8)
    with pytest.raises(zkp.ZkpError):
        ctx.paillier_enc_check(2048, 1, z, 0, z, z, c, None, None, ok)       # only one factor
    with pytest.raises(zkp.ZkpError):
        ctx.paillier_enc_check(2048, 1, z, 0, z, z, c, c, c, ok)             # both forms
    with pytest.raises(zkp.ZkpError):
        ctx.paillier_enc_check(2048, 1, z, 0, z, z, None, None, None, ok)    # neither


def test_modexp_rejects_short_strides(ctx, zkp):
    """a non-zero stride smaller than the element width would make the kernels read past the staged buffers"""
    b = np.zeros((2, 64), np.uint32); b[:, 0] = 3
    out = np.zeros_like(b)
    with pytest.raises(zkp.ZkpError):
        ctx.modexp(2048, 2048, 2, b, b, 32, b, 64, out)
    with pytest.raises(zkp.ZkpError):
        ctx.modexp(2048, 2048, 2, b, b, 64, b, 63, out)
    with pytest.raises(zkp.ZkpError):
        ctx.modmul(2048, 2, b, b, b, 1, out)


def test_enc_decrypts_with_the_fixture_secret_key(ctx):
    """Pins Enc to the standard g = 1+n Paillier definition WITHOUT the oracle: ciphertexts made by the GPU decrypt, with
    the secret key of the reference's fixture (range_proof_ni.rs:141-145), to the plaintext, c = r^n (mod n), and
    Paillier::open's randomness recovery (the reference's own test: correct_opening.rs:47-57) returns r."""
    p, q, n = H.fixture_key()
    nn, kw = n * n, 64
    lam = (p - 1) * (q - 1) // math.gcd(p - 1, q - 1)
    phi = (p - 1) * (q - 1)
    d = pm.Drbg(b"gpu-enc-decrypt")
    count = 12
    ms = [d.below(n) for _ in range(count)]; ms[0] = 0; ms[1] = 10; ms[2] = n - 1
    rs = [d.below(n) for _ in range(count)]; rs[3] = 1
    out = np.zeros((count, 2 * kw), np.uint32)
    ctx.paillier_enc(2048, count, L.ints_to_limbs([n], kw), 0, L.ints_to_limbs(ms, kw), L.ints_to_limbs(rs, kw), out)
    ok = np.zeros(count, np.uint8)
    ctx.paillier_enc_check(2048, count, L.ints_to_limbs([n], kw), 0, L.ints_to_limbs(ms, kw), L.ints_to_limbs(rs, kw), None, None, out, ok)
    assert ok.all()                                                      # verify_opening(ek, m, r, c)
    for m, r, c in zip(ms, rs, L.limbs_to_ints(out)):
        assert 0 <= c < nn
        u = pow(c, lam, nn)
        assert (u - 1) % n == 0
        assert ((u - 1) // n) * pow(lam, -1, n) % n == m                 # Dec(c) = L(c^lambda mod n^2) * mu mod n
        assert c % n == pow(r, n, n)                                     # c = (1 + m n) r^n  ->  c = r^n (mod n)
        assert pow(c % n, pow(n, -1, phi), n) == r                       # Paillier::open: the n-th root of c mod n


def test_even_modulus_deviation_is_pinned(ctx, oracle, zkp):
    """DOCUMENTED DEVIATION: Montgomery arithmetic needs an odd modulus.  For an even modulus the reference
    (GMP) still computes; the engine reports ZKP_ENONCANONICAL and leaves the outputs of those items untouched / zero, and
    proof-level entry points answer MALFORMED.  This test pins both behaviours next to each other."""
    kw = 64
    d = pm.Drbg(b"even-modulus")
    mod_even = (d.bits(2048) | (1 << 2047)) & ~1
    mod_odd = mod_even | 1
    base = [d.bits(2040), d.bits(2040)]; e = [d.bits(2048), d.bits(2048)]
    b, ee, m = L.ints_to_limbs(base, kw), L.ints_to_limbs(e, kw), L.ints_to_limbs([mod_even, mod_odd], kw)
    ref = oracle.modexp(2048, 2048, b, ee, kw, m, kw)
    assert L.limbs_to_ints(ref) == [pow(base[0], e[0], mod_even), pow(base[1], e[1], mod_odd)]      # what the reference returns
    out = np.full_like(b, 0xA5A5A5A5)
    with pytest.raises(zkp.ZkpError, match="status 2"):
        ctx.modexp(2048, 2048, 2, b, ee, kw, m, kw, out)
    # (host-pointer mode copies nothing back on an error status: the odd item is recomputed on its own)
    out1 = np.zeros((1, kw), np.uint32)
    ctx.modexp(2048, 2048, 1, b[1:], ee[1:], kw, m[1:], kw, out1)
    assert np.array_equal(out1[0], ref[1])
    # Enc under an even n: GMP computes a value, the engine writes zeros for that item
    n_even = (d.bits(1024) | (1 << 1023)) & ~1
    mm, rr = d.bits(256), d.bits(1000)
    nl, ml, rl = L.ints_to_limbs([n_even], 32), L.ints_to_limbs([mm], 32), L.ints_to_limbs([rr], 32)
    c_ref = oracle.paillier_enc(1024, nl, 0, ml, rl)
    assert L.limbs_to_int(c_ref[0]) == pm.enc(n_even, mm, rr) != 0
    c = np.full((1, 64), 7, np.uint32)
    ctx.paillier_enc(1024, 1, nl, 0, ml, rl, c)
    assert not c.any()
    # a RangeProofNi under an even key: the oracle (reference behaviour) evaluates it, the engine answers MALFORMED
    cases = H.build_range_case(b"even-key", [n_even], 1024, 1)
    pb, wt = H.fill_batch(cases, 1024, True, oracle)
    oracle.range_ni_prove(pb.struct(), wt.struct(), None, None, None)
    vo = np.full(1, 9, np.uint8); vg = np.full(1, 9, np.uint8)
    oracle.range_ni_verify(pb.struct(), vo)
    ctx.range_ni_verify(pb.struct(), vg, device=False)
    assert vo[0] == zkp.VERDICT_ACCEPT and vg[0] == zkp.VERDICT_MALFORMED


def test_structured_moduli_safe_and_fast_products(ctx, oracle):
    """W = 36: keys whose Orup multiple has lanes full of large digits fail k_setup's digit-sum test and run the SAFE Montgomery
    product (csrc/bigint29.hpp "column capacity"); operands with every digit at its maximum are the worst case for the FAST one.
    Both against the oracle, shared key (sliding-window ladder) and per-item keys (fixed windows, mixed in one wavefront)."""
    d = pm.Drbg(b"structured-moduli")
    # --- modexp, 2048 and 4096 bits: all-ones style moduli and bases / exponents of all ones
    for bits in (2048, 4096):
        nl = bits // 32
        mods = [(1 << bits) - 1, (1 << (bits - 1)) - 1, (1 << bits) - (1 << (bits // 2)) - 1, ((1 << bits) - 1) // 3 | 1,
                d.bits(bits) | 1 | (1 << (bits - 1)), (1 << bits) - 1 - (1 << 29), int("f" * (bits // 4 - 9) + "e" + "f" * 8, 16)]
        count = len(mods)
        bases = [(1 << bits) - 2, (1 << (bits - 1)) - 2, d.bits(bits), (1 << (bits - 2)) - 1, (1 << bits) - 1, d.bits(bits), (1 << bits) - 3]
        exps = [(1 << bits) - 1, d.bits(bits), (1 << bits) - 1, d.bits(bits), (1 << bits) - 1, d.bits(bits), d.bits(bits)]
        b, e, m = (L.ints_to_limbs(v, nl) for v in (bases, exps, mods))
        out = np.zeros_like(b)
        ctx.modexp(bits, bits, count, b, e, nl, m, nl, out)                      # per-item moduli
        assert np.array_equal(out, oracle.modexp(bits, bits, b, e, nl, m, nl))
        for i in (0, 2, 4):                                                      # each as a shared modulus with a shared exponent
            bb = L.ints_to_limbs([bases[j] for j in range(count)], nl)
            out = np.zeros_like(bb)
            ctx.modexp(bits, bits, count, bb, e[i:i + 1], 0, m[i:i + 1], 0, out)
            assert np.array_equal(out, oracle.modexp(bits, bits, bb, e[i:i + 1], 0, m[i:i + 1], 0))
    # --- Enc under structured keys n (n^2 is the modulus): shared key and per-item keys
    kw = 64
    ns = [(1 << 2048) - 1, (1 << 2047) + 1, (1 << 2048) - (1 << 1024) - 1, H.fixture_key()[2], (1 << 2040) - 1, d.bits(2048) | 1 | (1 << 2047)]
    ms = [d.bits(2048) for _ in ns]; rs = [(1 << 2048) - 1, d.bits(2048), (1 << 2047) - 1, (1 << 2048) - 1, d.bits(2048), (1 << 2048) - 1]
    nl_, ml, rl = L.ints_to_limbs(ns, kw), L.ints_to_limbs(ms, kw), L.ints_to_limbs(rs, kw)
    out = np.zeros((len(ns), 2 * kw), np.uint32)
    ctx.paillier_enc(2048, len(ns), nl_, kw, ml, rl, out)
    assert np.array_equal(out, oracle.paillier_enc(2048, nl_, kw, ml, rl))
    for i in (0, 2, 3):
        out = np.zeros((len(ns), 2 * kw), np.uint32)
        ctx.paillier_enc(2048, len(ns), nl_[i:i + 1], 0, ml, rl, out)
        assert np.array_equal(out, oracle.paillier_enc(2048, nl_[i:i + 1], 0, ml, rl))
