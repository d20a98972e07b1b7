"""Python model of the word-batched binary GCD / modular inverse (Pornin-style, K = 30 inner steps on 64-bit approximations),
with the same word-level passes and int64 range assertions as the kernel."""
import math, random
K = 30
MK = (1 << K) - 1
def i64(x):
    assert -(1 << 63) <= x < (1 << 63), x
    return x
def words(x, n): return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)]
def unwords(w): return sum(v << (32 * i) for i, v in enumerate(w))
def sunwords(w):                      # signed two's complement
    x = unwords(w); n = 32 * len(w)
    return x - (1 << n) if x >> (n - 1) else x
def bitlen_words(w, nlive):
    while nlive > 0 and w[nlive - 1] == 0: nlive -= 1
    return nlive, (0 if nlive == 0 else 32 * (nlive - 1) + w[nlive - 1].bit_length())
def extract(w, p, nbits, kw):         # bits [p, p+nbits) of the word array
    w0, off = p >> 5, p & 31
    g = lambda i: w[i] if 0 <= i < kw else 0
    v = (g(w0) | (g(w0 + 1) << 32) | (g(w0 + 2) << 64)) >> off
    return v & ((1 << nbits) - 1)
def lincomb_shift(f, g, A, B, n, signed_top=False):
    """(f*A + g*B) >> K over n words (A, B word arrays; if signed_top the top word is signed); returns (words[n], negative?)"""
    acc = 0; T = []
    for w in range(n):
        aw, bw = A[w], B[w]
        if signed_top and w == n - 1:
            aw = aw - (1 << 32) if aw >> 31 else aw; bw = bw - (1 << 32) if bw >> 31 else bw
        acc = i64(acc + f * aw + g * bw)
        T.append(acc & 0xFFFFFFFF); acc >>= 32
    T.append(acc & 0xFFFFFFFF)         # sign/top word
    out = [((T[w] >> K) | (T[w + 1] << (32 - K))) & 0xFFFFFFFF for w in range(n)]
    return out, acc < 0, T
def negate(W):
    c = 1; out = []
    for x in W:
        t = (x ^ 0xFFFFFFFF) + c; out.append(t & 0xFFFFFFFF); c = t >> 32
    return out
def wbgcd(y, m, kw, cof=True, stats=None):
    a, b = words(y, kw), words(m, kw)
    u, v = words(1, kw + 1), words(0, kw + 1)
    mw = words(m, kw) + [0]
    minv = (-pow(m, -1, 1 << K)) % (1 << K)
    na = nb = kw; iters = 0
    while True:
        na, la = bitlen_words(a, na); nb, lb = bitlen_words(b, nb)
        if la == 0: break
        n = max(la, lb)
        if n <= 64:
            xa, xb = unwords(a[:2]), unwords(b[:2])
        else:
            xa = (extract(a, n - 34, 34, kw) << 30) | (a[0] & MK)
            xb = (extract(b, n - 34, 34, kw) << 30) | (b[0] & MK)
        f0, g0, f1, g1 = 1, 0, 0, 1
        for _ in range(K):
            if xa & 1:
                if xa < xb: xa, xb, f0, g0, f1, g1 = xb, xa, f1, g1, f0, g0
                xa -= xb; f0 -= f1; g0 -= g1
            xa >>= 1; f1 <<= 1; g1 <<= 1
        assert abs(f0) + abs(g0) <= 1 << K and abs(f1) + abs(g1) <= 1 << K
        nw = max(na, nb)
        A2, nega, Ta = lincomb_shift(f0, g0, a, b, nw)
        B2, negb, Tb = lincomb_shift(f1, g1, a, b, nw)
        assert Ta[0] & MK == 0 and Tb[0] & MK == 0
        if nega: A2 = negate(A2); f0, g0 = -f0, -g0
        if negb: B2 = negate(B2); f1, g1 = -f1, -g1
        a[:nw] = A2; b[:nw] = B2
        na = nb = nw                      # either result may be as long as the longer input
        if cof:
            def upd(f, g):
                t0 = (f * u[0] + g * v[0]) & 0xFFFFFFFF
                c = (t0 * minv) & MK
                if c >> (K - 1): c -= 1 << K            # balanced
                acc = 0; T = []
                for w in range(kw + 1):
                    uw, vw = u[w], v[w]
                    if w == kw:
                        uw = uw - (1 << 32) if uw >> 31 else uw; vw = vw - (1 << 32) if vw >> 31 else vw
                    acc = i64(acc + f * uw + g * vw + c * mw[w])
                    T.append(acc & 0xFFFFFFFF); acc >>= 32
                T.append(acc & 0xFFFFFFFF)
                assert T[0] & MK == 0
                return [((T[w] >> K) | (T[w + 1] << (32 - K))) & 0xFFFFFFFF for w in range(kw + 1)]
            u2, v2 = upd(f0, g0), upd(f1, g1)
            u, v = u2, v2
            if stats is not None: stats["maxcof"] = max(stats["maxcof"], abs(sunwords(u)) // m, abs(sunwords(v)) // m)
        iters += 1
        assert iters < 2 * 32 * kw // K + 40
    g = unwords(b)
    if stats is not None: stats["iters"] = max(stats.get("iters", 0), iters)
    if not cof: return g, None
    if g != 1: return g, None
    vv = sunwords(v)
    assert (vv * y - 1) % m == 0
    # the kernel's finalisation (kernels_inv.hpp: k_modinv): add m while negative, subtract m while >= m, on kw words + sign word
    passes = 0
    while True:
        top = v[kw] - (1 << 32) if v[kw] >> 31 else v[kw]
        neg, ge = top < 0, top > 0
        if top == 0: ge = unwords(v[:kw]) >= m
        if not neg and not ge: break
        t = (unwords(v[:kw]) + (m if neg else -m))
        carry = t >> (32 * kw)                     # +1 / 0 / -1
        v = words(t & ((1 << (32 * kw)) - 1), kw) + [(v[kw] + carry) & 0xFFFFFFFF]
        passes += 1
        assert passes < 256
    if stats is not None: stats["final_passes"] = max(stats.get("final_passes", 0), passes)
    assert v[kw] == 0 and unwords(v[:kw]) == vv % m
    return g, unwords(v[:kw])

if __name__ == "__main__":
    rnd = random.Random(1)
    for kw in (2, 4, 64, 128):
        stats = {"maxcof": 0}
        for trial in range(40 if kw > 8 else 400):
            bits = 32 * kw
            m = rnd.getrandbits(bits) | 1 | (1 << (bits - 1)) if trial % 3 else rnd.getrandbits(rnd.randrange(3, bits)) | 1
            if m < 3: m = 3
            kind = trial % 7
            y = [rnd.randrange(1, m), 1, m - 1, 1 << rnd.randrange(0, m.bit_length() - 1), rnd.randrange(1, m) >> rnd.randrange(0, bits), m // 2, rnd.randrange(1, 1 << 20) % m or 1][kind]
            if trial % 11 == 0:
                p = rnd.getrandbits(bits // 3) | 1; m = (p * (rnd.getrandbits(bits - bits // 3 - 1) | 1)) | 1
                if m % p == 0: y = p * rnd.randrange(1, m // p)
            g, inv = wbgcd(y, m, kw, True, stats)
            assert g == math.gcd(y, m), (y, m)
            if g == 1: assert inv == pow(y, -1, m)
            g2, _ = wbgcd(y, m, kw, False)
            assert g2 == g
        print(kw, stats)
    print("ok")
