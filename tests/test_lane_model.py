"""CPU model of the wave-group Montgomery multiplication of csrc/bigint29.hpp (word-level CIOS on G lanes x W
limbs of 29 bits, circular column window, optional Orup multiple), checked against Python integers.  It documents
the invariants the kernel relies on:
  * no 64-bit column overflow.  W <= 31: for any operands.  W = 36 (72 products per column life): the SAFE product moves a
    column's upper word into the next column at half of its life and is exact for any operands; the FAST product is exact
    for any A and B provided every lane's limb sum of the modulus operand is <= COL_FAST_SN_LIMIT (k_setup tests the
    Orup multiple of every key and the ladders pick the product accordingly);
  * lane 0's bottom limb always zero (so the DPP pass-down needs no masking between groups);
  * result < 2M without any conditional subtraction, <= M when B == 1."""
import random

import pytest

B = 29
MASK = (1 << B) - 1


def fast_sn_limit(W):
    """bigint29.hpp COL_FAST_SN_LIMIT"""
    return ((1 << 64) - 1 - (1 << 36) - ((1 << B) + 16) * (W * (1 << B) + 16)) >> B


def to_limbs(x, n):
    return [(x >> (B * i)) & MASK for i in range(n)]


def from_limbs(l):
    return sum(v << (B * i) for i, v in enumerate(l))


def montmul(N, n1, G, W, A, Bv, orup, stats, safe=True, check=True):
    """mirror of montmul<G, ORUP, SAFE>: N = limbs of M (or of M~ = M*n1 when orup), n1 = -M^-1 mod 2^29"""
    c = [[0] * W for _ in range(G)]
    for s in range(G):
        for t in range(W):
            b = Bv[s * W + t]
            for j in range(G):
                for k in range(W):
                    c[j][(t + k) % W] += A[j * W + k] * b
            c0 = c[0][t] & 0xFFFFFFFF
            q = (c0 if orup else (c0 * n1) & 0xFFFFFFFF) & MASK
            lo = [0] * G
            for j in range(G):
                for k in range(W):
                    c[j][(t + k) % W] += N[j * W + k] * q
                    stats["maxcol"] = max(stats["maxcol"], c[j][(t + k) % W])
                v = c[j][t]
                lo[j] = v & MASK
                c[j][(t + 1) % W] += v >> B
                stats["maxcol"] = max(stats["maxcol"], c[j][(t + 1) % W])
            if check:
                assert lo[0] == 0                  # the value lane G-1 of the previous group would receive
            for j in range(G):
                c[j][t] = lo[j + 1] if j + 1 < G else 0
                if safe and 2 * W > 63:
                    km, kn = (t + W // 2) % W, (t + W // 2 + 1) % W
                    c[j][kn] += (c[j][km] >> 32) << 3
                    c[j][km] &= 0xFFFFFFFF
                    stats["maxcol"] = max(stats["maxcol"], c[j][kn])
    out, carries = [], []
    for j in range(G):
        cy, r = 0, []
        for k in range(W):
            v = c[j][k] + cy
            r.append(v & MASK)
            cy = v >> B
        out.append(r)
        carries.append(cy)
    for j in range(1, G):
        out[j][0] += carries[j - 1]
        assert not check or out[j][0] < (1 << B) + 64
    assert not check or carries[G - 1] == 0
    return [v for r in out for v in r]


@pytest.mark.parametrize("bits,G,W", [(2048, 2, 36), (4096, 4, 36), (8192, 8, 36), (300, 2, 36), (2048, 4, 18), (4096, 8, 18), (2048, 8, 9), (300, 4, 18)])
@pytest.mark.parametrize("orup", [False, True])
@pytest.mark.parametrize("safe", [True, False])
def test_word_level_cios_model(bits, G, W, orup, safe):
    rnd = random.Random(bits * 31 + G + W + orup)
    M = rnd.getrandbits(bits) | 1 | (1 << (bits - 1))
    L = G * W
    R = 1 << (B * L)
    n1 = (-pow(M, -1, 1 << B)) % (1 << B)
    Mt = M * n1
    assert Mt % (1 << B) == MASK and 4 * Mt < R      # Orup multiple fits with headroom
    N = to_limbs(Mt if orup else M, L)
    bound = 2 * (Mt if orup else M)
    Rinv = pow(R, -1, M)
    stats = {"maxcol": 0}
    for _ in range(2):
        a, b = rnd.randrange(bound), rnd.randrange(bound)
        r = montmul(N, n1, G, W, to_limbs(a, L), to_limbs(b, L), orup, stats, safe)
        v = from_limbs(r)
        assert v % M == a * b * Rinv % M and v < bound
        r2 = montmul(N, n1, G, W, r, r, orup, stats, safe)     # feed the almost-normalised output straight back
        assert from_limbs(r2) % M == v * v * Rinv % M
    if not orup:
        one = montmul(N, n1, G, W, to_limbs(rnd.randrange(2 * M), L), to_limbs(1, L), orup, stats, safe)
        assert from_limbs(one) <= M                            # montmul(x, 1) <= M: one equality test canonicalises
    assert stats["maxcol"] < (1 << 64)


def worst_operands(G, W):
    """every limb at its maximum, plus the slack of almost-normalised operands (limb 0 of a lane's block may reach 2^29 + 16)"""
    big = [MASK] * (G * W)
    for j in range(G):
        big[j * W] = MASK + 17
    return big


@pytest.mark.parametrize("G,W", [(4, 36), (2, 36), (8, 36), (8, 18)])
def test_safe_product_cannot_overflow_for_any_operands(G, W):
    stats = {"maxcol": 0}
    big = worst_operands(G, W)
    montmul([MASK] * (G * W), 1, G, W, big, big, True, stats, safe=True, check=False)
    assert (1 << 62) < stats["maxcol"] < (1 << 64)
    # analytic: at most W products before the hand-over and W after it (plus what the column kept, a carry and a digit)
    assert W * (MASK + 17) ** 2 + (1 << 32) + (1 << 36) < (1 << 64)


@pytest.mark.parametrize("G,W", [(4, 36), (2, 36)])
def test_fast_product_bound(G, W):
    """the FAST product (no hand-over) with A, B at their maximum: exact as long as every lane's limb sum of the modulus
    operand is within COL_FAST_SN_LIMIT; beyond it (an all-ones modulus) a column can exceed 64 bits in the model."""
    L = G * W
    lim = fast_sn_limit(W)
    assert 27 * (1 << B) < lim < 28 * (1 << B)
    big = worst_operands(G, W)
    # a modulus operand exactly at the limit: 27 limbs at the maximum, one partial, the rest zero, in every lane
    lane = [MASK] * 27 + [lim - 27 * MASK] + [0] * (W - 28)
    assert sum(lane) == lim and all(0 <= v <= MASK for v in lane)
    stats = {"maxcol": 0}
    montmul(lane * G, 1, G, W, big, big, True, stats, safe=False, check=False)
    assert stats["maxcol"] < (1 << 64)
    # analytic bound used by bigint29.hpp: max(b) * S_A + max(q) * S_N + carry/digit
    assert ((1 << B) + 16) * (W * (1 << B) + 16) + MASK * lim + (1 << 36) < (1 << 64)
    # ... and the bound is not vacuous: with an all-ones modulus operand the analytic bound fails
    assert ((1 << B) + 16) * (W * (1 << B) + 16) + MASK * (W * MASK) + (1 << 36) > (1 << 64)


def montmul2(N, G, W, A, Bv, stats):
    """mirror of montmul2<G> (bigint29.hpp): N = limbs of M~~ = M * n2, n2 = -M^-1 mod 2^58; W odd: (W-1)/2 pairs + one single step"""
    c = [[0] * W for _ in range(G)]

    def finish(t):
        lo = [0] * G
        for j in range(G):
            v = c[j][t % W]
            lo[j] = v & MASK
            c[j][(t + 1) % W] += v >> B
        assert lo[0] == 0
        for j in range(G):
            c[j][t % W] = lo[j + 1] if j + 1 < G else 0

    for s in range(G):
        t = 0
        while t + 1 < W:
            b0, b1 = Bv[s * W + t], Bv[s * W + t + 1]
            for j in range(G):
                c[j][t % W] += A[j * W] * b0
                c[j][(t + 1) % W] += A[j * W + 1] * b0 + A[j * W] * b1
            q0 = c[0][t % W] & MASK
            q1 = (c[0][(t + 1) % W] + (c[0][t % W] >> B)) & MASK          # does not wait for q0
            for j in range(G):
                for k in range(2, W):
                    c[j][(t + k) % W] += A[j * W + k] * b0
                for k in range(1, W - 1):
                    c[j][(t + 1 + k) % W] += A[j * W + k] * b1
                for k in range(W):
                    c[j][(t + k) % W] += N[j * W + k] * q0
                for k in range(W - 1):
                    c[j][(t + 1 + k) % W] += N[j * W + k] * q1
            finish(t)                                                      # slot t is column t + W from here on
            for j in range(G):
                c[j][t % W] += A[j * W + W - 1] * b1 + N[j * W + W - 1] * q1
            finish(t + 1)
            stats["maxcol"] = max(stats["maxcol"], max(max(r) for r in c))
            t += 2
        if W & 1:
            b = Bv[s * W + W - 1]
            for j in range(G):
                for k in range(W):
                    c[j][(W - 1 + k) % W] += A[j * W + k] * b
            q = c[0][W - 1] & MASK
            for j in range(G):
                for k in range(W):
                    c[j][(W - 1 + k) % W] += N[j * W + k] * q
            finish(W - 1)
            stats["maxcol"] = max(stats["maxcol"], max(max(r) for r in c))
    out, carries = [], []
    for j in range(G):
        cy, r = 0, []
        for k in range(W):
            v = c[j][k] + cy
            r.append(v & MASK)
            cy = v >> B
        out.append(r)
        carries.append(cy)
    for j in range(1, G):
        out[j][0] += carries[j - 1]
    assert carries[G - 1] == 0
    return [v for r in out for v in r]


@pytest.mark.parametrize("bits,G,W", [(4096, 16, 9), (8192, 32, 9), (2048, 8, 9), (900, 4, 9)])
def test_two_quotient_digits_per_step_model(bits, G, W):
    """the latency engine's product under the 58-bit Orup multiple: both digits of a pair of sub-steps come from the bottom two
    columns without a multiplication and without each other; same R as montmul, results < 2 M~~, no column near 2^64.  A modulus
    within 60 bits of the capacity (2048 bits in 8 x 9 limbs) has no room for M~~: k_setup's flag (ConstLayout::OFF_ST + 2)."""
    rnd = random.Random(bits + G)
    M = rnd.getrandbits(bits) | 1 | (1 << (bits - 1))
    L = G * W
    R = 1 << (B * L)
    n2 = (-pow(M, -1, 1 << (2 * B))) % (1 << (2 * B))
    Mt2 = M * n2
    assert Mt2 % (1 << (2 * B)) == (1 << (2 * B)) - 1
    fits = bits + 2 * B + 3 <= B * L                    # k_setup: mt2_ok
    assert not fits or 4 * (2 * Mt2) < R                # operands < 2 M~~ and R > 4 * that: results stay < 2 M~~
    if not fits:
        assert (bits, G) == (2048, 8)
        return
    N = to_limbs(Mt2, L)
    assert N[0] == MASK and N[1] == MASK
    Rinv = pow(R, -1, M)
    stats = {"maxcol": 0}
    n1 = n2 & MASK
    for _ in range(2):
        a, b = rnd.randrange(2 * Mt2), rnd.randrange(2 * Mt2)
        r = montmul2(N, G, W, to_limbs(a, L), to_limbs(b, L), stats)
        v = from_limbs(r)
        assert v % M == a * b * Rinv % M and v < 2 * Mt2
        # the single-digit product on the same multiple gives the same value limb for limb (montmul2 only reorders the schedule)
        assert r == montmul(N, n1, G, W, to_limbs(a, L), to_limbs(b, L), True, {"maxcol": 0}, safe=False)
        r2 = montmul2(N, G, W, r, r, stats)
        assert from_limbs(r2) % M == v * v * Rinv % M
    assert stats["maxcol"] < (1 << 63)


def sqr_sets(W):
    """bigint29.hpp montsqr: K_t as (k, doubled) pairs — the position itself undoubled, plus a tournament on the other positions"""
    H = W // 2
    K = []
    for t in range(W):
        ks = [(t, False)]
        for k in range(W):
            d = (k - t) % W
            take = (1 <= d <= H) if W & 1 else (1 <= d < H or (d == H and t < H))
            if take:
                ks.append((k, True))
        K.append(ks)
    return K


def montsqr(N, G, W, A, stats, check=True):
    """mirror of montsqr<G>: the Orup product of A with itself in which sub-step s multiplies only the limbs of K_(s mod W)"""
    K = sqr_sets(W)
    c = [[0] * W for _ in range(G)]
    for s in range(G):
        for t in range(W):
            b = A[s * W + t]
            for j in range(G):
                for k, dbl in K[t]:
                    c[j][(t + k) % W] += A[j * W + k] * (b + b if dbl else b)
                    stats["mads"] = stats.get("mads", 0) + 1
            q = c[0][t] & MASK
            lo = [0] * G
            for j in range(G):
                for k in range(W):
                    c[j][(t + k) % W] += N[j * W + k] * q
                    stats["mads"] = stats.get("mads", 0) + 1
                stats["maxcol"] = max(stats["maxcol"], max(c[j]))
                v = c[j][t]
                lo[j] = v & MASK
                c[j][(t + 1) % W] += v >> B
                stats["maxcol"] = max(stats["maxcol"], c[j][(t + 1) % W])
            assert not check or lo[0] == 0
            for j in range(G):
                c[j][t] = lo[j + 1] if j + 1 < G else 0
    out, carries = [], []
    for j in range(G):
        cy, r = 0, []
        for k in range(W):
            v = c[j][k] + cy
            r.append(v & MASK)
            cy = v >> B
        out.append(r)
        carries.append(cy)
    for j in range(1, G):
        out[j][0] += carries[j - 1]
        assert not check or out[j][0] < (1 << B) + 64
    assert not check or carries[G - 1] == 0
    return [v for r in out for v in r]


@pytest.mark.parametrize("W", [36, 18, 9])
def test_squaring_sets_form_a_tournament(W):
    """every unordered pair of DIFFERENT limb positions is taken by exactly one of its two sub-steps (and doubled); a position
    takes itself, undoubled: ordered pairs of limbs at equal positions keep both orders"""
    K = sqr_sets(W)
    for t in range(W):
        assert (t, False) in K[t] and all(dbl for k, dbl in K[t] if k != t)
        for k in range(W):
            if k != t:
                assert ((k, True) in K[t]) != ((t, True) in K[k])
    sizes = sorted(len(ks) for ks in K)
    assert sizes[0] >= (W + 1) // 2 and sizes[-1] <= W // 2 + 1


@pytest.mark.parametrize("bits,G,W", [(4096, 4, 36), (2048, 2, 36), (8192, 8, 36), (4096, 8, 18), (4096, 16, 9)])
def test_squaring_model_equals_the_product(bits, G, W):
    """montsqr(X) is montmul(X, X) on the Orup multiple: same value (here even limb for limb), 3/4 of the multiply-adds"""
    rnd = random.Random(bits + 7 * G)
    M = rnd.getrandbits(bits) | 1 | (1 << (bits - 1))
    L = G * W
    R = 1 << (B * L)
    n1 = (-pow(M, -1, 1 << B)) % (1 << B)
    Mt = M * n1
    N = to_limbs(Mt, L)
    Rinv = pow(R, -1, M)
    stats = {"maxcol": 0}
    x = to_limbs(rnd.randrange(2 * Mt), L)
    for _ in range(4):                                   # a chain of squarings: almost-normalised outputs fed straight back
        stats["mads"] = 0
        r = montsqr(N, G, W, x, stats)
        ref = montmul(N, n1, G, W, x, x, True, {"maxcol": 0}, safe=False)
        assert from_limbs(r) == from_limbs(ref) and from_limbs(r) % M == from_limbs(x) ** 2 * Rinv % M and from_limbs(r) < 2 * Mt
        x = r
    assert stats["maxcol"] < (1 << 64)
    assert 0.75 <= stats["mads"] / (2 * L * L) <= 0.78


@pytest.mark.parametrize("G,W", [(4, 36), (2, 36)])
def test_squaring_keeps_the_fast_column_bound(G, W):
    """operand limbs at their maximum and a modulus operand exactly at COL_FAST_SN_LIMIT: over a column's life the doubled and
    single products of a lane are at most 2 * 18 (or 2 * 17 + 2) limb products, the same total as the W products of montmul"""
    lim = fast_sn_limit(W)
    lane = [MASK] * 27 + [lim - 27 * MASK] + [0] * (W - 28)
    stats = {"maxcol": 0}
    montsqr(lane * G, G, W, worst_operands(G, W), stats, check=False)
    assert (1 << 63) < stats["maxcol"] < (1 << 64)
    K = sqr_sets(W)
    for col in range(W):                                 # multiplicity of limb products landing in one column of one lane
        mult = sum((2 if dbl else 1) for t in range(W) for k, dbl in K[t] if (t + k) % W == col)
        assert mult == W


def montsqr2(N, G, W, A, stats):
    """mirror of montsqr2<G> (bigint29.hpp): montmul2's schedule — both digits of a pair of sub-steps first — over the products the
    squaring tournament keeps"""
    K = sqr_sets(W)
    mult = {(t, k): (2 if dbl else 1) for t in range(W) for k, dbl in K[t]}
    c = [[0] * W for _ in range(G)]

    def finish(t):
        lo = [0] * G
        for j in range(G):
            v = c[j][t % W]
            lo[j] = v & MASK
            c[j][(t + 1) % W] += v >> B
        assert lo[0] == 0
        for j in range(G):
            c[j][t % W] = lo[j + 1] if j + 1 < G else 0

    def mads(j, tt, b, cols):
        for k in range(W):
            if (tt + k) in cols and (tt, k) in mult:
                c[j][(tt + k) % W] += A[j * W + k] * b * mult[(tt, k)]
                stats["mads"] = stats.get("mads", 0) + 1

    for s in range(G):
        t = 0
        while t + 1 < W:
            b0, b1 = A[s * W + t], A[s * W + t + 1]
            for j in range(G):
                mads(j, t, b0, (t, t + 1)); mads(j, t + 1, b1, (t + 1,))
            q0 = c[0][t % W] & MASK
            q1 = (c[0][(t + 1) % W] + (c[0][t % W] >> B)) & MASK
            for j in range(G):
                mads(j, t, b0, range(t + 2, t + W)); mads(j, t + 1, b1, range(t + 2, t + W))
                for k in range(W):
                    c[j][(t + k) % W] += N[j * W + k] * q0
                for k in range(W - 1):
                    c[j][(t + 1 + k) % W] += N[j * W + k] * q1
            finish(t)
            for j in range(G):
                mads(j, t + 1, b1, (t + W,))
                c[j][t % W] += N[j * W + W - 1] * q1
            finish(t + 1)
            stats["maxcol"] = max(stats["maxcol"], max(max(r) for r in c))
            t += 2
        if W & 1:
            tt = W - 1
            for j in range(G):
                mads(j, tt, A[s * W + tt], range(tt, tt + W))
            q = c[0][tt] & MASK
            for j in range(G):
                for k in range(W):
                    c[j][(tt + k) % W] += N[j * W + k] * q
            finish(tt)
    out, carries = [], []
    for j in range(G):
        cy, r = 0, []
        for k in range(W):
            v = c[j][k] + cy
            r.append(v & MASK)
            cy = v >> B
        out.append(r)
        carries.append(cy)
    for j in range(1, G):
        out[j][0] += carries[j - 1]
    assert carries[G - 1] == 0
    return [v for r in out for v in r]


@pytest.mark.parametrize("bits,G,W", [(4096, 16, 9), (8192, 32, 9), (900, 4, 9)])
def test_double_digit_squaring_model(bits, G, W):
    """montsqr2(X) == montmul2(X, X) (values; 5 + 9 instead of 9 + 9 multiply-adds per lane per sub-step), no column near 2^64"""
    rnd = random.Random(bits * 3 + G)
    M = rnd.getrandbits(bits) | 1 | (1 << (bits - 1))
    L = G * W
    n2 = (-pow(M, -1, 1 << (2 * B))) % (1 << (2 * B))
    Mt2 = M * n2
    assert bits + 2 * B + 3 <= B * L
    N = to_limbs(Mt2, L)
    stats = {"maxcol": 0}
    x = to_limbs(rnd.randrange(2 * Mt2), L)
    for _ in range(4):
        stats["mads"] = 0
        r = montsqr2(N, G, W, x, stats)
        assert from_limbs(r) == from_limbs(montmul2(N, G, W, x, x, {"maxcol": 0})) and from_limbs(r) < 2 * Mt2
        x = r
    assert stats["maxcol"] < (1 << 63)
    assert stats["mads"] == L * G * 5                    # the A half: 5 of 9 limbs per lane per sub-step
