"""The product of the five-wavefront-per-Enc kernel (csrc/kernels_basen_r2l.hpp: r2l5::product, one role = one wavefront = ONE lane group of
36 lanes x 2 limbs), stated lane by lane and instruction by instruction: the window of a lane is `bot` (the column that completes in this
sub-step) and `inn` (the column that opens with the limb shifted in from the neighbour lane, the addend of its first multiply-add); ONE
quotient digit per sub-step, read off lane 0 (v_readfirstlane: the digit is wave-uniform, every lane multiplies by it); the digits of
sub-steps 2 s and 2 s + 1 are parked in lane s of a register pair (v_writelane) and leave as the lane's two words of the digit area — the
layout B and D load Q in.  Checked against the n-sized Montgomery product on values (tests/basen_model.py): same result, same digits, every
column accumulator below 2^64, lanes 36 - 63 zero throughout."""
import random

import pytest

from basen_model import BaseN, B, LB, MASK

LANES, RW, RG, L = 64, 2, 36, 72
U64 = (1 << 64) - 1


def limbs(x):
    return [(x >> (LB * i)) & MASK for i in range(L)]


def product_lanes(X, Bst, Mt, cinit, capture=True):
    """X, Mt: 72 limbs each (lane j holds limbs 2 j, 2 j + 1; lanes >= 36 hold zeros); Bst: the staged operand's 72 limbs, broadcast reads;
    cinit: 72 initial column values (C3_i + (2^29 - Q_i) n1, or zeros).  -> (result limbs per lane, digit area words, max column)"""
    X0 = [X[2 * j] if j < RG else 0 for j in range(LANES)]
    X1 = [X[2 * j + 1] if j < RG else 0 for j in range(LANES)]
    N0 = [Mt[2 * j] if j < RG else 0 for j in range(LANES)]
    N1 = [Mt[2 * j + 1] if j < RG else 0 for j in range(LANES)]
    bot = [cinit[2 * j] if j < RG else 0 for j in range(LANES)]
    inn = [cinit[2 * j + 1] if j < RG else 0 for j in range(LANES)]
    qa, qb = [0] * LANES, [0] * LANES
    peak = 0
    for t in range(L):
        b = Bst[t]
        bot = [bot[j] + X0[j] * b for j in range(LANES)]                  # v_mad_u64_u32 on the accumulator
        top = [X1[j] * b + inn[j] for j in range(LANES)]                  # ... with the opening column as the addend
        q = bot[0] & MASK                                                  # v_and, v_readfirstlane: lane 0's bottom limb (M~ == -1 mod 2^29)
        if capture:
            (qa if t % 2 == 0 else qb)[t // 2] = q                         # v_writelane, lane t / 2
        bot = [bot[j] + N0[j] * q for j in range(LANES)]
        top = [top[j] + N1[j] * q for j in range(LANES)]
        peak = max(peak, max(bot), max(top))
        assert bot[0] & MASK == 0                                          # the digit has made lane 0's bottom limb zero
        top = [top[j] + (bot[j] >> LB) for j in range(LANES)]             # v_lshrrev_b64, v_lshl_add_u64
        peak = max(peak, max(top))
        # v_and_b32_dpp wave_shl:1 — lane j takes lane j + 1's low limb (lane 63: zero); lane 35 takes lane 36's, a zero
        inn = [(bot[j + 1] & MASK) if j + 1 < LANES else 0 for j in range(LANES)]
        bot = top
    assert peak <= U64
    assert all(v == 0 for v in bot[RG:]) and all(v == 0 for v in inn[RG:])
    R = []
    for j in range(LANES):
        t0 = bot[j]
        r0 = t0 & MASK
        t0 = inn[j] + (t0 >> LB)
        R.append([r0, t0 & MASK, t0 >> LB])
    # R[0] += the carry of the lane below (v_mov_b32_dpp wave_shr:1; lane 0 and the idle lanes take zero)
    out = []
    for j in range(RG):
        carry = R[j - 1][2] if j > 0 else 0
        out += [R[j][0] + carry, R[j][1]]
    assert R[RG - 1][2] == 0                                               # the value fits its 72 limbs
    area = []
    for s in range(RG):
        area += [qa[s], qb[s]]                                             # lane s stores its pair at words 2 s, 2 s + 1: digit t at word t
    return out, area, peak


def value_of(limbs_):
    return sum(v << (LB * i) for i, v in enumerate(limbs_))


@pytest.mark.parametrize("bits", [2048, 2047, 1200])
def test_the_lane_level_product_is_the_montgomery_product_with_its_digits(bits):
    rnd = random.Random(bits)
    n = rnd.getrandbits(bits) | 1 | (1 << (bits - 1))
    m = BaseN(n, 2)
    assert m.L == L
    Mt = limbs(m.Mt)
    for trial in range(6):
        x = rnd.randrange(2 * m.Mt) if trial else 2 * m.Mt - 1           # operands of a ladder stay below 2 M~ (a side) ...
        y = rnd.randrange(4 * m.Mt) if trial else 4 * m.Mt - 1           # ... and 4 M~ (b side)
        if trial == 2:
            x = (1 << (LB * L)) - 1                                       # every limb of the register operand all ones (an unreduced r next to a short key: anything below R')
        # a side: no initial columns, digits kept
        r, area, peak = product_lanes(limbs(x), limbs(y), Mt, [0] * L, capture=True)
        want, Q = m.redc(x * y)
        assert value_of(r) == want
        assert area == m.digits(Q)
        # b side: starts from the columns C3_i + (2^29 - Q_i) n1 of the a side's digits
        c3 = limbs(m.C3)
        cinit = [c3[i] + (B - area[i]) * m.n1 for i in range(L)]
        r2, _, peak2 = product_lanes(limbs(y % (1 << (LB * L))), limbs(x), Mt, cinit, capture=False)
        want2, _ = m.redc((y % (1 << (LB * L))) * x + m.q_term(Q))
        assert value_of(r2) == want2
        assert max(peak, peak2) <= U64
