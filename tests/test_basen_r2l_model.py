"""The five-group right-to-left ladder of the latency engine's one-Enc-per-wavefront kernel (csrc/kernels_basen_r2l.hpp), stated on values:
the slots exactly as the kernel runs them — A and C one product ahead of B, D and E, double-buffered by slot parity — on the n-sized
Montgomery product of tests/basen_model.py, against pow()."""
import random

import pytest

from basen_model import BaseN, B


def r2l_enc(m: BaseN, msg: int, r: int) -> int:
    n = m.n
    t = n.bit_length()
    bit = lambda k: 0 <= k < t and (n >> k) & 1

    def M(x, y, Q=None):
        """(x y [+ the quotient term of Q] + q M~) / R' and its own quotient digits"""
        return m.redc(x * y + (m.q_term(Q) if Q is not None else 0))

    one_a, one_b = m.one
    SA, SC, SE, DA, PC = [None, None], [None, None], [None, None], [None, None], [None, None]
    QA, QC = [None, None], [None, None]                 # the digits A / C left over their staged operand, by parity
    SB = m.RR[1]
    DA[1] = r
    SA[1] = m.RR[0]
    QQ = PX = RD = UU = None
    raw_a = raw_b = None
    for k in range(-1, t + 3):
        par, prev = k & 1, (k & 1) ^ 1
        fin1, fin2 = k == t + 1, k == t + 2
        actA, actB = k <= t - 2, 0 <= k <= t - 1
        actC, actD, actE = (1 <= k <= t - 1) or fin1, (2 <= k <= t) or fin2, (2 <= k <= t) or fin1
        res = {}
        # every product of the slot reads the state as it is at the START of the slot
        if actA:
            res["A"] = M(r if k < 0 else SA[par], SA[par])
        if actB:
            res["B"] = M(DA[prev], SB, QA[prev])[0]
        if actC:
            res["C"] = M(1, PX) if fin1 else M(PC[par], SC[par])
        if actD:
            res["D"] = M(1, QQ, QC_fin)[0] if fin2 else M(PC[prev], SB if bit(k - 1) else one_b, QC[prev])[0]
        if actE:
            res["E"] = M(msg, SE[t & 1])[0] if fin1 else M(QQ, SE[prev] if bit(k - 1) else one_a)[0]
        # ... and the results land at its end
        if actA:
            a_next, QA[par] = res["A"]
            SA[prev] = SE[prev] = a_next
            SC[prev] = a_next if bit(k + 1) else one_a
            if k < 0:
                PC[par] = a_next
            DA[prev] = 2 * a_next
        if actB:
            SB = res["B"]
            if k == 0:
                QQ = SB
        if actC:
            if fin1:
                raw_a, QC_fin = res["C"]
            else:
                PC[prev], QC[par] = res["C"]
                PX = PC[prev]
        if k == t:
            SE[t & 1] = PX
        if actE:
            if fin1:
                UU = res["E"]
            else:
                QQ = res["E"] + res["D"]
        if fin2:
            raw_b = res["D"] + UU
    # k_basen_finish: the raw pair -> the canonical residue
    a0 = raw_a % n
    kq = (raw_a - a0) // n
    bf = (raw_b + kq) % n
    return a0 + bf * n


@pytest.mark.parametrize("bits", [2048, 2047, 1300])
def test_five_group_ladder_equals_pow(bits):
    rnd = random.Random(bits)
    n = rnd.getrandbits(bits) | 1 | (1 << (bits - 1))
    m = BaseN(n, 2)
    nn = n * n
    for r, msg in ((rnd.getrandbits(2048), rnd.randrange(n)), (1, 0), (n - 1, n - 1), (0, 5)):
        assert r2l_enc(m, msg, r) == (1 + msg * n) * pow(r, n, nn) % nn
