"""Executable statement of the BASE-n form of Paillier's arithmetic modulo n^2 (csrc/kernels_basen.hpp):

    x  (mod n^2)   is held as   (a, b)   with   x = a + b * n  (mod n^2),

so that a product needs only n-sized Montgomery products — (a1 + b1 n)(a2 + b2 n) = a1 a2 + (a1 b2 + a2 b1) n, the b1 b2 term is a
multiple of n^2 — and the quotient of a1 a2 by n, which belongs to the b part, falls out of the Montgomery reduction of a1 a2 itself:

    a1 a2 + Q M~ = a' R'      (Q: the quotient digits of the systolic product, M~ = n * n1 the Orup multiple, R' = 2^(29 L))
 => a1 a2 = a' R' - (Q n1) n
 => x1 x2 / R'  =  a'  +  [ (a1 b2 + a2 b1 - Q n1) / R'  mod n ] * n          (mod n^2).

Everything here is plain Python integers and digit lists; tests/test_basen_model.py checks it against pow(), and the GPU tests check the
kernels against it through the diagnostics entry point."""

LB = 29
B = 1 << LB
MASK = B - 1
W = 36


class BaseN:
    """per-key constants of the pair form, as k_setup_basen computes them"""

    def __init__(self, n, G):
        assert n & 1
        self.n, self.G = n, G
        self.L = L = G * W
        self.R = R = 1 << (LB * L)
        self.n1 = (-pow(n, -1, B)) % B
        self.Mt = n * self.n1                      # Orup multiple: == -1 mod 2^29
        assert self.Mt % B == MASK and 4 * self.Mt < R
        S = (R - 1) // (B - 1)                     # every digit 1
        self.C3 = (-(self.n1 * B * S)) % n         # makes sum (B - Q_i) n1 B^i + C3 == -Q n1 (mod n)
        self.one = (R % n, (R // n) % n)           # Montgomery form of 1: R' = rho0 + rho1 n
        rr = (R * R) % (n * n)
        self.RR = (rr % n, rr // n)                # Montgomery form of R'
        self.R2n = (R * R) % n                     # plain mod-n constant (canonicalisation of the two halves)

    def digits(self, x):
        return [(x >> (LB * i)) & MASK for i in range(self.L)]

    # ---- the n-sized Montgomery product on the Orup multiple, with an initial column value and the quotient digits
    def redc(self, T):
        """(T + Q M~) / R' and Q, for T >= 0"""
        Q = (T * pow(-self.Mt, -1, self.R)) % self.R
        assert (T + Q * self.Mt) % self.R == 0
        return (T + Q * self.Mt) // self.R, Q

    def q_term(self, Q):
        """what the b side adds for the a side's quotient: sum (B - Q_i) n1 B^i + C3 == -Q n1 (mod n), column by column non-negative"""
        return sum((B - d) * self.n1 << (LB * i) for i, d in enumerate(self.digits(Q))) + self.C3

    def mul(self, x, y):
        """bn_mul: x y / R' (mod n^2)"""
        (a1, b1), (a2, b2) = x, y
        a, Q = self.redc(a1 * a2)
        bb1, _ = self.redc(a1 * b2 + self.q_term(Q))
        bb2, _ = self.redc(b1 * a2)
        return a, bb1 + bb2

    def sqr(self, x):
        a1, b1 = x
        a, Q = self.redc(a1 * a1)
        b, _ = self.redc(2 * a1 * b1 + self.q_term(Q))
        return a, b

    def value(self, x):
        return (x[0] + x[1] * self.n) % (self.n * self.n)

    def to_mont(self, r):
        return self.mul((r, 0), self.RR)

    def finish(self, xm, m):
        """Montgomery pair of y -> canonical (a0, bf) of y (1 + m n): multiply by the PLAIN pair (1, m), then split a' = a0 + k n"""
        a, b = self.mul((1, m), xm)
        a0 = a % self.n
        k = (a - a0) // self.n
        assert k < B and k == ((a0 - a) * self.n1) % B
        bf = (b + k) % self.n
        return a0, bf

    def enc(self, m, r, window_script=None):
        """(1 + m n) r^n mod n^2 by square-and-multiply on pairs"""
        x0 = self.to_mont(r)
        acc = None
        for bit in bin(self.n)[2:]:
            if acc is not None:
                acc = self.sqr(acc)
            if bit == "1":
                acc = x0 if acc is None else self.mul(x0, acc)
        a0, bf = self.finish(acc, m)
        return a0 + bf * self.n
