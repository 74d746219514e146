"""tools/gen_fe_scalar_exponent.py -- the exponent that takes the Fq2 inversion out of a lone final exponentiation.

libff's alt_bn128_final_exponentiation (w12.h: W12::final_exponentiation is the same chain) inverts f once, in its easy
part: f^-1 = conj(f) * N6^-1, N6 = f conj(f) in Fq6, N6^-1 = N6^(q^2) N6^(q^4) / n2, n2 in Fq2, 1 / n2 = conj(n2) / n1,
n1 = n2.c0^2 + n2.c1^2 in Fq.  The division by n1 is a binary GCD of ~20 000 instructions on the critical path of a kernel
whose other ~270 chain links are ~580 instructions each ("A lone wavefront").

Leave the division out: every later value of the chain is then the true value times a power of n1 -- an element of Fq,
which Frobenius maps and conjugations fix and products multiply -- and the end result is  FE(f) * n1^(2 K)  with K the
chain's exponent count in which a conjugation (the chain's way to invert a unitary element) does NOT change the sign:
    FIRST' = FIRST * n1^2,   exp_by_neg_z(a * t) = exp_by_neg_z(a) * t^zabs   (zabs = sum |digit_i| 2^i of the chain's
    width-3 NAF of z),   products add exponents, squarings double them.
The correction  n1^e,  e = -2 K mod (q - 1),  is ONE exponentiation in Fq by a fixed 254-bit exponent, independent of the
chain: the fourth wavefront of the workgroup (its SIMD is idle otherwise) runs it right to left beside the chain, one
squaring and one product per chain link, and the last link scales the twelve components by it.

This script walks the chain symbolically (K), checks the identity numerically on the big-int model of oracle/pymodel
(test infrastructure: imported by this generator only) and prints the words of e for w12.h.
"""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "pymodel"))
import bn254_model as M  # noqa: E402

P = M.P
D_P1, D_P3, D_M1, D_M3 = 0x4800120040011001, 0x0000804004000000, 0x0000000000000010, 0x0108000400880200
assert D_P1 + 3 * D_P3 - D_M1 - 3 * D_M3 == M.U
ZABS = D_P1 + 3 * D_P3 + D_M1 + 3 * D_M3


class Sym:
    """exponent of the stray scalar carried by a value of the chain"""

    def __init__(self, e):
        self.e = e

    def __mul__(self, o):
        return Sym(self.e + o.e)

    def sqr(self):
        return Sym(2 * self.e)

    def conj(self):
        return Sym(self.e)

    def frob(self, k):
        return Sym(self.e)


class Num:
    """the same interface on the model's Fq12 values"""

    def __init__(self, v):
        self.v = v

    def __mul__(self, o):
        return Num(M.f12_mul(self.v, o.v))

    def sqr(self):
        return Num(M.f12_sqr(self.v))

    def conj(self):
        return Num(M.f12_conj6(self.v))

    def frob(self, k):
        return Num(M.f12_pow(self.v, P ** k))


def exp_by_neg_z(a):
    """w12.h: w12_exp_by_neg_z_rows -- conj(a^z) with inverses taken by conjugation"""
    a3 = a.sqr() * a
    na, na3 = a.conj(), a3.conj()
    acc = a
    for i in range(61, -1, -1):
        acc = acc.sqr()
        if (D_P1 >> i) & 1:
            acc = acc * a
        elif (D_P3 >> i) & 1:
            acc = acc * a3
        elif (D_M1 >> i) & 1:
            acc = acc * na
        elif (D_M3 >> i) & 1:
            acc = acc * na3
    return acc.conj()


def hard_part(first):
    """w12.h: W12::final_exponentiation from FIRST on"""
    A = exp_by_neg_z(first)
    B = A.sqr()
    C = B.sqr()
    D = C * B
    E = exp_by_neg_z(D)
    F = E.sqr()
    G = exp_by_neg_z(F)
    H = D.conj()
    I = G.conj()
    J = I * E
    K = J * H
    L = K * B
    Mv = K * E
    N = Mv * first
    O = L.frob(1)
    Pv = O * N
    Q = K.frob(2)
    Rv = Q * Pv
    S = first.conj()
    T = S * L
    Uv = T.frob(3)
    return Uv * Rv


def main():
    k = hard_part(Sym(1)).e
    e_fix = (-2 * k) % (P - 1)
    # numerically: a unitary m (the easy part of a random f), a random t in Fq
    rng = random.Random(20261003)
    f = [(rng.randrange(P), rng.randrange(P)) for _ in range(6)]
    fi = M.f12_inv(f)
    c = M.f12_mul(M.f12_conj6(f), fi)
    m = M.f12_mul(M.f12_pow(c, P * P), c)
    assert M.f12_mul(m, M.f12_conj6(m)) == M.f12_one(), "m is unitary"
    t = rng.randrange(1, P)
    mt = [M.f2_scalar(x, t) for x in m]
    want = hard_part(Num(m)).v
    got = hard_part(Num(mt)).v
    tk = pow(t, k, P)
    assert got == [M.f2_scalar(x, tk) for x in want], "chain(m t) == chain(m) t^K"
    # the whole thing: FIRST' = FIRST * n1^2 with the division by n1 left out
    N6 = M.f12_mul(f, M.f12_conj6(f))
    n2 = M.f12_mul(M.f12_mul(N6, M.f12_pow(N6, P ** 2)), M.f12_pow(N6, P ** 4))
    assert all(x == (0, 0) for x in n2[1:]), "the norm lies in Fq2"
    n1 = (n2[0][0] ** 2 + n2[0][1] ** 2) % P
    corr = pow(n1, e_fix, P)
    first_scaled = [M.f2_scalar(x, n1 * n1 % P) for x in m]
    res = [M.f2_scalar(x, corr) for x in hard_part(Num(first_scaled)).v]
    assert res == want, "the correction restores libff's value"
    assert want == M.final_exponentiation(f), "and that value is f^(libff's exponent)"
    words = [(e_fix >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
    print("// generated by tools/gen_fe_scalar_exponent.py: e = -2 K mod (q - 1), %d bits" % e_fix.bit_length())
    print("static constexpr uint32_t W12_FE_SCALAR_EXP[8] = {" + ", ".join("0x%08xu" % w for w in words) + "};")
    print("static constexpr int W12_FE_SCALAR_EXP_BITS = %d;" % e_fix.bit_length())


if __name__ == "__main__":
    main()
