"""Independent big-int model of alt_bn128 (BN254) used ONLY to emit golden fixtures.

TEST INFRASTRUCTURE -- not product code.  Written from the curve equations, not
from libff: affine coordinates, schoolbook polynomial arithmetic for Fp12
(Fp2[w]/(w^6 - xi)), textbook optimal-ate Miller loop with affine slopes and a
generic square-and-multiply final exponentiation by the integer exponent that
libff's addition chain realises.  It is deliberately a *different* algorithm
from both the C restatement in oracle/bn254.c (Jacobian / Pippenger / tower
2-3-2 / projective line coefficients) and the HIP kernels, so agreement between
them pins the results.

Constants: SURVEY.md section 8 header (verified numerically there).
"""

P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
U = 4965661367192848881  # BN parameter z
ATE_LOOP = 6 * U + 2     # 29793968203157093288
B1 = 3
MONT_R = 1 << 256

G1_GEN = (1, 2)
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)

assert P == 36 * U**4 + 36 * U**3 + 24 * U**2 + 6 * U + 1
assert R == 36 * U**4 + 36 * U**3 + 18 * U**2 + 6 * U + 1


def inv(a, m=P):
    return pow(a, -1, m)


# ---------------------------------------------------------------- Fp2 = Fp[u]/(u^2+1)
def f2(a, b=0):
    return (a % P, b % P)


F2_ZERO = (0, 0)
F2_ONE = (1, 0)
XI = (9, 1)  # non-residue for the sextic twist / Fp6 / Fp12


def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_sqr(a):
    return f2_mul(a, a)


def f2_scalar(a, k):
    return ((a[0] * k) % P, (a[1] * k) % P)


def f2_inv(a):
    n = inv((a[0] * a[0] + a[1] * a[1]) % P)
    return ((a[0] * n) % P, (-a[1] * n) % P)


def f2_conj(a):
    return (a[0], (-a[1]) % P)


def f2_pow(a, e):
    r = F2_ONE
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_sqr(a)
        e >>= 1
    return r


TWIST_B = f2_mul((3, 0), f2_inv(XI))  # b' = 3 / xi


# ---------------------------------------------------------------- Fp12 = Fp2[w]/(w^6 - xi)
# element = list of 6 Fp2 coefficients g0..g5 (g_i * w^i)
def f12_one():
    return [F2_ONE] + [F2_ZERO] * 5


def f12_mul(a, b):
    t = [F2_ZERO] * 11
    for i in range(6):
        if a[i] == F2_ZERO:
            continue
        for j in range(6):
            if b[j] == F2_ZERO:
                continue
            t[i + j] = f2_add(t[i + j], f2_mul(a[i], b[j]))
    out = list(t[:6])
    for k in range(6, 11):
        out[k - 6] = f2_add(out[k - 6], f2_mul(t[k], XI))
    return out


def f12_sqr(a):
    return f12_mul(a, a)


def f12_pow(a, e):
    r = f12_one()
    nb = e.bit_length()
    for i in range(nb - 1, -1, -1):
        r = f12_sqr(r)
        if (e >> i) & 1:
            r = f12_mul(r, a)
    return r


def f12_conj6(a):
    """a^(q^6): w -> -w."""
    return [a[i] if i % 2 == 0 else f2_neg(a[i]) for i in range(6)]


def f12_inv(a):
    # a^{-1} = a^(q^12 - 2); slow but independent of any tower formula.
    return f12_pow(a, P**12 - 2)


def f12_to_libff_tower(a):
    """Fp12 poly coefficients -> libff Fp12_2over3over2 (c0=(g0,g2,g4), c1=(g1,g3,g5))."""
    return ((a[0], a[2], a[4]), (a[1], a[3], a[5]))


# ---------------------------------------------------------------- curves (affine, None = infinity)
def g1_is_on_curve(p):
    if p is None:
        return True
    x, y = p
    return (y * y - x * x * x - B1) % P == 0


def g1_neg(p):
    return None if p is None else (p[0], (-p[1]) % P)


def g1_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = (3 * x1 * x1) * inv(2 * y1) % P
    else:
        lam = (y2 - y1) * inv(x2 - x1) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def g1_mul(p, k):
    k %= R
    acc = None
    add = p
    while k:
        if k & 1:
            acc = g1_add(acc, add)
        add = g1_add(add, add)
        k >>= 1
    return acc


def g2_is_on_curve(p):
    if p is None:
        return True
    x, y = p
    return f2_sub(f2_sqr(y), f2_add(f2_mul(f2_sqr(x), x), TWIST_B)) == F2_ZERO


def g2_neg(p):
    return None if p is None else (p[0], f2_neg(p[1]))


def g2_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if f2_add(y1, y2) == F2_ZERO:
            return None
        lam = f2_mul(f2_scalar(f2_sqr(x1), 3), f2_inv(f2_scalar(y1, 2)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_sqr(lam), x1), x2)
    y3 = f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1)
    return (x3, y3)


def g2_mul(p, k):
    k %= R
    acc = None
    add = p
    while k:
        if k & 1:
            acc = g2_add(acc, add)
        add = g2_add(add, add)
        k >>= 1
    return acc


def msm(points, scalars, add, mul):
    acc = None
    for pt, s in zip(points, scalars):
        acc = add(acc, mul(pt, s))
    return acc


# ---------------------------------------------------------------- pairing (textbook optimal ate)
def _embed_fp(x):
    return [(x % P, 0)] + [F2_ZERO] * 5


def _line(T, Q, Pt):
    """Line through untwisted T,Q (points on the twist, affine Fp2) evaluated at
    Pt in G1, scaled by w^3 (a factor in Fp4 that the final exponentiation kills):
        l = xi*(lam*xT - yT) + yP * w^3 - lam*xP * w^4
    Returns (value, T+Q)."""
    xT, yT = T
    xQ, yQ = Q
    xP, yP = Pt
    if xT == xQ and yT == yQ:
        lam = f2_mul(f2_scalar(f2_sqr(xT), 3), f2_inv(f2_scalar(yT, 2)))
    else:
        assert xT != xQ
        lam = f2_mul(f2_sub(yQ, yT), f2_inv(f2_sub(xQ, xT)))
    out = [F2_ZERO] * 6
    out[0] = f2_mul(XI, f2_sub(f2_mul(lam, xT), yT))
    out[3] = (yP % P, 0)
    out[4] = f2_neg(f2_scalar(lam, xP))
    x3 = f2_sub(f2_sub(f2_sqr(lam), xT), xQ)
    y3 = f2_sub(f2_mul(lam, f2_sub(xT, x3)), yT)
    return out, (x3, y3)


def g2_frobenius(Q):
    """pi(x,y) on the twist: (conj(x)*xi^((p-1)/3), conj(y)*xi^((p-1)/2))."""
    gx = f2_pow(XI, (P - 1) // 3)
    gy = f2_pow(XI, (P - 1) // 2)
    return (f2_mul(f2_conj(Q[0]), gx), f2_mul(f2_conj(Q[1]), gy))


def miller_loop(Pt, Q):
    if Pt is None or Q is None:
        return f12_one()
    f = f12_one()
    T = Q
    nb = ATE_LOOP.bit_length()
    for i in range(nb - 2, -1, -1):
        l, T2 = _line(T, T, Pt)
        f = f12_mul(f12_sqr(f), l)
        T = T2
        if (ATE_LOOP >> i) & 1:
            l, T2 = _line(T, Q, Pt)
            f = f12_mul(f, l)
            T = T2
    Q1 = g2_frobenius(Q)
    Q2 = g2_neg(g2_frobenius(Q1))
    l, T2 = _line(T, Q1, Pt)
    f = f12_mul(f, l)
    T = T2
    l, T2 = _line(T, Q2, Pt)
    f = f12_mul(f, l)
    return f


def libff_final_exponent():
    """Integer exponent realised by libff's alt_bn128 final exponentiation:
    (q^6-1)(q^2+1) * [q^3(12z^3+6z^2+4z-1) + q^2(12z^3+6z^2+6z) + q(12z^3+6z^2+4z)
                      + (12z^3+12z^2+6z+1)]                       [upstream, recalled]
    The bracket equals 2z(6z^2+3z+1) * (q^4-q^2+1)/r (asserted below)."""
    z = U
    q = P
    l3 = 12 * z**3 + 6 * z**2 + 4 * z - 1
    l2 = 12 * z**3 + 6 * z**2 + 6 * z
    l1 = 12 * z**3 + 6 * z**2 + 4 * z
    l0 = 12 * z**3 + 12 * z**2 + 6 * z + 1
    hard = q**3 * l3 + q**2 * l2 + q * l1 + l0
    assert (q**4 - q**2 + 1) % R == 0
    assert hard == 2 * z * (6 * z**2 + 3 * z + 1) * ((q**4 - q**2 + 1) // R)
    return (q**6 - 1) * (q**2 + 1) * hard


FINAL_EXP = libff_final_exponent()


def final_exponentiation(f):
    return f12_pow(f, FINAL_EXP)


def reduced_pairing(Pt, Q):
    return final_exponentiation(miller_loop(Pt, Q))


# ---------------------------------------------------------------- libff helper rules
def libff_log2(n):
    """libff::log2 = ceil(log2(n)) with log2(0)=log2(1)=0   [upstream, recalled]."""
    r = 0 if (n & (n - 1)) == 0 else 1
    while n > 1:
        n >>= 1
        r += 1
    return r


def bdlo12_window(n):
    """c = L - (L/3 - 2) in size_t arithmetic (wraps for L<6)  [upstream, recalled]."""
    L = libff_log2(n)
    M = 1 << 64
    return (L - ((L // 3 - 2) % M)) % M


def to_mont(x, m=P):
    return (x * MONT_R) % m


def limbs_hex(x):
    """256-bit integer -> 64 hex chars, little-endian bytes (4 x u64 LE limbs)."""
    return x.to_bytes(32, "little").hex()
