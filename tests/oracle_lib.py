"""ctypes binding of oracle/liboracle_bn254.so for the tests, smoke() and bench.py's
cpu_baseline leg.  TEST INFRASTRUCTURE: never imported by legosnark_amd/.

All buffers are numpy uint64 arrays in libff layout (Montgomery LE limbs):
Fr/Fq: (..., 4); G1: (..., 12) = X|Y|Z; G2: (..., 24); Fq12: (..., 48).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
MONT = 1 << 256


def build(force=False):
    so = os.path.join(_ORACLE_DIR, "liboracle_bn254.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.oracle_bdlo12_window.restype = C.c_size_t
        _LIB.oracle_bdlo12_window.argtypes = [C.c_size_t]
        _LIB.oracle_libff_log2.restype = C.c_size_t
        _LIB.oracle_libff_log2.argtypes = [C.c_size_t]
        for g in ("g1", "g2"):
            getattr(_LIB, "oracle_%s_exp_window_size" % g).restype = C.c_size_t
            getattr(_LIB, "oracle_%s_exp_window_size" % g).argtypes = [C.c_size_t]
    return _LIB


def _p(a):
    assert a.dtype in (np.uint64, np.uint32) and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------ int <-> limb helpers
def int_to_limbs(x):
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def limbs_to_int(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1)
    return sum(int(a[i]) << (64 * i) for i in range(a.size))


def fq_mont(x):
    return int_to_limbs(x % P * MONT % P)


def fr_mont(x):
    return int_to_limbs(x % R * MONT % R)


def fr_mont_array(xs):
    out = np.zeros((len(xs), 4), dtype=np.uint64)
    for i, x in enumerate(xs):
        out[i] = fr_mont(x)
    return out


def g1_from_affine(pt, z=1):
    """Affine canonical (x,y) or None -> libff Jacobian Montgomery limbs (12,).
    z != 1 scales to (z^2 x, z^3 y, z) to exercise un-normalised inputs."""
    out = np.zeros(12, dtype=np.uint64)
    if pt is None:
        out[4:8] = fq_mont(1)  # libff zero = (0, 1, 0)
        return out
    x, y = pt
    out[0:4] = fq_mont(x * z * z)
    out[4:8] = fq_mont(y * z * z * z)
    out[8:12] = fq_mont(z)
    return out


def g2_from_affine(pt, z=(1, 0)):
    out = np.zeros(24, dtype=np.uint64)
    if pt is None:
        out[8:12] = fq_mont(1)
        return out

    def mul(a, b):
        return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)

    z2 = mul(z, z)
    z3 = mul(z2, z)
    X = mul(pt[0], z2)
    Y = mul(pt[1], z3)
    for k, v in enumerate((X[0], X[1], Y[0], Y[1], z[0], z[1])):
        out[4 * k:4 * k + 4] = fq_mont(v)
    return out


def g1_array(pts, zs=None):
    out = np.zeros((len(pts), 12), dtype=np.uint64)
    for i, pt in enumerate(pts):
        out[i] = g1_from_affine(pt, 1 if zs is None else zs[i])
    return out


def g2_array(pts, zs=None):
    out = np.zeros((len(pts), 24), dtype=np.uint64)
    for i, pt in enumerate(pts):
        out[i] = g2_from_affine(pt, (1, 0) if zs is None else zs[i])
    return out


# ------------------------------------------------------------ oracle calls
def g1_canonical_affine(pt):
    """(12,) Jacobian -> None or (x, y) canonical ints."""
    pt = np.ascontiguousarray(pt, dtype=np.uint64)
    out = np.zeros(8, dtype=np.uint64)
    inf = lib().og1_canonical_affine(_p(out), _p(pt))
    return None if inf else (limbs_to_int(out[0:4]), limbs_to_int(out[4:8]))


def g2_canonical_affine(pt):
    pt = np.ascontiguousarray(pt, dtype=np.uint64)
    out = np.zeros(16, dtype=np.uint64)
    inf = lib().og2_canonical_affine(_p(out), _p(pt))
    if inf:
        return None
    v = [limbs_to_int(out[4 * k:4 * k + 4]) for k in range(4)]
    return ((v[0], v[1]), (v[2], v[3]))


def _binop(name, width):
    def f(a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        r = np.zeros(width, dtype=np.uint64)
        getattr(lib(), name)(_p(r), _p(a), _p(b))
        return r
    return f


g1_add = _binop("og1_add", 12)
g1_mixed_add = _binop("og1_mixed_add", 12)
g2_add = _binop("og2_add", 24)
g2_mixed_add = _binop("og2_mixed_add", 24)
g1_mul = _binop("og1_mul", 12)   # (point, scalar_mont)
g2_mul = _binop("og2_mul", 24)
fq12_mul = _binop("ofq12_mul", 48)


def g1_dbl(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    r = np.zeros(12, dtype=np.uint64)
    lib().og1_dbl(_p(r), _p(a))
    return r


def g2_dbl(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    r = np.zeros(24, dtype=np.uint64)
    lib().og2_dbl(_p(r), _p(a))
    return r


def multi_exp(group, bases, scalars, chunks=1, threads=0, mode="mixed"):
    """mode: 'inner' | 'multi_exp' | 'mixed' (multi_exp_with_mixed_addition)."""
    w = 12 if group == "g1" else 24
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, w)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    n = min(len(bases), len(scalars))
    r = np.zeros(w, dtype=np.uint64)
    L = lib()
    if mode == "inner":
        getattr(L, "oracle_%s_multi_exp_inner" % group)(_p(r), _p(bases), _p(scalars), C.c_size_t(n))
    elif mode == "multi_exp":
        getattr(L, "oracle_%s_multi_exp" % group)(_p(r), _p(bases), _p(scalars), C.c_size_t(n),
                                                    C.c_size_t(chunks), C.c_int(threads))
    else:
        getattr(L, "oracle_%s_multi_exp_with_mixed_addition" % group)(
            _p(r), _p(bases), _p(scalars), C.c_size_t(n), C.c_size_t(chunks), C.c_int(threads))
    return r


def batch_exp(group, base, scalars, window=None):
    w = 12 if group == "g1" else 24
    base = np.ascontiguousarray(base, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    n = len(scalars)
    if window is None:
        window = getattr(lib(), "oracle_%s_exp_window_size" % group)(n)
    out = np.zeros((n, w), dtype=np.uint64)
    getattr(lib(), "oracle_%s_batch_exp" % group)(_p(out), _p(base), _p(scalars), C.c_size_t(n), C.c_size_t(window))
    return out


def g1_mul_batch(pts, scalars):
    """out[i] = scalars[i] * pts[i] with libff's scalar * point (og1_mul)."""
    pts = np.ascontiguousarray(pts, dtype=np.uint64).reshape(-1, 12)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros_like(pts)
    for i in range(len(pts)):
        lib().og1_mul(_p(out[i]), _p(pts[i]), _p(scalars[i]))
    return out


def mtxmultiexp(vals, rows, col_ptr, exps):
    vals = np.ascontiguousarray(vals, dtype=np.uint64).reshape(-1, 12)
    rows = np.ascontiguousarray(rows, dtype=np.uint32)
    col_ptr = np.ascontiguousarray(col_ptr, dtype=np.uint64)
    exps = np.ascontiguousarray(exps, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros((len(col_ptr) - 1, 12), dtype=np.uint64)
    lib().oracle_g1_mtxmultiexp(_p(out), _p(vals), _p(rows), _p(col_ptr), C.c_size_t(len(col_ptr) - 1), _p(exps))
    return out


def fr_cppoly_witness(v, r):
    v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
    w = np.zeros_like(v)
    lib().oracle_fr_cppoly_witness(_p(w), _p(v), _p(r), C.c_size_t(len(r)))
    return w


def fr_eval_mle(v, r):
    v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(4, dtype=np.uint64)
    lib().oracle_fr_eval_mle(_p(out), _p(v), _p(r), C.c_size_t(len(r)))
    return out


def fr_push_randomness(old, r):
    old = np.ascontiguousarray(old, dtype=np.uint64).reshape(-1, 4)
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
    cur = np.zeros((len(old) // 2, 4), dtype=np.uint64)
    lib().oracle_fr_push_randomness(_p(cur), _p(old), _p(r), C.c_size_t(len(old) // 2))
    return cur


def fr_sumcheck_round(tables, suff=None, pre=None, rho_j=None):
    tables = [np.ascontiguousarray(t, dtype=np.uint64).reshape(-1, 4) for t in tables]
    m, half = len(tables), len(tables[0]) // 2
    ptrs = (C.c_void_p * m)(*[t.ctypes.data for t in tables])
    out = np.zeros((m + (2 if rho_j is not None else 1), 4), dtype=np.uint64)
    sp = _p(np.ascontiguousarray(suff, dtype=np.uint64)) if suff is not None else None
    pp = _p(np.ascontiguousarray(pre, dtype=np.uint64)) if pre is not None else None
    rp = _p(np.ascontiguousarray(rho_j, dtype=np.uint64)) if rho_j is not None else None
    lib().oracle_fr_sumcheck_round(_p(out), sp, ptrs, C.c_size_t(m), C.c_size_t(half), pp, rp)
    return out


def fr_scale_upper(old, k):
    old = np.ascontiguousarray(old, dtype=np.uint64).reshape(-1, 4)
    cur = np.zeros((len(old) // 2, 4), dtype=np.uint64)
    lib().oracle_fr_scale_upper(_p(cur), _p(old), _p(np.ascontiguousarray(k, dtype=np.uint64)), C.c_size_t(len(old) // 2))
    return cur


def fr_eq_table(r):
    """DPBeta::compute_eq_tbl (mle.h:93-105): 2^len(r) entries."""
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros((1 << len(r), 4), dtype=np.uint64)
    lib().oracle_fr_eq_table(_p(out), _p(r), C.c_size_t(len(r)))
    return out


FR_TWO_ADICITY = 28
FR_GENERATOR = 5          # libff alt_bn128 Fr::multiplicative_generator


def fr_root_of_unity(log_n):
    """Python int: a primitive 2^log_n-th root of unity of Fr, libff get_root_of_unity's choice
    (root_of_unity = generator^((r-1)/2^s), squared s - log_n times)."""
    w = pow(FR_GENERATOR, (R - 1) >> FR_TWO_ADICITY, R)
    for _ in range(FR_TWO_ADICITY - log_n):
        w = w * w % R
    return w


def fr_domain_transform(a, omega, inverse=False, coset=None):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4).copy()
    log_n = len(a).bit_length() - 1
    cg = _p(np.ascontiguousarray(coset, dtype=np.uint64)) if coset is not None else None
    lib().oracle_fr_domain_transform(_p(a), C.c_size_t(log_n), _p(np.ascontiguousarray(omega, dtype=np.uint64)),
                                     C.c_int(1 if inverse else 0), cg)
    return a


def fr_step_domain_transform(a, big_log, small_log, omega, inverse=False, coset=None):
    """libfqfft step_radix2_domain FFT / iFFT / cosetFFT / icosetFFT on 2^big_log + 2^small_log values."""
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4).copy()
    assert len(a) == (1 << big_log) + (1 << small_log)
    cg = _p(np.ascontiguousarray(coset, dtype=np.uint64)) if coset is not None else None
    lib().oracle_fr_step_domain_transform(_p(a), C.c_size_t(big_log), C.c_size_t(small_log), _p(np.ascontiguousarray(omega, dtype=np.uint64)),
                                          C.c_int(1 if inverse else 0), cg)
    return a


def reduced_pairing(p, q):
    p = np.ascontiguousarray(p, dtype=np.uint64)
    q = np.ascontiguousarray(q, dtype=np.uint64)
    r = np.zeros(48, dtype=np.uint64)
    lib().oracle_reduced_pairing(_p(r), _p(p), _p(q))
    return r


G2_PRECOMP_WORDS = (2 + 3 * 102) * 8      # og2_precomp_t: QX, QY, 102 x {ell_0, ell_VW, ell_VV}


def precompute_g2(q):
    """libff alt_bn128_ate_precompute_G2 (oracle restatement) -> (G2_PRECOMP_WORDS,) uint64."""
    q = np.ascontiguousarray(q, dtype=np.uint64)
    r = np.zeros(G2_PRECOMP_WORDS, dtype=np.uint64)
    lib().oracle_precompute_g2(_p(r), _p(q))
    return r


def miller_loop_precomp(p, qpre):
    """libff miller_loop(precompute_G1(p), qpre)."""
    p = np.ascontiguousarray(p, dtype=np.uint64)
    qpre = np.ascontiguousarray(qpre, dtype=np.uint64)
    pp = np.zeros(8, dtype=np.uint64)
    lib().oracle_precompute_g1(_p(pp), _p(p))
    r = np.zeros(48, dtype=np.uint64)
    lib().oracle_miller_loop(_p(r), _p(pp), _p(qpre))
    return r


def fq12_unitary_inverse(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    r = np.zeros(48, dtype=np.uint64)
    lib().ofq12_unitary_inverse(_p(r), _p(a))
    return r


def fq12_inverse(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    r = np.zeros(48, dtype=np.uint64)
    lib().ofq12_inverse(_p(r), _p(a))
    return r


def fq12_frobenius(a, power):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    r = np.zeros(48, dtype=np.uint64)
    lib().ofq12_frobenius(_p(r), _p(a), C.c_uint(power))
    return r


def miller_loop_batch(ps, qs):
    ps = np.ascontiguousarray(ps, dtype=np.uint64).reshape(-1, 12)
    qs = np.ascontiguousarray(qs, dtype=np.uint64).reshape(-1, 24)
    out = np.zeros((len(ps), 48), dtype=np.uint64)
    lib().oracle_miller_loop_batch(_p(out), _p(ps), _p(qs), C.c_size_t(len(ps)))
    return out


def final_exponentiation(f):
    f = np.ascontiguousarray(f, dtype=np.uint64)
    r = np.zeros(48, dtype=np.uint64)
    lib().oracle_final_exponentiation(_p(r), _p(f))
    return r


def pairing_product(ps, qs):
    ps = np.ascontiguousarray(ps, dtype=np.uint64).reshape(-1, 12)
    qs = np.ascontiguousarray(qs, dtype=np.uint64).reshape(-1, 24)
    r = np.zeros(48, dtype=np.uint64)
    lib().oracle_pairing_product(_p(r), _p(ps), _p(qs), C.c_size_t(len(ps)))
    return r


def fq12_one():
    r = np.zeros(48, dtype=np.uint64)
    lib().ofq12_one(_p(r))
    return r


def fq12_mul(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    r = np.zeros(48, dtype=np.uint64)
    lib().ofq12_mul(_p(r), _p(a), _p(b))
    return r


def fq12_product(fs):
    acc = fq12_one()
    for f in np.asarray(fs, dtype=np.uint64).reshape(-1, 48):
        acc = fq12_mul(acc, f)
    return acc


def fq12_to_model(f):
    """libff tower limbs (48,) Montgomery -> model's 6 Fp2 poly coefficients (canonical)."""
    f = np.asarray(f, dtype=np.uint64).reshape(12, 4)
    Rinv = pow(MONT, -1, P)
    v = [limbs_to_int(f[i]) * Rinv % P for i in range(12)]
    # order: c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2 (each Fp2 = 2 Fp)
    fp2 = [(v[2 * i], v[2 * i + 1]) for i in range(6)]
    a0, a1, a2, b0, b1, b2 = fp2
    return [a0, b0, a1, b1, a2, b2]


def fq12_from_model(g):
    a0, b0, a1, b1, a2, b2 = g
    out = np.zeros(48, dtype=np.uint64)
    for i, c in enumerate((a0, a1, a2, b0, b1, b2)):
        out[8 * i:8 * i + 4] = fq_mont(c[0])
        out[8 * i + 4:8 * i + 8] = fq_mont(c[1])
    return out


def arith_bases(group, a, b, n):
    """out[i] = (a + i*b) * generator as un-normalised Jacobian points (n, 12|24)."""
    w = 12 if group == "g1" else 24
    out = np.zeros((n, w), dtype=np.uint64)
    am, bm = fr_mont(a), fr_mont(b)
    getattr(lib(), "oracle_%s_arith_bases" % group)(_p(out), _p(am), _p(bm), C.c_size_t(n))
    return out


def fr_dot(a, b):
    """sum_i a[i]*b[i] in Fr (Montgomery limbs in and out as a python int field value)."""
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    assert len(a) == len(b)
    out = np.zeros(4, dtype=np.uint64)
    lib().oracle_fr_dot(_p(out), _p(a), _p(b), C.c_size_t(len(a)))
    return limbs_to_int(out) * pow(MONT, -1, R) % R


def generator(group):
    w = 12 if group == "g1" else 24
    out = np.zeros(w, dtype=np.uint64)
    getattr(lib(), "o%s_one" % group)(_p(out))
    return out


def random_scalars(n, seed, bits=254):
    """Deterministic scalars uniform in [0, r) (or < 2^bits) as Montgomery limbs (n,4) + python ints."""
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 1 << 63, size=(n, 5), dtype=np.uint64)
    ints = []
    out = np.zeros((n, 4), dtype=np.uint64)
    for i in range(n):
        x = 0
        for j in range(5):
            x = (x << 63) | int(raw[i, j])
        x &= (1 << bits) - 1
        x %= R
        ints.append(x)
        out[i] = fr_mont(x)
    return out, ints
