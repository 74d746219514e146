// legosnark_amd/csrc/tower.h -- Fq6 = Fq2[v]/(v^3 - xi), Fq12 = Fq6[w]/(w^2 - v), xi = 9+u:
// the extension tower of alt_bn128's target group, laid out exactly like libff's
// Fp6_3over2 / Fp12_2over3over2 (GT = 384 B: c0.{c0,c1,c2}, c1.{c0,c1,c2}, each Fq2).
// Shared by the pairing kernels (pairing.hip) and the host side of the shim.
// Field results are canonical Montgomery residues, so they are bit-identical to libff's
// whatever multiplication schedule is used.
#pragma once
#include "fp.h"

namespace lsa {

LSA_HD Fq2 fq2_const(const uint32_t (&c)[2][8]) {
    Fq2 r;
#pragma unroll
    for (int i = 0; i < 8; i++) { r.c0.l[i] = c[0][i]; r.c1.l[i] = c[1][i]; }
    return r;
}

struct Fq6 {
    Fq2 c0, c1, c2;
    static LSA_HD Fq6 zero() { return {Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
    static LSA_HD Fq6 one() { return {Fq2::one(), Fq2::zero(), Fq2::zero()}; }
    LSA_HD bool operator==(const Fq6 &b) const { return c0 == b.c0 && c1 == b.c1 && c2 == b.c2; }
    friend LSA_HD Fq6 operator+(const Fq6 &a, const Fq6 &b) { return {a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2}; }
    friend LSA_HD Fq6 operator-(const Fq6 &a, const Fq6 &b) { return {a.c0 - b.c0, a.c1 - b.c1, a.c2 - b.c2}; }
    LSA_HD Fq6 neg() const { return {c0.neg(), c1.neg(), c2.neg()}; }
    // (c0, c1, c2) * v = (xi*c2, c0, c1)
    LSA_HD Fq6 mul_by_v() const { return {c2.mul_xi(), c0, c1}; }
    LSA_HD Fq6 mul_fq2(const Fq2 &k) const { return {c0 * k, c1 * k, c2 * k}; }
};

// Karatsuba-style 3-term product (6 Fq2 products)
LSA_HD_NOINLINE Fq6 fq6_mul(const Fq6 &a, const Fq6 &b) {
    Fq2 v0 = a.c0 * b.c0, v1 = a.c1 * b.c1, v2 = a.c2 * b.c2;
    Fq2 t0 = ((a.c1 + a.c2) * (b.c1 + b.c2) - v1 - v2).mul_xi() + v0;
    Fq2 t1 = (a.c0 + a.c1) * (b.c0 + b.c1) - v0 - v1 + v2.mul_xi();
    Fq2 t2 = (a.c0 + a.c2) * (b.c0 + b.c2) - v0 - v2 + v1;
    return {t0, t1, t2};
}
LSA_HD Fq6 operator*(const Fq6 &a, const Fq6 &b) { return fq6_mul(a, b); }

LSA_HD_NOINLINE Fq6 fq6_inverse(const Fq6 &a) {
    Fq2 t0 = a.c0.sqr(), t1 = a.c1.sqr(), t2 = a.c2.sqr();
    Fq2 t3 = a.c0 * a.c1, t4 = a.c0 * a.c2, t5 = a.c1 * a.c2;
    Fq2 c0 = t0 - t5.mul_xi();
    Fq2 c1 = t2.mul_xi() - t3;
    Fq2 c2 = t1 - t4;
    Fq2 t6 = (a.c0 * c0 + (a.c2 * c1 + a.c1 * c2).mul_xi()).inverse();
    return {c0 * t6, c1 * t6, c2 * t6};
}

template <int POWER>
LSA_HD Fq2 fq2_frobenius(const Fq2 &a) {
    if (POWER & 1) return a.conj();
    return a;
}
template <int POWER>
LSA_HD Fq6 fq6_frobenius(const Fq6 &a) {
    return {fq2_frobenius<POWER>(a.c0), fq2_frobenius<POWER>(a.c1) * fq2_const(LSA_FROB6_C1[POWER % 6]),
            fq2_frobenius<POWER>(a.c2) * fq2_const(LSA_FROB6_C2[POWER % 6])};
}

struct Fq12 {
    Fq6 c0, c1;
    static LSA_HD Fq12 one() { return {Fq6::one(), Fq6::zero()}; }
    LSA_HD bool operator==(const Fq12 &b) const { return c0 == b.c0 && c1 == b.c1; }
    LSA_HD bool operator!=(const Fq12 &b) const { return !(*this == b); }
    LSA_HD Fq12 unitary_inverse() const { return {c0, c1.neg()}; }
};

LSA_HD_NOINLINE Fq12 fq12_mul(const Fq12 &a, const Fq12 &b) {
    Fq6 aa = a.c0 * b.c0, bb = a.c1 * b.c1;
    Fq6 s = (a.c0 + a.c1) * (b.c0 + b.c1);
    return {aa + bb.mul_by_v(), s - aa - bb};
}
LSA_HD Fq12 operator*(const Fq12 &a, const Fq12 &b) { return fq12_mul(a, b); }

// complex squaring: (c0 + c1 w)^2 = (c0^2 + v c1^2) + 2 c0 c1 w
LSA_HD_NOINLINE Fq12 fq12_sqr(const Fq12 &a) {
    Fq6 ab = a.c0 * a.c1;
    Fq6 t = (a.c0 + a.c1) * (a.c0 + a.c1.mul_by_v()) - ab - ab.mul_by_v();
    return {t, ab + ab};
}

LSA_HD_NOINLINE Fq12 fq12_inverse(const Fq12 &a) {
    Fq6 t = fq6_inverse(a.c0 * a.c0 - (a.c1 * a.c1).mul_by_v());
    return {a.c0 * t, (a.c1 * t).neg()};
}

template <int POWER>
LSA_HD_NOINLINE Fq12 fq12_frobenius(const Fq12 &a) {
    return {fq6_frobenius<POWER>(a.c0), fq6_frobenius<POWER>(a.c1).mul_fq2(fq2_const(LSA_FROB12_C1[POWER % 12]))};
}

// libff Fp12::mul_by_024: a * (ell_0 + ell_VV v^2 + ell_VW v w), i.e. the sparse element
// Fp12(Fp6(ell_0, 0, ell_VV), Fp6(0, ell_VW, 0))  (13 Fq2 products instead of 18)
LSA_HD_NOINLINE Fq12 fq12_mul_by_024(const Fq12 &a, const Fq2 &e0, const Fq2 &eVW, const Fq2 &eVV) {
    // aa = a.c0 * (e0, 0, eVV)
    Fq2 a0e0 = a.c0.c0 * e0, a2eV = a.c0.c2 * eVV;
    Fq6 aa = {a0e0 + (a.c0.c1 * eVV).mul_xi(), a.c0.c1 * e0 + a2eV.mul_xi(), (a.c0.c0 + a.c0.c2) * (e0 + eVV) - a0e0 - a2eV};
    // bb = a.c1 * (0, eVW, 0)
    Fq6 bb = {(a.c1.c2 * eVW).mul_xi(), a.c1.c0 * eVW, a.c1.c1 * eVW};
    // s = (a.c0 + a.c1) * (e0, eVW, eVV)
    Fq6 s = fq6_mul(a.c0 + a.c1, Fq6{e0, eVW, eVV});
    return {aa + bb.mul_by_v(), s - aa - bb};
}

LSA_HD_NOINLINE Fq12 fq12_pow_u64(const Fq12 &a, uint64_t e) {
    Fq12 acc = Fq12::one();
    bool started = false;
    for (int i = 63; i >= 0; --i) {
        if (started) acc = fq12_sqr(acc);
        if ((e >> i) & 1) {
            acc = started ? fq12_mul(acc, a) : a;
            started = true;
        }
    }
    return acc;
}

}  // namespace lsa
