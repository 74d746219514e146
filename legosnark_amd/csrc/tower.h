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
// constants (generated as libff Montgomery limbs) in the representation B
template <class B> struct BaseConv {
    static LSA_HD B from(const Fq &v) { return B::from_mont256(v); }
};
template <> struct BaseConv<Fq> {
    static LSA_HD Fq from(const Fq &v) { return v; }
};
template <class B>
LSA_HD Fq2T<B> fq2_constT(const uint32_t (&c)[2][8]) {
    Fq2 r = fq2_const(c);
    return {BaseConv<B>::from(r.c0), BaseConv<B>::from(r.c1)};
}

template <class B>
struct Fq6T {
    using Fq2 = Fq2T<B>;
    using Fq6 = Fq6T<B>;
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

using Fq6 = Fq6T<Fq>;

// Karatsuba-style 3-term product (6 Fq2 products)
template <class B>
LSA_HD_NOINLINE Fq6T<B> fq6_mul(const Fq6T<B> &a, const Fq6T<B> &b) {
    using Fq2 = Fq2T<B>;
    Fq2 v0 = a.c0 * b.c0, v1 = a.c1 * b.c1, v2 = a.c2 * b.c2;
    Fq2 t0 = ((a.c1 + a.c2) * (b.c1 + b.c2) - v1 - v2).mul_xi() + v0;
    Fq2 t1 = (a.c0 + a.c1) * (b.c0 + b.c1) - v0 - v1 + v2.mul_xi();
    Fq2 t2 = (a.c0 + a.c2) * (b.c0 + b.c2) - v0 - v2 + v1;
    return {t0, t1, t2};
}
template <class B>
LSA_HD Fq6T<B> operator*(const Fq6T<B> &a, const Fq6T<B> &b) { return fq6_mul(a, b); }

template <class B>
LSA_HD_NOINLINE Fq6T<B> fq6_inverse(const Fq6T<B> &a) {
    using Fq2 = Fq2T<B>;
    Fq2 t0 = a.c0.sqr(), t1 = a.c1.sqr(), t2 = a.c2.sqr();
    Fq2 t3 = a.c0 * a.c1, t4 = a.c0 * a.c2, t5 = a.c1 * a.c2;
    Fq2 c0 = t0 - t5.mul_xi();
    Fq2 c1 = t2.mul_xi() - t3;
    Fq2 c2 = t1 - t4;
    Fq2 t6 = (a.c0 * c0 + (a.c2 * c1 + a.c1 * c2).mul_xi()).inverse();
    return {c0 * t6, c1 * t6, c2 * t6};
}

template <int POWER, class B>
LSA_HD Fq2T<B> fq2_frobenius(const Fq2T<B> &a) {
    if (POWER & 1) return a.conj();
    return a;
}
template <int POWER, class B>
LSA_HD Fq6T<B> fq6_frobenius(const Fq6T<B> &a) {
    return {fq2_frobenius<POWER>(a.c0), fq2_frobenius<POWER>(a.c1) * fq2_constT<B>(LSA_FROB6_C1[POWER % 6]),
            fq2_frobenius<POWER>(a.c2) * fq2_constT<B>(LSA_FROB6_C2[POWER % 6])};
}

template <class B>
struct Fq12T {
    using Fq6 = Fq6T<B>;
    using Fq12 = Fq12T<B>;
    Fq6 c0, c1;
    static LSA_HD Fq12 one() { return {Fq6::one(), Fq6::zero()}; }
    LSA_HD bool operator==(const Fq12 &b) const { return c0 == b.c0 && c1 == b.c1; }
    LSA_HD bool operator!=(const Fq12 &b) const { return !(*this == b); }
    LSA_HD Fq12 unitary_inverse() const { return {c0, c1.neg()}; }
};

using Fq12 = Fq12T<Fq>;

template <class B>
LSA_HD_NOINLINE Fq12T<B> fq12_mul(const Fq12T<B> &a, const Fq12T<B> &b) {
    using Fq6 = Fq6T<B>;
    Fq6 aa = a.c0 * b.c0, bb = a.c1 * b.c1;
    Fq6 s = (a.c0 + a.c1) * (b.c0 + b.c1);
    return {aa + bb.mul_by_v(), s - aa - bb};
}
template <class B>
LSA_HD Fq12T<B> operator*(const Fq12T<B> &a, const Fq12T<B> &b) { return fq12_mul(a, b); }

// complex squaring: (c0 + c1 w)^2 = (c0^2 + v c1^2) + 2 c0 c1 w
template <class B>
LSA_HD_NOINLINE Fq12T<B> fq12_sqr(const Fq12T<B> &a) {
    using Fq6 = Fq6T<B>;
    Fq6 ab = a.c0 * a.c1;
    Fq6 t = (a.c0 + a.c1) * (a.c0 + a.c1.mul_by_v()) - ab - ab.mul_by_v();
    return {t, ab + ab};
}

template <class B>
LSA_HD_NOINLINE Fq12T<B> fq12_inverse(const Fq12T<B> &a) {
    using Fq6 = Fq6T<B>;
    Fq6 t = fq6_inverse(a.c0 * a.c0 - (a.c1 * a.c1).mul_by_v());
    return {a.c0 * t, (a.c1 * t).neg()};
}

template <int POWER, class B>
LSA_HD_NOINLINE Fq12T<B> fq12_frobenius(const Fq12T<B> &a) {
    return {fq6_frobenius<POWER>(a.c0), fq6_frobenius<POWER>(a.c1).mul_fq2(fq2_constT<B>(LSA_FROB12_C1[POWER % 12]))};
}

// libff Fp12::mul_by_024: a * (ell_0 + ell_VV v^2 + ell_VW v w), i.e. the sparse element
// Fp12(Fp6(ell_0, 0, ell_VV), Fp6(0, ell_VW, 0))  (13 Fq2 products instead of 18)
template <class B>
LSA_HD_NOINLINE Fq12T<B> fq12_mul_by_024(const Fq12T<B> &a, const Fq2T<B> &e0, const Fq2T<B> &eVW, const Fq2T<B> &eVV) {
    using Fq2 = Fq2T<B>;
    using Fq6 = Fq6T<B>;
    // aa = a.c0 * (e0, 0, eVV)
    Fq2 a0e0 = a.c0.c0 * e0, a2eV = a.c0.c2 * eVV;
    Fq6 aa = {a0e0 + (a.c0.c1 * eVV).mul_xi(), a.c0.c1 * e0 + a2eV.mul_xi(), (a.c0.c0 + a.c0.c2) * (e0 + eVV) - a0e0 - a2eV};
    // bb = a.c1 * (0, eVW, 0)
    Fq6 bb = {(a.c1.c2 * eVW).mul_xi(), a.c1.c0 * eVW, a.c1.c1 * eVW};
    // s = (a.c0 + a.c1) * (e0, eVW, eVV)
    Fq6 s = fq6_mul(a.c0 + a.c1, Fq6{e0, eVW, eVV});
    return {aa + bb.mul_by_v(), s - aa - bb};
}

// a^2 for a in the cyclotomic subgroup (a^(q^6+1) = 1), Granger-Scott: three Fq4 squarings,
// 6 Fq2 products instead of 12.  Same value as fq12_sqr on such elements (libff
// cyclotomic_squared; checked in tests/cpp/test_tower29.cc).
template <class B>
LSA_HD_NOINLINE Fq12T<B> fq12_cyclotomic_sqr(const Fq12T<B> &a) {
    using Fq2 = Fq2T<B>;
    Fq2 z0 = a.c0.c0, z4 = a.c0.c1, z3 = a.c0.c2, z2 = a.c1.c0, z1 = a.c1.c1, z5 = a.c1.c2;
    Fq2 tmp = z0 * z1;
    Fq2 t0 = (z0 + z1) * (z0 + z1.mul_xi()) - tmp - tmp.mul_xi();
    Fq2 t1 = tmp + tmp;
    tmp = z2 * z3;
    Fq2 t2 = (z2 + z3) * (z2 + z3.mul_xi()) - tmp - tmp.mul_xi();
    Fq2 t3 = tmp + tmp;
    tmp = z4 * z5;
    Fq2 t4 = (z4 + z5) * (z4 + z5.mul_xi()) - tmp - tmp.mul_xi();
    Fq2 t5 = tmp + tmp;
    z0 = t0 - z0; z0 = z0 + z0; z0 = z0 + t0;          // 3 t0 - 2 z0
    z1 = t1 + z1; z1 = z1 + z1; z1 = z1 + t1;          // 3 t1 + 2 z1
    tmp = t5.mul_xi();
    z2 = tmp + z2; z2 = z2 + z2; z2 = z2 + tmp;        // 3 xi t5 + 2 z2
    z3 = t4 - z3; z3 = z3 + z3; z3 = z3 + t4;          // 3 t4 - 2 z3
    z4 = t2 - z4; z4 = z4 + z4; z4 = z4 + t2;          // 3 t2 - 2 z4
    z5 = t3 + z5; z5 = z5 + z5; z5 = z5 + t3;          // 3 t3 + 2 z5
    return {{z0, z4, z3}, {z2, z1, z5}};
}

template <class B>
LSA_HD_NOINLINE Fq12T<B> fq12_pow_u64(const Fq12T<B> &a, uint64_t e) {
    using Fq12 = Fq12T<B>;
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
// a^e for a in the cyclotomic subgroup (libff cyclotomic_exp)
template <class B>
LSA_HD_NOINLINE Fq12T<B> fq12_cyclotomic_pow_u64(const Fq12T<B> &a, uint64_t e) {
    Fq12T<B> acc = Fq12T<B>::one();
    bool started = false;
    for (int i = 63; i >= 0; --i) {
        if (started) acc = fq12_cyclotomic_sqr(acc);
        if ((e >> i) & 1) {
            acc = started ? fq12_mul(acc, a) : a;
            started = true;
        }
    }
    return acc;
}

// elt^(-z) for elt in the cyclotomic subgroup (libff alt_bn128_exp_by_neg_z: cyclotomic_exp + conj)
template <class B>
LSA_HD_NOINLINE Fq12T<B> fq12_exp_by_neg_z(const Fq12T<B> &a) {
    return fq12_cyclotomic_pow_u64(a, LSA_FINAL_EXP_Z).unitary_inverse();
}

// libff alt_bn128_final_exponentiation: first chunk (q^6-1)(q^2+1), last chunk by the
// Fuentes-Castaneda et al. addition chain (squarings inside the chain are cyclotomic).
template <class B>
LSA_HD_NOINLINE Fq12T<B> fq12_final_exponentiation(const Fq12T<B> &elt) {
    using P12 = Fq12T<B>;
    P12 A = elt.unitary_inverse();
    P12 Bv = fq12_inverse(elt);
    P12 C = A * Bv;
    P12 D = fq12_frobenius<2>(C);
    P12 first = D * C;
    A = fq12_exp_by_neg_z(first);
    Bv = fq12_cyclotomic_sqr(A);
    C = fq12_cyclotomic_sqr(Bv);
    D = C * Bv;
    P12 E = fq12_exp_by_neg_z(D);
    P12 F = fq12_cyclotomic_sqr(E);
    P12 G = fq12_exp_by_neg_z(F);
    P12 H = D.unitary_inverse();
    P12 I = G.unitary_inverse();
    P12 J = I * E;
    P12 K = J * H;
    P12 L = K * Bv;
    P12 M = K * E;
    P12 N = M * first;
    P12 O = fq12_frobenius<1>(L);
    P12 Pp = O * N;
    P12 Qq = fq12_frobenius<2>(K);
    P12 Rr = Qq * Pp;
    P12 S = first.unitary_inverse();
    P12 T = S * L;
    P12 U = fq12_frobenius<3>(T);
    return U * Rr;
}

}  // namespace lsa
