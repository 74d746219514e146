// legosnark_amd/csrc/smul_host.h -- HOST-side scalar multiplication of an ARBITRARY base (no device code).
//
// The reference's `Fr * G` on a point that is not a generator (the evaluation / opening terms of
// /root/reference/src/gadgets/poly.h:100-125, the commitment checks of src/prototools/commit.h:150-170: each base is met
// once or twice, so no table pays) ran 254 doublings + 63 general additions on a 4-bit window.  Here:
//
//   k = k1 + k2 lambda (mod r), |k1|, |k2| < 2^127      (glv.h: glv_decompose)
//   k P = k1 P + k2 phi(P),  phi(x, y) = (beta x, y)
//
// with both halves in width-5 NAF (odd digits in [-15, 15]: one addition per six bits on average) over the eight odd
// multiples P, 3P .. 15P and their images under phi (one product by beta each), sharing ONE chain of 127 doublings.  The
// table is built with co-Z additions, which leave all eight multiples on ONE Z: read as affine points of the isomorphic
// curve y^2 = x^3 + b Z^6 (neither the doubling nor the addition formulas involve b), they take MIXED additions, and the
// result's Z is multiplied by the common Z at the end -- no inversion.  About 1450 field products against 2950.  phi acts on G2 too (the twist y^2 = x^3 + b / xi has the same
// automorphism); which cube root of unity it is there is settled once, numerically, by `glv_beta<F>()` -- the candidate
// that fails lambda G = phi(G) on the group's generator is replaced by its square, and the process aborts if neither holds.
// The same group element as double-and-add (both groups have prime order r).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "ec.h"
#include "fp29.h"
#include "glv.h"
#include "tower.h"

namespace lsa {

template <class F> struct GlvGenerator;
template <> struct GlvGenerator<Fq> {
    static Jac<Fq> get() { const Fq one = Fq::one(); return {one, one + one, one}; }
};
template <> struct GlvGenerator<Fq2> {
    static Jac<Fq2> get() { return {fq2_const(LSA_G2_GEN_X), fq2_const(LSA_G2_GEN_Y), Fq2::one()}; }
};

static inline Fq glv_scale(const Fq &x, const Fq &b) { return x * b; }
static inline Fq2 glv_scale(const Fq2 &x, const Fq &b) { return {x.c0 * b, x.c1 * b}; }

// beta in Fq with phi(P) = (beta x, y) = lambda P on the group over F
template <class F>
static const Fq &glv_beta() {
    static const Fq beta = [] {
        constexpr uint32_t BETA29[9] = {0x0a337995u, 0x158d1d23u, 0x189c9b98u, 0x12fa4e45u, 0x185faadcu,       // smul.h, curves.h
                                        0x0176f16du, 0x0eed93bau, 0x14291140u, 0x000c0afeu};
        Fq b = F29::from_limbs(BETA29).to_mont256();
        const Jac<F> G = GlvGenerator<F>::get();
        constexpr uint32_t lam[8] = LSA_GLV_LAMBDA;
        Jac<F> want = Jac<F>::inf();
        for (int i = 255; i >= 0; --i) {
            want = jac_dbl(want);
            if ((lam[i >> 5] >> (i & 31)) & 1) want = jac_add(want, G);
        }
        for (int attempt = 0; attempt < 2; attempt++) {
            if (jac_eq(Jac<F>{glv_scale(G.X, b), G.Y, G.Z}, want)) return b;
            b = b.sqr();
        }
        fprintf(stderr, "legosnark_amd: no cube root of unity matches lambda on this group\n");
        abort();
        return b;
    }();
    return beta;
}

// width-5 NAF of m < 2^127: digits d[i] odd in [-15, 15] or 0, at most 128 of them; returns their count
static inline int wnaf5(unsigned __int128 m, int8_t d[130]) {
    int n = 0;
    while (m) {
        int v = 0;
        if (m & 1) {
            v = (int)(m & 31);
            if (v > 16) v -= 32;
            if (v > 0) m -= (unsigned)v; else m += (unsigned)(-v);
        }
        d[n++] = (int8_t)v;
        m >>= 1;
    }
    return n;
}

// Mixed addition (madd-2007-bl, 7M + 4S) of a Jacobian point and an affine one; complete.  Valid on every curve
// y^2 = x^3 + b' (the formulas, like jac_dbl's, do not involve b').
template <class F>
static inline Jac<F> jac_madd(const Jac<F> &a, const F &x2, const F &y2) {
    if (a.is_inf()) return {x2, y2, F::one()};
    const F Z1Z1 = a.Z.sqr();
    const F U2 = x2 * Z1Z1, S2 = y2 * (a.Z * Z1Z1);
    const F H = U2 - a.X, rh = S2 - a.Y;
    if (H.is_zero()) {
        if (rh.is_zero()) return jac_dbl(Jac<F>{x2, y2, F::one()});
        return Jac<F>::inf();
    }
    const F HH = H.sqr();
    const F I = HH.dbl().dbl();
    const F J = H * I;
    const F r = rh.dbl();
    const F V = a.X * I;
    const F X3 = r.sqr() - J - V.dbl();
    const F Y3 = r * (V - X3) - (a.Y * J).dbl();
    const F Z3 = (a.Z + H).sqr() - Z1Z1 - HH;
    return {X3, Y3, Z3};
}

// The eight odd multiples of P on a COMMON Z (co-Z additions, Meloni 2007): x[i], y[i] are the affine coordinates of
// (2i + 1) P on the isomorphic curve y^2 = x^3 + b zc^6, i.e. (2i + 1) P = (x[i], y[i], zc) in Jacobian form on the
// curve itself.  64M + 26S against 7 general additions (119 products) for a table whose entries then take MIXED
// additions.  false: a difference of abscissae vanished (P of small order -- never a point of the prime-order groups).
template <class F>
static inline bool odd_multiples_coz(const Jac<F> &P, F x[8], F y[8], F &zc) {
    // 2P together with P on 2P's Z: lambda = 2Y, X lambda^2 = 4 X Y^2 = S and Y lambda^3 = 8 Y^4 are values of the doubling
    const F A = P.X.sqr(), B = P.Y.sqr(), C = B.sqr();
    const F S = ((P.X + B).sqr() - A - C).dbl();
    const F E = A.dbl() + A;
    F dx = E.sqr() - S.dbl();
    const F C8 = C.dbl().dbl().dbl();
    F dy = E * (S - dx) - C8;
    F z = (P.Y * P.Z).dbl();
    if (z.is_zero()) return false;
    x[0] = S;
    y[0] = C8;
    F h[8];                                                   // Z of entry i = Z of entry i - 1 times h[i]
    for (int i = 1; i < 8; i++) {
        // (dx, dy) = 2P and entry i - 1 share z: entry i = their sum, 2P re-expressed on the sum's Z
        const F hx = dx - x[i - 1], ry = dy - y[i - 1];
        if (hx.is_zero()) return false;
        const F Cc = hx.sqr();
        const F W1 = dx * Cc, W2 = x[i - 1] * Cc;
        const F A1 = dy * (W1 - W2);
        x[i] = ry.sqr() - W1 - W2;
        y[i] = ry * (W1 - x[i]) - A1;
        dx = W1;
        dy = A1;
        h[i] = hx;
        z = z * hx;
    }
    zc = z;
    F lam = h[7];                                             // entry i scales by h[i + 1] ... h[7]
    for (int i = 6; i >= 0; --i) {
        const F l2 = lam.sqr();
        x[i] = x[i] * l2;
        y[i] = y[i] * (l2 * lam);
        if (i) lam = lam * h[i];
    }
    return true;
}

// k: canonical little-endian limbs of a scalar below r.  coz = false forces the degenerate-base route (tests).
template <class F>
static Jac<F> glv_mul_host(const Jac<F> &P, const uint64_t k[4], bool coz = true) {
    if (P.is_inf()) return P;
    uint32_t k32[8];
    for (int i = 0; i < 4; i++) { k32[2 * i] = (uint32_t)k[i]; k32[2 * i + 1] = (uint32_t)(k[i] >> 32); }
    const GlvSplit s = glv_decompose(k32);
    auto u128 = [](const uint32_t w[4]) {
        return (unsigned __int128)w[0] | ((unsigned __int128)w[1] << 32) | ((unsigned __int128)w[2] << 64) | ((unsigned __int128)w[3] << 96);
    };
    int8_t n1[130], n2[130];
    const int l1 = wnaf5(u128(s.k1), n1), l2 = wnaf5(u128(s.k2), n2);
    const int len = l1 > l2 ? l1 : l2;
    if (!len) return Jac<F>::inf();
    const Fq &beta = glv_beta<F>();
    F x[8], y1[8], ux[8], y1n[8], zc;
    if (coz && odd_multiples_coz(P, x, y1, zc)) {
        // the ladder runs on the isomorphic curve where the table is affine; the signs of the two halves go into
        // the ordinates once.  y1: ordinates for k1's positive digits, y2 = +-y1 for k2's
        for (int i = 0; i < 8; i++) {
            ux[i] = glv_scale(x[i], beta);
            y1n[i] = y1[i].neg();
        }
        const F *yp1 = s.neg1 ? y1n : y1, *ym1 = s.neg1 ? y1 : y1n;
        const F *yp2 = s.neg2 ? y1n : y1, *ym2 = s.neg2 ? y1 : y1n;
        Jac<F> R = Jac<F>::inf();
        for (int i = len - 1; i >= 0; --i) {
            R = jac_dbl(R);
            const int a = i < l1 ? n1[i] : 0, b = i < l2 ? n2[i] : 0;
            if (a > 0) R = jac_madd(R, x[a >> 1], yp1[a >> 1]); else if (a < 0) R = jac_madd(R, x[(-a) >> 1], ym1[(-a) >> 1]);
            if (b > 0) R = jac_madd(R, ux[b >> 1], yp2[b >> 1]); else if (b < 0) R = jac_madd(R, ux[(-b) >> 1], ym2[(-b) >> 1]);
        }
        if (R.is_inf()) return R;
        return {R.X, R.Y, R.Z * zc};
    }
    // degenerate base (small order): general additions over a Jacobian table
    Jac<F> T[8], U[8];
    const Jac<F> D = jac_dbl(P);
    T[0] = P;
    for (int i = 1; i < 8; i++) T[i] = jac_add(T[i - 1], D);
    for (int i = 0; i < 8; i++) {
        U[i] = {glv_scale(T[i].X, beta), s.neg2 ? T[i].Y.neg() : T[i].Y, T[i].Z};
        if (s.neg1) T[i].Y = T[i].Y.neg();
    }
    Jac<F> R = Jac<F>::inf();
    for (int i = len - 1; i >= 0; --i) {
        R = jac_dbl(R);
        const int a = i < l1 ? n1[i] : 0, b = i < l2 ? n2[i] : 0;
        if (a > 0) R = jac_add(R, T[a >> 1]); else if (a < 0) R = jac_add(R, jac_neg(T[(-a) >> 1]));
        if (b > 0) R = jac_add(R, U[b >> 1]); else if (b < 0) R = jac_add(R, jac_neg(U[(-b) >> 1]));
    }
    return R;
}

}  // namespace lsa
