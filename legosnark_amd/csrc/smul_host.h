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
#include "bn254_constants.h"
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

// ---------------------------------------------------------------------------------------------------------------
// G2 only: four-dimensional decomposition (Galbraith-Scott).  psi = twist o Frobenius o untwist,
//   psi(x, y) = (gamma_x conj(x), gamma_y conj(y))     (libff's mul_by_q: LSA_TWIST_MUL_BY_Q_X / _Y),
// acts on G2 as multiplication by mu = p mod r = 6 u^2 (u the BN parameter), so
//   k = k0 + k1 mu + k2 mu^2 + k3 mu^3 (mod r),  |k_i| < 2^66,
// and k P = sum k_i psi^i(P) over ONE chain of 66 doublings instead of 127: about 2 800 products in Fq against 3 500 for
// the two-dimensional split above.  The four tables must share ONE Z for mixed additions: the co-Z table's common Z is made
// REAL first (every entry scaled by conj(Z): the common Z becomes the norm, an element of Fq), and psi of (X, Y, N) with N
// in Fq is (gamma_x conj X, gamma_y conj Y, N) -- the same N.
// Decomposition: the rows of the Galbraith-Scott basis for BN curves,
//   (u+1, u, u, -2u), (2u+1, -u, -(u+1), -u), (2u, 2u+1, 2u+1, 2u+1), (u-1, 4u+2, -(2u-1), u-1),
// each orthogonal to (1, mu, mu^2, mu^3) modulo r, and Babai's rounding with the first row of the inverse as 2^256-scaled
// constants truncated towards zero: c_j = +-floor(k g_j / 2^256), k_i = k [i = 0] - sum_j c_j B[j][i].  The k_i are
// small (below 5.4 u < 2^65 on 2 * 10^5 random scalars; the bound of the rounding is 6u + 1), so the sums are taken modulo
// 2^128: only bits 256 .. 383 of the products k g_j are needed.  tests/cpp/test_smul.cc checks the identity
// sum k_i mu^i = k (mod r) in the scalar field and the products against double-and-add.
struct Gls4 { __int128 k[4]; };

static inline Gls4 gls4_decompose(const uint64_t k[4]) {
    typedef unsigned __int128 u128;
    constexpr __int128 U = (__int128)4965661367192848881ll;
    constexpr __int128 B[4][4] = {{U + 1, U, U, -2 * U}, {2 * U + 1, -U, -(U + 1), -U}, {2 * U, 2 * U + 1, 2 * U + 1, 2 * U + 1}, {U - 1, 4 * U + 2, -(2 * U - 1), U - 1}};
    constexpr uint64_t G[4][4] = {{0xd0cb46fd51906254ull, 0xc444fab18d269b9dull, 0x0000000000000000ull, 0x0000000000000000ull},
                                  {0x001378f5ee78976dull, 0x22df9f942d7d77c7ull, 0x3d00631561b25729ull, 0x0000000000000001ull},
                                  {0x36510546a93478abull, 0x916fcfca16bebbe4ull, 0x9e80318ab0d92b94ull, 0x0000000000000000ull},
                                  {0xf7ae23ce89afae7cull, 0xc444fab18d269b9aull, 0x0000000000000000ull, 0x0000000000000000ull}};
    constexpr int SIGN[4] = {1, 1, 1, -1};
    u128 c[4];
    for (int j = 0; j < 4; j++) {
        // limbs 4 and 5 of the 8-limb product k * G[j]
        uint64_t prod[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int a = 0; a < 4; a++) {
            uint64_t carry = 0;
            for (int b = 0; b < 4; b++) {
                const u128 t = (u128)k[a] * G[j][b] + prod[a + b] + carry;
                prod[a + b] = (uint64_t)t;
                carry = (uint64_t)(t >> 64);
            }
            prod[a + 4] = carry;
        }
        const u128 mag = (u128)prod[4] | ((u128)prod[5] << 64);
        c[j] = SIGN[j] > 0 ? mag : (u128)0 - mag;
    }
    Gls4 out;
    for (int i = 0; i < 4; i++) {
        u128 acc = i == 0 ? ((u128)k[0] | ((u128)k[1] << 64)) : (u128)0;
        for (int j = 0; j < 4; j++) acc -= c[j] * (u128)B[j][i];
        out.k[i] = (__int128)acc;
    }
    return out;
}

// psi on a Jacobian point whose Z lies in Fq (is its own conjugate)
static inline void gls_psi(const Fq2 &x, const Fq2 &y, Fq2 &px, Fq2 &py) {
    static const Fq2 gx = fq2_const(LSA_TWIST_MUL_BY_Q_X), gy = fq2_const(LSA_TWIST_MUL_BY_Q_Y);
    px = gx * x.conj();
    py = gy * y.conj();
}

// psi(G) = mu G on the generator of G2, mu = 6 u^2: checked once (false: callers take the two-dimensional path)
static inline bool gls4_ok() {
    static const bool ok = [] {
        const Jac<Fq2> G = GlvGenerator<Fq2>::get();
        const unsigned __int128 mu = (unsigned __int128)6 * 4965661367192848881ull * 4965661367192848881ull;
        Jac<Fq2> want = Jac<Fq2>::inf();
        for (int i = 127; i >= 0; --i) {
            want = jac_dbl(want);
            if ((mu >> i) & 1) want = jac_add(want, G);
        }
        Fq2 px, py;
        gls_psi(G.X, G.Y, px, py);                            // the generator is stored with Z = 1
        return G.Z == Fq2::one() && jac_eq(Jac<Fq2>{px, py, Fq2::one()}, want);
    }();
    return ok;
}

// k: canonical little-endian limbs of a scalar below r; P in G2 (the prime-order subgroup: psi acts as mu only there)
static Jac<Fq2> gls4_mul_host(const Jac<Fq2> &P, const uint64_t k[4]) {
    if (P.is_inf()) return P;
    Fq2 x[4][8], y[4][8], yn[4][8], zc;
    if (!gls4_ok() || !odd_multiples_coz(P, x[0], y[0], zc)) return glv_mul_host(P, k);
    const Gls4 s = gls4_decompose(k);
    int8_t naf[4][132];
    int len[4], top = 0;
    bool neg[4];
    for (int t = 0; t < 4; t++) {
        neg[t] = s.k[t] < 0;
        len[t] = wnaf5((unsigned __int128)(neg[t] ? -s.k[t] : s.k[t]), naf[t]);
        if (len[t] > top) top = len[t];
    }
    if (!top) return Jac<Fq2>::inf();
    // a real common Z: every entry times conj(zc)
    const Fq2 lam = zc.conj(), l2 = lam.sqr(), l3 = l2 * lam;
    const Fq N = zc.c0.sqr() + zc.c1.sqr();
    for (int i = 0; i < 8; i++) { x[0][i] = x[0][i] * l2; y[0][i] = y[0][i] * l3; }
    for (int t = 1; t < 4; t++)
        for (int i = 0; i < 8; i++) gls_psi(x[t - 1][i], y[t - 1][i], x[t][i], y[t][i]);
    for (int t = 0; t < 4; t++)
        for (int i = 0; i < 8; i++) {
            if (neg[t]) y[t][i] = y[t][i].neg();
            yn[t][i] = y[t][i].neg();
        }
    Jac<Fq2> R = Jac<Fq2>::inf();
    for (int i = top - 1; i >= 0; --i) {
        R = jac_dbl(R);
        for (int t = 0; t < 4; t++) {
            const int d = i < len[t] ? naf[t][i] : 0;
            if (d > 0) R = jac_madd(R, x[t][d >> 1], y[t][d >> 1]);
            else if (d < 0) R = jac_madd(R, x[t][(-d) >> 1], yn[t][(-d) >> 1]);
        }
    }
    if (R.is_inf()) return R;
    return {R.X, R.Y, Fq2{R.Z.c0 * N, R.Z.c1 * N}};
}

}  // namespace lsa
