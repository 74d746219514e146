// legosnark_amd/csrc/miller.h -- the optimal-ate Miller loop of alt_bn128 in two shapes:
//
//   miller_one()  one pairing per lane (k_miller, pairing.hip): libff's
//                 doubling_step_for_flipped_miller_loop / mixed_addition_step_for_flipped_miller_loop
//                 fused with the evaluation at P and the sparse product into f, so libff's
//                 ~20-KB G2_precomp table never exists.
//   WMiller       one pairing per WAVEFRONT on top of the W12 engine (w12.h): every round is one
//                 parallel Fq2-product phase -- 36 lanes for f*f or f*line, up to 6 more for
//                 the G2 point arithmetic and the line coefficients -- and one combine phase
//                 (anti-diagonal sums of f on 6 lanes, the point's additions on one lane).
//                 3 rounds per doubling step, 4 per addition step, ~10x shorter than the
//                 one-lane chain; what a verifier with a handful of pairings needs.
// Both follow libff's formulas value for value (the projective representative of the G2 point
// scales the lines, and miller_loop values are part of the API), so the results are the same
// canonical Fq12 elements.  Host + device; tests/cpp/test_w12.cc runs one against the other.
#pragma once
#include "ec.h"
#include "fs29.h"
#include "tower.h"
#include "w12.h"

namespace lsa {

using PB = Fs;
using P2 = Fq2T<PB>;
using P12 = Fq12T<PB>;

LSA_HD P2 load2(const Fq2 &v) { return {PB::from_mont256(v.c0), PB::from_mont256(v.c1)}; }

struct G2Proj { P2 X, Y, Z; };
struct Line { P2 e0, eVW, eVV; };

// libff doubling_step_for_flipped_miller_loop
LSA_HD_NOINLINE Line doubling_step(G2Proj &c, const PB &two_inv, const P2 &twist_b) {
    P2 X = c.X, Y = c.Y, Z = c.Z;
    P2 A = (X * Y).mul_fq(two_inv);
    P2 B = Y.sqr();
    P2 C = Z.sqr();
    P2 D = C + C + C;
    P2 E = twist_b * D;
    P2 F = E + E + E;
    P2 G = (B + F).mul_fq(two_inv);
    P2 H = (Y + Z).sqr() - (B + C);
    P2 I = E - B;
    P2 J = X.sqr();
    P2 E2 = E.sqr();
    c.X = A * (B - F);
    c.Y = G.sqr() - (E2 + E2 + E2);
    c.Z = B * H;
    return {I.mul_xi(), H.neg(), J + J + J};
}

// libff mixed_addition_step_for_flipped_miller_loop
LSA_HD_NOINLINE Line addition_step(const P2 &x2, const P2 &y2, G2Proj &c) {
    P2 X1 = c.X, Y1 = c.Y, Z1 = c.Z;
    P2 D = X1 - x2 * Z1;
    P2 E = Y1 - y2 * Z1;
    P2 F = D.sqr();
    P2 G = E.sqr();
    P2 H = D * F;
    P2 I = X1 * F;
    P2 J = H + Z1 * G - (I + I);
    c.X = D * J;
    c.Y = E * (I - J) - (H * Y1);
    c.Z = Z1 * H;
    return {(E * x2 - D * y2).mul_xi(), D, E.neg()};
}

LSA_HD P12 apply_line(const P12 &f, const Line &l, const PB &px, const PB &py) {
    return fq12_mul_by_024(f, l.e0, l.eVW.mul_fq(py), l.eVV.mul_fq(px));
}

LSA_HD int ate_bit(int i) {
    if (i >= 64) return (int)((LSA_ATE_LOOP_COUNT_HI >> (i - 64)) & 1);
    return (int)((LSA_ATE_LOOP_COUNT_LO >> i) & 1);
}

// The same loop count 6u + 2 in non-adjacent form: 66 signed digits, 22 of them non-zero (the binary expansion has 65 bits, 37
// of them set).  A Miller loop over these digits -- R <- 2R, f <- f^2 l_(R,R); at a digit +-1: f <- f l_(R,+-Q), R <- R +- Q --
// runs 65 doubling steps and 21 addition steps instead of 64 and 36.  Its value differs from libff's f_(6u+2,Q)(P) by vertical
// lines only (f_(a-1) = f_a l_([a]Q,-Q) / (v_([a-1]Q) v_Q)), which lie in Fq6 and are killed by the factor p^6 - 1 of the final
// exponent: the REDUCED pairing is the same element of GT, bit for bit.  Used where only GT values leave the library.
struct AteNaf { uint64_t plus_lo, plus_hi, minus_lo, minus_hi; int len, weight; };
constexpr AteNaf make_ate_naf() {
    uint64_t lo = LSA_ATE_LOOP_COUNT_LO, hi = LSA_ATE_LOOP_COUNT_HI;
    AteNaf r = {0, 0, 0, 0, 0, 0};
    int i = 0;
    while (lo | hi) {
        if (lo & 1) {
            const bool minus = (lo & 3) == 3;                   // k = 3 mod 4: digit -1, k <- k + 1
            if (minus) { lo += 1; if (lo == 0) hi += 1; (i < 64 ? r.minus_lo : r.minus_hi) |= 1ull << (i & 63); }
            else { lo -= 1; (i < 64 ? r.plus_lo : r.plus_hi) |= 1ull << (i & 63); }
            r.weight++;
        }
        lo = (lo >> 1) | (hi << 63);
        hi >>= 1;
        i++;
    }
    r.len = i;
    return r;
}
static constexpr AteNaf ATE_NAF = make_ate_naf();
static_assert(ATE_NAF.len == 66 && ATE_NAF.weight == 22 && ((ATE_NAF.plus_hi >> 1) & 1) == 1, "NAF of 6u + 2: 66 digits, top digit +1");
LSA_HD int ate_naf_digit(int i) {                               // i in [0, 65]
    const uint64_t pl = i < 64 ? ATE_NAF.plus_lo : ATE_NAF.plus_hi, mi = i < 64 ? ATE_NAF.minus_lo : ATE_NAF.minus_hi;
    return (int)((pl >> (i & 63)) & 1) - (int)((mi >> (i & 63)) & 1);
}

// libff to_affine_coordinates() on both inputs: O -> (0, 1, 0)
struct AffinePair { PB px, py; P2 qx, qy; };
LSA_HD_NOINLINE AffinePair miller_affine_inputs(const Jac<Fq> &P, const Jac<Fq2> &Q) {
    AffinePair r;
    if (P.Z.is_zero()) { r.px = PB::zero(); r.py = PB::one(); }
    else if (P.Z == Fq::one()) { r.px = PB::from_mont256(P.X); r.py = PB::from_mont256(P.Y); }
    else {
        PB zi = PB::from_mont256(P.Z).inverse(), zi2 = zi.sqr();
        r.px = PB::from_mont256(P.X) * zi2; r.py = PB::from_mont256(P.Y) * (zi2 * zi);
    }
    if (Q.Z.is_zero()) { r.qx = P2::zero(); r.qy = P2::one(); }
    else if (Q.Z == Fq2::one()) { r.qx = load2(Q.X); r.qy = load2(Q.Y); }
    else {
        P2 zi = load2(Q.Z).inverse(), zi2 = zi.sqr();
        r.qx = load2(Q.X) * zi2; r.qy = load2(Q.Y) * (zi2 * zi);
    }
    return r;
}

// precompute_G1 + precompute_G2 + miller_loop for one pair (libff layout in)
LSA_HD_NOINLINE P12 miller_one(const Jac<Fq> &P, const Jac<Fq2> &Q) {
    const AffinePair in = miller_affine_inputs(P, Q);
    const PB px = in.px, py = in.py;
    const P2 qx = in.qx, qy = in.qy;
    Fq ti;
#pragma unroll
    for (int i = 0; i < 8; i++) ti.l[i] = LSA_FQ_TWO_INV[i];
    const PB two_inv = PB::from_mont256(ti);
    const P2 twist_b = fq2_constT<PB>(LSA_TWIST_B);
    G2Proj R = {qx, qy, P2::one()};
    P12 f = P12::one();
    // bits of 6u+2 below the MSB (bit 64), MSB first
    for (int i = 63; i >= 0; --i) {
        Line l = doubling_step(R, two_inv, twist_b);
        f = fq12_sqr(f);
        f = apply_line(f, l, px, py);
        if (ate_bit(i)) {
            l = addition_step(qx, qy, R);
            f = apply_line(f, l, px, py);
        }
    }
    // Q1 = pi(Q), Q2 = -pi^2(Q)   (mul_by_q on affine points: Z stays 1)
    const P2 gx = fq2_constT<PB>(LSA_TWIST_MUL_BY_Q_X), gy = fq2_constT<PB>(LSA_TWIST_MUL_BY_Q_Y);
    P2 q1x = gx * qx.conj(), q1y = gy * qy.conj();
    P2 q2x = gx * q1x.conj(), q2y = (gy * q1y.conj()).neg();
    Line l = addition_step(q1x, q1y, R);
    f = apply_line(f, l, px, py);
    l = addition_step(q2x, q2y, R);
    f = apply_line(f, l, px, py);
    return f;
}

// ------------------------------------------------------------------------------------
// one pairing per wavefront
// ------------------------------------------------------------------------------------
static constexpr int WM_SIDE = 6;                          // side products per round
enum WMVar {                                               // Fq2 variables in LDS next to the W12 slots
    WM_RX, WM_RY, WM_RZ, WM_S,                             // current point, S = RY + RZ
    WM_QX, WM_QY, WM_Q1X, WM_Q1Y, WM_Q2X, WM_Q2Y,          // Q, pi(Q), -pi^2(Q)
    WM_PX, WM_PY, WM_TWB,                                  // (px,0), (py,0), twist_b
    WM_A, WM_B, WM_D, WM_E, WM_H, WM_NH, WM_J3, WM_BMF, WM_G,      // doubling step
    WM_DD, WM_EE, WM_NE, WM_F, WM_GG, WM_HH, WM_I, WM_ZG, WM_JJ, WM_IMJ,   // addition step
    WM_NVARS
};
static constexpr int WM_LDS_FQ2 = W12_LDS_FQ2 + WM_NVARS + WM_SIDE;

// Start-up of a wavefront / group engine, shared by TWO lanes: sel 0 brings the G1 point to
// affine form (libff to_affine_coordinates: O -> (0, 1)) and stores (px,0), (py,0), twist_b;
// sel 1 does the G2 point and derives pi(Q), -pi^2(Q) and R = Q.  Each needs one base-field
// inversion (Z, resp. the norm of Z) when its point is not normalised; written so that both lanes
// run that ~0.2 ms Fermat chain in the same instruction stream instead of one after the other.
LSA_HD void wm_setup(unsigned sel, bool valid, const Jac<Fq> *P, const Jac<Fq2> *Q, Fq2S *Vv) {
    PB z = PB::one();
    P2 qz = P2::one();
    bool need = false, inf = true;
    if (valid) {
        if (sel == 0) {
            inf = P->Z.is_zero();
            need = !inf && !(P->Z == Fq::one());
            if (need) z = PB::from_mont256(P->Z);
        } else {
            inf = Q->Z.is_zero();
            need = !inf && !(Q->Z == Fq2::one());
            if (need) { qz = load2(Q->Z); z = qz.c0.sqr() + qz.c1.sqr(); }
        }
    }
    PB zi = PB::one();
    if (need) zi = z.inverse();
    if (sel == 0) {
        PB px = PB::zero(), py = PB::one();
        if (!inf) {
            px = PB::from_mont256(P->X); py = PB::from_mont256(P->Y);
            if (need) { const PB zi2 = zi.sqr(); px = px * zi2; py = py * (zi2 * zi); }
        }
        Vv[WM_PX] = Fq2S{px, PB::zero()};
        Vv[WM_PY] = Fq2S{py, PB::zero()};
        Vv[WM_TWB] = fq2_constT<PB>(LSA_TWIST_B);
    } else {
        P2 qx = P2::zero(), qy = P2::one();
        if (!inf) {
            qx = load2(Q->X); qy = load2(Q->Y);
            if (need) {
                const P2 zinv = {qz.c0 * zi, (qz.c1 * zi).neg()};      // Fq2 inverse = conj / norm
                const P2 zi2 = zinv.sqr();
                qx = qx * zi2; qy = qy * (zi2 * zinv);
            }
        }
        const P2 gx = fq2_constT<PB>(LSA_TWIST_MUL_BY_Q_X), gy = fq2_constT<PB>(LSA_TWIST_MUL_BY_Q_Y);
        const P2 q1x = gx * qx.conj(), q1y = gy * qy.conj();
        Vv[WM_QX] = qx; Vv[WM_QY] = qy;
        Vv[WM_Q1X] = q1x; Vv[WM_Q1Y] = q1y;
        Vv[WM_Q2X] = gx * q1x.conj(); Vv[WM_Q2Y] = (gy * q1y.conj()).neg();
        Vv[WM_RX] = qx; Vv[WM_RY] = qy; Vv[WM_RZ] = P2::one();
        Vv[WM_S] = qy + P2::one();
    }
}

template <class X>
struct WMiller {
    W12<X> w;      // slot 0: f, slot 1: the line as a full element (zeros at w^1, w^2, w^5)
    Fq2S *V;       // WM_NVARS variables
    Fq2S *G;       // WM_SIDE side products of the current round
    enum { SF = 0, SL = 1 };

    struct Side { int8_t a[WM_SIDE], b[WM_SIDE]; int n; };

    // product phase: lanes < 36: slot(fa)[i] * slot(fb)[j] (when fmul); lanes 36 .. 36+n-1: side products
    LSA_HD void products(bool fmul, int fa, int fb, const Side sd) {
        Fq2S *A = w.slot(fa), *B = w.slot(fb), *Pp = w.P, *Vv = V, *Gg = G;
        w.x.par([=](unsigned lane) {
            const Fq2S *pa = Vv, *pb = Vv;
            Fq2S *pd = nullptr;
            if (lane < 36) {
                if (fmul) { pa = A + lane / 6; pb = B + lane % 6; pd = Pp + lane; }
            } else if ((int)lane - 36 < sd.n) {
                const int k = (int)lane - 36;
                pa = Vv + sd.a[k]; pb = Vv + sd.b[k]; pd = Gg + k;
            }
            if (pd) *pd = w12_fq2_mul(*pa, *pb);
        });
    }
    LSA_HD void set_line(const Fq2S &e0, const Fq2S &evw, const Fq2S &evv) const {
        Fq2S *L = w.slot(SL);
        L[0] = e0; L[3] = evw; L[4] = evv;      // w^0: ell_0, w^3: ell_VW * py, w^4: ell_VV * px
    }

    // ---- lazy Fq2 helpers of the combine phases.  Values live in LDS as Fq2S but follow the
    // engine's operand contract (w12.h: tight limbs, a-operands < 4p, b-operands < 20p)
    // instead of Fs's "< 2p"; every line carries its bound in multiples of p.
    static LSA_HD F29x2 ld(const Fq2S &v) { return {v.c0.v, v.c1.v}; }
    static LSA_HD Fq2S st(const F29x2 &v) { return {Fs{v.c0}, Fs{v.c1}}; }
    static LSA_HD F29x2 triple(const F29x2 &a) { return add_lazy(add_lazy(a, a), a).norm(); }             // [3x]
    static LSA_HD F29x2 halve2(const F29x2 &a) { return {f29_halve(a.c0), f29_halve(a.c1)}; }             // [x/2 + 0.5]
    static LSA_HD F29x2 csub2(const F29x2 &a) { return {condsub2(a.c0), condsub2(a.c1)}; }                // [<4] -> [<2]
    // xi * t for t < 2:  (9 t0 - t1 + 2p,  9 t1 + t0)   [< 20]
    static LSA_HD F29x2 xi_times(const F29x2 &t) {
        F29 a8, b8;
#pragma unroll
        for (int i = 0; i < 9; i++) { a8.l[i] = t.c0.l[i] << 3; b8.l[i] = t.c1.l[i] << 3; }
        a8 = w12_norm_u(a8);
        b8 = w12_norm_u(b8);
        return {sub_k<2>(add_lazy(a8, t.c0), t.c1), add_lazy(add_lazy(b8, t.c1), t.c0).norm()};
    }

    // f <- f^2 * line(2R), R <- 2R.   Invariant between rounds: RX, RY, RZ < 2, S = RY + RZ < 4.
    LSA_HD void doubling_round() {
        WMiller self = *this;
        Fq2S *Vv = V, *Gg = G;
        // round 1: f*f | X*Y, Y^2, Z^2, (Y+Z)^2, X^2
        products(true, SF, SF, Side{{WM_RX, WM_RY, WM_RZ, WM_S, WM_RX, 0}, {WM_RY, WM_RY, WM_RZ, WM_S, WM_RX, 0}, 5});
        w.x.par([=](unsigned lane) {
            if (lane < 12) w12_reduce_lane12(lane, self.w.P, self.w.slot(SF));
            else if (lane == 12) {
                const F29x2 B = ld(Gg[1]), C = ld(Gg[2]);
                const F29x2 H = sub_k<4>(ld(Gg[3]), add_lazy(B, C));        // (Y+Z)^2 - (B+C) + 4p   [<6]
                Vv[WM_A] = st(halve2(ld(Gg[0])));                            // X*Y/2                  [<1.5]
                Vv[WM_B] = Gg[1];
                Vv[WM_D] = st(triple(C));                                    // 3C                     [<6]
                Vv[WM_H] = st(H);
                Vv[WM_NH] = st(sub_k<6>(F29x2::zero(), H));                  // 6p - H                 [<=6]
                Vv[WM_J3] = st(triple(ld(Gg[4])));                           // 3 X^2                  [<6]
            }
        });
        // round 2: twist_b*D, B*H, py*(-H), px*(3J)      (the larger operand second)
        products(false, SF, SF, Side{{WM_TWB, WM_B, WM_PY, WM_PX, 0, 0}, {WM_D, WM_H, WM_NH, WM_J3, 0, 0}, 4});
        w.x.par([=](unsigned lane) {
            if (lane == 12) {
                const F29x2 E = ld(Gg[0]), B = ld(Vv[WM_B]);
                const F29x2 F = triple(E);                                   // 3E                     [<6]
                Vv[WM_E] = Gg[0];
                Vv[WM_G] = st(condsub4(halve2(add_lazy(B, F).norm())));      // (B+F)/2  [<4.5] -> [<4]
                Vv[WM_BMF] = st(sub_k<6>(B, F));                             // B - F + 6p             [<8]
                Vv[WM_RZ] = Gg[1];                                           // Z3 = B*H               [<2]
                self.set_line(st(xi_times(csub2(sub_k<2>(E, B)))), Gg[2], Gg[3]);   // xi*(E-B) [<20], [<2], [<2]
            }
        });
        // round 3: f*line | E^2, A*(B-F), G^2
        products(true, SF, SL, Side{{WM_E, WM_A, WM_G, 0, 0, 0}, {WM_E, WM_BMF, WM_G, 0, 0, 0}, 3});
        w.x.par([=](unsigned lane) {
            if (lane < 12) w12_reduce_lane12(lane, self.w.P, self.w.slot(SF));
            else if (lane == 12) {
                const F29x2 Y3 = csub2(condsub4(sub_k<6>(ld(Gg[2]), triple(ld(Gg[0])))));   // G^2 - 3E^2 + 6p [<8] -> [<2]
                Vv[WM_RX] = Gg[1];                                           // X3                     [<2]
                Vv[WM_RY] = st(Y3);
                Vv[WM_S] = st(add_lazy(Y3, ld(Vv[WM_RZ])).norm());           // [<4]
            }
        });
    }

    // f <- f * line(R + (x2,y2)), R <- R + (x2,y2);  x2, y2 < 2
    LSA_HD void addition_round(int x2, int y2) {
        WMiller self = *this;
        Fq2S *Vv = V, *Gg = G;
        // round 1: x2*Z1, y2*Z1
        products(false, SF, SF, Side{{(int8_t)x2, (int8_t)y2, 0, 0, 0, 0}, {WM_RZ, WM_RZ, 0, 0, 0, 0}, 2});
        w.x.par([=](unsigned lane) {
            if (lane == 12) {
                const F29x2 E = sub_k<2>(ld(Vv[WM_RY]), ld(Gg[1]));          // Y1 - y2 Z1 + 2p        [<4]
                Vv[WM_DD] = st(sub_k<2>(ld(Vv[WM_RX]), ld(Gg[0])));          // X1 - x2 Z1 + 2p        [<4]
                Vv[WM_EE] = st(E);
                Vv[WM_NE] = st(sub_k<4>(F29x2::zero(), E));                  // 4p - E                 [<=4]
            }
        });
        // round 2: D^2, E^2, E*x2, D*y2, D*py, px*(-E)
        products(false, SF, SF, Side{{WM_DD, WM_EE, WM_EE, WM_DD, WM_DD, WM_PX}, {WM_DD, WM_EE, (int8_t)x2, (int8_t)y2, WM_PY, WM_NE}, 6});
        w.x.par([=](unsigned lane) {
            if (lane == 12) {
                Vv[WM_F] = Gg[0];
                Vv[WM_GG] = Gg[1];
                self.set_line(st(xi_times(csub2(sub_k<2>(ld(Gg[2]), ld(Gg[3]))))), Gg[4], Gg[5]);
            }
        });
        // round 3: f*line | D*F, X1*F, Z1*G
        products(true, SF, SL, Side{{WM_DD, WM_RX, WM_RZ, 0, 0, 0}, {WM_F, WM_F, WM_GG, 0, 0, 0}, 3});
        w.x.par([=](unsigned lane) {
            if (lane < 12) w12_reduce_lane12(lane, self.w.P, self.w.slot(SF));
            else if (lane == 12) {
                const F29x2 H = ld(Gg[0]), I = ld(Gg[1]);
                const F29x2 J = sub_k<4>(add_lazy(H, ld(Gg[2])), add_lazy(I, I));   // H + Z1 G - 2I + 4p [<8]
                Vv[WM_HH] = Gg[0];
                Vv[WM_JJ] = st(J);
                Vv[WM_IMJ] = st(sub_k<8>(I, J));                             // I - J + 8p             [<10]
            }
        });
        // round 4: D*J, E*(I-J), H*Y1, Z1*H
        products(false, SF, SF, Side{{WM_DD, WM_EE, WM_HH, WM_RZ, 0, 0}, {WM_JJ, WM_IMJ, WM_RY, WM_HH, 0, 0}, 4});
        w.x.par([=](unsigned lane) {
            if (lane == 12) {
                const F29x2 Y3 = csub2(sub_k<2>(ld(Gg[1]), ld(Gg[2])));      // [<4] -> [<2]
                Vv[WM_RX] = Gg[0];
                Vv[WM_RY] = st(Y3);
                Vv[WM_RZ] = Gg[3];
                Vv[WM_S] = st(add_lazy(Y3, ld(Gg[3])).norm());               // [<4]
            }
        });
    }

    // f (slot 0) <- miller_loop(P, Q)
    LSA_HD void run(const Jac<Fq> &P, const Jac<Fq2> &Q) {
        Fq2S *Vv = V;
        Fq2S *F0 = w.slot(SF), *L = w.slot(SL);
        w.x.par([=](unsigned lane) {
            if (lane < 2) {
                wm_setup(lane, true, &P, &Q, Vv);
            } else if (lane >= 8 && lane < 14) {
                const unsigned k = lane - 8;
                F0[k] = k == 0 ? P2::one() : P2::zero();
                L[k] = P2::zero();
            }
        });
        for (int i = 63; i >= 0; --i) {
            doubling_round();
            if (ate_bit(i)) addition_round(WM_QX, WM_QY);
        }
        addition_round(WM_Q1X, WM_Q1Y);
        addition_round(WM_Q2X, WM_Q2Y);
    }
};

// ------------------------------------------------------------------------------------
// (Rounds 1-2 had two more engines here: six lanes per pairing, ten pairings per wavefront (G6Miller) and twelve lanes per
// pairing (G12Miller).  Every shape they served went to the fused kernel (tmiller.h: G2Pre + TabMillerP) in round 3;
// removed in round 5.  What remains below are the component products the table kernels still use.)
// ------------------------------------------------------------------------------------
// The unordered index pairs {t, u} behind coefficient k of a SQUARE: t + u = k (plain) or k + 6
// (wrapped, times xi).  k even: 4 pairs (two of them squares), k odd: 3 -- instead of the 6
// ordered pairs of a general product.  j-th pair of coefficient k:
LSA_HD void sqr_pair(int k, int j, int &t, int &u, bool &wrap) {
    const int hk = k >> 1;
    if (j <= hk) { t = j; u = k - j; wrap = false; }
    else { t = k + j - hk; u = k + 6 - t; wrap = true; }
}
LSA_HD int sqr_pair_count(int k) { return 4 - (k & 1); }
// component `part` of a*b: part 0: a0*b0 + a1*(KB p - b1), part 1: a0*b1 + a1*b0.  b's components < KB p.
template <int KB>
LSA_HD F29 g12_comp_mul(unsigned part, const Fq2S &a, const Fq2S &b) {
    const uint32_t pm = w12_mask(0u - part);
    const F29 nb1 = sub_k<KB>(F29::zero(), b.c1.v);
    F29 y0, y1;
#pragma unroll
    for (int l = 0; l < 9; l++) {
        y0.l[l] = (b.c1.v.l[l] & pm) | (b.c0.v.l[l] & ~pm);
        y1.l[l] = (b.c0.v.l[l] & pm) | (nb1.l[l] & ~pm);
    }
#if defined(LSA_F29_COLS)
    return f29_dot2_cols(a.c0.v, y0, a.c1.v, y1);
#endif
    return dot2(a.c0.v, y0, a.c1.v, y1);
}
LSA_HD Fs &g12_part(Fq2S &v, unsigned part) { return part ? v.c1 : v.c0; }


}  // namespace lsa
