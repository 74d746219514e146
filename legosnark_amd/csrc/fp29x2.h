// legosnark_amd/csrc/fp29x2.h -- Fq2 = Fq[u]/(u^2+1) over the 9 x 29-bit unsaturated-limb
// field of fp29.h, and the XYZZ point formulas of alt_bn128 G2 on top of it.
//
// An Fq2 product needs 4 limb-level products but only 2 Montgomery reductions: with 29-bit
// limbs the two products of each output component accumulate in the same 64-bit columns
// (dot2), so  c0 = a0*b0 + a1*(K*p - b1),  c1 = a0*b1 + a1*b0  cost 4*81 + 2*90 = 504
// v_mad_u64_u32 (Karatsuba on reduced values would be 3*171 = 513 plus carries).
//
// Bounds (per Fq component, in multiples of p; R = 2^261 ~ 169.6 p):
//   dot2 needs a0*b0 + a1*b1 < 169 p^2 and, per limb product, at most one loose operand.
//   Accumulator invariants:  X < 4,  Y < 4,  ZZ < 2,  ZZZ < 2, all limbs tight; infinity <=>
//   ZZ has all limbs zero.  One conditional subtraction of 4p (condsub4) after the X3
//   subtraction chain keeps them; every line below carries its bound.
#pragma once
#include "fp29.h"

namespace lsa {

// tight value < 8p  ->  same residue, < 4p
LSA_HD F29 condsub4(const F29 &t) {
    F29 d;
    int32_t c = 0;
    uint64_t pc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        pc += (uint64_t)F29::p(i) * 4u;
        uint32_t pl = (i < 8) ? ((uint32_t)pc & F29::MASK) : (uint32_t)pc;
        pc >>= 29;
        int32_t v = (int32_t)t.l[i] - (int32_t)pl + c;
        if (i < 8) { d.l[i] = (uint32_t)v & F29::MASK; c = v >> 29; }
        else d.l[i] = (uint32_t)v;
    }
    const uint32_t keep = (uint32_t)((int32_t)d.l[8] >> 31);   // all ones if t < 4p
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (t.l[i] & keep) | (d.l[i] & ~keep);
    return r;
}

struct F29x2 {
    F29 c0, c1;
    static LSA_HD F29x2 zero() { return {F29::zero(), F29::zero()}; }
    static LSA_HD F29x2 one() { return {F29::one(), F29::zero()}; }
    LSA_HD bool limbs_zero() const { return c0.limbs_zero() && c1.limbs_zero(); }
    LSA_HD bool is_zero_mod_p() const { return c0.is_zero_mod_p() && c1.is_zero_mod_p(); }
    LSA_HD F29x2 norm() const { return {c0.norm(), c1.norm()}; }
    LSA_HD F29x2 canonical() const { return {c0.canonical(), c1.canonical()}; }
};
LSA_HD F29x2 add_lazy(const F29x2 &a, const F29x2 &b) { return {add_lazy(a.c0, b.c0), add_lazy(a.c1, b.c1)}; }
template <int K>
LSA_HD F29x2 sub_k(const F29x2 &a, const F29x2 &b) { return {sub_k<K>(a.c0, b.c0), sub_k<K>(a.c1, b.c1)}; }
LSA_HD F29x2 condsub4(const F29x2 &a) { return {condsub4(a.c0), condsub4(a.c1)}; }

// a*b with b's components < KB*p.  Needs 2*A*KB < 169 for a's components < A*p.  [< 2; tight]
template <int KB>
LSA_HD F29x2 mul(const F29x2 &a, const F29x2 &b) {
    F29 nb1 = sub_k<KB>(F29::zero(), b.c1);              // KB*p - b1   [<= KB; tight]
    return {dot2(a.c0, b.c0, a.c1, nb1), dot2(a.c0, b.c1, a.c1, b.c0)};
}
// a^2 with components < KA*p (tight limbs).  Needs (2*KA)^2 < 169, i.e. KA <= 6.  [< 2; tight]
template <int KA>
LSA_HD F29x2 sqr(const F29x2 &a) {
    F29 s = add_lazy(a.c0, a.c1);                          // [< 2KA; loose]
    F29 d = sub_k<KA>(a.c0, a.c1);                         // a0 - a1 + KA*p   [< 2KA; tight]
    return {mul(s, d), mul(add_lazy(a.c0, a.c0), a.c1)};
}

// ------------------------------------------------------------------------------------
// Point formulas over any representation E of Fq2 that offers zero / one / limbs_zero / is_zero_mod_p / norm and the
// free functions add_lazy, sub_k<K>, condsub4, mul<KB>, sqr<KA> with the bounds above: F29x2 (both components in one
// lane) and F29h (fp29x2l.h: one component per lane of a pair; its predicates are uniform over the pair, so both
// lanes take the same branches).
template <class E>
struct AffE {
    E x, y;                                                // canonical (< p)
    LSA_HD bool is_inf() const { return x.limbs_zero() && y.limbs_zero(); }
};
template <class E>
struct XyzzE {
    E X, Y, ZZ, ZZZ;
    LSA_HD bool is_inf() const { return ZZ.limbs_zero(); }
    static LSA_HD XyzzE inf() { return {E::zero(), E::zero(), E::zero(), E::zero()}; }
};
using Aff29x2 = AffE<F29x2>;
using XYZZ29x2 = XyzzE<F29x2>;
struct AffPackedG2 { uint32_t w[4][8]; };                  // x.c0, x.c1, y.c0, y.c1 as 256-bit words (128 B)
LSA_HD Aff29x2 unpack_affine(const AffPackedG2 &q) {
    return {{F29::unpack256(q.w[0]), F29::unpack256(q.w[1])}, {F29::unpack256(q.w[2]), F29::unpack256(q.w[3])}};
}
LSA_HD F29x2 f29x2_from_mont256(const Fq2 &v) { return {F29::from_mont256(v.c0), F29::from_mont256(v.c1)}; }
LSA_HD Fq2 f29x2_to_mont256(const F29x2 &v) { return {v.c0.to_mont256(), v.c1.to_mont256()}; }
LSA_HD AffPackedG2 pack_affine_g2(const Aff<Fq2> &a) {
    AffPackedG2 r;
    if (a.is_inf()) {
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int i = 0; i < 8; i++) r.w[c][i] = 0;
    } else {
        F29::from_mont256(a.x.c0).canonical().pack256(r.w[0]);
        F29::from_mont256(a.x.c1).canonical().pack256(r.w[1]);
        F29::from_mont256(a.y.c0).canonical().pack256(r.w[2]);
        F29::from_mont256(a.y.c1).canonical().pack256(r.w[3]);
    }
    return r;
}

// 2*(x,y), affine canonical input (mdbl-2008-s-1)
template <class E>
LSA_HD XyzzE<E> g2_dbl_affine(const AffE<E> &b) {
    E U = add_lazy(b.y, b.y).norm();                                   // [<2; tight]
    E V = sqr<2>(U);
    E W = mul<2>(U, V);                                                // 2*2*2 = 8
    E S = mul<2>(b.x, V);
    E xx = sqr<1>(b.x);
    E M = add_lazy(add_lazy(xx, xx), xx).norm();                       // [<6; tight]
    E X3 = condsub4(sub_k<4>(sqr<6>(M), add_lazy(S, S)));              // M^2 - 2S + 4p <6 -> [<4]
    E Y3 = sub_k<2>(mul<6>(M, sub_k<4>(S, X3)), mul<1>(W, b.y));       // 2*6*6 = 72   [<4]
    return {X3, Y3, V, W};
}
// 2*P (dbl-2008-s-1)
template <class E>
LSA_HD XyzzE<E> g2_dbl(const XyzzE<E> &a) {
    if (a.is_inf()) return a;
    E U = condsub4(add_lazy(a.Y, a.Y).norm());                         // 2Y <8 -> [<4; tight]
    E V = sqr<4>(U);
    E W = mul<2>(U, V);                                                // 2*4*2 = 16
    E S = mul<2>(a.X, V);
    E xx = sqr<4>(a.X);
    E M = add_lazy(add_lazy(xx, xx), xx).norm();                       // [<6; tight]
    E X3 = condsub4(sub_k<4>(sqr<6>(M), add_lazy(S, S)));              // [<4]
    E Y3 = sub_k<2>(mul<6>(M, sub_k<4>(S, X3)), mul<4>(W, a.Y));       // 72 ; 2*2*4 = 16   [<4]
    return {X3, Y3, mul<2>(V, a.ZZ), mul<2>(W, a.ZZZ)};
}
// acc + (x2,y2), complete (madd-2008-s)
template <class E>
LSA_HD XyzzE<E> g2_madd(const XyzzE<E> &a, const AffE<E> &b) {
    if (b.is_inf()) return a;
    if (a.is_inf()) return {b.x, b.y, E::one(), E::one()};
    E U2 = mul<2>(b.x, a.ZZ);                                          // 2*1*2
    E S2 = mul<2>(b.y, a.ZZZ);
    E Pd = sub_k<4>(U2, a.X);                                          // [<6]
    E R = sub_k<4>(S2, a.Y);                                           // [<6]
    if (Pd.is_zero_mod_p()) {
        if (R.is_zero_mod_p()) return g2_dbl_affine(b);
        return XyzzE<E>::inf();
    }
    E PP = sqr<6>(Pd);                                                 // (12)(12) = 144
    E PPP = mul<2>(Pd, PP);                                            // 2*6*2 = 24
    E Q = mul<2>(a.X, PP);                                             // 2*4*2 = 16
    E X3 = condsub4(sub_k<6>(sqr<6>(R), add_lazy(PPP, add_lazy(Q, Q))));       // <8 -> [<4]
    E Y3 = sub_k<2>(mul<6>(R, sub_k<4>(Q, X3)), mul<2>(a.Y, PPP));             // 2*6*6 = 72 ; 16   [<4]
    return {X3, Y3, mul<2>(a.ZZ, PP), mul<2>(a.ZZZ, PPP)};
}
// a + b, complete (add-2008-s)
template <class E>
LSA_HD XyzzE<E> g2_add(const XyzzE<E> &a, const XyzzE<E> &b) {
    if (b.is_inf()) return a;
    if (a.is_inf()) return b;
    E U1 = mul<2>(a.X, b.ZZ);                                          // 2*4*2
    E U2 = mul<2>(b.X, a.ZZ);
    E S1 = mul<2>(a.Y, b.ZZZ);
    E S2 = mul<2>(b.Y, a.ZZZ);
    E Pd = sub_k<2>(U2, U1);                                           // [<4]
    E R = sub_k<2>(S2, S1);
    if (Pd.is_zero_mod_p()) {
        if (R.is_zero_mod_p()) return g2_dbl(a);
        return XyzzE<E>::inf();
    }
    E PP = sqr<4>(Pd);
    E PPP = mul<2>(Pd, PP);
    E Q = mul<2>(U1, PP);
    E X3 = condsub4(sub_k<6>(sqr<4>(R), add_lazy(PPP, add_lazy(Q, Q))));
    E Y3 = sub_k<2>(mul<6>(R, sub_k<4>(Q, X3)), mul<2>(S1, PPP));      // 2*4*6 = 48   [<4]
    return {X3, Y3, mul<2>(mul<2>(a.ZZ, b.ZZ), PP), mul<2>(mul<2>(a.ZZZ, b.ZZZ), PPP)};
}
// XYZZ -> libff Jacobian: Z = ZZZ, X' = X*ZZ^2, Y' = Y*ZZZ^2
LSA_HD Jac<Fq2> g2_to_jac(const XYZZ29x2 &a) {
    if (a.is_inf()) return Jac<Fq2>::inf();
    F29x2 Xj = mul<2>(a.X, sqr<2>(a.ZZ));
    F29x2 Yj = mul<2>(a.Y, sqr<2>(a.ZZZ));
    return {f29x2_to_mont256(Xj), f29x2_to_mont256(Yj), f29x2_to_mont256(a.ZZZ)};
}

}  // namespace lsa
