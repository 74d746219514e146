// legosnark_amd/csrc/ec.h -- short-Weierstrass (a = 0) point arithmetic over a generic
// field F (Fq for G1, Fq2 for G2), shared by kernels and host.
//
// Three representations:
//   Jac<F>    {X,Y,Z}      libff's in-memory layout for alt_bn128_G1/G2 (Z == 0 <=> O);
//                          what crosses the C-ABI (SURVEY.md section 8 header).
//   Aff<F>    {x,y}        device-resident bases; (0,0) encodes O (not on y^2 = x^3 + b).
//   XYZZ<F>   {X,Y,ZZ,ZZZ} bucket accumulators: x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; ZZ == 0 <=> O.
//                          Mixed add 8M+2S with no inversion, cheaper than Jacobian madd.
// All additions are COMPLETE (O, P+P, P+(-P) handled): LegoSNARK's CommScheme uses n
// copies of the generator as bases (/root/reference/src/prototools/commit.h:134-138), so
// doubling inside a bucket is the common case, not a corner.
#pragma once
#include "fp.h"

namespace lsa {

template <class F>
struct Aff {
    F x, y;
    LSA_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    static LSA_HD Aff inf() { return {F::zero(), F::zero()}; }
    LSA_HD Aff neg() const { return {x, y.neg()}; }
};

template <class F>
struct Jac {
    F X, Y, Z;
    LSA_HD bool is_inf() const { return Z.is_zero(); }
    static LSA_HD Jac inf() { return {F::zero(), F::one(), F::zero()}; }  // libff zero()
};

template <class F>
struct XYZZ {
    F X, Y, ZZ, ZZZ;
    LSA_HD bool is_inf() const { return ZZ.is_zero(); }
    static LSA_HD XYZZ inf() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
    static LSA_HD XYZZ from_affine(const Aff<F> &p) {
        if (p.is_inf()) return inf();
        return {p.x, p.y, F::one(), F::one()};
    }
    LSA_HD XYZZ neg() const { return {X, Y.neg(), ZZ, ZZZ}; }
};

// 2*(x,y) for an affine point (mdbl-2008-s-1)
template <class F>
LSA_HD XYZZ<F> xyzz_dbl_affine(const Aff<F> &p) {
    F U = p.y.dbl();
    F V = U.sqr();
    F W = U * V;
    F S = p.x * V;
    F xx = p.x.sqr();
    F M = xx.dbl() + xx;
    F X3 = M.sqr() - S.dbl();
    F Y3 = M * (S - X3) - W * p.y;
    return {X3, Y3, V, W};
}

// 2*P (dbl-2008-s-1)
template <class F>
LSA_HD_NOINLINE XYZZ<F> xyzz_dbl(const XYZZ<F> &p) {
    if (p.is_inf()) return p;
    F U = p.Y.dbl();
    F V = U.sqr();
    F W = U * V;
    F S = p.X * V;
    F xx = p.X.sqr();
    F M = xx.dbl() + xx;
    F X3 = M.sqr() - S.dbl();
    F Y3 = M * (S - X3) - W * p.Y;
    return {X3, Y3, V * p.ZZ, W * p.ZZZ};
}

// acc + (x2,y2)   (madd-2008-s), complete.
template <class F>
LSA_HD XYZZ<F> xyzz_madd(const XYZZ<F> &a, const Aff<F> &b) {
    if (b.is_inf()) return a;
    if (a.is_inf()) return {b.x, b.y, F::one(), F::one()};
    F U2 = b.x * a.ZZ;
    F S2 = b.y * a.ZZZ;
    F Pd = U2 - a.X;
    F R = S2 - a.Y;
    if (Pd.is_zero()) {
        if (R.is_zero()) return xyzz_dbl_affine(b);
        return XYZZ<F>::inf();
    }
    F PP = Pd.sqr();
    F PPP = Pd * PP;
    F Q = a.X * PP;
    F X3 = R.sqr() - PPP - Q.dbl();
    F Y3 = R * (Q - X3) - a.Y * PPP;
    return {X3, Y3, a.ZZ * PP, a.ZZZ * PPP};
}

// a + b   (add-2008-s), complete.
template <class F>
LSA_HD_NOINLINE XYZZ<F> xyzz_add(const XYZZ<F> &a, const XYZZ<F> &b) {
    if (b.is_inf()) return a;
    if (a.is_inf()) return b;
    F U1 = a.X * b.ZZ;
    F U2 = b.X * a.ZZ;
    F S1 = a.Y * b.ZZZ;
    F S2 = b.Y * a.ZZZ;
    F Pd = U2 - U1;
    F R = S2 - S1;
    if (Pd.is_zero()) {
        if (R.is_zero()) return xyzz_dbl(a);
        return XYZZ<F>::inf();
    }
    F PP = Pd.sqr();
    F PPP = Pd * PP;
    F Q = U1 * PP;
    F X3 = R.sqr() - PPP - Q.dbl();
    F Y3 = R * (Q - X3) - S1 * PPP;
    return {X3, Y3, a.ZZ * b.ZZ * PP, a.ZZZ * b.ZZZ * PPP};
}

// XYZZ -> Jacobian without inversion: Z = ZZZ, X' = X*ZZ^2, Y' = Y*ZZZ^2
// (Z^2 = ZZZ^2 = ZZ^3 so X'/Z^2 = X/ZZ and Y'/Z^3 = Y/ZZZ).
template <class F>
LSA_HD Jac<F> xyzz_to_jac(const XYZZ<F> &p) {
    if (p.is_inf()) return Jac<F>::inf();
    return {p.X * p.ZZ.sqr(), p.Y * p.ZZZ.sqr(), p.ZZZ};
}
// Jacobian -> XYZZ: ZZ = Z^2, ZZZ = Z^3
template <class F>
LSA_HD XYZZ<F> jac_to_xyzz(const Jac<F> &p) {
    if (p.is_inf()) return XYZZ<F>::inf();
    F zz = p.Z.sqr();
    return {p.X, p.Y, zz, zz * p.Z};
}

// Jacobian doubling (dbl-2009-l), 2M + 5S: used for the window fold (Horner).
template <class F>
LSA_HD_NOINLINE Jac<F> jac_dbl(const Jac<F> &p) {
    if (p.is_inf()) return p;
    F A = p.X.sqr();
    F B = p.Y.sqr();
    F C = B.sqr();
    F D = ((p.X + B).sqr() - A - C).dbl();
    F E = A.dbl() + A;
    F Fv = E.sqr();
    F X3 = Fv - D.dbl();
    F Y3 = E * (D - X3) - C.dbl().dbl().dbl();
    F Z3 = (p.Y * p.Z).dbl();
    return {X3, Y3, Z3};
}

// Jacobian general add (add-2007-bl), complete.
template <class F>
LSA_HD_NOINLINE Jac<F> jac_add(const Jac<F> &a, const Jac<F> &b) {
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    F Z1Z1 = a.Z.sqr(), Z2Z2 = b.Z.sqr();
    F U1 = a.X * Z2Z2, U2 = b.X * Z1Z1;
    F S1 = a.Y * (b.Z * Z2Z2), S2 = b.Y * (a.Z * Z1Z1);
    if (U1 == U2) {
        if (S1 == S2) return jac_dbl(a);
        return Jac<F>::inf();
    }
    F H = U2 - U1;
    F I = H.dbl().sqr();
    F J = H * I;
    F r = (S2 - S1).dbl();
    F V = U1 * I;
    F X3 = r.sqr() - J - V.dbl();
    F Y3 = r * (V - X3) - (S1 * J).dbl();
    F Z3 = ((a.Z + b.Z).sqr() - Z1Z1 - Z2Z2) * H;
    return {X3, Y3, Z3};
}

template <class F>
LSA_HD Jac<F> jac_neg(const Jac<F> &a) { return {a.X, a.Y.neg(), a.Z}; }

template <class F>
LSA_HD bool jac_eq(const Jac<F> &a, const Jac<F> &b) {
    if (a.is_inf()) return b.is_inf();
    if (b.is_inf()) return false;
    F Z1Z1 = a.Z.sqr(), Z2Z2 = b.Z.sqr();
    if (a.X * Z2Z2 != b.X * Z1Z1) return false;
    return a.Y * (b.Z * Z2Z2) == b.Y * (a.Z * Z1Z1);
}

// libff to_affine_coordinates(): O -> (0,1,0), else (X/Z^2, Y/Z^3, 1).
template <class F>
LSA_HD Jac<F> jac_normalize(const Jac<F> &a) {
    if (a.is_inf()) return Jac<F>::inf();
    F zi = a.Z.inverse();
    F zi2 = zi.sqr();
    return {a.X * zi2, a.Y * (zi2 * zi), F::one()};
}

}  // namespace lsa
