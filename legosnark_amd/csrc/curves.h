// legosnark_amd/csrc/curves.h -- curve traits shared by the MSM pipeline (msm.hip) and the
// fixed-base batch exponentiation (batch_exp.hip): which representation a group's device
// kernels compute in, how a device-resident base is stored, and the point operations.
#pragma once
#include "ec.h"
#include "fp29.h"
#include "fp29x2.h"
#include "quad29.h"

namespace lsa {

// G1 on 9 x 29-bit unsaturated limbs (fp29.h); bases packed to 64 B.
struct CurveG1 {
    using Field = Fq;
    using Base = AffPacked;
    using Acc = XYZZ29;
    static constexpr bool GLV = true;     // scalars split as k1 + k2*lambda (glv.h)
    static __device__ __forceinline__ Acc inf() { return Acc::inf(); }
    static __device__ __forceinline__ Acc madd(const Acc &a, const Base &b, bool negate, bool endo) {
        Aff29 q = unpack_affine(b);
        if (q.is_inf()) return a;
        if (negate) q.y = sub_k<1>(F29::zero(), q.y);                  // p - y
        if (endo) {                                                     // phi(x,y) = (beta*x, y)
            constexpr uint32_t BETA29[9] = {0x0a337995u, 0x158d1d23u, 0x189c9b98u, 0x12fa4e45u, 0x185faadcu,
                                            0x0176f16du, 0x0eed93bau, 0x14291140u, 0x000c0afeu};
            q.x = mul(q.x, F29::from_limbs(BETA29));                    // [<2p; tight]
        }
        return xyzz29_madd(a, q);
    }
    static __device__ __forceinline__ Acc add(const Acc &a, const Acc &b) { return xyzz29_add(a, b); }
    static __device__ __forceinline__ Acc dbl(const Acc &a) { return xyzz29_dbl(a); }
    static __device__ __forceinline__ Jac<Fq> to_jac(const Acc &a) { return xyzz29_to_jac(a); }
    static __device__ __forceinline__ Acc from_jac(const Jac<Fq> &p) {          // ZZ = Z^2, ZZZ = Z^3
        if (p.Z.is_zero()) return Acc::inf();
        F29 z = F29::from_mont256(p.Z), zz = sqr(z);
        return {F29::from_mont256(p.X), F29::from_mont256(p.Y), zz, mul(zz, z)};
    }
    static __device__ __forceinline__ Base from_affine(const Aff<Fq> &a) {
        Base r;
        if (a.is_inf()) {
#pragma unroll
            for (int i = 0; i < 8; i++) { r.x[i] = 0; r.y[i] = 0; }
        } else {
            F29::from_mont256(a.x).canonical().pack256(r.x);
            F29::from_mont256(a.y).canonical().pack256(r.y);
        }
        return r;
    }
};
// G2 on the same 29-bit-limb field (fp29x2.h): Fq2 products with fused reductions; bases
// packed to 128 B.  No GLV (the G2 endomorphism needs a 4-dimensional split): 16 windows.
struct CurveG2 {
    using Field = Fq2;
    using Base = AffPackedG2;
    using Acc = XYZZ29x2;
    static constexpr bool GLV = false;
    static __device__ __forceinline__ Acc inf() { return Acc::inf(); }
    static __device__ __forceinline__ Acc madd(const Acc &a, const Base &b, bool negate, bool /*endo*/) {
        Aff29x2 q = unpack_affine(b);
        if (q.is_inf()) return a;
        if (negate) q.y = sub_k<1>(F29x2::zero(), q.y);                // p - y per component
        return g2_madd(a, q);
    }
    static __device__ __forceinline__ Acc add(const Acc &a, const Acc &b) { return g2_add(a, b); }
    static __device__ __forceinline__ Acc dbl(const Acc &a) { return g2_dbl(a); }
    static __device__ __forceinline__ Jac<Fq2> to_jac(const Acc &a) { return g2_to_jac(a); }
    static __device__ __forceinline__ Acc from_jac(const Jac<Fq2> &p) {
        if (p.Z.is_zero()) return Acc::inf();
        F29x2 z = f29x2_from_mont256(p.Z), zz = sqr<2>(z);
        return {f29x2_from_mont256(p.X), f29x2_from_mont256(p.Y), zz, mul<2>(zz, z)};
    }
    static __device__ __forceinline__ Base from_affine(const Aff<Fq2> &a) { return pack_affine_g2(a); }
};
template <class F> struct CurveOf;
template <> struct CurveOf<Fq> { using type = CurveG1; };
template <> struct CurveOf<Fq2> { using type = CurveG2; };


}  // namespace lsa
