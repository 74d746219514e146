// legosnark_amd/csrc/quad29.h -- XYZZ29 doubling / addition with one point shared by a QUAD
// of lanes (device only).  Lane q of the quad computes the q-th independent field product of
// each dependency level; results are replicated with DPP quad_perm broadcasts.  State is
// replicated in the 4 lanes, so every branch below is quad-uniform.
#pragma once
#include "fp29.h"

namespace lsa {

template <int Q>
__device__ __forceinline__ F29 quad_bcast(const F29 &v) {
    F29 r;
#ifdef LSA_QUAD_USE_SHFL
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (uint32_t)__shfl((int)v.l[i], (int)((threadIdx.x & 60u) | Q), 64);
#else
    // v_mov_b32_dpp quad_perm:[Q,Q,Q,Q]; the result is made opaque to the optimizer so
    // that GCNDPPCombine cannot fold the broadcast into its consumer (folding it into the
    // v_sub of sub_k produced wrong low limbs on gfx950 / ROCm 7.2 -- tools/quad_check.hip).
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], Q * 0x55, 0xf, 0xf, false);
        asm volatile("" : "+v"(r.l[i]));
    }
#endif
    return r;
}
// per-lane 4-way select with bit masks (v_and_b32 / v_or_b32): v_cndmask_b32_e32 chains
// issue ~5.6x slower than plain VOP2 ops on gfx950 (tools/ubench_issue.hip)
__device__ __forceinline__ F29 quad_sel(unsigned q, const F29 &a0, const F29 &a1, const F29 &a2, const F29 &a3) {
    // a tree of three two-way bit selects (v_bfi_b32 / v_bitop3_b32: one instruction each) instead of
    // four ands and three ors per limb
    uint32_t lo = 0u - (q & 1u), hi = 0u - (q >> 1);
    asm volatile("" : "+v"(lo), "+v"(hi));                         // keep them as masks (no re-materialised compares)
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint32_t x = (a1.l[i] & lo) | (a0.l[i] & ~lo), y = (a3.l[i] & lo) | (a2.l[i] & ~lo);
        r.l[i] = (y & hi) | (x & ~hi);
    }
    return r;
}
// state replicated in the 4 lanes of the quad; branches below are quad-uniform
__device__ __forceinline__ XYZZ29 quad_dbl(const XYZZ29 &r, unsigned q) {
    if (r.is_inf()) return r;
    F29 U = add_lazy(r.Y, r.Y);
    F29 t = sqr(quad_sel(q, U, r.X, U, U));                               // L1: U^2 | X^2
    F29 V = quad_bcast<0>(t), xx = quad_bcast<1>(t);
    F29 M = add_lazy(add_lazy(xx, xx), xx).norm();
    t = mul(quad_sel(q, U, r.X, M, V), quad_sel(q, V, V, M, r.ZZ));      // L2: U*V | X*V | M^2 | V*ZZ
    F29 W = quad_bcast<0>(t), S = quad_bcast<1>(t), MM = quad_bcast<2>(t), ZZ3 = quad_bcast<3>(t);
    F29 X3 = sub_k<4>(MM, add_lazy(S, S));
    t = mul(quad_sel(q, M, W, W, W), quad_sel(q, sub_k<8>(S, X3), r.Y, r.ZZZ, r.ZZZ));   // L3: M*(S-X3) | W*Y | W*ZZZ
    F29 Y3a = quad_bcast<0>(t), WY = quad_bcast<1>(t), ZZZ3 = quad_bcast<2>(t);
    return {X3, sub_k<2>(Y3a, WY), ZZ3, ZZZ3};
}
__device__ __forceinline__ XYZZ29 quad_add(const XYZZ29 &a, const XYZZ29 &b, unsigned q) {
    if (b.is_inf()) return a;
    if (a.is_inf()) return b;
    F29 t = mul(quad_sel(q, a.X, b.X, a.Y, b.Y), quad_sel(q, b.ZZ, a.ZZ, b.ZZZ, a.ZZZ));   // L1: U1 | U2 | S1 | S2
    F29 U1 = quad_bcast<0>(t), U2 = quad_bcast<1>(t), S1 = quad_bcast<2>(t), S2 = quad_bcast<3>(t);
    F29 Pd = sub_k<2>(U2, U1), R = sub_k<2>(S2, S1);
    if (Pd.is_zero_mod_p()) {
        if (R.is_zero_mod_p()) return quad_dbl(a, q);      // (the quad's doubling: a third of the code of a lane-private one)
        return XYZZ29::inf();
    }
    t = mul(quad_sel(q, Pd, R, a.ZZ, a.ZZZ), quad_sel(q, Pd, R, b.ZZ, b.ZZZ));              // L2: P^2 | R^2 | ZZ1*ZZ2 | ZZZ1*ZZZ2
    F29 PP = quad_bcast<0>(t), RR = quad_bcast<1>(t), ZZ12 = quad_bcast<2>(t), ZZZ12 = quad_bcast<3>(t);
    t = mul(quad_sel(q, Pd, U1, ZZ12, ZZ12), PP);                                           // L3: P*PP | U1*PP | ZZ12*PP
    F29 PPP = quad_bcast<0>(t), Qv = quad_bcast<1>(t), ZZ3 = quad_bcast<2>(t);
    F29 X3 = sub_k<6>(RR, add_lazy(PPP, add_lazy(Qv, Qv)));
    t = mul(quad_sel(q, R, S1, ZZZ12, ZZZ12), quad_sel(q, sub_k<8>(Qv, X3), PPP, PPP, PPP)); // L4: R*(Q-X3) | S1*PPP | ZZZ12*PPP
    F29 Y3a = quad_bcast<0>(t), SP = quad_bcast<1>(t), ZZZ3 = quad_bcast<2>(t);
    return {X3, sub_k<2>(Y3a, SP), ZZ3, ZZZ3};
}

}  // namespace lsa

// ------------------------------------------------------------------------------------
// The same for G2 (Fq2 coordinates, fp29x2.h).  All four lanes run the same mul<K> with the
// level's largest bound constant K (a square is computed as a product so the instruction
// streams stay identical); bounds per line as in fp29x2.h.
// ------------------------------------------------------------------------------------
#include "fp29x2.h"

namespace lsa {

template <int Q>
__device__ __forceinline__ F29x2 quad_bcast(const F29x2 &v) { return {quad_bcast<Q>(v.c0), quad_bcast<Q>(v.c1)}; }
__device__ __forceinline__ F29x2 quad_sel(unsigned q, const F29x2 &a0, const F29x2 &a1, const F29x2 &a2, const F29x2 &a3) {
    return {quad_sel(q, a0.c0, a1.c0, a2.c0, a3.c0), quad_sel(q, a0.c1, a1.c1, a2.c1, a3.c1)};
}

__device__ __forceinline__ XYZZ29x2 quad_dbl(const XYZZ29x2 &r, unsigned q) {
    if (r.is_inf()) return r;
    F29x2 U = condsub4(add_lazy(r.Y, r.Y).norm());                        // [<4]
    F29x2 a = quad_sel(q, U, r.X, U, r.X);
    F29x2 t = mul<4>(a, a);                                               // L1: U^2 | X^2   (32)
    F29x2 V = quad_bcast<0>(t), xx = quad_bcast<1>(t);
    F29x2 M = add_lazy(add_lazy(xx, xx), xx).norm();                      // [<6]
    t = mul<6>(quad_sel(q, U, r.X, M, V), quad_sel(q, V, V, M, r.ZZ));    // L2: U*V | X*V | M^2 | V*ZZ   (<= 72)
    F29x2 W = quad_bcast<0>(t), S = quad_bcast<1>(t), MM = quad_bcast<2>(t), ZZ3 = quad_bcast<3>(t);
    F29x2 X3 = condsub4(sub_k<4>(MM, add_lazy(S, S)));                    // [<4]
    t = mul<6>(quad_sel(q, M, W, W, W), quad_sel(q, sub_k<4>(S, X3), r.Y, r.ZZZ, r.ZZZ));   // L3: M(S-X3) | W*Y | W*ZZZ
    F29x2 Y3a = quad_bcast<0>(t), WY = quad_bcast<1>(t), ZZZ3 = quad_bcast<2>(t);
    return {X3, sub_k<2>(Y3a, WY), ZZ3, ZZZ3};
}
__device__ __forceinline__ XYZZ29x2 quad_add(const XYZZ29x2 &a, const XYZZ29x2 &b, unsigned q) {
    if (b.is_inf()) return a;
    if (a.is_inf()) return b;
    F29x2 t = mul<2>(quad_sel(q, a.X, b.X, a.Y, b.Y), quad_sel(q, b.ZZ, a.ZZ, b.ZZZ, a.ZZZ));   // L1: U1 | U2 | S1 | S2
    F29x2 U1 = quad_bcast<0>(t), U2 = quad_bcast<1>(t), S1 = quad_bcast<2>(t), S2 = quad_bcast<3>(t);
    F29x2 Pd = sub_k<2>(U2, U1), R = sub_k<2>(S2, S1);                     // [<4]
    if (Pd.is_zero_mod_p()) {
        if (R.is_zero_mod_p()) return quad_dbl(a, q);
        return XYZZ29x2::inf();
    }
    t = mul<4>(quad_sel(q, Pd, R, a.ZZ, a.ZZZ), quad_sel(q, Pd, R, b.ZZ, b.ZZZ));                 // L2: P^2 | R^2 | ZZ1*ZZ2 | ZZZ1*ZZZ2
    F29x2 PP = quad_bcast<0>(t), RR = quad_bcast<1>(t), ZZ12 = quad_bcast<2>(t), ZZZ12 = quad_bcast<3>(t);
    t = mul<2>(quad_sel(q, Pd, U1, ZZ12, ZZ12), PP);                                              // L3: P*PP | U1*PP | ZZ12*PP
    F29x2 PPP = quad_bcast<0>(t), Qv = quad_bcast<1>(t), ZZ3 = quad_bcast<2>(t);
    F29x2 X3 = condsub4(sub_k<6>(RR, add_lazy(PPP, add_lazy(Qv, Qv))));                           // [<4]
    t = mul<6>(quad_sel(q, R, S1, ZZZ12, ZZZ12), quad_sel(q, sub_k<4>(Qv, X3), PPP, PPP, PPP));   // L4: R(Q-X3) | S1*PPP | ZZZ12*PPP
    F29x2 Y3a = quad_bcast<0>(t), SP = quad_bcast<1>(t), ZZZ3 = quad_bcast<2>(t);
    return {X3, sub_k<2>(Y3a, SP), ZZ3, ZZZ3};
}

}  // namespace lsa
