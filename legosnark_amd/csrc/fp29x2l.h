// legosnark_amd/csrc/fp29x2l.h -- Fq2 with ONE component per lane of a pair (even lane: c0, odd lane: c1), device only:
// the element type of the two-lane G2 bucket accumulation (msm.hip, k_accumulate_g2_pair).
//
// Why: an XYZZ accumulator over Fq2 is 72 words; with the products' column sums, the prefetched base and the
// temporaries of a mixed addition one lane needs ~330 registers, so the one-lane kernel runs at two wavefronts per SIMD
// with 316 B of scratch per lane.  Split by component every value halves: the pair runs the SAME instruction stream
// (the point formulas of fp29x2.h, instantiated with this type), each lane on its own component, and a product costs
// each lane one two-term reduction (dot2) plus the partner's operand components through DPP quad_perm moves:
//   c0 = a0*b0 + a1*(K p - b1)   (even lane: own a0, partner's a1)
//   c1 = a1*b0 + a0*b1           (odd lane:  own a1, partner's a0)
// i.e. dot2(own_a, b0, partner_a, odd ? b1 : K p - b1) with b0 / b1 broadcast inside the pair -- ~18 % more
// instructions per product than one lane doing both components, for half the registers and no scratch.
// Predicates (zero tests) are combined over the pair, so both lanes always take the same branch.
#pragma once
#include "fp29x2.h"

namespace lsa {

// DPP quad_perm controls: [1,0,3,2] swap inside pairs, [0,0,2,2] / [1,1,3,3] broadcast the even / odd lane's value
template <int CTRL>
__device__ __forceinline__ F29 h_dpp(const F29 &v) {
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], CTRL, 0xf, 0xf, false);
        asm volatile("" : "+v"(r.l[i]));                   // (not to be folded into the consumer: see quad29.h)
    }
    return r;
}
__device__ __forceinline__ uint32_t h_odd_mask() {
    uint32_t m = 0u - (threadIdx.x & 1u);
    asm volatile("" : "+v"(m));                            // kept as a mask (no re-materialised compares)
    return m;
}
__device__ __forceinline__ F29 h_blend(uint32_t odd, const F29 &if_odd, const F29 &if_even) {
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (if_odd.l[i] & odd) | (if_even.l[i] & ~odd);
    return r;
}
__device__ __forceinline__ bool h_both(bool mine) {
    const int other = __builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xf, 0xf, false);
    return mine && other != 0;
}

struct F29h {
    F29 v;                                                 // component (lane & 1) of the Fq2 value
    static __device__ __forceinline__ F29h zero() { return {F29::zero()}; }
    static __device__ __forceinline__ F29h one() { return {h_blend(h_odd_mask(), F29::zero(), F29::one())}; }
    __device__ __forceinline__ bool limbs_zero() const { return h_both(v.limbs_zero()); }
    __device__ __forceinline__ bool is_zero_mod_p() const { return h_both(v.is_zero_mod_p()); }
    __device__ __forceinline__ F29h norm() const { return {v.norm()}; }
};
__device__ __forceinline__ F29h add_lazy(const F29h &a, const F29h &b) { return {add_lazy(a.v, b.v)}; }
template <int K>
__device__ __forceinline__ F29h sub_k(const F29h &a, const F29h &b) { return {sub_k<K>(a.v, b.v)}; }
__device__ __forceinline__ F29h condsub4(const F29h &a) { return {condsub4(a.v)}; }

// a*b with b's components < KB*p; 2*A*KB < 169 for a's components < A*p.  [< 2; tight]
template <int KB>
__device__ __forceinline__ F29h mul(const F29h &a, const F29h &b) {
    const F29 ao = h_dpp<0xB1>(a.v);
    const F29 b0 = h_dpp<0xA0>(b.v), b1 = h_dpp<0xF5>(b.v);
    const F29 nb1 = sub_k<KB>(F29::zero(), b1);            // KB*p - b1   [<= KB; tight]
    return {dot2(a.v, b0, ao, h_blend(h_odd_mask(), b1, nb1))};
}
// a^2 with components < KA*p (tight limbs), (2*KA)^2 < 169:  c0 = (a0 + a1)(a0 - a1),  c1 = (2 a0) a1
template <int KA>
__device__ __forceinline__ F29h sqr(const F29h &a) {
    const uint32_t odd = h_odd_mask();
    const F29 ao = h_dpp<0xB1>(a.v);
    const F29 x = h_blend(odd, add_lazy(ao, ao), add_lazy(a.v, ao));      // 2 a0 | a0 + a1   [< 2KA; loose]
    const F29 y = h_blend(odd, a.v, sub_k<KA>(a.v, ao));                   // a1   | a0 - a1 + KA*p   [< 2KA; tight]
    return {mul(x, y)};
}

// this lane's halves of a packed base: x.c(lane & 1), y.c(lane & 1)
__device__ __forceinline__ AffE<F29h> unpack_affine_half(const AffPackedG2 &q, unsigned part) {
    return {{F29::unpack256(q.w[part])}, {F29::unpack256(q.w[2 + part])}};
}

}  // namespace lsa
