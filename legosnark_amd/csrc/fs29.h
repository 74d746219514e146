// legosnark_amd/csrc/fs29.h -- "strict" Fq on 9 x 29-bit limbs: the base field the pairing
// kernels' extension tower (tower.h: Fq2T<B>, Fq6T<B>, Fq12T<B>) is instantiated on.
//
// fp29.h is a lazy representation whose callers track bounds by hand (the MSM formulas).
// The tower has hundreds of additions, so here every value is kept tight and < 2p after
// every operation: products use the carry-free 29-bit Montgomery multiplication (inputs
// < 2p give outputs < 2p), sums and differences take one conditional subtraction of 2p
// done with bit masks (no v_cndmask).  Equality is equality mod p.
#pragma once
#include "fp29.h"
#include "inv29.h"
#include "fp29x2.h"

namespace lsa {

// namespace-scope names for F29's hidden friends (usable where a member named sqr / mul hides them)
LSA_HD F29 f29_sqr(const F29 &a) { return sqr(a); }
LSA_HD F29 f29_mul(const F29 &a, const F29 &b) { return mul(a, b); }

// sum_{t < N} a[t] * b[t] / 2^261 mod p with ONE reduction, written COLUMN-WISE (round 6): the 17 column sums of the
// schoolbook products are accumulated apart -- consecutive multiply-adds go to different 64-bit accumulators -- and the
// Montgomery factor m_k of column k is added into the eight later columns as soon as it is known.  fp29.h's mul / dot2 /
// dot4 run one accumulator through all columns, every multiply-add waiting for the one before it (a dependent
// v_mad_u64_u32 issues every 11 cycles in tools/ubench_clock.hip): the suspicion was that the pairing kernels, at one or two
// wavefronts per SIMD, ran at the speed of that chain.  MEASURED (-DLSA_F29_COLS, profiles/r06_v5_*): no -- the fused Miller
// kernel 1.03 -> 1.05 ms, k_g2_precomp 0.66 -> 0.66 ms, config 5 1.54 -> 1.58 ms; likewise the Fr kernels (fr29.h) and
// the MSM's accumulate kernel (fp29.h).  The serial forms stay; this one is kept behind the switch and in the host tests.
// Needs tight limbs everywhere and 9 N + 9 <= 64 products of < 2^58 per column; the result is < T / 2^261 + p for the
// integer sum T (< 2p for T < 169 p^2; callers with more state their bound).  Same value as the serial forms.  [tight]
template <int N>
LSA_HD F29 f29_dot_cols(const F29 (&a)[N], const F29 (&b)[N]) {
    static_assert(9 * N + 9 <= 64, "64-bit columns hold 64 products of 29-bit limbs");
    uint64_t col[17];
#pragma unroll
    for (int k = 0; k < 17; k++) col[k] = 0;
#pragma unroll
    for (int t = 0; t < N; t++)
#pragma unroll
        for (int i = 0; i < 9; i++)
#pragma unroll
            for (int j = 0; j < 9; j++) col[i + j] += (uint64_t)a[t].l[i] * b[t].l[j];
    uint64_t carry = 0;
    F29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        uint64_t acc = col[k] + carry;
        const uint32_t m = ((uint32_t)acc * F29::PINV) & F29::MASK;
        acc += (uint64_t)m * F29::p(0);
        carry = acc >> 29;
#pragma unroll
        for (int j = 1; j < 9; j++) col[k + j] += (uint64_t)m * F29::p(j);
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
        const uint64_t acc = col[k] + carry;
        r.l[k - 9] = (uint32_t)acc & F29::MASK;
        carry = acc >> 29;
    }
    r.l[8] = (uint32_t)carry;
    return r;
}
LSA_HD F29 f29_dot2_cols(const F29 &a0, const F29 &b0, const F29 &a1, const F29 &b1) {
    const F29 a[2] = {a0, a1}, b[2] = {b0, b1};
    return f29_dot_cols<2>(a, b);
}

// tight value < 4p -> same residue, < 2p
LSA_HD F29 condsub2(const F29 &t) {
    F29 d;
    int32_t c = 0;
    uint64_t pc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        pc += (uint64_t)F29::p(i) * 2u;
        uint32_t pl = (i < 8) ? ((uint32_t)pc & F29::MASK) : (uint32_t)pc;
        pc >>= 29;
        int32_t v = (int32_t)t.l[i] - (int32_t)pl + c;
        if (i < 8) { d.l[i] = (uint32_t)v & F29::MASK; c = v >> 29; }
        else d.l[i] = (uint32_t)v;
    }
    const uint32_t keep = (uint32_t)((int32_t)d.l[8] >> 31);   // all ones if t < 2p
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (t.l[i] & keep) | (d.l[i] & ~keep);
    return r;
}

// x/2 for a tight x: (x + p*(x odd)) >> 1.  Same field element as x * two_inv.
LSA_HD F29 f29_halve(const F29 &v) {
    const uint32_t odd = 0u - (v.l[0] & 1u);
    F29 t;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        uint32_t s = v.l[i] + (F29::p(i) & odd) + c;
        if (i < 8) { t.l[i] = s & F29::MASK; c = s >> 29; }
        else t.l[i] = s;
    }
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (t.l[i] >> 1) | (i < 8 ? ((t.l[i + 1] & 1u) << 28) : 0u);
    return r;
}

struct Fs {
    F29 v;   // tight, < 2p
    static LSA_HD Fs zero() { return {F29::zero()}; }
    static LSA_HD Fs one() { return {F29::one()}; }
    static LSA_HD Fs from_mont256(const Fq &x) { return {F29::from_mont256(x)}; }
    LSA_HD Fq to_mont256() const { return v.to_mont256(); }
    LSA_HD bool is_zero() const { return v.is_zero_mod_p(); }
    friend LSA_HD Fs operator*(const Fs &a, const Fs &b) { return {f29_mul(a.v, b.v)}; }
    LSA_HD Fs sqr() const { return {f29_sqr(v)}; }
    friend LSA_HD Fs operator+(const Fs &a, const Fs &b) { return {condsub2(add_lazy(a.v, b.v).norm())}; }
    friend LSA_HD Fs operator-(const Fs &a, const Fs &b) { return {condsub2(sub_k<2>(a.v, b.v))}; }
    LSA_HD Fs neg() const { return {condsub2(sub_k<2>(F29::zero(), v))}; }
    LSA_HD Fs dbl() const { return *this + *this; }
    LSA_HD Fs halve() const { return {f29_halve(v)}; }     // < 1.5p
    LSA_HD bool operator==(const Fs &b) const { return (*this - b).is_zero(); }
    LSA_HD bool operator!=(const Fs &b) const { return !(*this == b); }
    // constant-time binary GCD with 64-bit approximations (inv29.h): ~1/4 of the instructions of the power below,
    // the same instruction stream for every input (lanes that invert different values do not diverge); 0 -> 0
    LSA_HD_NOINLINE Fs inverse() const { return {f29_inverse(v)}; }
    // a^(p-2) (Fermat): the reference the host tests compare the one above with
    LSA_HD_NOINLINE Fs inverse_fermat() const {
        uint32_t e[8];
        uint64_t br = 2;
        for (int i = 0; i < 8; i++) {
            uint64_t x = (uint64_t)FqParams::MOD[i] - br;
            e[i] = (uint32_t)x;
            br = (x >> 32) & 1;
        }
        Fs acc = one();
        for (int i = 255; i >= 0; --i) {
            acc = acc.sqr();
            if ((e[i >> 5] >> (i & 31)) & 1) acc = acc * *this;
        }
        return acc;
    }
};

// 1 / (a0 + a1 u) = (a0 - a1 u) / (a0^2 + a1^2): one inversion in Fq.  Components < 2p, tight, in and out.
LSA_HD F29x2 f29x2_inverse(const F29x2 &a) {
    const F29 nrm = condsub2(add_lazy(sqr(a.c0), sqr(a.c1)).norm());                   // a0^2 + a1^2  [< 2p]
    const F29 ni = Fs{nrm}.inverse().v;
    return {mul(a.c0, ni), mul(sub_k<2>(F29::zero(), a.c1), ni)};
}

}  // namespace lsa
