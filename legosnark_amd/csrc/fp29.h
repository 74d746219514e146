// legosnark_amd/csrc/fp29.h -- Fq of alt_bn128 as 9 x 29-bit unsaturated limbs, the
// representation the MSM kernels compute in.
//
// Why: on gfx950 v_mad_u64_u32 (32x32+64 -> 64) issues in 4 cycles per wave64 -- exactly as
// much as one carry instruction (v_add_co / v_addc_co / v_lshl_add_u64), while plain
// v_add_u32 / v_and_b32 / shifts issue in 2 (tools/ubench_issue.hip).  With saturated 32-bit
// limbs a Montgomery product is 136 multiplies plus ~440 carry / move instructions.  With
// 29-bit limbs, 18 limb products (< 2^58 each, or < 2^60 for "loose" operands) fit a 64-bit
// accumulator, so a product-scanning Montgomery multiplication is 162 v_mad_u64_u32 +
// 9 v_mul_lo_u32 + one 64-bit shift and one mask per column: no carry chains at all.
//
// Montgomery radix R = 2^261.  Because 121*p < 2^261, operands up to 11p give results
// < 2p with no conditional subtraction; additions are lazy and subtractions add a static
// multiple of p (signed limb arithmetic + one carry pass).  Invariants are written as
// [value bound; limb bound] next to every operation.
//
// Device-resident bases and buckets use this form; conversion from/to libff's layout
// (8 x 32-bit limbs, R = 2^256) happens once at the boundary (from_mont256 / to_mont256).
#pragma once
#include "ec.h"

namespace lsa {

// A lane mask (all ones / zero) the optimiser must not see through: it rewrites (a & m) | (b & ~m) with m
// derived from a comparison into v_cndmask_b32, which issues ~5x slower than the bit operations on gfx950.
LSA_HD uint32_t lsa_mask(uint32_t m) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(m));
#endif
    return m;
}

struct F29 {
    static constexpr int W = 29;
    static constexpr uint32_t MASK = (1u << 29) - 1;
    static constexpr uint32_t PINV = 0x04866389u;      // -p^-1 mod 2^29
    static constexpr uint32_t PINV_POS = 0x1b799c77u;  //  p^-1 mod 2^29
    uint32_t l[9];

    static LSA_HD uint32_t p(int i) {
        constexpr uint32_t P29[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u,
                                     0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
        return P29[i];
    }
    static LSA_HD F29 from_limbs(const uint32_t (&c)[9]) {
        F29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = c[i];
        return r;
    }
    static LSA_HD F29 zero() {
        F29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = 0;
        return r;
    }
    static LSA_HD F29 one() {   // 2^261 mod p
        constexpr uint32_t C[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u,
                                   0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
        return from_limbs(C);
    }
    // exact all-limbs-zero test (used for the infinity encodings, which are written as
    // literal zeros; NOT a test of "== 0 mod p" -- see is_zero_mod_p)
    LSA_HD bool limbs_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) o |= l[i];
        return o == 0;
    }

    // carry pass: limbs become < 2^29 (top limb keeps the rest).  Input limbs are signed
    // 32-bit quantities in (-2^31, 2^31); the represented value must be >= 0.
    LSA_HD F29 norm() const {
        F29 r;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int32_t v = (int32_t)l[i] + c;
            r.l[i] = (uint32_t)v & MASK;
            c = v >> 29;
        }
        r.l[8] = (uint32_t)((int32_t)l[8] + c);
        return r;
    }

    // lazy sum: [a+b; limbs add]
    friend LSA_HD F29 add_lazy(const F29 &a, const F29 &b) {
        F29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
        return r;
    }
    // a - b + K*p, carry-normalised.  Requires b < K*p (value) so the result is positive,
    // and limbs a_i + (Kp)_i - b_i within int32.  [a - b + Kp; tight]
    template <int K>
    friend LSA_HD F29 sub_k(const F29 &a, const F29 &b);

    // Montgomery product a*b/2^261 mod p.  Requires limb products a_i*b_j <= 2^60 (both
    // operands "loose" < 2^30 is fine) and a*b < 121 p^2.  [< 2p; tight]
    // (-DLSA_FP29_COLS: mul, sqr and dot2 column-wise -- independent column accumulators, as fs29.h: f29_dot_cols.  An A/B
    // switch of round 6; the serial forms stay the default.)
    static LSA_HD F29 redc_cols(uint64_t (&col)[17]) {
        uint64_t carry = 0;
        F29 r;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            uint64_t acc = col[k] + carry;
            const uint32_t m = ((uint32_t)acc * PINV) & MASK;
            acc += (uint64_t)m * p(0);
            carry = acc >> 29;
#pragma unroll
            for (int j = 1; j < 9; j++) col[k + j] += (uint64_t)m * p(j);
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
            const uint64_t acc = col[k] + carry;
            r.l[k - 9] = (uint32_t)acc & MASK;
            carry = acc >> 29;
        }
        r.l[8] = (uint32_t)carry;
        return r;
    }
    friend LSA_HD F29 mul(const F29 &a, const F29 &b) {
#if defined(LSA_FP29_COLS)
        {
            uint64_t col[17];
#pragma unroll
            for (int k = 0; k < 17; k++) col[k] = 0;
#pragma unroll
            for (int i = 0; i < 9; i++)
#pragma unroll
                for (int j = 0; j < 9; j++) col[i + j] += (uint64_t)a.l[i] * b.l[j];
            return redc_cols(col);
        }
#endif
        uint64_t acc = 0;
        uint32_t m[9];
        F29 r;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * p(k - i);
            m[k] = ((uint32_t)acc * PINV) & MASK;
            acc += (uint64_t)m[k] * p(0);
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; i < 9; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
            for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * p(k - i);
            r.l[k - 9] = (uint32_t)acc & MASK;
            acc >>= 29;
        }
        r.l[8] = (uint32_t)acc;
        return r;
    }
    // a^2: cross terms once with doubled limbs (45 limb products instead of 81).  Input
    // limbs may be loose (< 2^30): 4*2^61 + 2^60 + 9*2^58 < 2^64 per column.  [< 2p; tight]
    friend LSA_HD F29 sqr(const F29 &a) {
        uint32_t d[9];
#pragma unroll
        for (int i = 0; i < 9; i++) d[i] = a.l[i] << 1;
#if defined(LSA_FP29_COLS)
        {
            uint64_t col[17];
#pragma unroll
            for (int k = 0; k < 17; k++) col[k] = 0;
#pragma unroll
            for (int i = 0; i < 9; i++) {
                col[2 * i] += (uint64_t)a.l[i] * a.l[i];
#pragma unroll
                for (int j = i + 1; j < 9; j++) col[i + j] += (uint64_t)a.l[i] * d[j];
            }
            return redc_cols(col);
        }
#endif
        uint64_t acc = 0;
        uint32_t m[9];
        F29 r;
#pragma unroll
        for (int k = 0; k < 17; k++) {
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int j = k - i;
                if (j > i && j < 9) acc += (uint64_t)a.l[i] * d[j];
            }
            if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
            if (k < 9) {
#pragma unroll
                for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * p(k - i);
                m[k] = ((uint32_t)acc * PINV) & MASK;
                acc += (uint64_t)m[k] * p(0);
            } else {
#pragma unroll
                for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * p(k - i);
                r.l[k - 9] = (uint32_t)acc & MASK;
            }
            acc >>= 29;
        }
        r.l[8] = (uint32_t)acc;
        return r;
    }
    // (a0*b0 + a1*b1) / 2^261 mod p with ONE reduction: both products accumulate in the same
    // 64-bit columns.  Needs a0*b0 + a1*b1 < 169 p^2 and at most one loose operand per
    // product (18*2^59 + 9*2^58 < 2^64).  [< 2p; tight]
    friend LSA_HD F29 dot2(const F29 &a0, const F29 &b0, const F29 &a1, const F29 &b1) {
#if defined(LSA_FP29_COLS)
        {
            uint64_t col[17];
#pragma unroll
            for (int k = 0; k < 17; k++) col[k] = 0;
#pragma unroll
            for (int i = 0; i < 9; i++)
#pragma unroll
                for (int j = 0; j < 9; j++) { col[i + j] += (uint64_t)a0.l[i] * b0.l[j]; col[i + j] += (uint64_t)a1.l[i] * b1.l[j]; }
            return redc_cols(col);
        }
#endif
        uint64_t acc = 0;
        uint32_t m[9];
        F29 r;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) {
                acc += (uint64_t)a0.l[i] * b0.l[k - i];
                acc += (uint64_t)a1.l[i] * b1.l[k - i];
            }
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * p(k - i);
            m[k] = ((uint32_t)acc * PINV) & MASK;
            acc += (uint64_t)m[k] * p(0);
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; i < 9; i++) {
                acc += (uint64_t)a0.l[i] * b0.l[k - i];
                acc += (uint64_t)a1.l[i] * b1.l[k - i];
            }
#pragma unroll
            for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * p(k - i);
            r.l[k - 9] = (uint32_t)acc & MASK;
            acc >>= 29;
        }
        r.l[8] = (uint32_t)acc;
        return r;
    }

    // (a0*b0 + a1*b1 + a2*b2 + a3*b3) / 2^261 mod p with ONE reduction.  All limbs tight (< 2^29, top limbs as their values
    // allow): 36 + 9 limb products of < 2^58 per column stay below 2^64.  With T = sum a_i*b_i the result is
    // < T / 2^261 + p, i.e. < 4p for T < 507 p^2 (callers state their T).  [tight]
    friend LSA_HD F29 dot4(const F29 &a0, const F29 &b0, const F29 &a1, const F29 &b1, const F29 &a2, const F29 &b2, const F29 &a3, const F29 &b3) {
        uint64_t acc = 0;
        uint32_t m[9];
        F29 r;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) {
                acc += (uint64_t)a0.l[i] * b0.l[k - i];
                acc += (uint64_t)a1.l[i] * b1.l[k - i];
                acc += (uint64_t)a2.l[i] * b2.l[k - i];
                acc += (uint64_t)a3.l[i] * b3.l[k - i];
            }
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * p(k - i);
            m[k] = ((uint32_t)acc * PINV) & MASK;
            acc += (uint64_t)m[k] * p(0);
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; i < 9; i++) {
                acc += (uint64_t)a0.l[i] * b0.l[k - i];
                acc += (uint64_t)a1.l[i] * b1.l[k - i];
                acc += (uint64_t)a2.l[i] * b2.l[k - i];
                acc += (uint64_t)a3.l[i] * b3.l[k - i];
            }
#pragma unroll
            for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * p(k - i);
            r.l[k - 9] = (uint32_t)acc & MASK;
            acc >>= 29;
        }
        r.l[8] = (uint32_t)acc;
        return r;
    }
    // cu*u + cv*v + K*p for small signed cu, cv and K >= 0 chosen by the caller so that the value is non-negative
    // (u, v tight); carry-normalised: limbs < 2^29, the value must stay below 2^261.  [tight]
    friend LSA_HD F29 lin2(const F29 &u, int cu, const F29 &v, int cv, int K) {
        F29 r;
        int64_t c = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
#if defined(__HIP_DEVICE_COMPILE__)
            // three signed 32 x 32 + 64 multiply-adds per limb (left to itself the compiler splits every term in two)
            int64_t t = c;
            asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(t) : "v"(K), "v"((int32_t)p(i)) : "vcc");
            asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(t) : "v"(cu), "v"((int32_t)u.l[i]) : "vcc");
            asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(t) : "v"(cv), "v"((int32_t)v.l[i]) : "vcc");
#else
            const int64_t t = (int64_t)cu * (int64_t)u.l[i] + (int64_t)cv * (int64_t)v.l[i] + (int64_t)K * (int64_t)p(i) + c;
#endif
            if (i < 8) { r.l[i] = (uint32_t)t & MASK; c = t >> 29; }
            else r.l[i] = (uint32_t)t;
        }
        return r;
    }

    // value == 0 (mod p) for a tight value < 16p.  Necessary condition first: a multiple
    // k*p has low limb k*p_0, so k = l0 * p_0^-1 mod 2^29 must be < 16 (false positives
    // 2^-25); the exact comparison runs only then.
    LSA_HD bool is_zero_mod_p() const {
        uint32_t k = (l[0] * PINV_POS) & MASK;
        if (k >= 16) return false;
        // compare with k*p limb by limb
        uint64_t c = 0;
        bool eq = true;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            c += (uint64_t)k * p(i);
            uint32_t want = (i < 8) ? ((uint32_t)c & MASK) : (uint32_t)c;
            eq = eq && (want == l[i]);
            c >>= 29;
        }
        return eq;
    }

    // fully reduced representative in [0, p) of a tight value < 16p
    LSA_HD F29 canonical() const {
        F29 t = *this;
#pragma unroll
        for (int s = 3; s >= 0; s--) {
            // d = t - (p << s): keep if non-negative
            F29 d;
            int32_t c = 0;
            uint64_t pc = 0;
#pragma unroll
            for (int i = 0; i < 9; i++) {
                pc += (uint64_t)p(i) << s;
                uint32_t pl = (i < 8) ? ((uint32_t)pc & MASK) : (uint32_t)pc;
                pc >>= 29;
                int32_t v = (int32_t)t.l[i] - (int32_t)pl + c;
                if (i < 8) { d.l[i] = (uint32_t)v & MASK; c = v >> 29; }
                else d.l[i] = (uint32_t)v;
            }
            bool neg = ((int32_t)d.l[8]) < 0;
#pragma unroll
            for (int i = 0; i < 9; i++) t.l[i] = neg ? t.l[i] : d.l[i];
        }
        return t;
    }

    // ---- boundary conversions ----------------------------------------------------------
    // 256-bit little-endian words -> 29-bit limbs (no arithmetic)
    static LSA_HD F29 unpack256(const uint32_t w[8]) {
        F29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int bit = 29 * i, j = bit >> 5, s = bit & 31;
            uint64_t two = (uint64_t)w[j] | ((uint64_t)(j + 1 < 8 ? w[j + 1] : 0) << 32);
            r.l[i] = (uint32_t)(two >> s) & MASK;
        }
        return r;
    }
    // tight value < 2^256 -> 8 x 32-bit words
    LSA_HD void pack256(uint32_t w[8]) const {
#pragma unroll
        for (int j = 0; j < 8; j++) w[j] = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int bit = 29 * i, j = bit >> 5, s = bit & 31;
            uint64_t v = (uint64_t)l[i] << s;
            w[j] |= (uint32_t)v;
            if (j + 1 < 8) w[j + 1] |= (uint32_t)(v >> 32);
        }
    }
    // libff Fq (x*2^256 mod p, canonical) -> x*2^261 mod p   [< 2p; tight]
    static LSA_HD F29 from_mont256(const Fq &v) {
        constexpr uint32_t C[9] = {0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u,
                                   0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};   // 2^266 mod p
        return mul(unpack256(v.l), from_limbs(C));
    }
    // x*2^261 (tight, < 11p) -> libff Fq (canonical x*2^256 mod p)
    LSA_HD Fq to_mont256() const {
        constexpr uint32_t D[9] = {0x058f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u,
                                   0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};   // 2^256 mod p
        F29 t = mul(*this, from_limbs(D)).canonical();
        Fq r;
        t.pack256(r.l);
        return r;
    }
};

template <int K>
LSA_HD F29 sub_k(const F29 &a, const F29 &b) {
    // K*p in tight limbs, folded at compile time
    F29 r;
    uint64_t pc = 0;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        pc += (uint64_t)F29::p(i) * (uint32_t)K;
        uint32_t kp = (i < 8) ? ((uint32_t)pc & F29::MASK) : (uint32_t)pc;
        pc >>= 29;
        int32_t v = (int32_t)a.l[i] + (int32_t)kp - (int32_t)b.l[i] + c;
        if (i < 8) { r.l[i] = (uint32_t)v & F29::MASK; c = v >> 29; }
        else r.l[i] = (uint32_t)v;
    }
    return r;
}

// ------------------------------------------------------------------------------------
// points over F29.  Invariants of an accumulator (X,Y,ZZ,ZZZ), all limbs tight:
//     X < 8p,  Y < 4p,  ZZ < 2p,  ZZZ < 2p;  infinity <=> ZZ has all limbs zero.
// Bases (x,y) are canonical (< p); (0,0) encodes infinity.
// ------------------------------------------------------------------------------------
struct Aff29 {
    F29 x, y;
    LSA_HD bool is_inf() const { return x.limbs_zero() && y.limbs_zero(); }
};
// packed device-resident base: canonical x*2^261, y*2^261 as 2 x 256-bit words (64 B)
struct AffPacked {
    uint32_t x[8], y[8];
};
struct XYZZ29 {
    F29 X, Y, ZZ, ZZZ;
    LSA_HD bool is_inf() const { return ZZ.limbs_zero(); }
    static LSA_HD XYZZ29 inf() { return {F29::zero(), F29::zero(), F29::zero(), F29::zero()}; }
};

LSA_HD Aff29 unpack_affine(const AffPacked &q) { return {F29::unpack256(q.x), F29::unpack256(q.y)}; }

// 2*(x,y), affine input (mdbl-2008-s-1).  Output within the accumulator invariants.
LSA_HD XYZZ29 xyzz29_dbl_affine(const Aff29 &b) {
    F29 U = add_lazy(b.y, b.y);                    // [2p; loose]
    F29 V = sqr(U);                                // [<2p]
    F29 W = mul(U, V);
    F29 S = mul(b.x, V);
    F29 xx = sqr(b.x);
    F29 M = add_lazy(add_lazy(xx, xx), xx).norm(); // [<6p; tight]
    F29 X3 = sub_k<4>(sqr(M), add_lazy(S, S));     // M^2 - 2S + 4p   [<6p]
    F29 Y3 = dot2(M, sub_k<8>(S, X3), W, sub_k<1>(F29::zero(), b.y));   // M(S-X3) + W(p-y): 60 + 2  [<2p]
    return {X3, Y3, V, W};
}

// 2*P (dbl-2008-s-1)
LSA_HD XYZZ29 xyzz29_dbl(const XYZZ29 &a) {
    if (a.is_inf()) return a;
    F29 U = add_lazy(a.Y, a.Y);                    // [<8p; loose]
    F29 V = sqr(U);
    F29 W = mul(U, V);
    F29 S = mul(a.X, V);
    F29 xx = sqr(a.X);
    F29 M = add_lazy(add_lazy(xx, xx), xx).norm(); // [<6p; tight]
    F29 X3 = sub_k<4>(sqr(M), add_lazy(S, S));     // [<6p]
    F29 Y3 = dot2(M, sub_k<8>(S, X3), W, sub_k<4>(F29::zero(), a.Y));   // M(S-X3) + W(4p-Y): 60 + 8  [<2p]
    return {X3, Y3, mul(V, a.ZZ), mul(W, a.ZZZ)};
}

// acc + (x2,y2), complete (madd-2008-s).  The hot operation: 8M + 2S, ~2300 instructions.
LSA_HD XYZZ29 xyzz29_madd(const XYZZ29 &a, const Aff29 &b) {
    if (b.is_inf()) return a;
    if (a.is_inf()) return {b.x, b.y, F29::one(), F29::one()};
    F29 U2 = mul(b.x, a.ZZ);                       // [<2p]
    F29 S2 = mul(b.y, a.ZZZ);
    F29 Pd = sub_k<8>(U2, a.X);                    // U2 - X1 + 8p   [<10p]
    F29 R = sub_k<4>(S2, a.Y);                     // S2 - Y1 + 4p   [<6p]
    if (Pd.is_zero_mod_p()) {
        if (R.is_zero_mod_p()) return xyzz29_dbl_affine(b);
        return XYZZ29::inf();
    }
    F29 PP = sqr(Pd);
    F29 PPP = mul(Pd, PP);
    F29 Q = mul(a.X, PP);
    F29 X3 = sub_k<6>(sqr(R), add_lazy(PPP, add_lazy(Q, Q)));          // R^2 - PPP - 2Q + 6p  [<8p]
    F29 Y3 = dot2(R, sub_k<8>(Q, X3), sub_k<4>(F29::zero(), a.Y), PPP);   // R(Q-X3) + (4p-Y1)PPP: 60 + 8  [<2p]
    return {X3, Y3, mul(a.ZZ, PP), mul(a.ZZZ, PPP)};
}

// a + b, complete (add-2008-s).  Inputs and output within the accumulator invariants.
LSA_HD XYZZ29 xyzz29_add(const XYZZ29 &a, const XYZZ29 &b) {
    if (b.is_inf()) return a;
    if (a.is_inf()) return b;
    F29 U1 = mul(a.X, b.ZZ);
    F29 U2 = mul(b.X, a.ZZ);
    F29 S1 = mul(a.Y, b.ZZZ);
    F29 S2 = mul(b.Y, a.ZZZ);
    F29 Pd = sub_k<2>(U2, U1);                     // [<4p]
    F29 R = sub_k<2>(S2, S1);
    if (Pd.is_zero_mod_p()) {
        if (R.is_zero_mod_p()) return xyzz29_dbl(a);
        return XYZZ29::inf();
    }
    F29 PP = sqr(Pd);
    F29 PPP = mul(Pd, PP);
    F29 Q = mul(U1, PP);
    F29 X3 = sub_k<6>(sqr(R), add_lazy(PPP, add_lazy(Q, Q)));
    F29 Y3 = dot2(R, sub_k<8>(Q, X3), sub_k<2>(F29::zero(), S1), PPP);    // R(Q-X3) + (2p-S1)PPP: 40 + 4  [<2p]
    return {X3, Y3, mul(mul(a.ZZ, b.ZZ), PP), mul(mul(a.ZZZ, b.ZZZ), PPP)};
}

LSA_HD XYZZ29 xyzz29_neg(const XYZZ29 &a) {
    if (a.is_inf()) return a;
    return {a.X, sub_k<4>(F29::zero(), a.Y), a.ZZ, a.ZZZ};   // 4p - Y  [<=4p]
}

// XYZZ29 -> libff Jacobian (Montgomery R = 2^256, canonical limbs): Z = ZZZ, X' = X*ZZ^2,
// Y' = Y*ZZZ^2.
LSA_HD Jac<Fq> xyzz29_to_jac(const XYZZ29 &a) {
    if (a.is_inf()) return Jac<Fq>::inf();
    F29 Xj = mul(a.X, sqr(a.ZZ));
    F29 Yj = mul(a.Y, sqr(a.ZZZ));
    return {Xj.to_mont256(), Yj.to_mont256(), a.ZZZ.to_mont256()};
}

}  // namespace lsa
