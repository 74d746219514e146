// legosnark_amd/csrc/fr29.h -- the scalar field Fr of alt_bn128 as 9 x 29-bit unsaturated limbs, for the NTT
// (ntt.hip).  Same construction as fp29.h (which is Fq: the MSM kernels' field): product-scanning Montgomery
// multiplication with R = 2^261, 18 limb products per 64-bit column, no carry chains -- 175 G products/s on gfx950
// against 100 G/s for the 8 x 32-bit CIOS of fp.h.
//
// How libff's values enter and leave without conversion products: an Fr value x travels as the canonical integer
// X = x * 2^256 mod r (libff's Montgomery form).  The NTT is linear, so the kernels read X as if it were the R = 2^261
// form of x' = x * 2^-5; every constant they multiply by (twiddles, coset powers, 1/n) IS in 2^261 form, so products and
// sums stay in that "shifted" form and the outputs, read back as integers, are y * 2^256 mod r: libff's bytes.
//
// Bounds: 121 r < 2^261, so mul(a, b) with a * b < 121 r^2 returns a value < 2r with tight limbs (< 2^29, the top limb
// carries the rest).  Sums and differences are carry-normalised; a butterfly output grows by 2r per stage and is brought
// back below 2r by the next product (a twiddle, the inter-pass twiddle, or the final scale constant).
#pragma once
#include <stdint.h>

#include "fp.h"

namespace lsa {

struct Fr29 {
    static constexpr uint32_t MASK = (1u << 29) - 1;
    static constexpr uint32_t RINV = 0x0fffffffu;          // -r^-1 mod 2^29
    uint32_t l[9];

    static LSA_HD uint32_t r(int i) {
        constexpr uint32_t R29[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u,
                                     0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
        return R29[i];
    }
    static LSA_HD Fr29 zero() {
        Fr29 z;
#pragma unroll
        for (int i = 0; i < 9; i++) z.l[i] = 0;
        return z;
    }
    static LSA_HD Fr29 one() {                              // 2^261 mod r
        constexpr uint32_t C[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu,
                                   0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
        Fr29 o;
#pragma unroll
        for (int i = 0; i < 9; i++) o.l[i] = C[i];
        return o;
    }
    // 256-bit little-endian words <-> limbs (shifts only).  pack256 needs a tight value < 2^256.
    static LSA_HD Fr29 unpack256(const uint32_t w[8]) {
        Fr29 x;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int bit = 29 * i, j = bit >> 5, s = bit & 31;
            const uint64_t two = (uint64_t)w[j] | ((uint64_t)(j + 1 < 8 ? w[j + 1] : 0) << 32);
            x.l[i] = (uint32_t)(two >> s) & MASK;
        }
        return x;
    }
    LSA_HD void pack256(uint32_t w[8]) const {
#pragma unroll
        for (int j = 0; j < 8; j++) w[j] = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int bit = 29 * i, j = bit >> 5, s = bit & 31;
            const uint64_t v = (uint64_t)l[i] << s;
            w[j] |= (uint32_t)v;
            if (j + 1 < 8) w[j + 1] |= (uint32_t)(v >> 32);
        }
    }
    // a * b / 2^261 mod r.  Limbs of both operands tight; a * b < 121 r^2.  [< 2r; tight]
    // (-DLSA_FR29_COLS: column-wise -- the 17 column sums accumulated apart, every Montgomery factor added into the eight later
    // columns as soon as it is known, no serial accumulator; fs29.h: f29_dot_cols.  An A/B switch of round 6: the sumcheck
    // round polynomial 0.476 -> 0.488 ms (two tables) / 0.86 -> 0.93 ms (three), NTT 2^24 2.41 ms either way.  Serial stays.)
    friend LSA_HD Fr29 mul(const Fr29 &a, const Fr29 &b) {
#if !defined(LSA_FR29_COLS)
        uint64_t acc = 0;
        uint32_t m[9];
        Fr29 o;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * r(k - i);
            m[k] = ((uint32_t)acc * RINV) & MASK;
            acc += (uint64_t)m[k] * r(0);
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; i < 9; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
            for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * r(k - i);
            o.l[k - 9] = (uint32_t)acc & MASK;
            acc >>= 29;
        }
        o.l[8] = (uint32_t)acc;
        return o;
#else
        uint64_t col[17];
#pragma unroll
        for (int k = 0; k < 17; k++) col[k] = 0;
#pragma unroll
        for (int i = 0; i < 9; i++)
#pragma unroll
            for (int j = 0; j < 9; j++) col[i + j] += (uint64_t)a.l[i] * b.l[j];
        uint64_t carry = 0;
        Fr29 o;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            uint64_t acc = col[k] + carry;
            const uint32_t m = ((uint32_t)acc * RINV) & MASK;
            acc += (uint64_t)m * r(0);
            carry = acc >> 29;
#pragma unroll
            for (int j = 1; j < 9; j++) col[k + j] += (uint64_t)m * r(j);
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
            const uint64_t acc = col[k] + carry;
            o.l[k - 9] = (uint32_t)acc & MASK;
            carry = acc >> 29;
        }
        o.l[8] = (uint32_t)carry;
        return o;
#endif
    }
    // a + b, carry-normalised.  [a + b; tight]
    friend LSA_HD Fr29 add(const Fr29 &a, const Fr29 &b) {
        Fr29 o;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const uint32_t v = a.l[i] + b.l[i] + c;
            if (i < 8) { o.l[i] = v & MASK; c = v >> 29; }
            else o.l[i] = v;
        }
        return o;
    }
    // a - b + 2r for b < 2r (the result is non-negative), carry-normalised.  [a + 2r; tight]
    friend LSA_HD Fr29 sub2r(const Fr29 &a, const Fr29 &b) {
        Fr29 o;
        int32_t c = 0;
        uint32_t rc = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            rc += r(i) * 2u;
            const uint32_t rl = (i < 8) ? (rc & MASK) : rc;
            rc >>= 29;
            const int32_t v = (int32_t)a.l[i] - (int32_t)b.l[i] + (int32_t)rl + c;
            if (i < 8) { o.l[i] = (uint32_t)v & MASK; c = v >> 29; }
            else o.l[i] = (uint32_t)v;
        }
        return o;
    }
    // a - b + 4r for b < 4r, carry-normalised.  [a + 4r; tight]
    friend LSA_HD Fr29 sub4r(const Fr29 &a, const Fr29 &b) {
        Fr29 o;
        int32_t c = 0;
        uint32_t rc = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            rc += r(i) * 4u;
            const uint32_t rl = (i < 8) ? (rc & MASK) : rc;
            rc >>= 29;
            const int32_t v = (int32_t)a.l[i] - (int32_t)b.l[i] + (int32_t)rl + c;
            if (i < 8) { o.l[i] = (uint32_t)v & MASK; c = v >> 29; }
            else o.l[i] = (uint32_t)v;
        }
        return o;
    }
    // ---- sums without carries (round 6, second session: the NTT's first stage of a stage pair).  A value whose limbs 0..7 may
    // exceed 29 bits ("loose") is still a valid operand of mul as long as they stay below 2^31 and the OTHER operand is tight:
    // a column is at most 9 * 2^31 * 2^29 + 9 * 2^58 + a carry < 2^64.
    // a + b limb by limb.  [a + b; limbs < 2^29 + those of a]
    friend LSA_HD Fr29 add_loose(const Fr29 &a, const Fr29 &b) {
        Fr29 o;
#pragma unroll
        for (int i = 0; i < 9; i++) o.l[i] = a.l[i] + b.l[i];
        return o;
    }
    // limb i of 3r "lifted": 3r written with limbs 0..7 >= 2^29 - 1 (limb 0: >= 2^29) and the top limb one less than its
    // normalised value -- the same integer (the borrowed 2^29 (i+1) of every limb is the lower neighbour's 2^29 more), so that
    // a_i + lift_i - b_i is non-negative for any tight b < 3r - 2^232 (a product's < 2r is) and NO borrow runs between limbs
    static LSA_HD uint32_t lift3r(int i) {
        uint64_t c = 0;
        uint32_t n = 0;
        for (int k = 0; k <= i; k++) {
            c += (uint64_t)r(k) * 3u;
            n = k < 8 ? (uint32_t)c & MASK : (uint32_t)c;
            c >>= 29;
        }
        return i == 0 ? n + (1u << 29) : (i < 8 ? n + MASK : n - 1u);
    }
    // a - b + 3r limb by limb, b tight and < 2r.  [a + 3r; limbs < 2^30 + those of a]
    friend LSA_HD Fr29 sub3r_loose(const Fr29 &a, const Fr29 &b) {
        Fr29 o;
#pragma unroll
        for (int i = 0; i < 9; i++) o.l[i] = a.l[i] + lift3r(i) - b.l[i];
        return o;
    }
    // the same, carry-normalised (unsigned carries: a's limbs may be loose, up to 3 * 2^29).  [a + 3r; tight]
    friend LSA_HD Fr29 sub3r_norm(const Fr29 &a, const Fr29 &b) {
        Fr29 o;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const uint32_t v = a.l[i] + lift3r(i) - b.l[i] + c;
            if (i < 8) { o.l[i] = v & MASK; c = v >> 29; }
            else o.l[i] = v;
        }
        return o;
    }
    // the representative in [0, r) of a tight value < 2r
    LSA_HD Fr29 canonical2() const {
        Fr29 d;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int32_t v = (int32_t)l[i] - (int32_t)r(i) + c;
            if (i < 8) { d.l[i] = (uint32_t)v & MASK; c = v >> 29; }
            else d.l[i] = (uint32_t)v;
        }
        const uint32_t keep = (uint32_t)((int32_t)d.l[8] >> 31);      // all ones if the value was < r
        Fr29 o;
#pragma unroll
        for (int i = 0; i < 9; i++) o.l[i] = (l[i] & keep) | (d.l[i] & ~keep);
        return o;
    }
    // libff Fr (x * 2^256, canonical) read as the limbs of the shifted form (see the header comment): no arithmetic
    static LSA_HD Fr29 from_words(const Fr &x) { return unpack256(x.l); }
    LSA_HD Fr to_words() const {                            // tight, < 2^256
        Fr x;
        pack256(x.l);
        return x;
    }
};

// Sums of products with ONE reduction: the 17 column sums of up to FOUR schoolbook products a (x) b of tight operands
// (a column takes at most 9 limb products of 58 bits per product: 36 * 2^58 < 2^64 together with the reduction's own
// terms), then a single Montgomery reduction -- 4 * 81 + 81 multiply-adds for four products instead of 4 * 162.  For a dot
// product sum v_j W_j the partial sums are what is wanted anyway.
struct Fr29Wide {
    uint64_t c[17];
};
LSA_HD Fr29Wide fr29_wide_zero() {
    Fr29Wide w;
#pragma unroll
    for (int k = 0; k < 17; k++) w.c[k] = 0;
    return w;
}
// w += a (x) b; at most four calls between fr29_wide_zero and fr29_wide_reduce
LSA_HD void fr29_wide_mac(Fr29Wide &w, const Fr29 &a, const Fr29 &b) {
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) w.c[i + j] += (uint64_t)a.l[i] * b.l[j];
}
// (the integer w stands for) / 2^261 mod r.  w < 121 r^2 in value.  [< 2r; tight]
LSA_HD Fr29 fr29_wide_reduce(const Fr29Wide &w) {
    uint64_t acc = 0;
    uint32_t m[9];
    Fr29 o;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        acc += w.c[k];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * Fr29::r(k - i);
        m[k] = ((uint32_t)acc * Fr29::RINV) & Fr29::MASK;
        acc += (uint64_t)m[k] * Fr29::r(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
        acc += w.c[k];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * Fr29::r(k - i);
        o.l[k - 9] = (uint32_t)acc & Fr29::MASK;
        acc >>= 29;
    }
    o.l[8] = (uint32_t)acc;
    return o;
}

}  // namespace lsa
