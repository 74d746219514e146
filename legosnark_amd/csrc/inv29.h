// legosnark_amd/csrc/inv29.h -- inversion in Fq by a constant-time binary GCD with 64-bit approximations
// (T. Pornin, "Optimized Binary GCD for Modular Inversion", 2020, Algorithm 2), host and device.
//
// Why: a^(p-2) is 254 squarings + ~127 products of ONE lane -- ~87 000 instructions, 0.17 ms of every lone final
// exponentiation (its Fq6 inversion ends in one Fq inversion) and most of every point normalisation.  Round 3 measured a
// classical binary extended Euclid on the device and rejected it: ~760 data-dependent trips of multi-word arithmetic, and
// lanes that invert different values diverge.  This variant has neither problem: 17 outer iterations, each running 31
// branch-free inner steps on 64-bit APPROXIMATIONS of (a, b) (their low 31 and top 33 bits), which yield a 2 x 2 matrix
// of factors |f|, |g| <= 2^31 that is then applied once to the full-size values -- ~1 500 instructions per outer
// iteration, the same instruction stream whatever the input.
//
// Invariants (p odd, 0 < y < p):  a = u*y, b = v*y (mod p), b odd, a, b >= 0.  One outer iteration replaces
//   (a, b) <- ((a f0 + b g0) / 2^31, (a f1 + b g1) / 2^31)      (exact divisions; a negative result is negated with its row)
//   (u, v) <- ((u f0 + v g0) / 2^31, (u f1 + v g1) / 2^31) mod p (Montgomery-style: a multiple of p clears the low 31 bits)
// and after 2*254 - 1 = 507 <= 17 * 31 inner steps a = 0, b = gcd = 1, so v = y^-1.  y = 0 returns 0.
#pragma once
#include "fp29.h"

namespace lsa {

struct Inv256 { uint32_t w[8]; };

// r = x * f for a 256-bit x and f <= 2^31, as 9 words
LSA_HD void inv_mul_small(const uint32_t x[8], uint32_t f, uint32_t r[9]) {
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)x[i] * f;
        r[i] = (uint32_t)c;
        c >>= 32;
    }
    r[8] = (uint32_t)c;
}
// r = (neg ? -r : r) over 9 words (two's complement)
LSA_HD void inv_cond_neg9(uint32_t r[9], uint32_t neg_mask) {
    uint64_t c = neg_mask & 1u;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        c += (uint64_t)(r[i] ^ neg_mask);
        r[i] = (uint32_t)c;
        c >>= 32;
    }
}
// t = x * f + y * g with signed factors given as magnitude (<= 2^31) and sign mask; 9-word two's complement
LSA_HD void inv_lin(const uint32_t x[8], uint32_t fm, uint32_t fs, const uint32_t y[8], uint32_t gm, uint32_t gs, uint32_t t[9]) {
    uint32_t p0[9], p1[9];
    inv_mul_small(x, fm, p0);
    inv_mul_small(y, gm, p1);
    inv_cond_neg9(p0, fs);
    inv_cond_neg9(p1, gs);
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        c += (uint64_t)p0[i] + p1[i];
        t[i] = (uint32_t)c;
        c >>= 32;
    }
}
// 8 words of the 9-word two's complement t shifted right by 31 (arithmetic), and the sign of t as a mask
LSA_HD uint32_t inv_shr31(const uint32_t t[9], uint32_t r[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = (t[i] >> 31) | (t[i + 1] << 1);
    return 0u - (t[8] >> 31);
}

// y^-1 mod p for canonical y (< p) given as 8 little-endian 32-bit words; 0 -> 0
LSA_HD Inv256 inv_mod_p(const Inv256 &y) {
    constexpr uint32_t NPINV32 = 0xe4866389u;          // -p^-1 mod 2^32
    uint32_t a[8], b[8], u[8], v[8], pw[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = y.w[i]; b[i] = FqParams::MOD[i]; pw[i] = FqParams::MOD[i]; u[i] = 0; v[i] = 0; }
    u[0] = 1;
#pragma unroll 1
    for (int iter = 0; iter < 17; iter++) {
        // ---- 64-bit approximations: the low 31 bits, and the 33 bits below the top bit of max(a, b)
        uint64_t ah, bh;
        {
            // top non-zero word index j of a | b (as masks: sel[i] all ones for i == j)
            uint32_t seen = 0, hi_a = 0, hi_b = 0, mid_a = 0, mid_b = 0, top = 0, small = 0;
#pragma unroll
            for (int i = 7; i >= 0; i--) {
                const uint32_t m = a[i] | b[i];
                const uint32_t nz = 0u - (uint32_t)(m != 0);
                const uint32_t sel = nz & ~seen;               // first non-zero word from the top
                seen |= nz;
                top |= m & sel;
                hi_a |= a[i] & sel; hi_b |= b[i] & sel;
                if (i >= 1) { mid_a |= a[i - 1] & sel; mid_b |= b[i - 1] & sel; }
                if (i <= 1) small |= sel;                       // both values fit 64 bits: the approximation is the value
            }
            small |= ~seen;                                     // a | b == 0
            unsigned s = 0;                                     // leading zeros of the top word (32 for a zero word)
            {
                uint32_t x = top;
                s = 32;
#pragma unroll
                for (int sh = 16; sh >= 1; sh >>= 1) { const uint32_t t2 = x >> sh; const uint32_t nzm = 0u - (uint32_t)(t2 != 0); s -= sh & nzm; x = (t2 & nzm) | (x & ~nzm); }
                s -= x & 1u;
            }
            // the 33 bits below bit n = 32 j + 32 - s of the larger value: words (j, j - 1) shifted right by 31 - s
            const unsigned r = 31 - (s & 31);                   // 0 .. 31
            const uint64_t ta = ((((uint64_t)hi_a << 32) | mid_a) >> r) & 0x1ffffffffull;
            const uint64_t tb = ((((uint64_t)hi_b << 32) | mid_b) >> r) & 0x1ffffffffull;
            const uint64_t exact_a = ((uint64_t)a[1] << 32) | a[0], exact_b = ((uint64_t)b[1] << 32) | b[0];
            const uint64_t approx_a = (ta << 31) | (a[0] & 0x7fffffffu), approx_b = (tb << 31) | (b[0] & 0x7fffffffu);
            const uint64_t sm = (uint64_t)0 - (uint64_t)(small & 1u);
            ah = (exact_a & sm) | (approx_a & ~sm);
            bh = (exact_b & sm) | (approx_b & ~sm);
        }
        // ---- 31 inner steps on the approximations.  After k steps |f|, |g| <= 2^k: the first 30 steps keep the factors in
        // 32-bit registers (half the instructions of 64-bit ones on this machine), the last one runs on 64 bits because
        // 2^31 itself does occur (g1 after 31 halvings of an even a).
        int64_t f0, g0, f1, g1;
        {
            uint32_t F0 = 1, G0 = 0, F1 = 0, G1 = 1;
#pragma unroll 1
            for (int j = 0; j < 30; j++) {
                const uint32_t odd = 0u - ((uint32_t)ah & 1u);
                const uint32_t sw = odd & (0u - (uint32_t)(ah < bh));
                const uint64_t sw64 = (uint64_t)(int64_t)(int32_t)sw, odd64 = (uint64_t)(int64_t)(int32_t)odd;
                const uint64_t t = (ah ^ bh) & sw64; ah ^= t; bh ^= t;
                uint32_t ti = (F0 ^ F1) & sw; F0 ^= ti; F1 ^= ti;
                ti = (G0 ^ G1) & sw; G0 ^= ti; G1 ^= ti;
                ah -= bh & odd64;
                F0 -= F1 & odd;
                G0 -= G1 & odd;
                ah >>= 1;
                F1 <<= 1;
                G1 <<= 1;
            }
            f0 = (int64_t)(int32_t)F0; g0 = (int64_t)(int32_t)G0; f1 = (int64_t)(int32_t)F1; g1 = (int64_t)(int32_t)G1;
        }
        {
            const uint64_t odd = (uint64_t)0 - (ah & 1u);
            const uint64_t lt = (uint64_t)0 - (uint64_t)(ah < bh);
            const uint64_t sw = odd & lt;
            uint64_t t = (ah ^ bh) & sw; ah ^= t; bh ^= t;
            int64_t ti = (f0 ^ f1) & (int64_t)sw; f0 ^= ti; f1 ^= ti;
            ti = (g0 ^ g1) & (int64_t)sw; g0 ^= ti; g1 ^= ti;
            ah -= bh & odd;
            f0 -= f1 & (int64_t)odd;
            g0 -= g1 & (int64_t)odd;
            ah >>= 1;
            f1 <<= 1;
            g1 <<= 1;
        }
        // ---- apply the matrix to (a, b) and (u, v)
        uint32_t f0s = (uint32_t)(f0 >> 63), g0s = (uint32_t)(g0 >> 63), f1s = (uint32_t)(f1 >> 63), g1s = (uint32_t)(g1 >> 63);
        const uint32_t f0m = (uint32_t)((f0 ^ (int64_t)(int32_t)f0s) - (int64_t)(int32_t)f0s), g0m = (uint32_t)((g0 ^ (int64_t)(int32_t)g0s) - (int64_t)(int32_t)g0s);
        const uint32_t f1m = (uint32_t)((f1 ^ (int64_t)(int32_t)f1s) - (int64_t)(int32_t)f1s), g1m = (uint32_t)((g1 ^ (int64_t)(int32_t)g1s) - (int64_t)(int32_t)g1s);
        uint32_t t0[9], t1[9], na[8], nb[8];
        inv_lin(a, f0m, f0s, b, g0m, g0s, t0);
        inv_lin(a, f1m, f1s, b, g1m, g1s, t1);
        const uint32_t sa = inv_shr31(t0, na), sb = inv_shr31(t1, nb);
        {   // |a|, |b|: a negative row is negated, and its factors with it
            uint64_t c = sa & 1u, d = sb & 1u;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                c += (uint64_t)(na[i] ^ sa); a[i] = (uint32_t)c; c >>= 32;
                d += (uint64_t)(nb[i] ^ sb); b[i] = (uint32_t)d; d >>= 32;
            }
            f0s ^= sa; g0s ^= sa; f1s ^= sb; g1s ^= sb;
        }
        // (u, v): t = u f + v g, made divisible by 2^31 with a multiple of p, shifted, brought back to [0, p)
        auto upd = [&](uint32_t fm, uint32_t fs, uint32_t gm, uint32_t gs, uint32_t out[8]) {
            uint32_t t[9];
            inv_lin(u, fm, fs, v, gm, gs, t);
            const uint32_t k = (t[0] * NPINV32) & 0x7fffffffu;
            uint64_t c = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                c += (uint64_t)pw[i] * k + t[i];
                t[i] = (uint32_t)c;
                c >>= 32;
            }
            t[8] += (uint32_t)c;
            uint32_t r[8];
            const uint32_t neg = inv_shr31(t, r);               // value in (-p, 2p)
            // r += p if negative
            uint64_t cc = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) { cc += (uint64_t)r[i] + (pw[i] & neg); r[i] = (uint32_t)cc; cc >>= 32; }
            // r -= p if r >= p
            uint32_t d2[8];
            uint64_t br = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) { const uint64_t x = (uint64_t)r[i] - pw[i] - br; d2[i] = (uint32_t)x; br = (x >> 32) & 1; }
            const uint32_t ge = 0u - (uint32_t)(br == 0);
#pragma unroll
            for (int i = 0; i < 8; i++) out[i] = (d2[i] & ge) | (r[i] & ~ge);
        };
        uint32_t nu[8], nv[8];
        upd(f0m, f0s, g0m, g0s, nu);
        upd(f1m, f1s, g1m, g1s, nv);
#pragma unroll
        for (int i = 0; i < 8; i++) { u[i] = nu[i]; v[i] = nv[i]; }
    }
    Inv256 r;
    // b == 1 for y != 0 (gcd); for y == 0 the answer is 0 like the power's
    uint32_t yz = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) yz |= y.w[i];
    const uint32_t keep = 0u - (uint32_t)(yz != 0);
#pragma unroll
    for (int i = 0; i < 8; i++) r.w[i] = v[i] & keep;
    return r;
}

// (x * 2^261)^-1 in the same form: x^-1 * 2^261, for a tight value < 16p.  [< 2p; tight]
LSA_HD F29 f29_inverse(const F29 &a) {
    Inv256 y;
    a.canonical().pack256(y.w);
    const Inv256 z = inv_mod_p(y);                       // x^-1 * 2^-261
    constexpr uint32_t R3[9] = {0x0e2312b2u, 0x16c05ca2u, 0x0bc84389u, 0x1cdf310bu, 0x11adafddu, 0x032e568eu, 0x1d6ae48cu, 0x10d4cd1fu, 0x0026c2d2u};   // 2^783 mod p
    return mul(F29::unpack256(z.w), F29::from_limbs(R3));     // x^-1 2^-261 * 2^783 / 2^261
}

}  // namespace lsa
