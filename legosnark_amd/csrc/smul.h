// legosnark_amd/csrc/smul.h -- one variable-base scalar multiplication k*P on G1 with the GLV
// split and a signed 4-bit window: the per-lane body of k_smul_g1 (scalar_mul.hip), kept in a
// header so that the host tests (tests/cpp/test_smul.cc) run the very same code.
#pragma once
#include "fp29.h"
#include "glv.h"

namespace lsa {

static constexpr int SMUL_TBL = 8;           // table entries 1P .. 8P

// Signed 4-bit digits of k < 2^127 without a carry chain: with k' = k + 0x0888..8 (an 8 in every
// nibble but the top one), digit_j = nibble_j(k') - 8 in [-8, 7] for j < 31, digit_31 =
// nibble_31(k') in [0, 8], and sum digit_j 16^j = k.
LSA_HD void recode_nibbles(const uint32_t k[4], uint32_t out[4]) {
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        c += (uint64_t)k[i] + (i == 3 ? 0x08888888u : 0x88888888u);
        out[i] = (uint32_t)c;
        c >>= 32;
    }
}

// k*P for the canonical scalar s (< r).  T: SMUL_TBL entries of scratch owned by the caller.
LSA_HD XYZZ29 smul_glv(const Aff29 &P, const uint32_t s[8], XYZZ29 *T) {
    constexpr uint32_t BETA29[9] = {0x0a337995u, 0x158d1d23u, 0x189c9b98u, 0x12fa4e45u, 0x185faadcu,
                                    0x0176f16du, 0x0eed93bau, 0x14291140u, 0x000c0afeu};
    uint32_t any = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) any |= s[w];
    if (P.is_inf() || any == 0) return XYZZ29::inf();
    const GlvSplit g = glv_decompose(s);
    uint32_t dg[2][4];
    recode_nibbles(g.k1, dg[0]);
    recode_nibbles(g.k2, dg[1]);
    // table d*P, d = 1..8
    XYZZ29 t = {P.x, P.y, F29::one(), F29::one()};
    T[0] = t;
    t = xyzz29_dbl_affine(P);
    T[1] = t;
#pragma unroll 1
    for (int d = 2; d < SMUL_TBL; d++) {
        t = xyzz29_madd(t, P);
        T[d] = t;
    }
    XYZZ29 acc = XYZZ29::inf();
#pragma unroll 1
    for (int j = 31; j >= 0; --j) {
        if (j != 31) {
#pragma unroll 1
            for (int r = 0; r < 4; r++) acc = xyzz29_dbl(acc);
        }
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            const int d = (int)((dg[h][j >> 3] >> ((j & 7) * 4)) & 15u) - (j == 31 ? 0 : 8);
            if (d == 0) continue;
            const bool neg = (d < 0) != (h ? g.neg2 : g.neg1);
            XYZZ29 q = T[(d < 0 ? -d : d) - 1];
            if (h) q.X = mul(q.X, F29::from_limbs(BETA29));        // phi(x, y) = (beta x, y)
            if (neg) q.Y = sub_k<4>(F29::zero(), q.Y);
            acc = xyzz29_add(acc, q);
        }
    }
    return acc;
}

}  // namespace lsa
