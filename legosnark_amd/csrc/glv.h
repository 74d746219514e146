// legosnark_amd/csrc/glv.h -- GLV decomposition for alt_bn128 G1.
//
// G1 has the efficiently computable endomorphism phi(x, y) = (beta*x, y) = lambda*(x, y)
// with beta^3 = 1 in Fq and lambda^2 + lambda + 1 = 0 in Fr.  Every scalar splits as
// k = k1 + k2*lambda (mod r) with |k1|, |k2| < 2^127, so an n-term MSM over 254-bit scalars
// becomes a 2n-term MSM over 127-bit scalars: half as many windows, hence half the bucket
// reduction and half the doubling chain of the final fold; phi costs one field product on
// the gathered x.  The group element computed is unchanged (G1 has prime order r).
//
// Constants (derived and checked with the independent big-int model that also produces the
// golden vectors: phi(Q) == lambda*Q, lattice basis by
// the extended Euclidean algorithm on (r, lambda)):
//   v1 = (a1, -|b1|), v2 = (a2, b2), det = +r;
//   c1 = floor(k*g1 / 2^256), c2 = floor(k*g2 / 2^256), g1 = floor(b2*2^256/r), g2 = floor(|b1|*2^256/r);
//   k1 = k - c1*a1 - c2*a2,   k2 = c1*|b1| - c2*b2.
// Truncating (instead of rounding) c1, c2 keeps |k1| <= |a1|+|a2| < 2^127 and
// |k2| <= |b1|+|b2| < 2^127.
#pragma once
#include "fp.h"

namespace lsa {

// low NO limbs of a (NA limbs) * b (NB limbs)
template <int NA, int NB, int NO>
LSA_HD void mul_limbs_lo(const uint32_t *a, const uint32_t *b, uint32_t *out) {
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < NO; k++) {
        uint64_t lo = carry & 0xffffffffu, hi = carry >> 32;   // column sum split to avoid overflow
#pragma unroll
        for (int i = 0; i < NA; i++) {
            int j = k - i;
            if (j >= 0 && j < NB) {
                uint64_t p = (uint64_t)a[i] * b[j];
                lo += p & 0xffffffffu;
                hi += p >> 32;
            }
        }
        out[k] = (uint32_t)lo;
        carry = hi + (lo >> 32);
    }
}

struct GlvSplit {
    uint32_t k1[4], k2[4];   // magnitudes (< 2^127)
    bool neg1, neg2;
};

// lambda, canonical (non-Montgomery) limbs
#define LSA_GLV_LAMBDA {0xb99c90ddu, 0x8b17ea66u, 0x8d8daaa7u, 0x5bfc4108u, 0x41a91758u, 0xb3c4d79du, 0u, 0u}

LSA_HD GlvSplit glv_decompose(const uint32_t k[8]) {
    constexpr uint32_t G1c[3] = {0xc7e0b3d7u, 0xd91d232eu, 0x00000002u};
    constexpr uint32_t G2c[5] = {0x391eb18du, 0x7a7bd9d4u, 0xa773d2cfu, 0x4ccef014u, 0x00000002u};
    constexpr uint32_t A1[2] = {0x94d213e3u, 0x89d32568u};
    constexpr uint32_t A2[4] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u};
    constexpr uint32_t B1[4] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u};   // |b1|
    constexpr uint32_t B2[2] = {0x94d213e3u, 0x89d32568u};
    uint32_t g1[3], g2[5], a1[2], a2[4], b1[4], b2[2];
#pragma unroll
    for (int i = 0; i < 3; i++) g1[i] = G1c[i];
#pragma unroll
    for (int i = 0; i < 5; i++) g2[i] = G2c[i];
#pragma unroll
    for (int i = 0; i < 2; i++) { a1[i] = A1[i]; b2[i] = B2[i]; }
#pragma unroll
    for (int i = 0; i < 4; i++) { a2[i] = A2[i]; b1[i] = B1[i]; }
    uint32_t t1[11], t2[13];
    mul_limbs_lo<8, 3, 11>(k, g1, t1);
    mul_limbs_lo<8, 5, 13>(k, g2, t2);
    const uint32_t *c1 = t1 + 8;   // 2 limbs (t1[10] == 0)
    const uint32_t *c2 = t2 + 8;   // 4 limbs (t2[12] == 0)
    // 160-bit two's-complement arithmetic
    uint32_t p1[5], p2[5], q1[5], q2[5];
    mul_limbs_lo<2, 2, 5>(c1, a1, p1);
    mul_limbs_lo<4, 4, 5>(c2, a2, p2);
    mul_limbs_lo<2, 4, 5>(c1, b1, q1);
    mul_limbs_lo<4, 2, 5>(c2, b2, q2);
    uint32_t r1[5], r2[5];
    int64_t br = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        int64_t v = (int64_t)k[i] - (int64_t)p1[i] - (int64_t)p2[i] + br;
        r1[i] = (uint32_t)v;
        br = v >> 32;
    }
    br = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        int64_t v = (int64_t)q1[i] - (int64_t)q2[i] + br;
        r2[i] = (uint32_t)v;
        br = v >> 32;
    }
    GlvSplit s;
    s.neg1 = (r1[4] >> 31) != 0;
    s.neg2 = (r2[4] >> 31) != 0;
    uint64_t c = 1;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t w = s.neg1 ? ~r1[i] : r1[i];
        if (s.neg1) { c += w; w = (uint32_t)c; c >>= 32; }
        s.k1[i] = w;
    }
    c = 1;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t w = s.neg2 ? ~r2[i] : r2[i];
        if (s.neg2) { c += w; w = (uint32_t)c; c >>= 32; }
        s.k2[i] = w;
    }
    return s;
}

}  // namespace lsa
