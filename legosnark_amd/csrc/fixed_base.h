// legosnark_amd/csrc/fixed_base.h -- HOST-side fixed-base scalar multiplication (no device code).
//
// The reference multiplies a GENERATOR by a single scalar at most of its host scalar-multiplication sites:
// `LFr * LG1::one()` / `LFr * LG2::one()` in Comm's operators and CommScheme::commit's blinding term
// (/root/reference/src/prototools/commit.h:43-44,162-163), polytools.h:126-133, and CPPoly::verify's
// `pts[i] * LG2::one()` per variable (src/gadgets/poly.h:117).  libff computes each with double-and-add (254 doublings
// + ~127 additions); for a base known in advance the doublings can be tabulated once:
//
//   k = sum_w d_w 2^(8w),  d_w in [-128, 128]  (32 signed 8-bit digits, carry-propagating recoding; k < 2^254)
//   k G = sum_w sign(d_w) T[w][|d_w| - 1],     T[w][j] = (j + 1) 2^(8w) G  stored affine
//
// i.e. at most 32 mixed additions (XYZZ accumulator, ec.h) and no doubling: the same group element as double-and-add.
// The table is 32 x 128 affine points (256 KiB for G1, 512 KiB for G2), built with mixed additions along each row
// (the row's base normalised first) and ONE field inversion for all 4096 entries (Montgomery's trick over ZZZ).
#pragma once
#include <cstdint>
#include <vector>

#include "ec.h"

namespace lsa {

template <class F>
struct FixedBaseTable {
    static constexpr int WBITS = 8, NWIN = 32, HALF = 128;
    std::vector<Aff<F>> t;                                   // [NWIN][HALF]

    void build(const Jac<F> &base) {
        std::vector<XYZZ<F>> acc((size_t)NWIN * HALF);
        Jac<F> b = jac_normalize(base);
        for (int w = 0; w < NWIN; w++) {
            const Aff<F> ba = b.is_inf() ? Aff<F>::inf() : Aff<F>{b.X, b.Y};
            XYZZ<F> cur = XYZZ<F>::from_affine(ba);
            acc[(size_t)w * HALF] = cur;
            for (int j = 1; j < HALF; j++) {
                cur = xyzz_madd(cur, ba);                    // (j + 1) * 2^(8w) * G   (j = 1: the complete addition doubles)
                acc[(size_t)w * HALF + j] = cur;
            }
            if (w + 1 < NWIN) b = jac_normalize(xyzz_to_jac(xyzz_dbl(cur)));   // 256 * 2^(8w) * G
        }
        // x = X / ZZ = X * ZZ^2 / ZZZ^2, y = Y / ZZZ; all 1 / ZZZ from one inversion
        const size_t n = acc.size();
        std::vector<F> pre(n);
        F run = F::one();
        for (size_t i = 0; i < n; i++) { pre[i] = run; if (!acc[i].is_inf()) run = run * acc[i].ZZZ; }
        F inv = run.inverse();
        t.assign(n, Aff<F>::inf());
        for (size_t i = n; i-- > 0;) {
            if (acc[i].is_inf()) continue;
            const F izzz = inv * pre[i];
            inv = inv * acc[i].ZZZ;
            const F s = acc[i].ZZ * izzz;                    // ZZ / ZZZ = 1 / Z
            t[i] = Aff<F>{acc[i].X * s.sqr(), acc[i].Y * izzz};
        }
    }

    // k: canonical (non-Montgomery) little-endian limbs of a scalar below 2^254
    Jac<F> mul(const uint64_t k[4]) const {
        XYZZ<F> acc = XYZZ<F>::inf();
        unsigned carry = 0;
        for (int w = 0; w < NWIN; w++) {
            const unsigned raw = (unsigned)((k[w >> 3] >> (8 * (w & 7))) & 0xffu) + carry;   // 0 .. 256
            int d;
            if (raw > (unsigned)HALF) { d = (int)raw - 256; carry = 1; } else { d = (int)raw; carry = 0; }
            if (d > 0) acc = xyzz_madd(acc, t[(size_t)w * HALF + (size_t)(d - 1)]);
            else if (d < 0) acc = xyzz_madd(acc, t[(size_t)w * HALF + (size_t)(-d - 1)].neg());
        }
        // (k < 2^254: the top digit is at most 63 + 1, so no carry leaves the last window)
        return xyzz_to_jac(acc);
    }
};

}  // namespace lsa
