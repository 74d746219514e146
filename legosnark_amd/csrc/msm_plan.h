// legosnark_amd/csrc/msm_plan.h -- the digit plan of the wide-window MSM pipelines (msm.hip, msm_compact.hip):
// where the pre-shifted copies of a resident base table sit, which windows a call of a given size uses, and the
// signed-digit recoding of one scalar.  Shared so that every pipeline cuts a scalar the same way.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "bn254_constants.h"
#include "fp.h"

namespace lsa {

struct WidePlan {
    unsigned nwin;        // digits per scalar
    unsigned c;           // widest window: 2^(c-1) buckets
    unsigned copy_step;   // window k gathers from table copy k * copy_step
    unsigned start[32];   // window k covers bits [start[k], start[k] + width[k])
    unsigned width[32];
};

// Copy j of the table holds 2^(pos[j]) * P.  The positions are the starts of 13 (12 from 6*2^20
// points on) wide windows that split the 255 scalar bits as evenly as possible (8 x 20 + 5 x 19
// bits; 3 x 22 + 9 x 21) plus the midpoint of each: wide digits use every other copy, narrow
// digits (10 or 9 bits; 11 or 10) all of them.  Even widths matter: a short window concentrates
// its digits on few buckets, and the longest bucket list bounds the accumulate kernel.
struct TableGrid {
    unsigned ncopies;
    unsigned pos[33];     // pos[ncopies] = 255
};
inline TableGrid table_grid(size_t n_table) {
    TableGrid t;
    const unsigned nbig = n_table >= ((size_t)6 << 20) ? 12u : 13u;
    const unsigned base = 255 / nbig, rem = 255 % nbig;
    unsigned bit = 0;
    for (unsigned k = 0; k < nbig; k++) {
        const unsigned w = base + (k < rem ? 1u : 0u);
        t.pos[2 * k] = bit;
        t.pos[2 * k + 1] = bit + (w + 1) / 2;
        bit += w;
    }
    t.ncopies = 2 * nbig;
    t.pos[t.ncopies] = 255;
    return t;
}

// big: 13 (12) wide digits over every other copy; otherwise all 26 (24) positions as narrow digits
inline WidePlan wide_plan_for(size_t n_table, bool big) {
    const TableGrid t = table_grid(n_table);
    WidePlan pl = {};
    pl.copy_step = big ? 2 : 1;
    pl.nwin = t.ncopies / pl.copy_step;
    pl.c = 0;
    for (unsigned k = 0; k < pl.nwin; k++) {
        pl.start[k] = t.pos[k * pl.copy_step];
        pl.width[k] = t.pos[(k + 1) * pl.copy_step] - pl.start[k];
        if (pl.width[k] > pl.c) pl.c = pl.width[k];
    }
    return pl;
}

#if defined(__HIPCC__)
// Signed digits of one scalar, produced one window at a time (the carry chain is sequential) and
// handed to `use(k, sd)` with sd = +-(segment * B + |digit|), 0 for a zero digit -- no digit
// array: the ranking pass and the scatter pass both recompute them from the scalar (a Montgomery
// reduction and a few shifts per scalar) instead of writing 4 B per digit to HBM and reading
// them back twice.
template <class Use>
__device__ __forceinline__ void wide_digits(const Fr &scalar, const WidePlan &pl, uint32_t seg_base, Use use) {
    uint32_t s[8];
    scalar.to_canonical(s);
    // balanced representative: s > (r-1)/2 is recoded as -(r - s), i.e. the digits of r - s with
    // every sign flipped.  Same sum; "small negative" scalars (r - 1, r - 2, ...: ten of their
    // thirteen digits would be the digits of r, the same ten buckets for every such scalar) become
    // small digits, and the top window never exceeds a quarter of its range.
    bool flip;
    {
        uint32_t t[8];
        uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {                  // t = r - s
            const uint64_t x = (uint64_t)LSA_R[i] - s[i] - br;
            t[i] = (uint32_t)x;
            br = (x >> 32) & 1;
        }
        br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) br = (((uint64_t)t[i] - s[i] - br) >> 32) & 1;      // borrow out <=> t < s <=> 2 s > r
        flip = br != 0;
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = flip ? t[i] : s[i];
    }
    // consume the limbs through a 64-bit bit buffer (static limb index: no register-array indexing)
    uint64_t buf = 0;
    unsigned have = 0, k = 0;
    uint32_t carry = 0;
#pragma unroll
    for (int limb = 0; limb < 8; limb++) {
        buf |= (uint64_t)s[limb] << have;
        have += 32;
        while (k < pl.nwin && (have >= pl.width[k] || limb == 7)) {
            const unsigned width = pl.width[k];
            uint32_t d = (uint32_t)buf & ((1u << width) - 1);
            buf >>= width;
            have = have >= width ? have - width : 0;
            d += carry;
            int32_t sd;
            // the top window is never recoded: scalars are < 2^254 and the windows cover 255 bits, so
            // its raw value is < 2^(width-1) and the carry keeps it <= 2^(width-1) <= B
            if (k + 1 < pl.nwin && d >= (1u << (width - 1))) { sd = (int32_t)d - (int32_t)(1u << width); carry = 1; }
            else { sd = (int32_t)d; carry = 0; }
            if (sd != 0) { const int32_t m = (int32_t)seg_base + (sd < 0 ? -sd : sd); sd = (sd < 0) != flip ? -m : m; }
            use(k, sd);
            k++;
        }
    }
}

#endif

}  // namespace lsa
