// legosnark_amd/csrc/ntt_core.h -- the radix-2 NTT over Fr as at most THREE passes over the data (ntt.hip), written as
// per-element functions shared by the kernel (one workgroup per tile, strided loops, a barrier between stages) and the
// host test that runs the same functions tile by tile (tests/cpp/test_ntt_core.cc).
//
// Replaces libfqfft's basic_radix2_domain<Fr>::{FFT, iFFT, cosetFFT, icosetFFT} (_basic_radix2_FFT /
// _multiply_by_coset) as called by /root/reference/src/gadgets/lipmaa.cc:68-81,102-175: X[k] = sum_i a[i] omega^(ik),
// natural order in and out.  Fr values are canonical Montgomery residues, so any correct schedule gives libfqfft's bytes.
//
// Decomposition (Cooley-Tukey, applied twice): n = n1 n2 n3 = 2^(l1 + l2 + l3), m = n2 n3.
//   pass 1  for every column i' < m:        n1-point NTT over a[i1 m + i'] (stride m), root omega^m,
//           then the twiddle omega^(i' k1);                       in place: Y[k1 m + i']
//   pass 2  for every (k1, i3 < n3):        n2-point NTT over Y[k1 m + i2 n3 + i3] (stride n3), root omega^(n1 n3),
//           then the twiddle omega^(n1 i3 k2);                    in place: Z[k1 m + k2 n3 + i3]
//   pass 3  for every (k1, k2):             n3-point NTT over the contiguous Z[k1 m + k2 n3 + i3], root omega^(n1 n2);
//           X[k1 + n1 k2 + n1 n2 k3], times the final scale (1/n, coset powers)      out of place (a transposition)
// Transforms of up to 2^10 points are pass 3 alone, up to 2^16 passes 1 and 3.  A tile is 2^10 elements: C = 2^10 / 2^l
// neighbouring columns (rows in pass 3), so every global access is a run of C x 32 bytes.  Inside a tile each column is
// a decimation-in-time NTT: rows are loaded in bit-reversed order, stage s combines rows 2^s apart.
//
// Arithmetic: fr29.h (29-bit limbs, R = 2^261).  Constants are stored as canonical 256-bit words of their 2^261 form:
//   W[j]    = rho^j, rho = omega^(n / 2^lmax), j < 2^(lmax - 1)          the butterflies' twiddles of every pass
//   Tlo[e], Thi[e]:  omega^e = Thi[e >> h] * Tlo[e & (2^h - 1)]           pass 1's twiddles (two-level: n of them)
//   T2[k2 n3 + i3] = omega^(n1 i3 k2)                                     pass 2's twiddles (m = n2 n3 of them: one table)
//   T1[k1 m + i']  = omega^(i' k1)                                        pass 1's twiddles as ONE table of n entries where that is
//                                                                         affordable (a product per element less than the two-level form)
//   Glo, Ghi: c * g^i the same way                                        coset powers (forward: on load; inverse: with 1/n on store)
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "fr29.h"

namespace lsa {

static constexpr unsigned NTT_TILE_LOG = 10;

struct NttPlan {
    unsigned L, l1, l2, l3;       // n = 2^L = 2^(l1 + l2 + l3); l1 = 0: no pass 1, l2 = 0: no pass 2
    unsigned lmax;                // W has 2^(lmax - 1) entries (lmax >= 1)
    unsigned h;                   // Tlo has 2^h entries, Thi 2^(L - h)
    unsigned tile_log;
};
// tile_log = NTT_TILE_LOG in the library; the host test also runs smaller tiles so that small transforms take all passes
LSA_HD NttPlan ntt_plan(unsigned L, unsigned tile_log = NTT_TILE_LOG) {
    NttPlan p = {};
    p.L = L;
    p.tile_log = tile_log;
    if (L <= tile_log) { p.l3 = L; }
    else if (L <= 2 * tile_log - 4 || L <= tile_log + 1) { p.l3 = L / 2; p.l1 = L - p.l3; }          // two passes, each at most tile_log
    else { p.l1 = (L + 2) / 3; p.l2 = (L - p.l1 + 1) / 2; p.l3 = L - p.l1 - p.l2; }
    // a pass of length 2^l takes C = 2^(tile_log - l) columns; the passes' column counts must offer that many:
    // pass 2 needs l2 + l3 >= tile_log, pass 3 needs l1 + l3 >= tile_log (or fewer columns: see ntt_pass_cols)
    p.lmax = p.l1 > p.l2 ? p.l1 : p.l2;
    if (p.l3 > p.lmax) p.lmax = p.l3;
    if (p.lmax == 0) p.lmax = 1;
    p.h = (L + 1) / 2;
    return p;
}

struct NttPass {
    unsigned kind;                // 1, 2, 3
    unsigned l, logC;             // sub-transform length 2^l, 2^logC columns per tile
    unsigned tiles;               // workgroups
};
LSA_HD NttPass ntt_pass(const NttPlan &p, unsigned kind) {
    NttPass q = {};
    q.kind = kind;
    q.l = kind == 1 ? p.l1 : (kind == 2 ? p.l2 : p.l3);
    unsigned avail = kind == 1 ? p.L - p.l1 : (kind == 2 ? p.l3 : p.l1);       // log2 of the columns that lie next to each other
    unsigned want = p.tile_log > q.l ? p.tile_log - q.l : 0;
    q.logC = want < avail ? want : avail;
    q.tiles = 1u << (p.L - q.l - q.logC);
    return q;
}
LSA_HD unsigned ntt_tile_words(const NttPass &q) {           // LDS words of a tile's data (pass 3 pads its columns by one element)
    return ((1u << (q.l + q.logC)) + (q.kind == 3 ? (1u << q.logC) : 0u)) * 9u;
}

struct NttArgs {
    const Fr *src;
    Fr *dst;
    NttPlan plan;
    NttPass pass;
    const uint32_t *W;            // the butterflies' twiddles as limbs (9 words per entry: no unpacking per butterfly)
    const Fr *Tlo, *Thi;          // 2^261-form words
    const Fr *T2;                 // pass 2's twiddles, or null: two-level look-up (tests)
    const Fr *T1;                 // pass 1's twiddles omega^(col k) at [k 2^(L - l1) + col] (n of them), or null: two-level look-up
    const Fr *Glo, *Ghi;          // coset / scale tables or null
    const Fr *Gfull;              // the same powers c g^i as ONE table of n entries, or null: the two-level look-up
    unsigned gh;                  // Glo has 2^gh entries
    int pre_scale;                // first pass: a[i] *= G(i) on load
    int post_scale;               // last pass: X[k] *= G(k) on store (tables), else X[k] *= cst
    Fr cst;                       // 2^261-form words of 1 (forward) or 1/n (inverse), for transforms without a coset
};

LSA_HD unsigned ntt_brev(unsigned x, unsigned bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? __builtin_bitreverse32(x) >> (32u - bits) : 0u;      // v_bfrev_b32 (the loop below: ~5 instructions per bit and element)
#else
    unsigned r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
#endif
}
LSA_HD Fr29 ntt_lds_get(const uint32_t *lds, unsigned idx) {
    Fr29 v;
#pragma unroll
    for (int i = 0; i < 9; i++) v.l[i] = lds[idx * 9u + i];
    return v;
}
LSA_HD void ntt_lds_put(uint32_t *lds, unsigned idx, const Fr29 &v) {
#pragma unroll
    for (int i = 0; i < 9; i++) lds[idx * 9u + i] = v.l[i];
}
LSA_HD Fr29 ntt_two_level(const Fr *lo, const Fr *hi, unsigned hbits, uint64_t e) {
    return mul(Fr29::from_words(lo[e & ((1ull << hbits) - 1)]), Fr29::from_words(hi[e >> hbits]));
}

// element x of tile w: where it lives in memory (input side) and in the tile
struct NttSlot {
    uint64_t gaddr;               // index into src
    unsigned idx;                 // LDS element index
};
LSA_HD NttSlot ntt_load_slot(const NttArgs &a, unsigned w, unsigned x) {
    const NttPlan &p = a.plan;
    const unsigned l = a.pass.l, logC = a.pass.logC, C = 1u << logC, len = 1u << l;
    NttSlot s;
    if (a.pass.kind == 1) {
        const unsigned c = x & (C - 1), jr = x >> logC, j = ntt_brev(jr, l);
        const uint64_t col = (uint64_t)w * C + c;                            // i'
        s.gaddr = ((uint64_t)j << (p.L - p.l1)) + col;
        s.idx = x;
    } else if (a.pass.kind == 2) {
        const unsigned c = x & (C - 1), jr = x >> logC, j = ntt_brev(jr, l);
        const uint64_t q = (uint64_t)w * C + c;                              // k1 * n3 + i3
        const uint64_t k1 = q >> p.l3, i3 = q & ((1ull << p.l3) - 1);
        s.gaddr = (k1 << (p.l2 + p.l3)) + ((uint64_t)j << p.l3) + i3;
        s.idx = x;
    } else {
        const unsigned j = x & (len - 1), c = x >> l;
        const unsigned per_k2 = 1u << (p.l1 - logC);                         // tiles per value of k2
        const uint64_t k2 = w / per_k2, k1 = (uint64_t)(w % per_k2) * C + c;
        s.gaddr = (k1 << (p.l2 + p.l3)) + (k2 << p.l3) + j;
        s.idx = c * (len + 1) + ntt_brev(j, l);
    }
    return s;
}
LSA_HD void ntt_tile_load(const NttArgs &a, unsigned w, unsigned x, uint32_t *lds) {
    const NttSlot s = ntt_load_slot(a, w, x);
    Fr29 v = Fr29::from_words(a.src[s.gaddr]);
    if (a.pre_scale) v = mul(v, a.Gfull ? Fr29::from_words(a.Gfull[s.gaddr]) : ntt_two_level(a.Glo, a.Ghi, a.gh, s.gaddr));      // [< 2r]
    ntt_lds_put(lds, s.idx, v);
}
// butterfly b (< tile / 2) of stage s; the W table is read from memory (4 KB for 8-stage passes: it lives in the L1 / L2
// caches; keeping it out of LDS lets four workgroups share a CU instead of three)
LSA_HD void ntt_tile_stage(const NttArgs &a, unsigned s, unsigned b, uint32_t *lds) {
    const unsigned l = a.pass.l, logC = a.pass.logC, len = 1u << l;
    unsigned c, jb;
    if (a.pass.kind != 3) { c = b & ((1u << logC) - 1); jb = b >> logC; }
    else { jb = b & ((len >> 1) - 1); c = b >> (l - 1); }
    const unsigned pos = jb & ((1u << s) - 1), j0 = ((jb >> s) << (s + 1)) | pos, j1 = j0 + (1u << s);
    const unsigned i0 = a.pass.kind != 3 ? (j0 << logC) + c : c * (len + 1) + j0;
    const unsigned i1 = a.pass.kind != 3 ? (j1 << logC) + c : c * (len + 1) + j1;
    const Fr29 u = ntt_lds_get(lds, i0);
    Fr29 t = ntt_lds_get(lds, i1);
    if (s != 0) t = mul(t, ntt_lds_get(a.W, pos << (a.plan.lmax - 1 - s)));        // stage 0: the twiddle is 1 and t < 2r already
    ntt_lds_put(lds, i0, add(u, t));
    ntt_lds_put(lds, i1, sub2r(u, t));
}
// stages s and s + 1 together on the four elements j0, j0 + 2^s, j0 + 2^(s+1), j0 + 2^(s+1) + 2^s of group gidx (< tile / 4):
// the same four products and the same values as two calls of ntt_tile_stage per pair, with one LDS round trip and one
// barrier instead of two (a radix-4 step written as two radix-2 steps in registers)
LSA_HD void ntt_tile_stage2(const NttArgs &a, unsigned s, unsigned gidx, uint32_t *lds) {
    const unsigned l = a.pass.l, logC = a.pass.logC, len = 1u << l;
    unsigned c, jg;
    if (a.pass.kind != 3) { c = gidx & ((1u << logC) - 1); jg = gidx >> logC; }
    else { jg = gidx & ((len >> 2) - 1); c = gidx >> (l - 2); }
    const unsigned h = 1u << s, pos = jg & (h - 1), j0 = ((jg >> s) << (s + 2)) | pos;
    const unsigned base = a.pass.kind != 3 ? c : c * (len + 1), sh = a.pass.kind != 3 ? logC : 0u;
    const unsigned i0 = base + (j0 << sh), i1 = base + ((j0 + h) << sh), i2 = base + ((j0 + 2 * h) << sh), i3 = base + ((j0 + 3 * h) << sh);
    Fr29 x0 = ntt_lds_get(lds, i0), x1 = ntt_lds_get(lds, i1), x2 = ntt_lds_get(lds, i2), x3 = ntt_lds_get(lds, i3);
    if (s != 0) {                                                                  // stage s: both pairs share the twiddle of position pos
        const Fr29 w = ntt_lds_get(a.W, pos << (a.plan.lmax - 1 - s));
        x1 = mul(x1, w);
        x3 = mul(x3, w);
    }
    // stage s + 1: (y0, y2) at position pos, (y1, y3) at position pos + 2^s of a 2^(s+2)-point butterfly
    const unsigned shw = a.plan.lmax - 2 - s;
    if (s != 0) {
        // the first stage's sums WITHOUT carries (fr29.h: add_loose / sub3r_loose, 9 and 18 instructions instead of 27 and 36):
        // x1, x3 are fresh products (tight, < 2r), x0, x2 tight values out of LDS; the sums' limbs stay below 3 * 2^29, which
        // the products by the (tight) twiddles and the carry-normalising second stage accept.  Values: an output is below
        // (its x0 or x2) + 6r -- 8r after the pass's first pair, + 6r per later pair, < 32r after five, far below the 121 r a
        // product allows.
        const Fr29 y0 = add_loose(x0, x1), y1 = sub3r_loose(x0, x1), y2 = add_loose(x2, x3), y3 = sub3r_loose(x2, x3);
        const Fr29 t3 = mul(y3, ntt_lds_get(a.W, (pos + h) << shw));
        const Fr29 t2 = mul(y2, ntt_lds_get(a.W, pos << shw));
        ntt_lds_put(lds, i0, add(y0, t2));
        ntt_lds_put(lds, i2, sub2r(y0, t2));
        ntt_lds_put(lds, i1, add(y1, t3));
        ntt_lds_put(lds, i3, sub3r_norm(y1, t3));
        return;
    }
    // s == 0, the first pair of stages of a pass: no twiddle in stage 0, pos = 0 in every group, the twiddle of (y0, y2) is 1 -- one
    // product instead of four.  y2 < 4r then instead of < 2r: values grow to < 8r here.
    const Fr29 y0 = add(x0, x1), y1 = sub2r(x0, x1), y2 = add(x2, x3), y3 = sub2r(x2, x3);
    const Fr29 t3 = mul(y3, ntt_lds_get(a.W, (pos + h) << shw));
    ntt_lds_put(lds, i0, add(y0, y2));
    ntt_lds_put(lds, i2, sub4r(y0, y2));
    ntt_lds_put(lds, i1, add(y1, t3));
    ntt_lds_put(lds, i3, sub2r(y1, t3));
}
LSA_HD void ntt_tile_store(const NttArgs &a, unsigned w, unsigned x, const uint32_t *lds) {
    const NttPlan &p = a.plan;
    const unsigned l = a.pass.l, logC = a.pass.logC, C = 1u << logC, len = 1u << l;
    const unsigned c = x & (C - 1), k = x >> logC;
    if (a.pass.kind == 1) {
        const uint64_t col = (uint64_t)w * C + c;
        const uint64_t at = ((uint64_t)k << (p.L - p.l1)) + col;
        const Fr29 tw = a.T1 ? Fr29::from_words(a.T1[at]) : ntt_two_level(a.Tlo, a.Thi, p.h, col * k);
        a.dst[at] = mul(ntt_lds_get(lds, x), tw).to_words();                         // < 2r < 2^256
    } else if (a.pass.kind == 2) {
        const uint64_t q = (uint64_t)w * C + c;
        const uint64_t k1 = q >> p.l3, i3 = q & ((1ull << p.l3) - 1);
        const Fr29 tw = a.T2 ? Fr29::from_words(a.T2[((uint64_t)k << p.l3) + i3]) : ntt_two_level(a.Tlo, a.Thi, p.h, (i3 * k) << p.l1);
        const Fr29 v = mul(ntt_lds_get(lds, x), tw);
        a.dst[(k1 << (p.l2 + p.l3)) + ((uint64_t)k << p.l3) + i3] = v.to_words();
    } else {
        const unsigned per_k2 = 1u << (p.l1 - logC);
        const uint64_t k2 = w / per_k2, k1 = (uint64_t)(w % per_k2) * C + c;
        const uint64_t kout = k1 + (k2 << p.l1) + ((uint64_t)k << (p.l1 + p.l2));
        const Fr29 g = a.post_scale ? (a.Gfull ? Fr29::from_words(a.Gfull[kout]) : ntt_two_level(a.Glo, a.Ghi, a.gh, kout)) : Fr29::from_words(a.cst);
        a.dst[kout] = mul(ntt_lds_get(lds, c * (len + 1) + k), g).canonical2().to_words();
    }
}

}  // namespace lsa
