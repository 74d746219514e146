// legosnark_amd/csrc/tmiller.h -- the optimal-ate Miller loop of alt_bn128 split the way libff
// splits it: a G2-only precomputation that produces the line coefficients of every step
// (libff alt_bn128_ate_precompute_G2 -> alt_bn128_ate_G2_precomp, ~100 coefficient triples per Q)
// and a Miller loop that only runs the Fq12 chain f <- f^2 * line(P) over such a table
// (libff alt_bn128_ate_miller_loop / alt_bn128_ate_double_miller_loop).
//
// Call sites this serves in the reference: keys keep G2_precomp values and re-use them in every
// verification -- /root/reference/src/gadgets/subspace.cc:48,66-70 (C_precomp, a_precomp built at
// key generation), :152-166 (verifyLin3or4: miller_loop / double_miller_loop on them),
// /root/reference/src/gadgets/lipmaa.h:95-96 and lipmaa.cc:194-196, /root/reference/src/gadgets/poly.h:97-121.
// With the table resident, a Miller loop is 64 squarings + 102 sparse products of f (166 dependent
// rounds) instead of the 344 dependent rounds of G2 point arithmetic the fused loops of miller.h
// carry (3 per doubling step, 4 per addition step).
//
//   G12Pre     twelve lanes per G2 point, five points per wavefront: the point arithmetic of
//              G12Miller (miller.h) with P = (1, 1), so that the "evaluated" line of every step IS
//              libff's coefficient triple (ell_0, ell_VW, ell_VV); emits one table per point.
//   TabMiller  twelve lanes per ACCUMULATOR, four accumulators per wavefront, 16 helper lanes: an
//              accumulator f is shared by up to TM_MAXM pairs of one product (f <- f^2 * prod_i line_i:
//              the squaring is paid once per product chunk, and prod_i miller_loop(P_i, Q_i) is the
//              same Fq12 element however it is associated); lane (k, part) owns one Fq component of
//              coefficient k of f; the helper lanes scale the next line by (px, py) in the same
//              instruction stream; table rows stream global memory -> registers -> LDS two uses ahead.
// Same formulas, same lazy-bounds contract and same canonical values as miller.h.  Host + device;
// tests/cpp/test_tmiller.cc runs both against miller_one and the tower code.
#pragma once
#include "miller.h"

namespace lsa {

static constexpr int ATE_NUM_COEFFS = 102;                        // 64 doublings + 36 additions + 2 Frobenius steps
static constexpr int TM_LINE_WORDS = 54;                          // {ell_0, ell_VW, ell_VV} as 3 Fq2S of 18 words
static constexpr int TM_TAB_FQ2 = ATE_NUM_COEFFS * 3 + 2;         // + the affine point (QX, QY) libff keeps beside the coefficients
static constexpr int TM_TAB_WORDS = TM_TAB_FQ2 * 18;              // 22 176 B per point (internal form: 29-bit limbs, values < 2p)
static constexpr int G2_PRECOMP_FQ2 = 2 + 3 * ATE_NUM_COEFFS;     // public form: QX, QY, coefficients; 64-B libff Fq2 each
static constexpr int G2_PRECOMP_BYTES = G2_PRECOMP_FQ2 * 64;      // 19 712 B

// ------------------------------------------------------------------------------------
// G2 line tables
// ------------------------------------------------------------------------------------
template <class X>
struct G12Pre {
    X &x;
    Fq2S *mem;          // G12_LDS_FQ2 elements (the layout of G12Miller; its F / XF / T slots stay unused)
    using GM = G12Miller<X>;
    using WM = WMiller<X>;

    // one round of the point arithmetic: side products of step `op`, then its combine on lane 0 of the group;
    // after the rounds that complete a line (1: doubling, 4: addition) lanes (k < 3, part) write it out
    LSA_HD void round(int op, int x2, int y2, uint32_t *const *out, int entry) {
        Fq2S *m = mem;
        const typename GM::Step st = GM::step_of(op, x2, y2);
        x.par([=](unsigned lane) {
            const unsigned g = lane / 12, k = (lane % 12) >> 1, part = lane & 1;
            if (g >= (unsigned)G12_GROUPS) return;
            Fq2S *base = m + g * G12_STRIDE;
            if ((int)k < st.sd.n) g12_part(base[G12_G + k], part) = Fs{g12_comp_mul<20>(part, base[G12_V + st.sd.a[k]], base[G12_V + st.sd.b[k]])};
        });
        x.par([=](unsigned lane) {
            const unsigned g = lane / 12, k = (lane % 12) >> 1, part = lane & 1;
            if (g >= (unsigned)G12_GROUPS || part || k) return;
            Fq2S *base = m + g * G12_STRIDE;
            GM::combine_op(op, base + G12_V, base + G12_G, base + G12_L);
        });
        if (op == 1 || op == 4) {
            x.par([=](unsigned lane) {
                const unsigned g = lane / 12, k = (lane % 12) >> 1, part = lane & 1;
                if (g >= (unsigned)G12_GROUPS || k >= 3 || !out[g]) return;
                const Fs v = g12_part(m[g * G12_STRIDE + G12_L + k], part);
                uint32_t *d = out[g] + entry * TM_LINE_WORDS + k * 18 + part * 9;
#pragma unroll
                for (int i = 0; i < 9; i++) d[i] = v.v.l[i];
            });
        }
    }

    // tables out[g] (TM_TAB_WORDS words each, null: idle group) <- precompute_G2(Q[g])
    LSA_HD void run(const Jac<Fq2> *Q, unsigned count, uint32_t *const *out) {
        Fq2S *m = mem;
        x.par([=](unsigned lane) {
            const unsigned g = lane / 12, k = (lane % 12) >> 1, part = lane & 1;
            if (g >= (unsigned)G12_GROUPS || part || k >= 2) return;
            Fq2S *Vv = m + g * G12_STRIDE + G12_V;
            if (k == 0) {
                Vv[WM_PX] = Fq2S{Fs::one(), Fs::zero()};          // P = (1, 1): ell_VW * py = ell_VW, ell_VV * px = ell_VV
                Vv[WM_PY] = Fq2S{Fs::one(), Fs::zero()};
                Vv[WM_TWB] = fq2_constT<Fs>(LSA_TWIST_B);
            } else {
                wm_setup(1, g < count, nullptr, Q + g, Vv);
                if (out[g]) {                                      // libff keeps the affine point beside the coefficients
                    uint32_t *d = out[g] + ATE_NUM_COEFFS * TM_LINE_WORDS;
                    const Fq2S qx = Vv[WM_QX], qy = Vv[WM_QY];
#pragma unroll
                    for (int i = 0; i < 9; i++) { d[i] = qx.c0.v.l[i]; d[9 + i] = qx.c1.v.l[i]; d[18 + i] = qy.c0.v.l[i]; d[27 + i] = qy.c1.v.l[i]; }
                }
            }
        });
        int entry = 0;
#pragma unroll 1
        for (int ph = 0; ph < 66; ph++) {
            const bool dbl = ph < 64;
            const bool add = dbl ? ate_bit(63 - ph) != 0 : true;
            const int x2 = ph == 64 ? WM_Q1X : (ph == 65 ? WM_Q2X : WM_QX), y2 = x2 + 1;
            const int first = dbl ? 0 : 3, last = add ? 7 : 3;
#pragma unroll 1
            for (int op = first; op < last; op++) {
                round(op, x2, y2, out, entry);
                if (op == 1 || op == 4) entry++;
            }
        }
    }
};

// ------------------------------------------------------------------------------------
// Miller loops over line tables
// ------------------------------------------------------------------------------------
static constexpr int TM_CHUNKS = 4;        // accumulators per wavefront (48 lanes) + 16 helper lanes
static constexpr int TM_MAXM = 4;          // pairs sharing one accumulator
enum {                                     // Fq2S slots of one accumulator in LDS
    TM_F = 0, TM_XF = 6, TM_T = 12,        // f, xi*f, the round's result
    TM_RAW = 18,                           // ring of 3 table rows {ell_0, ell_VW, ell_VV}
    TM_SC = 27,                            // ring of 2 scaled rows {ell_VW * py, ell_VV * px}
    TM_PXY = 31,                           // (px, 0), (py, 0) per pair
    TM_STRIDE = TM_PXY + 2 * TM_MAXM
};
static constexpr int TM_ZERO = TM_CHUNKS * TM_STRIDE;             // one shared zero
static constexpr int TM_LDS_FQ2 = TM_ZERO + 1;

// One term per (lane, j): Fq2S operands A, B in LDS (offsets into the engine's memory) and a weight.  A lane's
// result is component `part` of sum_j weight_j * A_j * B_j; main lanes and helper lanes run the same
// instruction stream on different terms (a missing term is 0 * 0).
struct TMTerm { int a, b; uint32_t w2; };

// sum_{t < N} a[t] * b[t] / 2^261 mod p with ONE reduction.  Needs sum a_t b_t < 169 p^2, tight limbs
// everywhere and 9 N + 9 <= 64 limb products per column (N <= 6).  [< 2p; tight]
template <int N>
LSA_HD F29 dotn(const F29 (&a)[N], const F29 (&b)[N]) {
    static_assert(9 * N + 9 <= 64, "64-bit columns hold 64 products of 29-bit limbs");
    uint64_t acc = 0;
    uint32_t m[9];
    F29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++)
#pragma unroll
            for (int t = 0; t < N; t++) acc += (uint64_t)a[t].l[i] * b[t].l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * F29::p(k - i);
        m[k] = ((uint32_t)acc * F29::PINV) & F29::MASK;
        acc += (uint64_t)m[k] * F29::p(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++)
#pragma unroll
            for (int t = 0; t < N; t++) acc += (uint64_t)a[t].l[i] * b[t].l[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * F29::p(k - i);
        r.l[k - 9] = (uint32_t)acc & F29::MASK;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}
// the two limb-level operand pairs behind component `part` of a*b (b's components < KB p): part 0:
// a0*b0 + a1*(KB p - b1), part 1: a0*b1 + a1*b0
template <int KB>
LSA_HD void tm_comp_operands(unsigned part, const Fq2S &a, const Fq2S &b, F29 &x0, F29 &y0, F29 &x1, F29 &y1) {
    const uint32_t pm = w12_mask(0u - part);
    const F29 nb1 = sub_k<KB>(F29::zero(), b.c1.v);
    x0 = a.c0.v;
    x1 = a.c1.v;
#pragma unroll
    for (int l = 0; l < 9; l++) {
        y0.l[l] = (b.c1.v.l[l] & pm) | (b.c0.v.l[l] & ~pm);
        y1.l[l] = (b.c0.v.l[l] & pm) | (nb1.l[l] & ~pm);
    }
}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void tm_store_word(uint32_t *p, uint32_t v) { *w12_lds(p) = v; }
#else
inline void tm_store_word(uint32_t *p, uint32_t v) { *p = v; }
#endif

// precompute_G1 for one pair (libff to_affine_coordinates: O -> (0, 1)); neg: use -P (the Miller value of
// (-P, Q) is the conjugate -- libff unitary_inverse -- of that of (P, Q): conjugation negates the odd
// powers of w and only ell_VW * py sits on one)
LSA_HD void tm_setup_g1(bool valid, const Jac<Fq> *P, bool neg, Fq2S *pxy) {
    PB px = PB::zero(), py = PB::one();
    if (valid && !P->Z.is_zero()) {
        px = PB::from_mont256(P->X); py = PB::from_mont256(P->Y);
        if (!(P->Z == Fq::one())) {
            const PB zi = PB::from_mont256(P->Z).inverse(), zi2 = zi.sqr();
            px = px * zi2; py = py * (zi2 * zi);
        }
    }
    if (neg) py = py.neg();
    pxy[0] = Fq2S{px, PB::zero()};
    pxy[1] = Fq2S{py, PB::zero()};
}

template <class X>
struct TabMiller {
    X &x;
    Fq2S *mem;                        // TM_LDS_FQ2 elements
    const uint32_t *const *tab;       // [TM_CHUNKS * TM_MAXM] table of pair i of accumulator c (never null: a missing
                                      // pair points at the identity table, whose rows are the line (1, 0, 0))
    using WM = WMiller<X>;

    // term j of `lane` in a round of `mode` (1: F*F, 2: F * line(use u)); helper lanes: the scaling of use `scale`
    static LSA_HD TMTerm term(int mode, unsigned lane, int j, unsigned M, int u, int scale) {
        TMTerm t = {TM_ZERO, TM_ZERO, 0u};
        if (lane < 12u * TM_CHUNKS) {
            const int base = (int)(lane / 12) * TM_STRIDE, k = (int)((lane % 12) >> 1);
            if (mode == 1) {
                if (j < sqr_pair_count(k)) {
                    int ti, ui;
                    bool wrap;
                    sqr_pair(k, j, ti, ui, wrap);
                    t.a = base + (wrap ? TM_XF : TM_F) + ti;
                    t.b = base + TM_F + ui;
                    t.w2 = ti == ui ? 0u : ~0u;
                }
            } else if (mode == 2 && j < 3) {
                int ai = k - (j == 0 ? 0 : j + 2);                      // line coefficients sit at w^0, w^3, w^4
                const bool wrap = ai < 0;
                if (wrap) ai += 6;
                t.a = base + (wrap ? TM_XF : TM_F) + ai;
                t.b = j == 0 ? base + TM_RAW + 3 * (u % 3) : base + TM_SC + 2 * (u % 2) + (j - 1);
            }
        } else if (scale >= 0 && j == 0) {
            const int h = (int)lane - 12 * TM_CHUNKS, base = (h >> 2) * TM_STRIDE, which = (h >> 1) & 1;   // 0: ell_VW * py, 1: ell_VV * px
            t.a = base + TM_RAW + 3 * (scale % 3) + 1 + which;
            t.b = base + TM_PXY + 2 * (scale % (int)M) + (1 - which);
        }
        return t;
    }
    // where the lane's result goes (null: nowhere)
    static LSA_HD Fs *dest(Fq2S *m, int mode, unsigned lane, int scale) {
        const unsigned part = lane & 1;
        if (lane < 12u * TM_CHUNKS) return mode ? &g12_part(m[(lane / 12) * TM_STRIDE + TM_T + ((lane % 12) >> 1)], part) : nullptr;
        if (scale < 0) return nullptr;
        const unsigned h = lane - 12u * TM_CHUNKS;
        return &g12_part(m[(h >> 2) * TM_STRIDE + TM_SC + 2 * ((unsigned)scale % 2u) + ((h >> 1) & 1)], part);
    }

    // mode 1: T <- F*F, 2: T <- F * line(use u), 0: nothing;  then F, XF <- T.
    // scale >= 0: helper lanes scale the row of use `scale`;  load >= 0: row of use `load` -> RAW ring.
    // A use u is (entry u / M, pair u % M).
    LSA_HD void round(int mode, unsigned M, int u, int scale, int load) {
        Fq2S *m = mem;
        const uint32_t *const *tb = tab;
        x.par([=](unsigned lane) {
            // ---- table row prefetch: global -> registers now, -> LDS at the end of the phase
            uint32_t row[4] = {0, 0, 0, 0};
            if (load >= 0) {
                const unsigned pair = (unsigned)load % M, entry = (unsigned)load / M;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const unsigned wd = lane + 64u * j, c = wd / TM_LINE_WORDS, wi = wd % TM_LINE_WORDS;
                    if (c < (unsigned)TM_CHUNKS) row[j] = tb[c * TM_MAXM + pair][entry * TM_LINE_WORDS + wi];
                }
            }
            const unsigned part = lane & 1;
            Fs *dst = dest(m, mode, lane, scale);
            if (mode == 2) {
                // three products, ONE reduction: a < 20p (xi * F) in at most two terms, b < 2p:
                // 4 * 40 + 2 * 4 = 168 < 169 p^2
                F29 xa[6], yb[6];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const TMTerm t = term(2, lane, j, M, u, scale);
                    tm_comp_operands<2>(part, w12_load(m + t.a), w12_load(m + t.b), xa[2 * j], yb[2 * j], xa[2 * j + 1], yb[2 * j + 1]);
                }
                const Fs r = {dotn<6>(xa, yb)};
                if (dst) w12_store(dst, r);
            } else {
                // four (three, one) products reduced one by one, summed with their weights (6 in all) and
                // brought back under 2p by a Montgomery product with 1
                F29 sum = F29::zero();
#pragma unroll 1
                for (int j = 0; j < 4; j++) {
                    const TMTerm t = term(mode, lane, j, M, u, scale);
                    const F29 x1 = g12_comp_mul<2>(part, w12_load(m + t.a), w12_load(m + t.b));
                    const uint32_t w2 = w12_mask(t.w2);
#pragma unroll
                    for (int l = 0; l < 9; l++) sum.l[l] += x1.l[l] + (x1.l[l] & w2);
                }
                const Fs r = {mul(w12_norm_u(sum), F29::one())};
                if (dst) w12_store(dst, r);
            }
            if (load >= 0) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const unsigned wd = lane + 64u * j, c = wd / TM_LINE_WORDS, wi = wd % TM_LINE_WORDS;
                    if (c < (unsigned)TM_CHUNKS) tm_store_word(reinterpret_cast<uint32_t *>(m + c * TM_STRIDE + TM_RAW + 3 * ((unsigned)load % 3u)) + wi, row[j]);
                }
            }
        });
        if (mode) {
            x.par([=](unsigned lane) {
                if (lane >= 12u * TM_CHUNKS || (lane & 1)) return;
                const unsigned c = lane / 12, k = (lane % 12) >> 1;
                Fq2S *base = m + c * TM_STRIDE;
                const Fq2S tv = base[TM_T + k];
                base[TM_F + k] = tv;
                base[TM_XF + k] = WM::st(WM::xi_times(WM::ld(tv)));                          // [< 20]
            });
        }
    }

    // accumulator c < nacc: F <- prod_{i < cnt[c]} miller_loop(+-P[c][i], table[c][i]);  M = max cnt
    LSA_HD void run(const Jac<Fq> *const *P, const uint8_t *neg, const unsigned *cnt, unsigned M) {
        Fq2S *m = mem;
        x.par([=](unsigned lane) {
            if (lane == 63) m[TM_ZERO] = Fq2S::zero();
            if (lane < 12u * TM_CHUNKS) {
                if (lane & 1) return;
                const unsigned c = lane / 12, k = (lane % 12) >> 1;
                Fq2S *base = m + c * TM_STRIDE;
                const Fq2S f0 = k == 0 ? P2::one() : P2::zero();
                base[TM_F + k] = f0;
                base[TM_XF + k] = WM::st(WM::xi_times(WM::ld(f0)));
            } else {
                const unsigned h = lane - 12u * TM_CHUNKS, c = h >> 2, i = h & 3;              // TM_MAXM == 4 pairs per accumulator
                tm_setup_g1(i < cnt[c], P[c * TM_MAXM + i], neg[c * TM_MAXM + i] != 0, m + c * TM_STRIDE + TM_PXY + 2 * i);
            }
        });
        const int U = ATE_NUM_COEFFS * (int)M;
        int u = 0, ns = 0, nl = 0;
        // what the helper lanes and the prefetch do beside the main work of a round that is about to consume use u
        auto side = [&](int &sc, int &ld) {
            sc = (ns < U && ns < nl && ns < u + 2) ? ns : -1;       // its row was loaded in an earlier round; SC ring of 2
            ld = (nl < U && nl < u + 3) ? nl : -1;                  // RAW ring of 3
            if (sc >= 0) ns++;
            if (ld >= 0) nl++;
        };
        int sc, ld;
        side(sc, ld);
        round(0, M, 0, sc, ld);                                     // row 0
        int entry = 0;
#pragma unroll 1
        for (int ph = 0; ph < 66; ph++) {
            const bool dbl = ph < 64;
            const int lines = dbl ? 1 + ate_bit(63 - ph) : 1;
            if (dbl) {
                side(sc, ld);
                round(1, M, u, sc, ld);
            }
#pragma unroll 1
            for (int li = 0; li < lines; li++, entry++) {
#pragma unroll 1
                for (unsigned i = 0; i < M; i++) {
                    side(sc, ld);
                    round(2, M, u, sc, ld);
                    u++;
                }
            }
        }
    }
    LSA_HD Fq12S result(unsigned c) const {
        Fq12S t;
        for (int k = 0; k < 6; k++) w12_tower_ref(t, k) = mem[c * TM_STRIDE + TM_F + k];
        return t;
    }
};

}  // namespace lsa
