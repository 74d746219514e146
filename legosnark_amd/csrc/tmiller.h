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
//   G2Pre      twelve lanes per G2 point, five points per wavefront: lane (k, part) computes one Fq component of
//              the k-th Fq2 product of a round; the additions between rounds run component-wise on the two lanes of
//              pair 0.  P = (1, 1), so that each step's line IS libff's coefficient triple (ell_0, ell_VW, ell_VV).
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
static constexpr int TM_LINE_WORDS = 54;                          // a row {ell_0, ell_VW, ell_VV} in LDS: 3 Fq2S of 18 words
static constexpr int TM_ROW_WORDS = 48;                           // the same row in a table: 6 x 256-bit packed components
static constexpr int TM_TAB_WORDS = ATE_NUM_COEFFS * TM_ROW_WORDS + 32;   // 19 712 B per point: x * 2^261 mod p, ell_0 < 2p, ell_VW / ell_VV
                                                                  // < 4p (< 2^256), then QX, QY
static constexpr int G2_PRECOMP_FQ2 = 2 + 3 * ATE_NUM_COEFFS;     // public form: QX, QY, coefficients; 64-B libff Fq2 each
static constexpr int G2_PRECOMP_BYTES = G2_PRECOMP_FQ2 * 64;      // 19 712 B
static_assert(ATE_NUM_COEFFS * TM_ROW_WORDS + 32 == TM_TAB_WORDS, "table layout");

// limb `wi % 9` of component `wi / 9` (LDS word wi of a row) out of the packed row
LSA_HD uint32_t tm_row_element(const uint32_t *row, unsigned wi) {
    const unsigned c = wi / 9, i = wi % 9, bit = 29 * i, j = bit >> 5, sh = bit & 31;
    const uint32_t lo = row[c * 8 + j], hi = j + 1 < 8 ? row[c * 8 + j + 1] : 0u;
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) & F29::MASK;
}

// ------------------------------------------------------------------------------------
// G2 line tables.  The G2 side of a Miller loop is a chain of 344 dependent rounds (3 per doubling step, 4
// per addition step) of at most five independent Fq2 products.  A wavefront issues one instruction per four
// cycles whatever its lanes do, so the chain's time is the instruction count of the wavefront that runs it:
// per round one fused two-product reduction per lane (component `part` of product k), then the additions
// of the step formulas on ONE component per lane (lanes (0, part) of the group; the only cross-component
// operation, xi * t, recomputes the partner's t).  ~650 instructions per round; one lane per Fq2 value for
// the additions (miller.h, G12Miller) took ~1100, two lanes per POINT with everything in registers (no
// LDS, no barriers, 32 points per wavefront) 5.1 K / 6.4 K per doubling / addition step -- 2.5 x longer.
// Formulas and lazy bounds are WMiller::doubling_round / addition_round's (libff's step formulas).
// ------------------------------------------------------------------------------------
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

enum G2PVar {                                  // Fq2S slots of one point
    GP_X, GP_Y, GP_Z, GP_S,                    // R = (X, Y, Z): X, Z < 2p, Y < 4p (round 6; GP_S: unused since H = 2 Y Z)
    GP_QX, GP_QY, GP_Q1X, GP_Q1Y, GP_Q2X, GP_Q2Y, GP_TWB, GP_ONE,
    GP_A, GP_B, GP_D, GP_H, GP_E, GP_G, GP_BMF, GP_XIT,           // doubling step
    GP_DD, GP_EE, GP_F, GP_GG, GP_HH, GP_I, GP_J, GP_IMJ,         // addition step
    GP_L0, GP_L1, GP_L2,                       // the line being assembled
    GP_PX, GP_PY, GP_S1, GP_S2,                // fused kernel: (px, 0), (py, 0) of the pair's G1 point; ell_VW * py, ell_VV * px
    GP_P0, GP_P1, GP_P2, GP_P3, GP_P4, GP_P5,  // the round's products
    GP_NQX, GP_NQY,                            // -Q (the signed-digit loop's subtractions)
    GP_LIFT0, GP_LIFT1, GP_LIFT2, GP_LIFT3,    // the products' negation offsets K p, lifted (w12.h: w12_comp_mul_lift): 7 rows of nine words
    GP_STRIDE
};
// the bounds K (units of p) the products of G2Pre's rounds keep for their b-operands; kb below holds INDICES into this list
static constexpr int GP_KLIST[7] = {2, 4, 5, 6, 8, 10, 40};
static constexpr int gp_ki(int K) { return K == 2 ? 0 : K == 4 ? 1 : K == 5 ? 2 : K == 6 ? 3 : K == 8 ? 4 : K == 10 ? 5 : 6; }
static_assert(GP_KLIST[gp_ki(40)] == 40 && GP_KLIST[gp_ki(5)] == 5 && GP_KLIST[gp_ki(10)] == 10, "index of a bound");
static constexpr int GP_GROUPS = 5;
static constexpr int GP_LDS_FQ2 = GP_GROUPS * GP_STRIDE;

// which step of the ate loop table entry e belongs to: 0: a doubling step, 1: the addition of Q, 2: of pi(Q), 3: of -pi^2(Q)
LSA_HD int tm_entry_kind(int e) {
    int idx = 0;
    for (int i = 63; i >= 0; --i) {
        if (idx == e) return 0;
        idx++;
        if (ate_bit(i)) {
            if (idx == e) return 1;
            idx++;
        }
    }
    return e == idx ? 2 : 3;
}

// the signed-digit loop (miller.h: ate_naf_digit): 65 doubling steps, an addition of Q (kind 1) or of -Q (kind 4) at the non-zero
// digits below the top one, then the two Frobenius steps -- 88 entries.  No table has this shape: fused kernel only.
static constexpr int NAF_NUM_ENTRIES = 65 + 21 + 2;
LSA_HD int tm_naf_entry_kind(int e) {
    int idx = 0;
    for (int i = 64; i >= 0; --i) {
        if (idx == e) return 0;
        idx++;
        const int d = ate_naf_digit(i);
        if (d) {
            if (idx == e) return d > 0 ? 1 : 4;
            idx++;
        }
    }
    return e == idx ? 2 : 3;
}

// the same schedule walked incrementally: next() returns the kind of entry 0, 1, 2, ...
struct TmSchedule {
    int bit = 63, tail = 0;
    bool add_pending = false;
    LSA_HD int next() {
        if (bit < 0) return 2 + tail++;
        if (add_pending) { add_pending = false; bit--; return 1; }
        if (ate_bit(bit)) add_pending = true;
        else bit--;
        return 0;
    }
};

template <class X, int NG = GP_GROUPS>
struct G2Pre {
    X &x;
    Fq2S *mem;          // NG * GP_STRIDE elements
    // the <= 5 products of a round as packed slot lists (byte k = operand of product k): a lane picks its operand with
    // a shift -- an int8 array indexed by the lane lives in scratch, two dependent scratch loads per round
    // kb: byte k = a bound K_k (in units of p) of product k's b-operand -- its c1 is negated as K_k p - b1.  Round 6: bounds are
    // kept PER PRODUCT (2 * bound(a) * K_k < 169 for each, listed beside the table) instead of one "a < 4p, b < 20p" for all of
    // them, which is what forced a conditional subtraction onto most of the values a step formula produces.
    struct Prod { uint64_t a, b, kb; int n; };
    static constexpr uint64_t pack5(int v0, int v1 = 0, int v2 = 0, int v3 = 0, int v4 = 0, int v5 = 0) {
        return (uint64_t)v0 | (uint64_t)v1 << 8 | (uint64_t)v2 << 16 | (uint64_t)v3 << 24 | (uint64_t)v4 << 32 | (uint64_t)v5 << 40;
    }

    // ops 0-2: the rounds of a doubling step, 3-6: of an addition step with the point at slots (x2, x2 + 1).
    // scaled (the fused kernel): the round after the one that fixes ell_VW and ell_VV also multiplies them by (py, px) on
    // lanes that are free there, so that the consumer gets its line ready to use.
    // Bounds (units of p; a x b -> 2 a K):  X, Z, every product < 2;  Y < 4;  (px, 0), (py, 0), the point's coordinates < 2.
    //   0: X Y  2x4=16   Y Y  4x4=32   Z Z  2x2   Y Z  4x2=16   X X  2x2
    //   1: b' D  1x6=12   B H  2x4=16   ell_VW py  4x2=16   ell_VV px  6x2=24
    //   2: E E  2x2   A (B-F)  1.5x8=24   G G  4.5x5=45   1 * xi(E-B)  1x40=80
    //   3: x2 Z, y2 Z  2x2
    //   4: D D  4x4=32   E E  6x6=72   E x2  6x2=24   D y2  4x2=16   ell_VW py  4x2   ell_VV px  6x2
    //   5: D F  4x2   X F  2x2   Z G  2x2   1 * xi(E x2 - D y2)  1x40=80
    //   6: D J  4x8=64   E (I-J)  6x10=120   H Y  2x4=16   Z H  2x2
    static LSA_HD Prod products_of(int op, int x2, bool scaled) {
        const int X2 = x2, Y2 = x2 + 1;
        switch (op) {
        case 0: return {pack5(GP_X, GP_Y, GP_Z, GP_Y, GP_X), pack5(GP_Y, GP_Y, GP_Z, GP_Z, GP_X), pack5(gp_ki(4), gp_ki(4), gp_ki(2), gp_ki(2), gp_ki(2)), 5};
        case 1: return {pack5(GP_TWB, GP_B, GP_L1, GP_L2), pack5(GP_D, GP_H, GP_PY, GP_PX), pack5(gp_ki(6), gp_ki(4), gp_ki(2), gp_ki(2)), scaled ? 4 : 2};
        case 2: return {pack5(GP_E, GP_A, GP_G, GP_ONE), pack5(GP_E, GP_BMF, GP_G, GP_XIT), pack5(gp_ki(2), gp_ki(8), gp_ki(5), gp_ki(40)), 4};
        case 3: return {pack5(X2, Y2), pack5(GP_Z, GP_Z), pack5(gp_ki(2), gp_ki(2)), 2};
        case 4: return {pack5(GP_DD, GP_EE, GP_EE, GP_DD, GP_L1, GP_L2), pack5(GP_DD, GP_EE, X2, Y2, GP_PY, GP_PX), pack5(gp_ki(4), gp_ki(6), gp_ki(2), gp_ki(2), gp_ki(2), gp_ki(2)), scaled ? 6 : 4};
        case 5: return {pack5(GP_DD, GP_X, GP_Z, GP_ONE), pack5(GP_F, GP_F, GP_GG, GP_XIT), pack5(gp_ki(2), gp_ki(2), gp_ki(2), gp_ki(40)), 4};
        default: return {pack5(GP_DD, GP_EE, GP_HH, GP_Z), pack5(GP_J, GP_IMJ, GP_Y, GP_HH), pack5(gp_ki(8), gp_ki(10), gp_ki(4), gp_ki(2)), 4};
        }
    }
    // component c of slot v
    static LSA_HD F29 ld(const Fq2S *V, int v, unsigned c) { return w12_load(&w12_comp(const_cast<Fq2S &>(V[v]), c)).v; }
    static LSA_HD void st(Fq2S *V, int v, unsigned c, const F29 &val) { w12_store(&w12_comp(V[v], c), Fs{val}); }
    static LSA_HD F29 triple(const F29 &a) { return add_lazy(add_lazy(a, a), a).norm(); }
    // component c of xi * t for t = (t_c, t_o) < 4p tight: c = 0: 9 t0 - t1 + 4p, c = 1: 9 t1 + t0   [< 40; tight]
    static LSA_HD F29 xi_comp(unsigned c, const F29 &tc, const F29 &to) {
        F29 t8;
#pragma unroll
        for (int l = 0; l < 9; l++) t8.l[l] = tc.l[l] << 3;
        const uint32_t pm = w12_mask(0u - c);
        const F29 neg = sub_k<4>(F29::zero(), to);
        F29 sel;
#pragma unroll
        for (int l = 0; l < 9; l++) sel.l[l] = (to.l[l] & pm) | (neg.l[l] & ~pm);
        return w12_norm_u(add_lazy(add_lazy(w12_norm_u(t8), tc), sel));
    }
    // the additions after the products of round `op`, on component c (one lane each)
    // tab: a table is being written -- its rows must be below 4p (256-bit packing, the table kernels' operand bound), which only
    // ell_VV of a doubling step is not by itself
    static LSA_HD void combine(int op, unsigned c, Fq2S *V, bool tab) {
        const unsigned o = c ^ 1u;
        switch (op) {
        case 0: {
            const F29 B = ld(V, GP_P1, c), C = ld(V, GP_P2, c), YZ = ld(V, GP_P3, c);
            const F29 H = add_lazy(YZ, YZ).norm();                                  // (Y+Z)^2 - (B+C) = 2 Y Z       [<4]
            st(V, GP_A, c, f29_halve(ld(V, GP_P0, c)));                             // X Y / 2                [<1.5]
            st(V, GP_B, c, B);
            st(V, GP_D, c, triple(C));                                              // 3C                     [<6]
            st(V, GP_H, c, H);
            st(V, GP_L1, c, sub_k<4>(F29::zero(), H));                              // ell_VW = -H            [<=4]
            const F29 L2 = triple(ld(V, GP_P4, c));                                 // ell_VV = 3 X^2         [<6]
            st(V, GP_L2, c, tab ? condsub4(L2) : L2);
        } break;
        case 1: {
            const F29 E = ld(V, GP_P0, c), B = ld(V, GP_B, c);
            const F29 F = triple(E);                                                // 3E                     [<6]
            st(V, GP_E, c, E);
            st(V, GP_G, c, f29_halve(add_lazy(B, F).norm()));                       // (B+F)/2                [<4.5]
            st(V, GP_BMF, c, sub_k<6>(B, F));                                       // B - F + 6p             [<8]
            st(V, GP_Z, c, ld(V, GP_P1, c));                                        // Z3 = B H               [<2]
            const F29 tc = sub_k<2>(E, B);                                          // E - B + 2p             [<4]
            const F29 to = sub_k<2>(ld(V, GP_P0, o), ld(V, GP_B, o));               // the other component of the same
            st(V, GP_XIT, c, xi_comp(c, tc, to));                                   // xi (E - B)             [<40]
            st(V, GP_S1, c, ld(V, GP_P2, c));                                       // (scaled mode: ell_VW * py, ell_VV * px; else unused)
            st(V, GP_S2, c, ld(V, GP_P3, c));
        } break;
        case 2: {
            st(V, GP_X, c, ld(V, GP_P1, c));
            st(V, GP_Y, c, condsub4(sub_k<6>(ld(V, GP_P2, c), triple(ld(V, GP_P0, c)))));   // G^2 - 3E^2 + 6p [<8] -> [<4]
            st(V, GP_L0, c, ld(V, GP_P3, c));                                       // ell_0 = xi (E - B), reduced  [<2]
        } break;
        case 3: {
            const F29 E = sub_k<2>(ld(V, GP_Y, c), ld(V, GP_P1, c));                // Y1 - y2 Z1 + 2p        [<6]
            const F29 D = sub_k<2>(ld(V, GP_X, c), ld(V, GP_P0, c));                // X1 - x2 Z1 + 2p        [<4]
            st(V, GP_DD, c, D);
            st(V, GP_EE, c, E);
            st(V, GP_L1, c, D);                                                     // ell_VW = D             [<4]
            const F29 L2 = sub_k<6>(F29::zero(), E);                                // ell_VV = -E            [<=6]
            st(V, GP_L2, c, tab ? condsub4(L2) : L2);
        } break;
        case 4: {
            st(V, GP_F, c, ld(V, GP_P0, c));
            st(V, GP_GG, c, ld(V, GP_P1, c));
            const F29 tc = sub_k<2>(ld(V, GP_P2, c), ld(V, GP_P3, c));              // E x2 - D y2 + 2p       [<4]
            const F29 to = sub_k<2>(ld(V, GP_P2, o), ld(V, GP_P3, o));
            st(V, GP_XIT, c, xi_comp(c, tc, to));                                   //                        [<40]
            st(V, GP_S1, c, ld(V, GP_P4, c));
            st(V, GP_S2, c, ld(V, GP_P5, c));
        } break;
        case 5: {
            const F29 H = ld(V, GP_P0, c), I = ld(V, GP_P1, c);
            const F29 J = sub_k<4>(add_lazy(H, ld(V, GP_P2, c)), add_lazy(I, I));   // H + Z1 G - 2I + 4p     [<8]
            st(V, GP_HH, c, H);
            st(V, GP_J, c, J);
            st(V, GP_IMJ, c, sub_k<8>(I, J));                                       // I - J + 8p             [<10]
            st(V, GP_L0, c, ld(V, GP_P3, c));                                       // ell_0, reduced         [<2]
        } break;
        default: {
            st(V, GP_X, c, ld(V, GP_P0, c));
            st(V, GP_Y, c, sub_k<2>(ld(V, GP_P1, c), ld(V, GP_P2, c)));             // E (I - J) - H Y1 + 2p  [<4]
            st(V, GP_Z, c, ld(V, GP_P3, c));
        } break;
        }
    }

    // one round; after the rounds that complete a line (2: doubling, 5: addition) lanes (k < 3, part) write it out: into the
    // table out[g] (packed, global memory) and / or as three Fq2S at rows[g] (LDS: the consumer's row ring)
    LSA_HD void round(int op, int x2, uint32_t *const *out, int entry, Fq2S *const *rows = nullptr, bool scaled = false) {
        Fq2S *m = mem;
        const Prod pr = products_of(op, x2, scaled);
        const bool tab = out != nullptr;
        x.par([=](unsigned lane) {
            const unsigned g = lane / 12, k = (lane % 12) >> 1, part = lane & 1;
            if (g >= (unsigned)NG || (int)k >= pr.n) return;
            Fq2S *V = m + g * GP_STRIDE;
            // (bounds per product: products_of)
            const Fs r = {w12_comp_mul_lift(part, w12_load(V + (unsigned)((pr.a >> (8 * k)) & 0xffu)), w12_load(V + (unsigned)((pr.b >> (8 * k)) & 0xffu)),
                                            reinterpret_cast<const uint32_t *>(V + GP_LIFT0) + 9u * (unsigned)((pr.kb >> (8 * k)) & 0xffu))};
            w12_store(&w12_comp(V[GP_P0 + k], part), r);
        });
        x.par([=](unsigned lane) {
            const unsigned g = lane / 12, k = (lane % 12) >> 1, part = lane & 1;
            if (g >= (unsigned)NG || k) return;
            combine(op, part, m + g * GP_STRIDE, tab);
        });
        if (op == 2 || op == 5) {
            x.par([=](unsigned lane) {
                const unsigned g = lane / 12, k = (lane % 12) >> 1, part = lane & 1;
                if (g >= (unsigned)NG || k >= 3) return;
                const F29 v = ld(m + g * GP_STRIDE, GP_L0 + (int)k, part);
                // the consumer's row: {ell_0, ell_VW, ell_VV}, or in scaled mode {ell_0, ell_VW * py, ell_VV * px}
                if (rows && rows[g]) w12_store(&w12_comp(rows[g][k], part), Fs{scaled && k ? ld(m + g * GP_STRIDE, GP_S1 + (int)k - 1, part) : v});
                if (!out || !out[g]) return;
                uint32_t w[8];
                v.pack256(w);                                                      // < 4p < 2^256
                uint32_t *d = out[g] + entry * TM_ROW_WORDS + (2 * k + part) * 8;
#pragma unroll
                for (int l = 0; l < 8; l++) d[l] = w[l];
            });
        }
    }
    // the rounds of table entry `entry` (its kind: tm_entry_kind)
    LSA_HD void entry_rounds(int kind, int entry, uint32_t *const *out, Fq2S *const *rows, bool scaled = false) {
        const int x2 = kind == 2 ? GP_Q1X : (kind == 3 ? GP_Q2X : (kind == 4 ? GP_NQX : GP_QX));
        const int first = kind == 0 ? 0 : 3, last = kind == 0 ? 3 : 7;
#pragma unroll 1
        for (int op = first; op < last; op++) round(op, x2, out, entry, rows, scaled);
    }
    // scaled mode: the affine G1 point of each pair (libff precompute_G1; neg: -P, the conjugate Miller value)
    LSA_HD void setup_g1(const Jac<Fq> *const *Pp, const uint8_t *neg, unsigned count) {
        Fq2S *m = mem;
        x.par([=](unsigned lane) {
            const unsigned g = lane / 12, k = lane % 12;
            if (g >= (unsigned)NG || k != 2) return;
            tm_setup_g1(g < count, Pp[g], g < count && neg[g] != 0, m + g * GP_STRIDE + GP_PX);
        });
    }

    // libff to_affine_coordinates (O -> (0, 1)), pi(Q), -pi^2(Q), R = Q for the points Q[g], g < count
    LSA_HD void setup(const Jac<Fq2> *const *Qp, unsigned count, uint32_t *const *out) {
        Fq2S *m = mem;
        x.par([=](unsigned lane) {
            const unsigned g = lane / 12, k = lane % 12;
            if (g >= (unsigned)NG || k) return;
            const Jac<Fq2> *Q = g < count ? Qp[g] : nullptr;
            Fq2S *V = m + g * GP_STRIDE;
            // libff to_affine_coordinates (O -> (0, 1)), pi(Q), -pi^2(Q)
            P2 qx = P2::zero(), qy = P2::one();
            if (Q && !Q->Z.is_zero()) {
                qx = load2(Q->X); qy = load2(Q->Y);
                if (!(Q->Z == Fq2::one())) {
                    const P2 zi = load2(Q->Z).inverse(), zi2 = zi.sqr();
                    qx = qx * zi2; qy = qy * (zi2 * zi);
                }
            }
            const P2 gx = fq2_constT<PB>(LSA_TWIST_MUL_BY_Q_X), gy = fq2_constT<PB>(LSA_TWIST_MUL_BY_Q_Y);
            const P2 q1x = gx * qx.conj(), q1y = gy * qy.conj();
            V[GP_QX] = qx; V[GP_QY] = qy;
            V[GP_Q1X] = q1x; V[GP_Q1Y] = q1y;
            V[GP_Q2X] = gx * q1x.conj(); V[GP_Q2Y] = (gy * q1y.conj()).neg();
            V[GP_NQX] = qx; V[GP_NQY] = qy.neg();
            V[GP_X] = qx; V[GP_Y] = qy; V[GP_Z] = P2::one();
            V[GP_S] = qy + P2::one();
            V[GP_TWB] = fq2_constT<PB>(LSA_TWIST_B);
            V[GP_ONE] = P2::one();
            for (int t = 0; t < 7; t++) w12_lift_kp(GP_KLIST[t], reinterpret_cast<uint32_t *>(V + GP_LIFT0) + 9 * t);
            if (out && out[g]) {                                   // libff keeps the affine point beside the coefficients
                uint32_t *d = out[g] + ATE_NUM_COEFFS * TM_ROW_WORDS;
                qx.c0.v.pack256(d); qx.c1.v.pack256(d + 8); qy.c0.v.pack256(d + 16); qy.c1.v.pack256(d + 24);
            }
        });
    }
    // tables out[g] (TM_TAB_WORDS words each, null: idle group) <- precompute_G2(Q[g])
    LSA_HD void run(const Jac<Fq2> *const *Qp, unsigned count, uint32_t *const *out) {
        setup(Qp, count, out);
        TmSchedule sch;
#pragma unroll 1
        for (int e = 0; e < ATE_NUM_COEFFS; e++) entry_rounds(sch.next(), e, out, nullptr);     // ONE call site: the rounds' code once
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
#if defined(LSA_F29_COLS)
    return f29_dot_cols<N>(a, b);     // (fs29.h: an A/B switch of round 6 -- measured equal, the serial form stays)
#endif
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
                    if (c < (unsigned)TM_CHUNKS) row[j] = tm_row_element(tb[c * TM_MAXM + pair] + entry * TM_ROW_WORDS, wi);
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
                // four (three, one) products in TWO fused reductions of two each, an off-diagonal pair's a-operand doubled: as
                // TabMillerP::round below (round 6: 8 products' multiply-adds + 2 reductions instead of 9 + 5; a group's sum may
                // reach 320 p^2 -- result < 2.9p --, both groups < 4.9p, two conditional subtractions restore < 2p)
                F29 grp[2];
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    F29 xa[4], yb[4];
#pragma unroll
                    for (int jj = 0; jj < 2; jj++) {
                        const TMTerm t = term(mode, lane, 2 * g + jj, M, u, scale);
                        tm_comp_operands<2>(part, w12_load(m + t.a), w12_load(m + t.b), xa[2 * jj], yb[2 * jj], xa[2 * jj + 1], yb[2 * jj + 1]);
                        const uint32_t w2 = w12_mask(t.w2);
#pragma unroll
                        for (int l = 0; l < 9; l++) { xa[2 * jj].l[l] += xa[2 * jj].l[l] & w2; xa[2 * jj + 1].l[l] += xa[2 * jj + 1].l[l] & w2; }
                        xa[2 * jj] = w12_norm_u(xa[2 * jj]);
                        xa[2 * jj + 1] = w12_norm_u(xa[2 * jj + 1]);
                    }
                    grp[g] = dotn<4>(xa, yb);
                }
                const Fs r = {condsub2(condsub4(w12_norm_u(add_lazy(grp[0], grp[1]))))};
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

    // f = 1, the affine G1 points
    LSA_HD void setup(const Jac<Fq> *const *P, const uint8_t *neg, const unsigned *cnt) {
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
    }
    // accumulator c < nacc: F <- prod_{i < cnt[c]} miller_loop(+-P[c][i], table[c][i]);  M = max cnt
    LSA_HD void run(const Jac<Fq> *const *P, const uint8_t *neg, const unsigned *cnt, unsigned M) {
        setup(P, neg, cnt);
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

// ------------------------------------------------------------------------------------
// The Fq12 chain of the fused kernel: NC accumulators per wavefront, ONE pair each, every line delivered READY TO USE
// ({ell_0, ell_VW * py, ell_VV * px}, by the G2 wavefront of the workgroup) into slot entry % TP_RING of the accumulator's row
// ring.  No helper lanes, no scaling rounds, no global loads: five accumulators fill sixty lanes.
// ------------------------------------------------------------------------------------
// component c of xi * t for t = (t_c, t_o) < 2p tight: c = 0: 9 t0 - t1 + 2p, c = 1: 9 t1 + t0   [< 20; tight]
LSA_HD F29 wt_xi_comp(unsigned c, const F29 &tc, const F29 &to) {
    F29 t8;
#pragma unroll
    for (int l = 0; l < 9; l++) t8.l[l] = tc.l[l] << 3;
    const uint32_t pm = w12_mask(0u - c);
    const F29 neg = sub_k<2>(F29::zero(), to);
    F29 sel;
#pragma unroll
    for (int l = 0; l < 9; l++) sel.l[l] = (to.l[l] & pm) | (neg.l[l] & ~pm);
    return w12_norm_u(add_lazy(add_lazy(w12_norm_u(t8), tc), sel));
}
// the value of the lane next door (2i <-> 2i + 1): one DPP move per limb
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ F29 wt_swap(const F29 &a) {
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.l[i], 0xB1, 0xf, 0xf, false);   // quad_perm:[1,0,3,2]
        asm volatile("" : "+v"(r.l[i]));       // opaque: GCNDPPCombine must not fold the move into its consumer (quad29.h)
    }
    return r;
}
#endif

// (TP_RING rows of three Fq2S: the G2 wavefront may run that many entries ahead of the chain)
static constexpr int TP_RING = 6;
enum { TP_F = 0, TP_XF = 6, TP_T = 12, TP_RAW = 18, TP_STRIDE = TP_RAW + 3 * TP_RING };
template <class X, int NC>
struct TabMillerP {
    static_assert(12 * NC <= 64, "one wavefront");
    static constexpr int ZERO = NC * TP_STRIDE;
    static constexpr int LDS_FQ2 = ZERO + 1;
    X &x;
    Fq2S *mem;                        // LDS_FQ2 elements
    using WM = WMiller<X>;

    LSA_HD void setup() {
        Fq2S *m = mem;
        x.par([=](unsigned lane) {
            if (lane == 63) m[ZERO] = Fq2S::zero();
            if (lane >= 12u * NC || (lane & 1)) return;
            const unsigned c = lane / 12, k = (lane % 12) >> 1;
            const Fq2S f0 = k == 0 ? P2::one() : P2::zero();
            m[c * TP_STRIDE + TP_F + k] = f0;
            m[c * TP_STRIDE + TP_XF + k] = WM::st(WM::xi_times(WM::ld(f0)));
        });
    }
    // mode 1: f <- f*f, 2: f <- f * line(entry)
    // On the device the round is ONE phase: a lane's result is component `part` of coefficient k, its neighbour (DPP
    // quad_perm) holds the other component, so each lane derives its own component of xi * T and stores both F and XF -- the
    // wavefront's loads of F / XF all precede these stores in program order (one wavefront, in-order LDS queue).  The host
    // build (lanes run one after the other) keeps T and a second phase.
    LSA_HD void round(int mode, int entry) {
        Fq2S *m = mem;
        x.par([=](unsigned lane) {
            if (lane >= 12u * NC) return;
            const unsigned part = lane & 1;
            const int base = (int)(lane / 12) * TP_STRIDE, k = (int)((lane % 12) >> 1);
            F29 res;
            if (mode == 2) {
                // three products, ONE reduction: a < 20p (xi * f) in at most two terms, b < 2p: 4 * 40 + 2 * 4 = 168 < 169 p^2
                F29 xa[6], yb[6];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    int ai = k - (j == 0 ? 0 : j + 2);                      // line coefficients sit at w^0, w^3, w^4
                    const bool wrap = ai < 0;
                    if (wrap) ai += 6;
                    tm_comp_operands<2>(part, w12_load(m + base + (wrap ? TP_XF : TP_F) + ai), w12_load(m + base + TP_RAW + 3 * (entry % TP_RING) + j),
                                        xa[2 * j], yb[2 * j], xa[2 * j + 1], yb[2 * j + 1]);
                }
                res = dotn<6>(xa, yb);
            } else {
                // the 4 (k even) or 3 unordered pairs of a square in TWO fused reductions of two pairs each (round 6; one
                // reduction per pair and a product by 1 before: 9 products' worth of multiply-adds and 5 reductions, now 8 and 2).
                // An off-diagonal pair counts twice: its a-operand is doubled (< 40p when it is xi * f).  A group's integer
                // sum may then exceed the 169 p^2 that guarantee a result below 2p -- worst case two doubled wrapped pairs,
                // 2 * 2 * 40p * 2p = 320 p^2 -- which only loosens the result's bound: T / 2^261 + p < 2.9p, the 64-bit
                // columns hold 9 * 4 + 9 products of tight limbs whatever the values.  Both groups: < 4.9p, two conditional
                // subtractions bring the coefficient back under 2p.
                F29 grp[2];
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    F29 xa[4], yb[4];
#pragma unroll
                    for (int jj = 0; jj < 2; jj++) {
                        const int j = 2 * g + jj;
                        int ao = ZERO, bo = ZERO;
                        uint32_t w2 = 0;
                        if (j < sqr_pair_count(k)) {
                            int ti, ui;
                            bool wrap;
                            sqr_pair(k, j, ti, ui, wrap);
                            ao = base + (wrap ? TP_XF : TP_F) + ti;
                            bo = base + TP_F + ui;
                            w2 = ti == ui ? 0u : ~0u;
                        }
                        tm_comp_operands<2>(part, w12_load(m + ao), w12_load(m + bo), xa[2 * jj], yb[2 * jj], xa[2 * jj + 1], yb[2 * jj + 1]);
                        w2 = w12_mask(w2);
#pragma unroll
                        for (int l = 0; l < 9; l++) { xa[2 * jj].l[l] += xa[2 * jj].l[l] & w2; xa[2 * jj + 1].l[l] += xa[2 * jj + 1].l[l] & w2; }
                        xa[2 * jj] = w12_norm_u(xa[2 * jj]);
                        xa[2 * jj + 1] = w12_norm_u(xa[2 * jj + 1]);
                    }
                    grp[g] = dotn<4>(xa, yb);
                }
                res = condsub2(condsub4(w12_norm_u(add_lazy(grp[0], grp[1]))));
            }
#if defined(__HIP_DEVICE_COMPILE__)
            const F29 other = wt_swap(res);
            const F29 xf = wt_xi_comp(part, res, other);                                      // [< 20]
            w12_store(&g12_part(m[base + TP_F + k], part), Fs{res});
            w12_store(&g12_part(m[base + TP_XF + k], part), Fs{xf});
#else
            w12_store(&g12_part(m[base + TP_T + k], part), Fs{res});
#endif
        });
#if !defined(__HIP_DEVICE_COMPILE__)
        x.par([=](unsigned lane) {
            if (lane >= 12u * NC || (lane & 1)) return;
            const unsigned c = lane / 12, k = (lane % 12) >> 1;
            Fq2S *base = m + c * TP_STRIDE;
            const Fq2S tv = base[TP_T + k];
            base[TP_F + k] = tv;
            base[TP_XF + k] = WM::st(WM::xi_times(WM::ld(tv)));                              // [< 20]
        });
#endif
    }
    // one table entry (kind 0: a doubling step)
    LSA_HD void entry(int kind, int e) {
#pragma unroll 1
        for (int r = kind == 0 ? 0 : 1; r < 2; r++) round(r ? 2 : 1, e);                     // (one call site)
    }
    LSA_HD Fq2S *row(unsigned c, int slot) const { return mem + c * TP_STRIDE + TP_RAW + 3 * slot; }
    LSA_HD Fq12S result(unsigned c) const {
        Fq12S t;
        for (int k = 0; k < 6; k++) w12_tower_ref(t, k) = mem[c * TP_STRIDE + TP_F + k];
        return t;
    }
};

// ------------------------------------------------------------------------------------
// One accumulator per WAVEFRONT (WTab): the latency shape.  A wavefront issues one instruction per four
// cycles whatever its lanes do, so the time of a Miller loop is its instruction count: here every lane owns
// ONE Fq component of ONE partial product of the round (42 lanes for f*f from unordered coefficient pairs, 36
// for f * line), twelve lanes then sum the anti-diagonals -- ~900 instructions per round against 1400 (f * line)
// and 2400 (f * f) when a lane owns a whole coefficient (TabMiller above, four accumulators per wavefront:
// the throughput shape).  Used while there are fewer accumulators than about two per SIMD.
// ------------------------------------------------------------------------------------
enum {                                     // Fq2S slots of the wavefront's accumulator in LDS
    WT_F = 0, WT_XF = 6, WT_P = 12,        // f; xi * f; partial products [k * 4 + j]
    WT_RAW = 36, WT_SC = 45, WT_PXY = 49,  // as TM_RAW / TM_SC / TM_PXY
    WT_ZERO = WT_PXY + 2 * TM_MAXM,
    WT_LDS_FQ2 = WT_ZERO + 1
};

// What a lane does in a round depends on the lane alone: computed once into a two-word descriptor per lane (LDS).
//   word 0 (f*f):      a | b << 8 | dst << 16 | active << 25 | counts twice << 27        (Fq2S slot numbers)
//   word 1 (f * line): a | b << 8 | dst << 16 | active << 25 | b-is-ell_0 << 26          (b without its ring offset)
// The wrapped terms (i + j >= 6) take their a-operand from xi * f, kept beside f.  Every coefficient sums FOUR slots:
// lanes without a product of their own write the zeros of the absent terms (a = b = the zero slot).
LSA_HD void wt_make_desc(unsigned lane, uint32_t *d) {
    d[0] = d[1] = 0;
    const int q = (int)(lane >> 1);
    if (lane < 42u) {
        // pair q of the 21 unordered coefficient pairs: coefficient k owns 4 (k even) or 3 of them
        const int k = q < 4 ? 0 : q < 7 ? 1 : q < 11 ? 2 : q < 14 ? 3 : q < 18 ? 4 : 5;
        const int j = q - (k < 1 ? 0 : k < 2 ? 4 : k < 3 ? 7 : k < 4 ? 11 : k < 5 ? 14 : 18);
        int ti, ui;
        bool wrap;
        sqr_pair(k, j, ti, ui, wrap);
        d[0] = (uint32_t)((wrap ? WT_XF : WT_F) + ti) | (uint32_t)(WT_F + ui) << 8 | (uint32_t)(WT_P + 4 * k + j) << 16 | 1u << 25 | (uint32_t)(ti != ui) << 27;
    } else if (lane < 48u) {
        const int k = 2 * (q - 21) + 1;                             // the fourth slot of the odd coefficients: zero
        d[0] = (uint32_t)WT_ZERO | (uint32_t)WT_ZERO << 8 | (uint32_t)(WT_P + 4 * k + 3) << 16 | 1u << 25;
    }
    if (lane < 36u) {
        const int k = q / 3, j = q % 3;
        int ai = k - (j == 0 ? 0 : j + 2);                          // line coefficients sit at w^0, w^3, w^4
        const bool wrap = ai < 0;
        if (wrap) ai += 6;
        const int b = j == 0 ? WT_RAW : WT_SC + (j - 1);
        d[1] = (uint32_t)((wrap ? WT_XF : WT_F) + ai) | (uint32_t)b << 8 | (uint32_t)(WT_P + 4 * k + j) << 16 | 1u << 25 | (uint32_t)(j == 0) << 26;
    } else if (lane < 48u) {
        const int k = q - 18;                                       // the fourth slot of every coefficient: zero
        d[1] = (uint32_t)WT_ZERO | (uint32_t)WT_ZERO << 8 | (uint32_t)(WT_P + 4 * k + 3) << 16 | 1u << 25;
    }
}
template <class X>
struct WTabMiller {
    X &x;
    Fq2S *mem;                        // WT_LDS_FQ2 elements
    const uint32_t *const *tab;       // [TM_MAXM] table of pair i (never null)
    uint32_t *desc;                   // 64 x 2 words (wt_make_desc)

    // component `part` of coefficient k of the round's result: the sum of its four slots, < 12p -> < 2p
    static LSA_HD F29 coeff_part(const Fq2S *m, unsigned k, unsigned part) {
        F29 sum = F29::zero();
#pragma unroll
        for (int j = 0; j < 4; j++) sum = add_lazy(sum, w12_load(&w12_comp(const_cast<Fq2S &>(m[WT_P + 4 * k + j]), part)).v);
        return f29_mul(w12_norm_u(sum), F29::one());
    }

    // mode 1: f <- f*f, 2: f <- f * line(use u), 0: nothing.  scale / load as in TabMiller::round.
    LSA_HD void round(int mode, unsigned M, int u, int scale, int load) {
        Fq2S *m = mem;
        const uint32_t *const *tb = tab;
        const uint32_t *ds = desc;
        // uniform ring offsets
        const int r3u = 3 * (u % 3), r2u = 2 * (u % 2);
        const int sa = scale >= 0 ? WT_RAW + 3 * (scale % 3) + 1 : WT_ZERO, sb = scale >= 0 ? WT_PXY + 2 * (scale % (int)M) + 1 : WT_ZERO,
                  sd = WT_SC + 2 * (scale >= 0 ? scale % 2 : 0);
        x.par([=](unsigned lane) {
            uint32_t row = 0;
            if (load >= 0 && lane < (unsigned)TM_LINE_WORDS) row = tm_row_element(tb[(unsigned)load % M] + ((unsigned)load / M) * TM_ROW_WORDS, lane);
            const unsigned part = lane & 1;
            const uint32_t dd = mode ? ds[2 * lane + (mode - 1)] : 0u;
            int ao = (int)(dd & 0xffu), bo = (int)((dd >> 8) & 0xffu), dsto = (int)((dd >> 16) & 0xffu);
            bool active = ((dd >> 25) & 1u) != 0;
            const uint32_t w2 = w12_mask(0u - ((dd >> 27) & 1u));
            if (mode == 2) bo += ((dd >> 26) & 1u) ? r3u : (((dd >> 8) & 0xffu) == (uint32_t)WT_ZERO ? 0 : r2u);
            if (!active) { ao = WT_ZERO; bo = WT_ZERO; }
            if (lane >= 60u && scale >= 0) {                               // ell_VW * py (lanes 60, 61), ell_VV * px (62, 63)
                const int which = (int)((lane >> 1) & 1);
                ao = sa + which; bo = sb - which; dsto = sd + which;
                active = true;
            }
            // a < 20p (xi * f) or < 4p (a table coefficient), b < 2p: 2 * 20 * 2 = 80 < 169
            F29 r = w12_comp_mul<2>(part, w12_load(m + ao), w12_load(m + bo));
#pragma unroll
            for (int l = 0; l < 9; l++) r.l[l] += r.l[l] & w2;             // an off-diagonal pair of a square counts twice
            if (active) w12_store(&w12_comp(m[dsto], part), Fs{r});
            if (load >= 0 && lane < (unsigned)TM_LINE_WORDS) tm_store_word(reinterpret_cast<uint32_t *>(m + WT_RAW + 3 * ((unsigned)load % 3u)) + lane, row);
        });
        if (mode) {
            x.par([=](unsigned lane) {
                // twelve results, but EVERY lane computes one (lane l the same as lane l mod 12): with only twelve lanes on,
                // a lone wavefront runs this phase 1.1-2.1x slower, depending on the CU (w12.h: w12_pin) -- only the
                // stores are predicated
                const unsigned k = (lane >> 1) % 6u, part = lane & 1;
                F29 mine = coeff_part(m, k, part);
#if defined(__HIP_DEVICE_COMPILE__)
                const F29 other = wt_swap(mine);
#else
                const F29 other = coeff_part(m, k, part ^ 1u);
#endif
                F29 xf = wt_xi_comp(part, mine, other);                                           // [< 20]
#if defined(__HIP_DEVICE_COMPILE__)
                w12_pin(mine);
                w12_pin(xf);
#endif
                if (lane < 12) {
                    w12_store(&w12_comp(m[WT_F + k], part), Fs{mine});
                    w12_store(&w12_comp(m[WT_XF + k], part), Fs{xf});
                }
            });
        }
    }

    // f <- prod_{i < cnt} miller_loop(+-P[i], table[i]);  M = max(cnt, 1)
    LSA_HD void run(const Jac<Fq> *const *P, const uint8_t *neg, unsigned cnt, unsigned M) {
        Fq2S *m = mem;
        uint32_t *dsc = desc;
        x.par([=](unsigned lane) {
            wt_make_desc(lane, dsc + 2 * lane);
            if (lane == 63) m[WT_ZERO] = Fq2S::zero();
            if (lane < 6) {
                const Fq2S f0 = lane == 0 ? P2::one() : P2::zero();
                m[WT_F + lane] = f0;
                m[WT_XF + lane] = WMiller<X>::st(WMiller<X>::xi_times(WMiller<X>::ld(f0)));
            }
            else if (lane >= 60) {
                const unsigned i = lane - 60;                               // TM_MAXM == 4
                tm_setup_g1(i < cnt, P[i], neg[i] != 0, m + WT_PXY + 2 * i);
            }
        });
        const int U = ATE_NUM_COEFFS * (int)M;
        int u = 0, ns = 0, nl = 0;
        auto side = [&](int &sc, int &ld) {
            sc = (ns < U && ns < nl && ns < u + 2) ? ns : -1;
            ld = (nl < U && nl < u + 3) ? nl : -1;
            if (sc >= 0) ns++;
            if (ld >= 0) nl++;
        };
        int sc, ld;
        side(sc, ld);
        round(0, M, 0, sc, ld);
#pragma unroll 1
        for (int ph = 0; ph < 66; ph++) {
            const bool dbl = ph < 64;
            const int lines = dbl ? 1 + ate_bit(63 - ph) : 1;
            if (dbl) {
                side(sc, ld);
                round(1, M, u, sc, ld);
            }
#pragma unroll 1
            for (int li = 0; li < lines; li++) {
#pragma unroll 1
                for (unsigned i = 0; i < M; i++) {
                    side(sc, ld);
                    round(2, M, u, sc, ld);
                    u++;
                }
            }
        }
    }
    LSA_HD Fq12S result() const {
        Fq12S t;
        for (int k = 0; k < 6; k++) w12_tower_ref(t, k) = mem[WT_F + k];
        return t;
    }
};

static constexpr int RT_MAXM = 2;
enum { RT_F = 0, RT_PXY = 6, RT_LINES = RT_PXY + RT_MAXM, RT_F2 = RT_LINES + RT_MAXM * ATE_NUM_COEFFS * 3, RT_LDS_FQ2 = RT_F2 + 6 };   // (RT_F2: the accumulator's other home)
// The 66 steps of a Miller loop over a table (64 doubling steps with one or two lines each, two closing lines) cut into K <= 8
// contiguous ranges for the K workgroups that share one accumulator (rt_miller_run): range [start[r], start[r + 1]) costs its
// workgroup 63 - start[r] squarings (none from step 64 on) + M line products per line; the smallest bound T that K ranges
// cover the loop with, ranges grown from the end.
struct RtSplit { uint8_t start[9]; };
inline RtSplit rt_split(unsigned M, unsigned K) {
    auto lines = [](int ph) { return ph < 64 ? 1 + ate_bit(63 - ph) : 1; };
    RtSplit best{};
    for (int T = 1; T < 1024; T++) {
        int b = 66, starts[9], n = 0;
        starts[0] = 66;
        for (unsigned r = 0; r < K && b > 0; r++) {
            int a = b, ln = 0;
            while (a > 0) {
                const int na = a - 1, sq = na < 64 ? 63 - na : 0;
                if (sq + (int)M * (ln + lines(na)) > T) break;
                ln += lines(na);
                a = na;
            }
            if (a == b) break;
            starts[++n] = a;
            b = a;
        }
        if (b != 0) continue;
        // n ranges (n <= K) from the end; workgroups beyond n get empty ranges at the front
        for (unsigned r = 0; r <= K; r++) {
            const int back = (int)K - (int)r;              // start[r] = the start of the range `back` places from the end
            best.start[r] = (uint8_t)(back <= n ? starts[back] : 0);
        }
        best.start[K] = 66;
        return best;
    }
    for (unsigned r = 0; r <= K; r++) best.start[r] = (uint8_t)(r == 0 ? 0 : 66);      // (not reached: T = 64 + 102 M always covers)
    return best;
}

#if defined(__HIP_DEVICE_COMPILE__)
// ------------------------------------------------------------------------------------------------------------------
// The same job on the row engine of w12.h (192 lanes, up to RT_MAXM pairs per accumulator): the latency shape for the
// verifiers' lone pairing checks.  WTabMiller's round is two phases (42 + 12 lanes, an LDS round trip and a barrier
// between them, the line of the next round scaled by four helper lanes on the side): ~780 instructions and two barriers
// per round, 2.1 us.  Here
//   * the lines are made ONCE, before the loop: all 102 rows of a pair's table are unpacked, ell_VW scaled by py and
//     ell_VV by px (612 values per pair, four batches of one Fq product per lane) and parked in LDS (22 KB per pair);
//   * a round is ONE row product (w12_rows): f <- f * f, or f <- f * line with the line's three coefficients (0, 3, 4
//     of the polynomial basis: libff mul_by_024) as a sparse second factor -- ~600 instructions, 1.15 us.
// Same field elements as WTabMiller's (products commute, values leave canonical), 64 + 102 M rounds.
// ------------------------------------------------------------------------------------------------------------------
// m: RT_LDS_FQ2 values in LDS; tab, P, neg: RT_MAXM entries each (pairs >= cnt: the identity table); result in m[RT_F .. RT_F + 5].
// [lo, hi) != [0, 66): ONE OF K workgroups that share the loop.  With F_s the product of the lines of step s, the loop computes
// f = prod_s F_s^(2^(63 - s)) (times the two closing lines) by f <- f^2 * F_s; the factors of a SUBSET of the steps alone
// obey the same recurrence -- g <- g^2 every step, g <- g * F_s on the workgroup's own steps -- and the K results
// multiply to f.  The squarings are the chain, the 102 M line products are shared out, on K otherwise idle CUs; `own`
// (104 bytes of LDS) marks the entries of the own steps.
// (round 6, second session) WHICH steps a workgroup takes: a contiguous range [lo, hi) of the 66, not a residue class.  A
// workgroup's accumulator is 1 until its first own step, so its chain starts THERE -- 63 - lo squarings instead of 64 for
// everybody -- and the ranges are cut (rt_split, on the host) so that squarings + line products come out even: 65 links
// instead of 77 for one pair per accumulator, 73 instead of 90 for two.
__device__ __forceinline__ void rt_miller_run(Fq2S *m, const uint32_t *const *tab, const Jac<Fq> *const *P, const uint8_t *neg, unsigned cnt, unsigned M,
                                              unsigned lo, unsigned hi, uint8_t *own) {
    const unsigned lane = threadIdx.x;
    if (lane < 66) {                                      // lane = step: its entries are [first, first + lines)
        unsigned first = 0;
        for (unsigned s = 0; s < lane; s++) first += s < 64 ? 1u + (unsigned)ate_bit(63 - (int)s) : 1u;
        const unsigned lines = lane < 64 ? 1u + (unsigned)ate_bit(63 - (int)lane) : 1u;
        for (unsigned l = 0; l < lines; l++) own[first + l] = (uint8_t)(lane >= lo && lane < hi);
    }
    {   // (px, py) of every pair, computed by every lane (no sparse EXEC mask: w12_pin), stored by one
        const unsigned i = lane & (RT_MAXM - 1);
        Fq2S pxy[2];
        tm_setup_g1(i < cnt, P[i], neg[i] != 0, pxy);
        w12_pin(pxy[0].c0.v);
        w12_pin(pxy[1].c0.v);
        if (lane < (unsigned)RT_MAXM) m[RT_PXY + lane] = Fq2S{pxy[0].c0, pxy[1].c0};        // (px, py) as the two halves of one slot
        if (lane < 6) m[RT_F + lane] = lane == 0 ? Fq2S::one() : Fq2S::zero();
    }
    __syncthreads();
    // the lines: value (pair i, entry e, component c) = table word group c of row e, times py (ell_VW) or px (ell_VV)
    for (unsigned idx = lane; idx < M * (unsigned)ATE_NUM_COEFFS * 6u; idx += 192u) {
        const unsigned i = idx / (ATE_NUM_COEFFS * 6u), rem = idx % (ATE_NUM_COEFFS * 6u), e = rem / 6u, c = rem % 6u;
        if (!own[e]) continue;
        const uint32_t *src = tab[i] + e * TM_ROW_WORDS + c * 8;
        uint32_t w[8];
#pragma unroll
        for (int q = 0; q < 8; q++) w[q] = src[q];
        F29 v = F29::unpack256(w);
        if (c >= 2) {
            const Fs s = w12_load(&w12_comp(m[RT_PXY + i], c < 4 ? 1u : 0u));       // ell_VW * py, ell_VV * px
            v = mul(v, s.v);
        }
        w12_store(&w12_comp(m[RT_LINES + (i * ATE_NUM_COEFFS + e) * 3 + (c >> 1)], c & 1u), Fs{v});
    }
    __syncthreads();
    // the accumulator alternates between two homes (RT_F, RT_F2): a link's result is never written over one of its factors, so a
    // link needs ONE barrier (w12_rows, NOALIAS); an odd number of links ends with a copy back to RT_F
    unsigned u = 0, cur = 0;
    auto home = [m](unsigned which) { return m + (which ? (int)RT_F2 : (int)RT_F); };
#pragma unroll 1
    for (int ph = 0; ph < 66; ph++) {
        const bool dbl = ph < 64;
        const int lines = dbl ? 1 + ate_bit(63 - ph) : 1;
        if (dbl && (unsigned)ph > lo) { w12_rows<W12_MUL, false, true>(home(cur ^ 1u), home(cur), home(cur), nullptr); cur ^= 1u; }   // (up to its first own step the accumulator is 1)
        const bool mine = (unsigned)ph >= lo && (unsigned)ph < hi;
#pragma unroll 1
        for (int li = 0; li < lines; li++, u++) {
            if (!mine) continue;
#pragma unroll 1
            for (unsigned i = 0; i < M; i++) { w12_rows<W12_LINE, false, true>(home(cur ^ 1u), home(cur), m + RT_LINES + (i * ATE_NUM_COEFFS + u) * 3, nullptr); cur ^= 1u; }
        }
    }
    if (cur) {
        if (lane < 6) m[RT_F + lane] = m[RT_F2 + lane];
        __syncthreads();
    }
}
#endif

}  // namespace lsa
