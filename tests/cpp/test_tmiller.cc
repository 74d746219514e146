// Host unit test of csrc/tmiller.h: the G2 line-table engine (G2Pre) against the one-lane step
// formulas of miller.h, and the table-driven Miller loop with shared accumulators (TabMiller)
// against products of one-lane Miller loops, executed phase by phase over emulated lane ids.
#include <cstdio>
#include <random>
#include <vector>
#include "tmiller.h"
using namespace lsa;
static std::mt19937_64 rng(11);
static int fails = 0;
#define CHECK(c, msg) do { if (!(c)) { if (fails < 20) printf("FAIL %s (line %d)\n", msg, __LINE__); fails++; } } while (0)
static Fq rand_fq() {
    for (;;) {
        Fq r;
        for (int i = 0; i < 4; i++) { uint64_t x = rng(); r.l[2 * i] = (uint32_t)x; r.l[2 * i + 1] = (uint32_t)(x >> 32); }
        r.l[7] &= 0x3fffffffu;
        bool lt = false;
        for (int i = 7; i >= 0; --i) if (r.l[i] != FqParams::MOD[i]) { lt = r.l[i] < FqParams::MOD[i]; break; }
        if (lt) return r;
    }
}
struct LoopExec {
    template <class F> void par(F f) { for (unsigned l = 0; l < 64; l++) f(l); }
    unsigned nlanes() const { return 64; }
};
static Fq2S tab_fq2(const std::vector<uint32_t> &t, int idx) {      // Fq2 number idx of a packed table
    Fq2S r;
    r.c0.v = F29::unpack256(&t[idx * 16]);
    r.c1.v = F29::unpack256(&t[idx * 16 + 8]);
    return r;
}
// libff alt_bn128_ate_precompute_G2 with the one-lane step formulas of miller.h
static std::vector<Line> expected_lines(const Jac<Fq2> &Q, P2 &qx, P2 &qy) {
    Jac<Fq> dummy = {Fq::one(), Fq::one(), Fq::one()};
    const AffinePair in = miller_affine_inputs(dummy, Q);
    qx = in.qx; qy = in.qy;
    Fq ti;
    for (int i = 0; i < 8; i++) ti.l[i] = LSA_FQ_TWO_INV[i];
    const PB two_inv = PB::from_mont256(ti);
    const P2 twist_b = fq2_constT<PB>(LSA_TWIST_B);
    G2Proj R = {qx, qy, P2::one()};
    std::vector<Line> out;
    for (int i = 63; i >= 0; --i) {
        out.push_back(doubling_step(R, two_inv, twist_b));
        if (ate_bit(i)) out.push_back(addition_step(qx, qy, R));
    }
    const P2 gx = fq2_constT<PB>(LSA_TWIST_MUL_BY_Q_X), gy = fq2_constT<PB>(LSA_TWIST_MUL_BY_Q_Y);
    P2 q1x = gx * qx.conj(), q1y = gy * qy.conj();
    P2 q2x = gx * q1x.conj(), q2y = (gy * q1y.conj()).neg();
    out.push_back(addition_step(q1x, q1y, R));
    out.push_back(addition_step(q2x, q2y, R));
    return out;
}
// rt_split: K contiguous ranges of the 66 steps that cover the loop, cost per range (63 - start squarings + M per line) even
static void check_rt_split() {
    auto lines = [](int ph) { return ph < 64 ? 1 + ate_bit(63 - ph) : 1; };
    int total = 0;
    for (int ph = 0; ph < 66; ph++) total += lines(ph);
    CHECK(total == ATE_NUM_COEFFS, "66 steps hold the 102 table entries");
    for (unsigned M = 1; M <= (unsigned)RT_MAXM; M++)
        for (unsigned K : {1u, 2u, 8u}) {
            const RtSplit s = rt_split(M, K);
            bool ok = s.start[0] == 0 && s.start[K] == 66;
            int worst = 0;
            for (unsigned r = 0; r < K; r++) {
                ok = ok && s.start[r] <= s.start[r + 1];
                int ln = 0;
                for (int ph = s.start[r]; ph < s.start[r + 1]; ph++) ln += lines(ph);
                const int sq = s.start[r] < 64 && s.start[r + 1] > s.start[r] ? 63 - s.start[r] : 0;
                if (sq + (int)M * ln > worst) worst = sq + (int)M * ln;
            }
            CHECK(ok, "rt_split covers the loop with ordered ranges");
            if (K == 1) CHECK(worst == 63 + (int)M * ATE_NUM_COEFFS, "one workgroup: the whole loop");
            if (K == 8) CHECK(worst <= (M == 1 ? 65 : 73), "eight workgroups: 65 links for one pair per accumulator, 73 for two");
        }
}
int main() {
    LoopExec ex;
    check_rt_split();
    // ---- line tables
    const unsigned nq = 5;
    Jac<Fq2> Qs[GP_GROUPS];
    for (unsigned g = 0; g < nq; g++)
        Qs[g] = {{rand_fq(), rand_fq()}, {rand_fq(), rand_fq()}, g % 2 ? Fq2::one() : Fq2{rand_fq(), rand_fq()}};
    Qs[3].Z = Fq2::zero();
    std::vector<std::vector<uint32_t>> tabs(nq + 1, std::vector<uint32_t>(TM_TAB_WORDS, 0xdeadbeefu));
    {
        std::vector<Fq2S> lds(GP_LDS_FQ2);
        uint32_t *out[GP_GROUPS];
        for (unsigned g = 0; g < (unsigned)GP_GROUPS; g++) out[g] = g < nq ? tabs[g].data() : nullptr;
        const Jac<Fq2> *Qp[GP_GROUPS];
        for (unsigned g = 0; g < (unsigned)GP_GROUPS; g++) Qp[g] = &Qs[g < nq ? g : 0];
        G2Pre<LoopExec> pre{ex, lds.data()};
        pre.run(Qp, nq, out);
    }
    for (unsigned g = 0; g < nq; g++) {
        P2 qx, qy;
        const std::vector<Line> want = expected_lines(Qs[g], qx, qy);
        CHECK((int)want.size() == ATE_NUM_COEFFS, "102 coefficient triples");
        bool ok = true;
        for (int e = 0; e < ATE_NUM_COEFFS; e++)
            ok = ok && tab_fq2(tabs[g], 3 * e) == want[e].e0 && tab_fq2(tabs[g], 3 * e + 1) == want[e].eVW && tab_fq2(tabs[g], 3 * e + 2) == want[e].eVV;
        CHECK(ok, "line table");
        CHECK(tab_fq2(tabs[g], 3 * ATE_NUM_COEFFS) == qx && tab_fq2(tabs[g], 3 * ATE_NUM_COEFFS + 1) == qy, "affine Q beside the table");
    }
    // the identity table: every row is the line (1, 0, 0)
    std::vector<uint32_t> &ident = tabs[nq];
    for (int i = 0; i < TM_TAB_WORDS; i++) ident[i] = 0;
    for (int e = 0; e < ATE_NUM_COEFFS; e++) F29::one().pack256(&ident[e * TM_ROW_WORDS]);
    // ---- table-driven Miller loops, accumulators of 1..4 pairs, some terms conjugated
    for (unsigned M = 1; M <= (unsigned)TM_MAXM; M++) {
        std::vector<Fq2S> lds(TM_LDS_FQ2);
        Jac<Fq> Ps[TM_CHUNKS * TM_MAXM];
        const Jac<Fq> *Pp[TM_CHUNKS * TM_MAXM];
        const uint32_t *tp[TM_CHUNKS * TM_MAXM];
        uint8_t neg[TM_CHUNKS * TM_MAXM];
        unsigned qi[TM_CHUNKS * TM_MAXM];
        unsigned cnt[TM_CHUNKS];
        for (int c = 0; c < TM_CHUNKS; c++) {
            cnt[c] = c == 0 ? M : (unsigned)(rng() % (M + 1));          // an empty accumulator stays 1
            for (unsigned i = 0; i < (unsigned)TM_MAXM; i++) {
                const int s = c * TM_MAXM + (int)i;
                Ps[s] = {rand_fq(), rand_fq(), (rng() & 1) ? Fq::one() : rand_fq()};
                if (c == 1 && i == 0) Ps[s].Z = Fq::zero();
                Pp[s] = &Ps[s];
                neg[s] = (uint8_t)(rng() & 1);
                qi[s] = (unsigned)(rng() % nq);
                tp[s] = i < cnt[c] ? tabs[qi[s]].data() : ident.data();
            }
        }
        TabMiller<LoopExec> tm{ex, lds.data(), tp};
        tm.run(Pp, neg, cnt, M);
        for (int c = 0; c < TM_CHUNKS; c++) {
            Fq12S want = Fq12S::one();
            for (unsigned i = 0; i < cnt[c]; i++) {
                const int s = c * TM_MAXM + (int)i;
                Fq12S f = miller_one(Ps[s], Qs[qi[s]]);
                if (neg[s]) f = f.unitary_inverse();
                want = fq12_mul(want, f);
            }
            CHECK(tm.result((unsigned)c) == want, "table-driven miller product");
        }
    }
    // ---- the fused shape: the G2 engine one entry ahead of the Fq12 chain, lines (already scaled by the pair's G1 point)
    // handed over in the row ring, no table
    {
        constexpr int NP = 5;
        using TP = TabMillerP<LoopExec, NP>;
        const unsigned np = 4;                                          // one idle pair
        std::vector<Fq2S> g2mem(NP * GP_STRIDE), tpmem(TP::LDS_FQ2);
        Jac<Fq> Ps[NP];
        const Jac<Fq> *Pp[NP];
        const Jac<Fq2> *Qp[NP];
        uint8_t neg[NP];
        Fq2S *rows[TP_RING][NP];
        for (int c = 0; c < NP; c++) {
            Ps[c] = {rand_fq(), rand_fq(), c == 1 ? Fq::one() : rand_fq()};
            if (c == 2) Ps[c].Z = Fq::zero();
            Pp[c] = &Ps[c];
            neg[c] = (uint8_t)(c & 1);
            Qp[c] = &Qs[c % (int)nq];
            for (int s = 0; s < TP_RING; s++) rows[s][c] = tpmem.data() + c * TP_STRIDE + TP_RAW + 3 * s;
        }
        std::vector<std::vector<uint32_t>> tabs2(np, std::vector<uint32_t>(TM_TAB_WORDS, 0xdeadbeefu));
        uint32_t *tout[NP];
        for (int c = 0; c < NP; c++) tout[c] = (unsigned)c < np ? tabs2[c].data() : nullptr;
        G2Pre<LoopExec, NP> pre{ex, g2mem.data()};
        TP tp{ex, tpmem.data()};
        pre.setup(Qp, np, tout);
        pre.setup_g1(Pp, neg, np);
        tp.setup();
        for (int e = -1; e < ATE_NUM_COEFFS; e++) {
            if (e + 1 < ATE_NUM_COEFFS) pre.entry_rounds(tm_entry_kind(e + 1), e + 1, tout, rows[(e + 1) % TP_RING], true);
            if (e >= 0) tp.entry(tm_entry_kind(e), e);
        }
        for (unsigned c = 0; c < np; c++) {
            Fq12S f = miller_one(Ps[c], Qs[c % nq]);
            if (neg[c]) f = f.unitary_inverse();
            CHECK(tp.result(c) == f, "fused G2 + Fq12 chain");
            // the table written on the way is the UNSCALED one
            bool same = true;
            for (int i = 0; i < TM_TAB_WORDS; i++) same = same && tabs2[c][i] == tabs[c % nq][i];
            CHECK(same, "table emitted by the fused engine");
        }
    }
    // ---- the fused shape over the SIGNED digits of the loop count (88 entries, additions of Q and of -Q): not libff's Miller
    // value -- it differs by vertical lines, elements of Fq6 -- but the same reduced pairing
    {
        CHECK(tm_naf_entry_kind(0) == 0 && tm_naf_entry_kind(NAF_NUM_ENTRIES - 2) == 2 && tm_naf_entry_kind(NAF_NUM_ENTRIES - 1) == 3, "signed-digit schedule ends");
        int kinds[5] = {0, 0, 0, 0, 0};
        for (int e = 0; e < NAF_NUM_ENTRIES; e++) kinds[tm_naf_entry_kind(e)]++;
        CHECK(kinds[0] == 65 && kinds[1] + kinds[4] == 21 && kinds[2] == 1 && kinds[3] == 1, "signed-digit schedule counts");
        // the digits are the loop count
        {
            unsigned __int128 v = 0;
            for (int i = 65; i >= 0; --i) v = 2 * v + (unsigned __int128)(__int128)ate_naf_digit(i);
            CHECK((uint64_t)v == LSA_ATE_LOOP_COUNT_LO && (uint64_t)(v >> 64) == LSA_ATE_LOOP_COUNT_HI, "signed digits sum to 6u + 2");
            for (int i = 0; i < 65; i++) CHECK(!(ate_naf_digit(i) && ate_naf_digit(i + 1)), "non-adjacent");
        }
        constexpr int NP = 5;
        using TP = TabMillerP<LoopExec, NP>;
        const unsigned np = 5;
        std::vector<Fq2S> g2mem(NP * GP_STRIDE), tpmem(TP::LDS_FQ2);
        Jac<Fq> Ps[NP];
        Jac<Fq2> Qc[NP];
        const Jac<Fq> *Pp[NP];
        const Jac<Fq2> *Qp[NP];
        uint8_t neg[NP];
        Fq2S *rows[TP_RING][NP];
        // (the identity holds on the curve -- it is the group law that makes R - Q the point [a - 1]Q: multiples of the generators,
        // un-normalised Jacobian; the other blocks of this test feed the step formulas arbitrary coordinates)
        Jac<Fq> g1 = {Fq::one(), Fq::one() + Fq::one(), Fq::one()};
        Jac<Fq2> g2;
        for (int i = 0; i < 8; i++) { g2.X.c0.l[i] = LSA_G2_GEN_X[0][i]; g2.X.c1.l[i] = LSA_G2_GEN_X[1][i]; g2.Y.c0.l[i] = LSA_G2_GEN_Y[0][i]; g2.Y.c1.l[i] = LSA_G2_GEN_Y[1][i]; }
        g2.Z = Fq2::one();
        auto times = [](auto pt, uint64_t k) {
            auto acc = pt;
            bool have = false;
            for (int b = 63; b >= 0; --b) {
                if (have) acc = jac_dbl(acc);
                if ((k >> b) & 1) { acc = have ? jac_add(acc, pt) : pt; have = true; }
            }
            return acc;
        };
        for (int c = 0; c < NP; c++) {
            Ps[c] = times(g1, rng() | 1);
            Qc[c] = times(g2, rng() | 1);
            Pp[c] = &Ps[c];
            neg[c] = (uint8_t)(c & 1);
            Qp[c] = &Qc[c];
            for (int s = 0; s < TP_RING; s++) rows[s][c] = tpmem.data() + c * TP_STRIDE + TP_RAW + 3 * s;
        }
        uint32_t *tout[NP] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        G2Pre<LoopExec, NP> pre{ex, g2mem.data()};
        TP tp{ex, tpmem.data()};
        pre.setup(Qp, np, tout);
        pre.setup_g1(Pp, neg, np);
        tp.setup();
        for (int e = -1; e < NAF_NUM_ENTRIES; e++) {
            if (e + 1 < NAF_NUM_ENTRIES) pre.entry_rounds(tm_naf_entry_kind(e + 1), e + 1, tout, rows[(e + 1) % TP_RING], true);
            if (e >= 0) tp.entry(tm_naf_entry_kind(e), e);
        }
        for (unsigned c = 0; c < np; c++) {
            Fq12S f = miller_one(Ps[c], *Qp[c]);
            if (neg[c]) f = f.unitary_inverse();
            CHECK(!(tp.result(c) == f), "the signed-digit Miller value is a different representative");
            CHECK(fq12_final_exponentiation(tp.result(c)) == fq12_final_exponentiation(f), "signed-digit loop: the same reduced pairing");
        }
    }
    // ---- one accumulator per wavefront
    for (unsigned cnt = 0; cnt <= (unsigned)TM_MAXM; cnt++) {
        std::vector<Fq2S> lds(WT_LDS_FQ2);
        Jac<Fq> Ps[TM_MAXM];
        const Jac<Fq> *Pp[TM_MAXM];
        const uint32_t *tp[TM_MAXM];
        uint8_t neg[TM_MAXM];
        unsigned qi[TM_MAXM];
        for (unsigned i = 0; i < (unsigned)TM_MAXM; i++) {
            Ps[i] = {rand_fq(), rand_fq(), (rng() & 1) ? Fq::one() : rand_fq()};
            if (cnt == 3 && i == 1) Ps[i].Z = Fq::zero();
            Pp[i] = &Ps[i];
            neg[i] = (uint8_t)(rng() & 1);
            qi[i] = (unsigned)(rng() % nq);
            tp[i] = i < cnt ? tabs[qi[i]].data() : ident.data();
        }
        std::vector<uint32_t> desc(256);
        WTabMiller<LoopExec> wt{ex, lds.data(), tp, desc.data()};
        wt.run(Pp, neg, cnt, cnt ? cnt : 1);
        Fq12S want = Fq12S::one();
        for (unsigned i = 0; i < cnt; i++) {
            Fq12S f = miller_one(Ps[i], Qs[qi[i]]);
            if (neg[i]) f = f.unitary_inverse();
            want = fq12_mul(want, f);
        }
        CHECK(wt.result() == want, "wavefront table-driven miller product");
    }
    printf(fails ? "FAILED (%d)\n" : "PASS\n", fails);
    return fails ? 1 : 0;
}
