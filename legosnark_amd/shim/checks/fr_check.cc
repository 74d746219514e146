// fr_check -- the Fr kernels of the C-ABI against the reference's OWN templates.
//
// The O(2^d) field loops of the reference's provers are header code that lives IN the reference tree and compiles
// against the shim: MultiVPolyT::evalMLE (/root/reference/src/prototools/polytools.h:207-234), the witness recursion of
// CPPoly::prove (src/gadgets/poly.h:55-67), DPMle::pushRandomness / getMLEPoly (src/prototools/mle.h:199-226),
// DPBeta::compute_eq_tbl / pushRandomness / getBetaPoly (mle.h:36-53,74-82,93-105) and CPSumcheck::make_new_h_poly (src/gadgets/sumcheck.h:85-106).
// This program runs those unchanged functions on random inputs and compares every output with what lsa_fr_eval_mle,
// lsa_fr_cppoly_witness, lsa_fr_fold, lsa_fr_scale_upper, lsa_fr_eq_table and lsa_fr_sumcheck_round return for the same inputs --
// field elements byte for byte (both sides hold canonical Montgomery residues), the witness coefficients through the
// reference's own multiExpMA over DISTINCT random bases (the recursion's vector is a local of CPPoly::prove: its only
// observable outputs are pf.witness[i] / pf.witnessa[i]) and, for d <= 6, also term by term on the host.
//   usage: fr_check [max d = 16] [repetitions per shape = 2]
// Exit code = number of mismatching shapes; one JSON line of counts.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "globl.h"
#include "mle.h"
#include "poly.h"
#include "polytools.h"
#include "sumcheck.h"

using namespace std;

static bool same(const LFr &a, const LFr &b) { return memcmp(&a, &b, sizeof(LFr)) == 0; }
static Ins rnd(size_t n) {
    Ins v(n);
    for (auto &x : v) x = LFr::random_element();
    return v;
}
// a sprinkling of the values the kernels treat specially
static Ins rnd_with_edges(size_t n) {
    Ins v = rnd(n);
    if (n >= 8) { v[1] = LFr::zero(); v[2] = LFr::one(); v[3] = -LFr::one(); v[n - 1] = LFr::zero(); }
    return v;
}
#define LSA_OK_OR_DIE(call)                                                           \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, lsa_last_error()); exit(100); } \
    } while (0)

// a CommScheme whose bases are distinct random points: CPPoly::prove's multiExpMA(g1s, tmp_e) then pins tmp_e itself
// (with n copies of the generator, commit.h:134-138, it would only pin the sum of the coefficients)
struct RandBasesScheme : public CommScheme {
    void keygen(long _n) override {
        n = _n;
        g1s = cputil::simpleBatchExp<LG1, LFr>(LG1::one(), rnd((size_t)n));
        g2s.assign(1, LG2::one());
    }
};

int main(int argc, char **argv) {
    default_ec_pp::init_public_params();
    const size_t dmax = argc > 1 ? (size_t)atoi(argv[1]) : 16;
    const int reps = argc > 2 ? atoi(argv[2]) : 2;
    size_t bad = 0, n_eval = 0, n_wit = 0, n_fold = 0, n_scale = 0, n_round = 0, n_eq = 0;

    // ---- MultiVPolyT::evalMLE (polytools.h:207-234) vs lsa_fr_eval_mle
    for (size_t d = 1; d <= dmax; d++)
        for (int rep = 0; rep < reps; rep++) {
            const Ins v = rep ? rnd_with_edges(size_t(1) << d) : rnd(size_t(1) << d), r = rnd(d);
            const LFr want = MultiVPolyT::evalMLE(v, r);
            LFr got;
            LSA_OK_OR_DIE(lsa_fr_eval_mle(v.data(), d, r.data(), &got, 0));
            n_eval++;
            if (!same(want, got)) { bad++; fprintf(stderr, "evalMLE d=%zu mismatch\n", d); }
        }

    // ---- CPPoly::prove (poly.h:45-91) vs lsa_fr_cppoly_witness + the same multiExpMA
    {
        const size_t dw = dmax < 12 ? dmax : 12;
        RandBasesScheme scm;
        scm.keygen(long(1) << dw);
        CPPoly cppoly(&scm);
        const vector<LG1> g1s = scm.getBases1();
        for (size_t d = 1; d <= dw; d++) {
            const Ins v = rnd_with_edges(size_t(1) << d), r = rnd(d);
            PolyPf pf;
            CommOut unused;
            cppoly.prove(v, unused, r, pf);                        // the reference's recursion + ladder
            Ins w(size_t(1) << d);
            LSA_OK_OR_DIE(lsa_fr_cppoly_witness(v.data(), d, r.data(), w.data(), 0));
            size_t start = 0;
            bool ok = same(w.back(), LFr::zero());                  // the reference's w_coeffs[2^d - 1] is value-initialised and never written
            for (size_t i = 0; i < d; i++) {
                const size_t m = size_t(1) << (d - i - 1);
                const Ins seg(w.begin() + start, w.begin() + start + m);
                const LG1 mine = multiExpMA<LG1>(g1s, seg);
                ok = ok && mine == pf.witness[i] && (i == 0 || mine == pf.witnessa[i]);
                if (d <= 6) {                                       // and on the host, term by term
                    LG1 host = LG1::zero();
                    for (size_t p = 0; p < m; p++) host = host + seg[p] * g1s[p];
                    ok = ok && host == pf.witness[i];
                }
                start += m;
            }
            n_wit++;
            if (!ok) { bad++; fprintf(stderr, "CPPoly::prove witness d=%zu mismatch\n", d); }
        }
    }

    // ---- the sumcheck prover's dynamic-programming tables, round by round (sumcheck.cc:60-75):
    //      make_new_h_poly / getMLEPoly / getBetaPoly vs lsa_fr_sumcheck_round; DPMle::pushRandomness vs lsa_fr_fold;
    //      DPBeta::pushRandomness vs lsa_fr_scale_upper
    {
        CommScheme scm;
        scm.keygen(1);
        CPPoly cppoly(&scm);
        CPSumcheck sc(&scm, &cppoly);
        for (size_t d = 1; d <= dmax; d++)
            for (size_t m = 1; m <= 3; m++)
                for (int with_beta = 0; with_beta < 2; with_beta++) {
                    if (d > 12 && !(m == 2 && with_beta) && d != dmax) continue;      // the large sizes: the prover's own shape, and every shape once at dmax
                    const size_t n = size_t(1) << d;
                    const Ins rho = rnd(d);
                    shared_ptr<DPBeta> beta = with_beta ? make_shared<DPBeta>(d, rho) : static_pointer_cast<DPBeta>(make_shared<DPBetaDummy>());
                    vector<shared_ptr<DPMle>> mles;
                    vector<Ins> mine(m);                                 // this side's tables (host buffers handed to the library)
                    for (size_t t = 0; t < m; t++) {
                        mine[t] = t == 1 ? rnd_with_edges(n) : rnd(n);
                        mles.push_back(make_shared<DPMle>(d, n, mine[t]));
                    }
                    Ins suff = with_beta ? beta->beta_suff_rho_cur : Ins();   // after precomputeAll (mle.h:121-137)
                    bool ok = true;
                    if (with_beta && m == 1) {
                        // the table precomputeAll starts from (compute_eq_tbl, mle.h:93-105, run here on its own) and the first
                        // suffix table derived from it (mle.h:132-137) vs lsa_fr_eq_table(variant 0) + lsa_fr_scale_upper
                        Ins eq(n), tmp(n), got(n), first(n / 2);
                        DPBeta::compute_eq_tbl(d, eq, tmp, rho);
                        LSA_OK_OR_DIE(lsa_fr_eq_table(rho.data(), d, 0, got.data(), 0));
                        for (size_t p = 0; p < n; p++) ok = ok && same(eq[p], got[p]);
                        LSA_OK_OR_DIE(lsa_fr_scale_upper(got.data(), n / 2, &beta->rhoInvs[0], first.data(), 0));
                        for (size_t p = 0; p < n / 2; p++) ok = ok && same(beta->beta_suff_rho_cur[p], first[p]);
                        n_eq++;
                        if (!ok) fprintf(stderr, "eq table d=%zu mismatch\n", d);
                    }
                    for (size_t j = 0; j < d && ok; j++) {
                        const size_t half = size_t(1) << (d - j - 1);
                        const PolyT h = sc.make_new_h_poly(d, j, beta, mles);
                        const void *tabs[4];
                        for (size_t t = 0; t < m; t++) tabs[t] = mine[t].data();
                        const LFr pre = with_beta ? beta->getBetaPre(j - 1) : LFr::one();
                        const bool use_suff = with_beta && j + 1 <= d - 1;        // getBetaSuff(j + 1, p) is one() beyond (mle.h:57-62)
                        Ins got(m + 2, LFr::zero());
                        LSA_OK_OR_DIE(lsa_fr_sumcheck_round(use_suff ? suff.data() : nullptr, tabs, m, half, &pre, with_beta ? &rho[j] : nullptr, got.data(), 0));
                        const size_t have = with_beta ? m + 2 : m + 1;
                        for (size_t k = 0; k < h.vRepr.size() || k < have; k++) {
                            const LFr a = k < h.vRepr.size() ? h.vRepr[k] : LFr::zero(), b = k < have ? got[k] : LFr::zero();
                            ok = ok && same(a, b);
                        }
                        n_round++;
                        // the tables themselves: getVTable / getMLEPoly / getBetaPoly against this side's arrays
                        for (size_t t = 0; t < m && ok; t++)
                            for (size_t p = 0; p < half; p += (half > 64 ? half / 61 : 1)) {
                                const PolyT mp = mles[t]->getMLEPoly(j, p);
                                ok = ok && same(mles[t]->getVTable(j, p), mine[t][p]) && same(mp.vRepr[0], mine[t][p]) &&
                                     same(mp.vRepr[1], mine[t][p + half] - mine[t][p]);
                            }
                        if (with_beta)
                            for (size_t p = 0; p < half && ok; p += (half > 64 ? half / 61 : 1)) {
                                const PolyT bp = beta->getBetaPoly(j, p);
                                const LFr s = use_suff ? suff[p] : LFr::one(), f1 = LFr::one();
                                ok = ok && same(bp.vRepr[0], (f1 - rho[j]) * (pre * s)) && same(bp.vRepr[1], ((f1 + f1) * rho[j] - f1) * (pre * s));
                            }
                        if (j + 1 == d) break;
                        // next round (sumcheck.cc:69-74)
                        const LFr rj = LFr::random_element();
                        beta->pushRandomness(rj, j);
                        for (auto &mle : mles) mle->pushRandomness(rj, j);
                        for (size_t t = 0; t < m; t++) {
                            LSA_OK_OR_DIE(lsa_fr_fold(mine[t].data(), half, &rj, mine[t].data(), 0));
                            n_fold++;
                            for (size_t p = 0; p < half && ok; p++) ok = ok && same(mles[t]->getVTable(j + 1, p), mine[t][p]);
                        }
                        if (with_beta && j + 2 <= d - 1) {                         // mle.h:45-53: no update at the last steps
                            const size_t q = size_t(1) << (d - j - 2);
                            LSA_OK_OR_DIE(lsa_fr_scale_upper(suff.data(), q, &beta->rhoInvs[j + 1], suff.data(), 0));
                            n_scale++;
                            for (size_t p = 0; p < q && ok; p++) ok = ok && same(beta->beta_suff_rho_cur[p], suff[p]);
                        }
                    }
                    if (!ok) { bad++; fprintf(stderr, "sumcheck tables d=%zu m=%zu beta=%d mismatch\n", d, m, with_beta); }
                }
    }

    printf("{\"check\": \"fr kernels vs the reference's own templates\", \"max_d\": %zu, \"evalMLE\": %zu, \"cppoly_witness\": %zu, "
           "\"sumcheck_rounds\": %zu, \"folds\": %zu, \"suffix_updates\": %zu, \"eq_tables\": %zu, \"mismatching_shapes\": %zu}\n",
           dmax, n_eval, n_wit, n_round, n_fold, n_scale, n_eq, bad);
    return bad > 99 ? 99 : (int)bad;
}
