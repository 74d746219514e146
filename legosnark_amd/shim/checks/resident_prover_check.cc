// resident_prover_check -- CPHad's prover with its vectors resident on the device, against the reference's own prover.
//
// The unchanged CPHad::prove (/root/reference/src/gadgets/hadamardsc.cc:54-98 -> CPSumcheck::prove, sumcheck.cc:12-125,
// CPPoly::prove, poly.h:45-91) keeps a, b, c = a o b in host vectors: its O(2^d) field loops run on one host core and every
// multiExpMA re-uploads its scalars.  resident_prove() below is the same protocol written against the C-ABI: the three
// vectors are uploaded once, and per proof
//   * MultiVPolyT::evalMLE                      -> lsa_fr_eval_mle          (3 x)
//   * CPPoly::prove's recursion + MSM ladder    -> lsa_fr_cppoly_witness, lsa_msm_run_async / lsa_msm_run_segments_async (3 x)
//   * DPBeta::precomputeAll                     -> lsa_fr_eq_table (variant 0) + lsa_fr_scale_upper
//   * make_new_h_poly, per round                -> lsa_fr_sumcheck_round    (coefficients come back: 4 Fr)
//   * DPMle / DPBeta::pushRandomness            -> lsa_fr_fold, lsa_fr_scale_upper
//   * CommScheme::commit of a and b             -> lsa_commit_run_async     (G1 + G2 over one sort)
// while the O(d) host work of the protocol -- commitments to single field elements, the sigma proofs -- is the reference's own
// code (CommScheme::commit(In), PolyT::commit, ZKEqProof, ZKPrdProof), called in the reference's order and overlapped with
// the MSMs the device is still running; PolyT::evalAsPolyOn's linear combinations of commitments are evaluated in the
// exponent (the same group elements: the comparison below would say otherwise).
//
// Both provers are fed the same inputs and the same randomness: this program is built with -DLSA_SHIM_TEST_SEED (the
// shim's random_element then reads a seeded generator, which is re-seeded before either run), so every proof element can be
// compared: r, witness / witnessa, uProdEvalCm, and in the sumcheck proof r, hCom, the d equality proofs, polycm_g,
// polycm_evalg, both poly proofs and the product proof.  The resident proof is then handed to the reference's unchanged
// CPHad::verify.
//   usage: resident_prover_check [d = 12] [inputs: squares (the hadamard example's u[i] = i, c = i^2) | random]
// One JSON line; exit code 0 iff every element is equal and the verifier accepts.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "commit.h"
#include "globl.h"
#include "hadamardsc.h"

#ifndef LSA_SHIM_TEST_SEED
#error "build with -DLSA_SHIM_TEST_SEED: both provers must read the same random stream"
#endif

using namespace std;
using Clock = chrono::steady_clock;
static double ms_since(Clock::time_point t0) { return chrono::duration<double, milli>(Clock::now() - t0).count(); }

#define HIP_OR_DIE(call)                                                                                          \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); exit(100); }           \
    } while (0)
#define LSA_OR_DIE(call)                                                                                          \
    do {                                                                                                          \
        int rc_ = (call);                                                                                         \
        if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, lsa_last_error()); exit(100); }             \
    } while (0)

struct DevBuf {
    char *p = nullptr;
    explicit DevBuf(size_t bytes) { HIP_OR_DIE(hipMalloc((void **)&p, bytes ? bytes : 32)); }
    ~DevBuf() { (void)hipFree(p); }
    DevBuf(const DevBuf &) = delete;
    char *fr(size_t i) const { return p + 32 * i; }
    char *g1(size_t i) const { return p + 96 * i; }
};
static void to_device(const DevBuf &d, const void *h, size_t bytes) { HIP_OR_DIE(hipMemcpy(d.p, h, bytes, hipMemcpyHostToDevice)); }
static void to_host(void *h, const char *d, size_t bytes) { HIP_OR_DIE(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); }

// the CommScheme's bases (commit.h:128-138: n copies of the generators) as resident handles, and the prover's three vectors
struct ResidentState {
    size_t d, n;
    lsa_bases *g1 = nullptr, *g2 = nullptr;
    DevBuf a, b, c;
    // a prover's workspace, allocated once like the vectors (hipMalloc / hipFree inside a proof would each cost a device-wide wait)
    DevBuf rho, rsc, vals, eq, suff, ta, tb, w_c, w_a, w_b, wit_c, wit_a, wit_b, cm, kcm;
    ResidentState(size_t _d)
        : d(_d), n(size_t(1) << _d), a(32 * n), b(32 * n), c(32 * n), rho(32 * d), rsc(32 * d), vals(32 * 4), eq(32 * n), suff(32 * (n / 2)), ta(32 * (n / 2)),
          tb(32 * (n / 2)), w_c(32 * n), w_a(32 * n), w_b(32 * n), wit_c(96 * d), wit_a(96 * d), wit_b(96 * d), cm(96 * 2), kcm(192 * 2) {}
};

// CPPoly::prove (poly.h:45-91) on a resident vector: the recursion, then the ladder witness[i] = MSM(g1s, w[start_i ..
// start_i + 2^(d-1-i))) -- the long rungs one call each, the rest as one segmented call.  Asynchronous after the recursion.
static void resident_cppoly_prove(const ResidentState &st, const char *d_v, const char *d_point, const DevBuf &d_w, const DevBuf &d_wit) {
    const size_t d = st.d;
    LSA_OR_DIE(lsa_fr_cppoly_witness(d_v, d, d_point, d_w.p, 1));
    vector<uint64_t> starts(d + 1, 0);
    for (size_t i = 0; i < d; i++) starts[i + 1] = starts[i] + (uint64_t(1) << (d - 1 - i));
    size_t first_small = 0;
    while (first_small < d && (size_t(1) << (d - 1 - first_small)) > (size_t(1) << 15)) first_small++;
    if (!lsa_bases_has_table(st.g1)) first_small = d;
    for (size_t i = 0; i < first_small; i++)
        LSA_OR_DIE(lsa_msm_run_async(st.g1, 0, d_w.fr(starts[i]), size_t(1) << (d - 1 - i), d_wit.g1(i)));
    if (first_small < d)
        LSA_OR_DIE(lsa_msm_run_segments_async(st.g1, 0, d_w.p, starts.data() + first_small, d - first_small, d_wit.g1(first_small)));
}
static PolyPf poly_pf_from(const DevBuf &d_wit, size_t d) {
    PolyPf pf;
    pf.witness.resize(d);
    pf.witnessa.resize(d);
    to_host(pf.witness.data(), d_wit.p, 96 * d);
    for (size_t i = 1; i < d; i++) pf.witnessa[i] = pf.witness[i];        // poly.h:84-86: the same sum, issued twice upstream
    return pf;
}

// CPHad::prove with resident vectors.  `in` is the reference's prover input (commitments to c, a, b: hadamard.cc:88-90).
static HadPf *resident_prove(const ResidentState &st, CommScheme *scm, const CPPIn &in, double *ms_device_wait) {
    const size_t d = st.d, n = st.n;
    HadPf *pf = new HadPf((long)d);
    for (size_t i = 0; i < d; i++) pf->r[i] = CommRand::random_element();                        // hadamardsc.cc:84-86
    const Ins &rho = pf->r;
    const DevBuf &d_rho = st.rho, &d_rsc = st.rsc, &d_one = st.vals;
    to_device(d_rho, rho.data(), 32 * d);

    // ---- cppolyProve for c (hadamardsc.cc:33-38): evaluation, its commitment (one draw); the MSM work is issued below
    LFr ans_c;
    LSA_OR_DIE(lsa_fr_eval_mle(st.c.p, d, d_rho.p, d_one.p, 1));
    to_host(&ans_c, d_one.p, 32);
    CommOut uProdEvalCmOut = scm->commit(ans_c);

    // ---- CPSumcheck::prove (sumcheck.cc:12-125) with rho = pf->r, y = ans_c
    SumcheckRand r(d);
    for (size_t i = 0; i < d; i++) r[i] = CommRand::random_element();                            // sumcheck.cc:46-48
    to_device(d_rsc, r.data(), 32 * d);
    // DPBeta (mle.h:21-25,121-137): inverses of rho, the eq table as the reference builds it, the first suffix table
    Ins rhoInvs(d);
    for (size_t i = 0; i < d; i++) rhoInvs[i] = rho[i].inverse();
    const DevBuf &d_eq = st.eq, &d_suff = st.suff, &d_ta = st.ta, &d_tb = st.tb;
    LSA_OR_DIE(lsa_fr_eq_table(d_rho.p, d, 0, d_eq.p, 1));
    LSA_OR_DIE(lsa_fr_scale_upper(d_eq.p, n / 2, &rhoInvs[0], d_suff.p, 1));
    vector<PolyT> h(d);
    Scalars z(d + 1);
    z[0] = ans_c;
    LFr pre = LFr::one();
    for (size_t i = 0; i < d; i++) {
        const size_t half = size_t(1) << (d - i - 1);
        const void *tabs[2] = {i ? d_ta.p : st.a.p, i ? d_tb.p : st.b.p};
        Scalars coeffs(4);
        LSA_OR_DIE(lsa_fr_sumcheck_round(i + 1 <= d - 1 ? d_suff.p : nullptr, tabs, 2, half, &pre, &rho[i], coeffs.data(), 1));
        h[i] = PolyT(coeffs);
        z[i + 1] = h[i].eval(r[i]);
        if (i + 1 < d) {                                                                         // sumcheck.cc:69-74
            pre = pre * eqbit(r[i], rho[i]);                                                     // mle.h:42
            if (i + 2 < d) LSA_OR_DIE(lsa_fr_scale_upper(d_suff.p, size_t(1) << (d - i - 2), &rhoInvs[i + 1], d_suff.p, 1));   // mle.h:45-53
            LSA_OR_DIE(lsa_fr_fold(tabs[0], half, d_rsc.fr(i), d_ta.p, 1));                      // mle.h:199-210; round 0 leaves a, b intact
            LSA_OR_DIE(lsa_fr_fold(tabs[1], half, d_rsc.fr(i), d_tb.p, 1));
        }
    }
    // the evaluations of a and b at r (sumcheck.cc:104: computeAnswer), needed on the host further down
    LFr ans_ab[2];
    for (int t = 0; t < 2; t++) {
        LSA_OR_DIE(lsa_fr_eval_mle(t ? st.b.p : st.a.p, d, d_rsc.p, d_one.fr(1 + t), 1));
        to_host(&ans_ab[t], d_one.fr(1 + t), 32);
    }

    // ---- everything that is an MSM, queued now and running while the host does the sigma protocols:
    //      CPPoly::prove for c at rho, for a and b at r (poly.h:45-91); commitPoly of a and b (poly.h:30-32 -> commit.h:149-158)
    const DevBuf &d_w = st.w_c, &d_w2 = st.w_a, &d_w3 = st.w_b, &d_wit_c = st.wit_c, &d_wit_a = st.wit_a, &d_wit_b = st.wit_b, &d_cm = st.cm, &d_kcm = st.kcm;
    resident_cppoly_prove(st, st.c.p, d_rho.p, d_w, d_wit_c);
    resident_cppoly_prove(st, st.a.p, d_rsc.p, d_w2, d_wit_a);
    resident_cppoly_prove(st, st.b.p, d_rsc.p, d_w3, d_wit_b);
    LSA_OR_DIE(lsa_commit_run_async(st.g1, st.g2, st.a.p, n, d_cm.p, d_kcm.p));
    LSA_OR_DIE(lsa_commit_run_async(st.g1, st.g2, st.b.p, n, d_cm.p + 96, d_kcm.p + 192));

    // ---- the host side of sumcheck.cc:77-111, in its order (every commit(In) and every sigma proof draws randomness)
    vector<CommOuts> hComOut(d);
    for (size_t i = 0; i < d; i++) hComOut[i] = h[i].commit(scm);
    SumcheckPf::EqProofs eqPfs(d);
    CommOuts zComOut(d + 1);
    zComOut[0] = uProdEvalCmOut;
    // PolyT::evalAsPolyOn(hComOut[i], pt) (polytools.h:99-109) is sum_k pt^k * commit(h_k): with commit(x) = (x G1, x G2; r_x; x)
    // (commit.h:160-167) that is the commitment (h(pt) G1, h(pt) G2; sum_k pt^k r_k; h(pt)) -- the same group elements from two
    // fixed-base products instead of six variable-base ones per evaluation (60 evaluations: 20 ms of one host core at d = 20)
    auto eval_at = [](const PolyT &hp, const CommOuts &hc, const In &pt) {
        In val = In::zero(), rr = In::zero(), pw = In::one();
        for (size_t k = 0; k < hc.size(); k++) { val = val + pw * hp.vRepr[k]; rr = rr + pw * hc[k].r; pw = pw * pt; }
        return CommOut(Comm(val * LG1::one(), val * LG2::one()), rr, val);
    };
    for (size_t i = 0; i < d; i++) {
        const CommOut at0 = eval_at(h[i], hComOut[i], In::zero()), at1 = eval_at(h[i], hComOut[i], In::one());
        eqPfs[i] = make_shared<ZKEqProof>(scm, at0 + at1, zComOut[i]);
        zComOut[i + 1] = eval_at(h[i], hComOut[i], r[i]);
    }
    CommOuts cmout_eval(2);
    for (int t = 0; t < 2; t++) cmout_eval[t] = scm->commit(ans_ab[t]);
    const In betaEval = evalBetaOnPoint(rho, r);
    const CommOut lhsProd = cmout_eval[0] * betaEval;
    ZKPrdProof prdPf(scm, lhsProd, cmout_eval[1], zComOut[d]);

    // ---- collect what the device computed meanwhile
    const auto t_wait = Clock::now();
    LSA_OR_DIE(lsa_synchronize());
    *ms_device_wait = ms_since(t_wait);
    pf->polyProof = poly_pf_from(d_wit_c, d);
    pf->uProdEvalCm = uProdEvalCmOut.c;
    vector<PolyPf> polypf = {poly_pf_from(d_wit_a, d), poly_pf_from(d_wit_b, d)};
    LG1 cm[2];
    LG2 kcm[2];
    to_host(cm, d_cm.p, sizeof cm);
    to_host(kcm, d_kcm.p, sizeof kcm);
    vector<Comms> hCom(d);
    for (size_t i = 0; i < d; i++) hCom[i] = CommOut::toComms(hComOut[i]);
    Comms polycm_g = {Comm(cm[0], kcm[0]), Comm(cm[1], kcm[1])};
    pf->sumcheckPf = new SumcheckPf(r, hCom, eqPfs, polycm_g, CommOut::toComms({lhsProd, cmout_eval[1]}), polypf, prdPf);
    return pf;
}

// ---- proof comparison, element by element
static size_t g_diff = 0;
static void expect(bool ok, const char *what, long i = -1) {
    if (ok) return;
    if (g_diff++ < 12) fprintf(stderr, "proof element differs: %s[%ld]\n", what, i);
}
static void same_comm(const Comm &x, const Comm &y, const char *what, long i) { expect(x.c == y.c && x.kc == y.kc, what, i); }
static void same_polypf(const PolyPf &x, const PolyPf &y, const char *what) {
    expect(x.witness.size() == y.witness.size() && x.witnessa.size() == y.witnessa.size(), what);
    for (size_t i = 0; i < x.witness.size() && i < y.witness.size(); i++) expect(x.witness[i] == y.witness[i] && x.witnessa[i] == y.witnessa[i], what, (long)i);
}
static size_t compare_proofs(const HadPf *x, const HadPf *y) {
    g_diff = 0;
    expect(x->r == y->r, "HadPf.r");
    same_polypf(x->polyProof, y->polyProof, "HadPf.polyProof");
    same_comm(x->uProdEvalCm, y->uProdEvalCm, "HadPf.uProdEvalCm", -1);
    const SumcheckPf *s = x->sumcheckPf, *t = y->sumcheckPf;
    expect(s->r == t->r, "SumcheckPf.r");
    expect(s->hCom.size() == t->hCom.size(), "SumcheckPf.hCom.size");
    for (size_t i = 0; i < s->hCom.size() && i < t->hCom.size(); i++) {
        expect(s->hCom[i].size() == t->hCom[i].size(), "SumcheckPf.hCom[i].size", (long)i);
        for (size_t k = 0; k < s->hCom[i].size() && k < t->hCom[i].size(); k++) same_comm(s->hCom[i][k], t->hCom[i][k], "SumcheckPf.hCom", (long)(4 * i + k));
    }
    expect(s->eqPfs.size() == t->eqPfs.size(), "SumcheckPf.eqPfs.size");
    for (size_t i = 0; i < s->eqPfs.size() && i < t->eqPfs.size(); i++) {
        const ZKEqProof &p = *s->eqPfs[i], &q = *t->eqPfs[i];
        expect(p.c == q.c && p.z == q.z && p.a == q.a && p.c0 == q.c0 && p.c1 == q.c1, "SumcheckPf.eqPfs", (long)i);
    }
    for (int i = 0; i < 2; i++) {
        same_comm(s->polycm_g[i], t->polycm_g[i], "SumcheckPf.polycm_g", i);
        same_comm(s->polycm_evalg[i], t->polycm_evalg[i], "SumcheckPf.polycm_evalg", i);
        same_polypf(s->polypf_g[i], t->polypf_g[i], i ? "SumcheckPf.polypf_g[1]" : "SumcheckPf.polypf_g[0]");
    }
    const ZKPrdProof &p = s->finalPrdPf, &q = t->finalPrdPf;
    expect(p.c == q.c && p.z1 == q.z1 && p.z2 == q.z2 && p.z3 == q.z3 && p.z4 == q.z4 && p.z5 == q.z5 && p.c0 == q.c0 && p.c1 == q.c1 && p.cPrd == q.cPrd,
           "SumcheckPf.finalPrdPf");
    return g_diff;
}

int main(int argc, char **argv) {
    default_ec_pp::init_public_params();
    const size_t d = argc > 1 ? (size_t)atoi(argv[1]) : 12;
    const bool random_inputs = argc > 2 && !strcmp(argv[2], "random");
    const size_t n = size_t(1) << d;
    const uint64_t seed = getenv("LSA_SEED") ? strtoull(getenv("LSA_SEED"), nullptr, 0) : 20261003ull;

    Ins a(n), b(n), c(n);
    for (size_t i = 0; i < n; i++) {
        if (random_inputs) { a[i] = LFr::random_element(); b[i] = LFr::random_element(); }
        else a[i] = b[i] = LFr::one() * (long)i;                                                  // hadamard.cc:130-135
        c[i] = a[i] * b[i];
    }
    // the reference's set-up (hadamard.cc:82-96)
    CommScheme *scm = new CommScheme;
    scm->keygen((long)n);
    CPPIn proverInput;
    CPVIn verifInput;
    CPInputFmt::init_no_pub(proverInput, verifInput, scm, {c, a, b});
    CPHad had(scm, new CPPoly(scm));
    auto pBm = make_shared<Benchmark>();
    had.setBenchmark(pBm, "CPHadSumcheck");
    HadKey *crs = had.keygen(new HadRel(n));

    // the resident side's set-up: the bases as handles, the three vectors uploaded once
    auto t0 = Clock::now();
    ResidentState st(d);
    {
        const vector<LG1> g1s = scm->getBases1();
        const vector<LG2> g2s(n, LG2::one());                                                     // commit.h:137-138
        LSA_OR_DIE(lsa_g1_bases_create(g1s.data(), n, 0, &st.g1));
        LSA_OR_DIE(lsa_g2_bases_create(g2s.data(), n, 0, &st.g2));
    }
    const double ms_handles = ms_since(t0);
    t0 = Clock::now();
    to_device(st.a, a.data(), 32 * n);
    to_device(st.b, b.data(), 32 * n);
    to_device(st.c, c.data(), 32 * n);
    const double ms_upload = ms_since(t0);

    // the unchanged prover
    lsa_test_reseed(seed);
    t0 = Clock::now();
    HadPf *ref = had.prove(crs, proverInput);
    const double ms_ref = ms_since(t0);

    // the resident prover on the same random stream (twice: the second run has warm workspaces, like any later proof)
    double ms_res[2], ms_wait[2];
    HadPf *mine = nullptr;
    for (int rep = 0; rep < 2; rep++) {
        lsa_test_reseed(seed);
        t0 = Clock::now();
        mine = resident_prove(st, scm, proverInput, &ms_wait[rep]);
        ms_res[rep] = ms_since(t0);
    }
    const size_t diffs = compare_proofs(ref, mine);
    const bool accepted = had.verify(crs, verifInput, mine);

    printf("{\"resident_prover\": {\"d\": %zu, \"inputs\": \"%s\", \"proof_equal\": %s, \"differing_elements\": %zu, \"reference_verifier_accepts\": %s, "
           "\"reference_prove_ms\": %.2f, \"prove_ms\": %.2f, \"prove_ms_first\": %.2f, \"of_which_waiting_for_the_device_ms\": %.2f, "
           "\"handles_ms\": %.1f, \"upload_ms\": %.2f}}\n",
           d, random_inputs ? "random" : "squares", diffs == 0 ? "true" : "false", diffs, accepted ? "true" : "false", ms_ref, ms_res[1], ms_res[0], ms_wait[1],
           ms_handles, ms_upload);
    lsa_bases_destroy(st.g1);
    lsa_bases_destroy(st.g2);
    return diffs == 0 && accepted ? 0 : 1;
}
