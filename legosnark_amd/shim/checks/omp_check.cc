// legosnark_amd/shim/checks/omp_check.cc -- the reference's -DMULTICORE=ON configuration through the shim
// (/root/reference/CMakeLists.txt:35-39,57-59,78-80: -fopenmp -DMULTICORE=1).  What that configuration changes:
//   * multiExpMA / simplesparsemexp pass chunks = omp_get_max_threads() (src/utils/globl.h:51-55,67-71,
//     src/utils/sparsemexp.cc:5-9,16-20) -- the GPU must return the SAME point whatever `chunks` says;
//   * the four `#pragma omp parallel for` loops of src/gadgets/lipmaa.cc:125-172 run this header's Fr operators on
//     several threads at once.
// This program is compiled by legosnark_amd/shim/Makefile in BOTH builds (with -fopenmp -DMULTICORE=1 into
// build/reference_mc, without into build/reference, where the pragmas are inert) against the reference's unchanged
// headers and liblegobasic.a, and checks:
//   1. multiExpMA<LG1> / <LG2> (the reference's own template; prints "NCHUNKS : <threads>") against libff::multi_exp
//      with chunks = 1 and against the host sum of k_i * P_i -- the Jacobian bytes after normalisation;
//   2. the lipmaa.cc loop bodies over shim Fr vectors in a parallel loop against the same bodies run serially;
//   3. random_element() drawn concurrently: canonical (below r), no duplicates between threads' pools;
//   4. host `Fr * G1`, `Fr * G2` (generator and other bases) concurrently against serial results;
//   5. copies of ONE deferred GT value evaluated from all threads at once: one GPU evaluation, identical bytes;
//   6. commitments (MSMs through the C-ABI) issued from several threads at once: serialised by the shim's lock,
//      each equal to its serial result.
// Prints one JSON line; exit status = number of failures.
#include <cstdio>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "globl.h"
#ifdef MULTICORE
#include <omp.h>
#endif

using namespace std;

static int fails = 0;
#define CHECK(c, msg) do { if (!(c)) { fprintf(stderr, "FAIL %s (line %d)\n", msg, __LINE__); fails++; } } while (0)

template <class G>
static bool same_point(G a, G b) {
    a.to_affine_coordinates();
    b.to_affine_coordinates();
    return (a.is_zero() && b.is_zero()) || (memcmp((const void *)&a, (const void *)&b, sizeof(G)) == 0);
}

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 3000;
    def_ec::init_public_params();
#ifdef MULTICORE
    const int threads = omp_get_max_threads();
    const bool multicore = true;
#else
    const int threads = 1;
    const bool multicore = false;
#endif
    // ---- 1. multiExpMA with chunks = threads
    {
        vector<LFr> xs(n);
        vector<LG1> g1(n);
        vector<LG2> g2(n / 8 + 3);
        for (auto &x : xs) x = LFr::random_element();
        xs[0] = LFr::zero();
        xs[1] = LFr::one();
        LG1 p = LFr::random_element() * LG1::one();
        for (size_t i = 0; i < n; i++) { g1[i] = p; p = p + LG1::one(); }
        g1[2] = LG1::zero();
        LG2 q = LFr::random_element() * LG2::one();
        for (auto &e : g2) { e = q; q = q.dbl() + LG2::one(); }
        const LG1 a1 = multiExpMA<LG1>(g1, xs);
        const LG1 b1 = libff::multi_exp<LG1, LFr, libff::multi_exp_method_BDLO12>(g1.begin(), g1.end(), xs.begin(), xs.end(), 1);
        CHECK(same_point(a1, b1), "multiExpMA<G1>(chunks = threads) == multi_exp(chunks = 1)");
        vector<LFr> xs2(xs.begin(), xs.begin() + g2.size());
        const LG2 a2 = multiExpMA<LG2>(g2, xs2);
        const LG2 b2 = libff::multi_exp<LG2, LFr, libff::multi_exp_method_BDLO12>(g2.begin(), g2.end(), xs2.begin(), xs2.end(), 1);
        CHECK(same_point(a2, b2), "multiExpMA<G2>(chunks = threads) == multi_exp(chunks = 1)");
        LG2 h2 = LG2::zero();
        for (size_t i = 0; i < g2.size(); i++) h2 = h2 + xs2[i] * g2[i];
        CHECK(same_point(a2, h2), "multiExpMA<G2> == host sum");
        const size_t m = n < 200 ? n : 200;
        vector<LG1> gs(g1.begin(), g1.begin() + m);
        vector<LFr> ss(xs.begin(), xs.begin() + m);
        LG1 h1 = LG1::zero();
        for (size_t i = 0; i < m; i++) h1 = h1 + ss[i] * gs[i];
        CHECK(same_point(multiExpMA<LG1>(gs, ss), h1), "multiExpMA<G1> == host sum");
    }
    // ---- 2. the loop bodies of lipmaa.cc:125-172 over Fr
    {
        const size_t m = 1 << 16;
        vector<LFr> aA(m), aB(m), aC(m), H(m), Hs(m), acc(m), accs(m);
        for (size_t i = 0; i < m; i++) { aA[i] = LFr::random_element(); aB[i] = LFr::random_element(); aC[i] = LFr::random_element(); }
        const LFr d1 = LFr::random_element(), d2 = LFr::random_element();
        for (size_t i = 0; i < m; i++) { accs[i] = d2 * aA[i] + d1 * aB[i]; Hs[i] = aA[i] * aB[i]; Hs[i] = (Hs[i] - aC[i]); accs[i] += Hs[i]; }
#ifdef MULTICORE
#pragma omp parallel for
#endif
        for (size_t i = 0; i < m; ++i) { acc[i] = d2 * aA[i] + d1 * aB[i]; }
#ifdef MULTICORE
#pragma omp parallel for
#endif
        for (size_t i = 0; i < m; ++i) { H[i] = aA[i] * aB[i]; }
#ifdef MULTICORE
#pragma omp parallel for
#endif
        for (size_t i = 0; i < m; ++i) { H[i] = (H[i] - aC[i]); }
#ifdef MULTICORE
#pragma omp parallel for
#endif
        for (size_t i = 0; i < m; ++i) { acc[i] += H[i]; }
        CHECK(memcmp((const void *)acc.data(), (const void *)accs.data(), m * sizeof(LFr)) == 0, "lipmaa.cc loop bodies: parallel == serial");
        // inverse() and the decimal constructor are used by the same file's set-up code
        vector<LFr> inv(4096), invs(4096);
        for (size_t i = 0; i < inv.size(); i++) invs[i] = aA[i].inverse() * LFr("12345678901234567890");
#ifdef MULTICORE
#pragma omp parallel for
#endif
        for (size_t i = 0; i < inv.size(); ++i) inv[i] = aA[i].inverse() * LFr("12345678901234567890");
        CHECK(memcmp((const void *)inv.data(), (const void *)invs.data(), inv.size() * sizeof(LFr)) == 0, "Fr::inverse: parallel == serial");
    }
    // ---- 3. random_element() from every thread
    {
        const size_t m = 1 << 14;
        vector<LFr> r(m);
#ifdef MULTICORE
#pragma omp parallel for
#endif
        for (size_t i = 0; i < m; ++i) r[i] = LFr::random_element();
        set<string> seen;
        bool canonical = true;
        for (auto &x : r) {
            seen.insert(string((const char *)&x, sizeof x));
            canonical = canonical && (x + LFr::zero() == x) && Fr<def_ec>(x.as_bigint()) == x;
        }
        CHECK(seen.size() == m, "random_element: concurrent draws are distinct");
        CHECK(canonical, "random_element: canonical residues");
    }
    // ---- 4. host scalar multiplications
    {
        const size_t m = 256;
        vector<LFr> k(m);
        for (auto &x : k) x = LFr::random_element();
        k[0] = LFr::zero(); k[1] = LFr::one(); k[2] = -LFr::one(); k[3] = LFr(65535); k[4] = LFr(65536);
        const LG1 B1 = LFr::random_element() * LG1::one();
        const LG2 B2 = LFr::random_element() * LG2::one();
        vector<LG1> s1(m), p1(m), sb1(m), pb1(m);
        vector<LG2> s2(m), p2(m), sb2(m), pb2(m);
        for (size_t i = 0; i < m; i++) { s1[i] = k[i] * LG1::one(); s2[i] = k[i] * LG2::one(); sb1[i] = k[i] * B1; sb2[i] = k[i] * B2; }
#ifdef MULTICORE
#pragma omp parallel for
#endif
        for (size_t i = 0; i < m; ++i) { p1[i] = k[i] * LG1::one(); p2[i] = k[i] * LG2::one(); pb1[i] = k[i] * B1; pb2[i] = k[i] * B2; }
        bool ok = true;
        for (size_t i = 0; i < m; i++) ok = ok && same_point(s1[i], p1[i]) && same_point(s2[i], p2[i]) && same_point(sb1[i], pb1[i]) && same_point(sb2[i], pb2[i]);
        CHECK(ok, "Fr * G1 / Fr * G2 on the host: parallel == serial");
        // against the group law directly: (k + 1) G = k G + G
        ok = true;
        for (size_t i = 5; i < 40; i++) ok = ok && same_point((k[i] + LFr::one()) * LG1::one(), p1[i] + LG1::one()) && same_point((k[i] + LFr::one()) * LG2::one(), p2[i] + LG2::one());
        CHECK(ok, "Fr * generator: (k + 1) G == k G + G");
    }
    // ---- 5. copies of one deferred GT value looked at from every thread
    {
        const LFr a = LFr::random_element(), b = LFr::random_element();
        const auto P = def_ec::precompute_G1(a * LG1::one());
        const auto Q = def_ec::precompute_G2(b * LG2::one());
        const auto P2 = def_ec::precompute_G1((a * b) * LG1::one());
        const auto Q2 = def_ec::precompute_G2(LG2::one());
        const auto lhs = def_ec::miller_loop(P, Q), rhs = def_ec::miller_loop(P2, Q2);
        const LGT v = def_ec::final_exponentiation(lhs * rhs.unitary_inverse());
        const int copies = 32;
        vector<LGT> c(copies, v);
        vector<int> is_one(copies, 0);
#ifdef MULTICORE
#pragma omp parallel for
#endif
        for (int i = 0; i < copies; ++i) is_one[i] = (c[i] == LGT::one()) ? 1 : 0;
        bool ok = true;
        for (int i = 0; i < copies; i++) ok = ok && is_one[i] == 1 && memcmp((const void *)&c[i].val(), (const void *)&v.val(), sizeof(lsa::Fq12)) == 0;
        CHECK(ok, "one deferred GT value, many readers: e(aG, bH) / e(abG, H) == 1 for every copy");
    }
    // ---- 6. MSMs issued from several threads (not something the reference does; the shim serialises them)
    {
        const int calls = 16;
        const size_t m = 600;
        vector<vector<LG1>> bases(calls, vector<LG1>(m));
        vector<vector<LFr>> sc(calls, vector<LFr>(m));
        vector<LG1> serial(calls), par(calls);
        LG1 p = LFr::random_element() * LG1::one();
        for (int c = 0; c < calls; c++)
            for (size_t i = 0; i < m; i++) { bases[c][i] = p; p = p + LG1::one(); sc[c][i] = LFr::random_element(); }
        for (int c = 0; c < calls; c++) serial[c] = libff::multi_exp<LG1, LFr, libff::multi_exp_method_BDLO12>(bases[c].begin(), bases[c].end(), sc[c].begin(), sc[c].end(), 1);
#ifdef MULTICORE
#pragma omp parallel for
#endif
        for (int c = 0; c < calls; ++c) par[c] = libff::multi_exp<LG1, LFr, libff::multi_exp_method_BDLO12>(bases[c].begin(), bases[c].end(), sc[c].begin(), sc[c].end(), (size_t)threads);
        bool ok = true;
        for (int c = 0; c < calls; c++) ok = ok && same_point(serial[c], par[c]);
        CHECK(ok, "multi_exp from several threads at once == serial");
    }
    printf("{\"omp_check\": {\"multicore\": %s, \"threads\": %d, \"n\": %zu, \"failures\": %d}}\n", multicore ? "true" : "false", threads, n, fails);
    return fails;
}
