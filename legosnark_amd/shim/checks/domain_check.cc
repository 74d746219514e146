// legosnark_amd/shim/checks/domain_check.cc -- evaluation domains whose size is not a power of two, through the
// reference's own code: libfqfft's step radix-2 domain (shim/libfqfft/evaluation_domain/get_evaluation_domain.hpp over
// lsa_fr_ntt_step) must (1) transform between coefficients and values at its points, (2) carry the Lipmaa Hadamard
// gadget (src/gadgets/lipmaa.cc: keygen over the domain's Lagrange basis, prove with iFFT / cosetFFT / icosetFFT and
// divide_by_Z_on_coset, verify) at n = 3 * 2^k exactly as the power-of-two domains do: an honest proof is accepted, a
// proof for a wrong product is rejected (n = 8 runs beside them as the control).
// Prints one line per check and a final JSON line; exit status = number of failures.
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "globl.h"
#include "commit.h"
#include "lipmaa.h"
#include "benchmark.h"

static int fails = 0;
static void check(bool ok, const std::string &what) { printf("%s %s\n", ok ? "ok  " : "FAIL", what.c_str()); if (!ok) fails++; }

static LFr horner(const std::vector<LFr> &c, const LFr &x) {
    LFr acc = LFr::zero();
    for (size_t i = c.size(); i-- > 0;) acc = acc * x + c[i];
    return acc;
}

static void transforms(size_t m) {
    auto dom = libfqfft::get_evaluation_domain<LFr>(m);
    const std::string tag = "m = " + std::to_string(m) + ": ";
    check(dom->m == m, tag + "a domain of exactly this size");
    std::vector<LFr> a(m);
    for (auto &x : a) x = LFr::random_element();
    std::vector<LFr> v = a;
    dom->FFT(v);
    bool ok = true;
    for (size_t k = 0; k < m; k += (m > 256 ? 37 : 1)) ok = ok && v[k] == horner(a, dom->get_domain_element(k));
    ok = ok && v[m - 1] == horner(a, dom->get_domain_element(m - 1));
    check(ok, tag + "FFT gives the values at get_domain_element(k)");
    dom->iFFT(v);
    check(v == a, tag + "iFFT(FFT(a)) == a");
    const LFr g = LFr::multiplicative_generator;
    v = a;
    dom->cosetFFT(v, g);
    ok = true;
    for (size_t k = 0; k < m; k += (m > 256 ? 41 : 1)) ok = ok && v[k] == horner(a, g * dom->get_domain_element(k));
    check(ok, tag + "cosetFFT gives the values at g * get_domain_element(k)");
    dom->icosetFFT(v, g);
    check(v == a, tag + "icosetFFT(cosetFFT(a)) == a");
    // the Lagrange coefficients at a random point interpolate the values
    std::vector<LFr> vals = a;
    dom->FFT(vals);
    const LFr t = LFr::random_element();
    const std::vector<LFr> L = dom->evaluate_all_lagrange_polynomials(t);
    LFr acc = LFr::zero();
    for (size_t k = 0; k < m; k++) acc += L[k] * vals[k];
    check(acc == horner(a, t), tag + "sum L_k(t) FFT(a)[k] == a(t)");
}

static void lipmaa(size_t n) {
    const std::string tag = "Lipmaa Hadamard, n = " + std::to_string(n) + ": ";
    // (CPHadL::prove reads its FIRST input for both factors -- src/gadgets/lipmaa.cc:115-119 sets aA and aB from aPts --
    // so the statement it can prove honestly is a o a = c, with the one commitment to a on both sides)
    Ins a(n), c(n);
    for (size_t i = 0; i < n; i++) { a[i] = In::random_element(); c[i] = a[i] * a[i]; }
    InterpCommScheme ics;
    CPHadL cphadl;
    auto pBm = std::make_shared<Benchmark>();
    cphadl.setBenchmark(pBm, "CPHadLipmaa");
    auto interp = new Interpolator(n);
    const IScalar chi = IScalar::random_element(), gamma = IScalar::random_element();
    ics.keygen(n, *interp, chi, gamma);
    cphadl.keygen(n, *interp, chi, gamma);
    auto cmOuta = ics.commit(a), cmOutb = cmOuta, cmOutc = ics.commit(c);
    auto pf = cphadl.prove(cmOuta, cmOutb, cmOutc);
    check(cphadl.verify(pf, cmOuta.c, cmOutb.c, cmOutc.c), tag + "an honest proof is accepted");
    Ins c2 = c;
    c2[n / 3] += In::one();
    auto cmOutc2 = ics.commit(c2);
    auto pf2 = cphadl.prove(cmOuta, cmOutb, cmOutc2);
    check(!cphadl.verify(pf2, cmOuta.c, cmOutb.c, cmOutc2.c), tag + "a proof for a wrong product is rejected");
    check(!cphadl.verify(pf, cmOuta.c, cmOutb.c, cmOutc2.c), tag + "the honest proof does not carry over to another commitment");
    delete interp;
}

int main(int argc, char **argv) {
    default_ec_pp::init_public_params();
    for (size_t m : {size_t(3), size_t(6), size_t(12), size_t(96), size_t(768), size_t(5120), size_t((1u << 16) + (1u << 11))}) transforms(m);
    {
        auto d = libfqfft::get_evaluation_domain<LFr>(700);
        check(d->m == 768, "get_evaluation_domain(700): 512 + 188 rounds to the step domain of 512 + 256");
    }
    lipmaa(8);                                             // a power of two beside them
    lipmaa(12);
    lipmaa(768);
    if (argc > 1) lipmaa((size_t)atol(argv[1]));
    printf("{\"domain_check\": {\"failures\": %d}}\n", fails);
    return fails;
}
