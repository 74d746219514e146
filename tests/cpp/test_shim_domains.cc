// Host test of the shim's evaluation domains (shim/libfqfft/evaluation_domain/get_evaluation_domain.hpp): which domain
// libfqfft's get_evaluation_domain selects for a size, and the O(m) helpers of the step radix-2 domain (m = 2^b + 2^s)
// against their definitions on the domain's points -- no GPU call (the transforms themselves are checked on the GPU:
// tests/test_fr_vec_gpu.py through the C-ABI, shim/checks/shim_check.cc through this class).
#include <cstdio>
#include <memory>
#include <vector>

#include "libff/lsa_libff.hpp"
#include "libfqfft/evaluation_domain/get_evaluation_domain.hpp"

using namespace libff;
typedef alt_bn128_Fr Fr_;

static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

static Fr_ horner(const std::vector<Fr_> &c, const Fr_ &x) {
    Fr_ acc = Fr_::zero();
    for (size_t i = c.size(); i-- > 0;) acc = acc * x + c[i];
    return acc;
}

static void check_domain(size_t min_size, size_t want_m, bool want_step) {
    auto dom = libfqfft::get_evaluation_domain<Fr_>(min_size);
    const size_t m = dom->m;
    CHECK(m == want_m);
    const bool is_step = dynamic_cast<libfqfft::step_radix2_domain<Fr_> *>(dom.get()) != nullptr;
    CHECK(is_step == want_step);
    std::vector<Fr_> pts(m);
    for (size_t k = 0; k < m; k++) pts[k] = dom->get_domain_element(k);
    if (is_step) {
        auto *sd = static_cast<libfqfft::step_radix2_domain<Fr_> *>(dom.get());
        CHECK(sd->big_m + sd->small_m == m && sd->big_m == (size_t(1) << (libff::log2(m) - 1)));
        CHECK((dom->omega ^ (unsigned long)(2 * sd->big_m)) == Fr_::one() && (dom->omega ^ (unsigned long)sd->big_m) == -Fr_::one());
        CHECK(pts[1] == dom->omega.squared() && pts[sd->big_m] == dom->omega);
    } else {
        CHECK(m == 1 || pts[1] == dom->omega);
    }
    // distinct points, all roots of Z; Z monic of degree m: Z(t) = prod (t - x_k)
    const Fr_ t = Fr_::random_element();
    Fr_ prod = Fr_::one();
    bool roots = true, distinct = true;
    for (size_t k = 0; k < m; k++) {
        roots = roots && dom->compute_vanishing_polynomial(pts[k]).is_zero();
        for (size_t j = 0; j < k && m <= 96; j++) distinct = distinct && pts[j] != pts[k];
        prod *= t - pts[k];
    }
    CHECK(roots);
    CHECK(distinct);
    CHECK(dom->compute_vanishing_polynomial(t) == prod);
    // Lagrange coefficients interpolate: sum_k L_k(t) f(x_k) = f(t) for deg f < m; indicator on a domain point
    std::vector<Fr_> f(m);
    for (auto &c : f) c = Fr_::random_element();
    const std::vector<Fr_> L = dom->evaluate_all_lagrange_polynomials(t);
    Fr_ acc = Fr_::zero(), sum = Fr_::zero();
    for (size_t k = 0; k < m; k++) { acc += L[k] * horner(f, pts[k]); sum += L[k]; }
    CHECK(L.size() == m && acc == horner(f, t));
    CHECK(sum == Fr_::one());
    for (size_t at : {size_t(0), m / 2, m - 1}) {
        const std::vector<Fr_> e = dom->evaluate_all_lagrange_polynomials(pts[at]);
        bool ind = true;
        for (size_t i = 0; i < m; i++) ind = ind && e[i] == (i == at ? Fr_::one() : Fr_::zero());
        CHECK(ind);
    }
    // add_poly_Z: H += c Z as coefficient vectors
    {
        std::vector<Fr_> H(m + 1);
        for (auto &c : H) c = Fr_::random_element();
        const Fr_ before = horner(H, t), c = Fr_::random_element();
        dom->add_poly_Z(c, H);
        CHECK(horner(H, t) == before + c * dom->compute_vanishing_polynomial(t));
    }
    // divide_by_Z_on_coset: P[k] <- P[k] / Z(g x_k)
    {
        std::vector<Fr_> P(m), Q;
        for (auto &c : P) c = Fr_::random_element();
        Q = P;
        dom->divide_by_Z_on_coset(Q);
        bool ok = true;
        for (size_t k = 0; k < m; k++) ok = ok && Q[k] * dom->compute_vanishing_polynomial(Fr_::multiplicative_generator * pts[k]) == P[k];
        CHECK(ok);
    }
}

int main() {
    // powers of two: the basic radix-2 domain of that size
    check_domain(2, 2, false);
    check_domain(64, 64, false);
    // 2^b + 2^s: the step domain of that very size (small_m = 1, a middle one, big_m / 2)
    check_domain(3, 3, true);
    check_domain(5, 5, true);
    check_domain(6, 6, true);
    check_domain(12, 12, true);
    check_domain(33, 33, true);
    check_domain(40, 40, true);
    check_domain(96, 96, true);
    check_domain(768, 768, true);
    // any other size: the part above the top power of two is rounded up to a power of two
    check_domain(7, 8, false);          // 4 + 3 -> 4 + 4
    check_domain(11, 12, true);         // 8 + 3 -> 8 + 4
    check_domain(13, 16, false);        // 8 + 5 -> 8 + 8
    check_domain(700, 768, true);       // 512 + 188 -> 512 + 256
    check_domain(1000, 1024, false);    // 512 + 488 -> 512 + 512
    check_domain(67, 68, true);         // 64 + 3 -> 64 + 4
    // beyond the 2-adicity: refused with a message, never replaced by another domain
    {
        bool threw = false;
        try { (void)libfqfft::get_evaluation_domain<Fr_>((size_t(1) << 28) + 1); } catch (const std::exception &e) { threw = true; }
        CHECK(threw);
        threw = false;
        try { (void)libfqfft::get_evaluation_domain<Fr_>(0); } catch (const std::exception &e) { threw = true; }
        CHECK(threw);
    }
    printf(fails ? "FAILED %d\n" : "PASS\n", fails);
    return fails ? 1 : 0;
}
