// Radix-2 evaluation domain with libfqfft's interface
// (libfqfft/evaluation_domain/{evaluation_domain,get_evaluation_domain}.hpp), enough for
// /root/reference/src/prototools/interp.h:62-78 and src/gadgets/lipmaa.cc:46-175 to compile,
// link and run.  FFT / iFFT / cosetFFT / icosetFFT forward to the GPU NTT (lsa_fr_ntt,
// csrc/ntt.hip); the O(m) helpers (Lagrange coefficients,
// vanishing polynomial) are inline host glue.  Unlike libfqfft, sizes are always rounded up to
// a power of two (basic radix-2 domain); the extended / step domains are not provided.
#pragma once
#include <memory>
#include <stdexcept>
#include <vector>

#include "../../libff/lsa_libff.hpp"

namespace libfqfft {

template <typename FieldT>
class evaluation_domain {
public:
    const size_t m;
    FieldT omega;
    evaluation_domain(const size_t m_) : m(m_), omega(libff::get_root_of_unity<FieldT>(m_)) {}
    virtual ~evaluation_domain() {}

    void check(const std::vector<FieldT> &a) const { if (a.size() != m) throw std::invalid_argument("evaluation_domain: expected a.size() == m"); }
    size_t log_m() const { size_t l = 0; while ((size_t(1) << l) < m) l++; return l; }
    void ntt(std::vector<FieldT> &a, int inverse, const FieldT *g) {
        check(a);
        libff::lsa_shim::GpuLock lock;
        libff::lsa_require(lsa_fr_ntt(a.data(), log_m(), &omega, inverse, g, 0), "evaluation_domain FFT");
    }
    virtual void FFT(std::vector<FieldT> &a) { ntt(a, 0, nullptr); }
    virtual void iFFT(std::vector<FieldT> &a) { ntt(a, 1, nullptr); }
    virtual void cosetFFT(std::vector<FieldT> &a, const FieldT &g) { ntt(a, 0, &g); }
    virtual void icosetFFT(std::vector<FieldT> &a, const FieldT &g) { ntt(a, 1, &g); }
    virtual std::vector<FieldT> evaluate_all_lagrange_polynomials(const FieldT &t) {
        std::vector<FieldT> u(m, FieldT::zero());
        if ((t ^ (unsigned long)m) == FieldT::one()) {
            FieldT w = FieldT::one();
            for (size_t i = 0; i < m; ++i) { if (w == t) { u[i] = FieldT::one(); return u; } w *= omega; }
        }
        const FieldT Z = (t ^ (unsigned long)m) - FieldT::one();
        FieldT l = Z * FieldT((unsigned long)m).inverse();
        FieldT r = FieldT::one();
        // u[i] = l_i / (t - r_i), l_i = l omega^i, r_i = omega^i -- libfqfft inverts every denominator on its own; here
        // all m share ONE inversion (prefix products forward, peel backward): the same field elements, three products
        // per entry instead of an inversion (m = 2^20: seconds -> a tenth of one).  No denominator is zero (t is not
        // in the domain: handled above).
        std::vector<FieldT> pre(m);
        FieldT acc = FieldT::one();
        for (size_t i = 0; i < m; ++i) { u[i] = t - r; pre[i] = acc; acc *= u[i]; r *= omega; }
        FieldT inv = acc.inverse();
        std::vector<FieldT> ls(m);
        for (size_t i = 0; i < m; ++i) { ls[i] = l; l *= omega; }
        for (size_t i = m; i-- > 0;) {
            const FieldT d = u[i];
            u[i] = ls[i] * (inv * pre[i]);
            inv *= d;
        }
        return u;
    }
    virtual FieldT get_domain_element(const size_t idx) { return omega ^ (unsigned long)idx; }
    virtual FieldT compute_vanishing_polynomial(const FieldT &t) { return (t ^ (unsigned long)m) - FieldT::one(); }
    virtual void add_poly_Z(const FieldT &coeff, std::vector<FieldT> &H) {
        if (H.size() != m + 1) throw std::invalid_argument("add_poly_Z: expected H.size() == m+1");
        H[m] += coeff;
        H[0] -= coeff;
    }
    virtual void divide_by_Z_on_coset(std::vector<FieldT> &P) {
        const FieldT zi = compute_vanishing_polynomial(FieldT::multiplicative_generator).inverse();
        for (auto &x : P) x *= zi;
    }
};

template <typename FieldT>
std::shared_ptr<evaluation_domain<FieldT>> get_evaluation_domain(const size_t min_size) {
    // libfqfft picks basic_radix2 for a power of two and extended / step / arithmetic-sequence
    // domains (a different m, omega and Lagrange basis) otherwise.  Only the basic radix-2 domain
    // exists here; every size the built examples request is a power of two
    // (src/prototools/interp.h:62, src/gadgets/lipmaa.cc:102 with n = 2^d).  Rounding another
    // size up would silently produce keys and proofs that differ from a libfqfft build, so it is
    // refused instead.
    if (min_size == 0 || (min_size & (min_size - 1)) != 0)
        throw std::invalid_argument("get_evaluation_domain: only power-of-two sizes (basic radix-2 domain) are supported; "
                                    "libfqfft would select an extended/step/arithmetic domain for this size");
    if (min_size > (size_t(1) << 28)) throw std::invalid_argument("get_evaluation_domain: size exceeds the 2-adicity of Fr (2^28)");
    return std::make_shared<evaluation_domain<FieldT>>(min_size);
}

}  // namespace libfqfft
