// Host-only radix-2 evaluation domain with libfqfft's interface
// (libfqfft/evaluation_domain/{evaluation_domain,get_evaluation_domain}.hpp), enough for
// /root/reference/src/prototools/interp.h:62-78 and src/gadgets/lipmaa.cc:46-175 to compile,
// link and run.  OUT OF SCOPE for the GPU path (Fr NTT, Lipmaa only -- SURVEY.md section
// 8f rank 4).  Unlike libfqfft, sizes are always rounded up to a power of two (basic radix-2
// domain); the extended / step domains are not provided.
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

    static void fft_inplace(std::vector<FieldT> &a, const FieldT &w) {
        const size_t n = a.size();
        for (size_t i = 1, j = 0; i < n; i++) {          // bit reversal
            size_t bit = n >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) std::swap(a[i], a[j]);
        }
        for (size_t len = 2; len <= n; len <<= 1) {
            FieldT wl = w;
            for (size_t k = len; k < n; k <<= 1) wl = wl.squared();
            for (size_t i = 0; i < n; i += len) {
                FieldT x = FieldT::one();
                for (size_t j = 0; j < len / 2; j++) {
                    FieldT u = a[i + j], v = a[i + j + len / 2] * x;
                    a[i + j] = u + v;
                    a[i + j + len / 2] = u - v;
                    x *= wl;
                }
            }
        }
    }
    void check(const std::vector<FieldT> &a) const { if (a.size() != m) throw std::invalid_argument("evaluation_domain: expected a.size() == m"); }
    virtual void FFT(std::vector<FieldT> &a) { check(a); fft_inplace(a, omega); }
    virtual void iFFT(std::vector<FieldT> &a) {
        check(a);
        fft_inplace(a, omega.inverse());
        const FieldT sconst = FieldT((unsigned long)m).inverse();
        for (auto &x : a) x *= sconst;
    }
    virtual void cosetFFT(std::vector<FieldT> &a, const FieldT &g) {
        FieldT u = FieldT::one();
        for (auto &x : a) { x *= u; u *= g; }
        FFT(a);
    }
    virtual void icosetFFT(std::vector<FieldT> &a, const FieldT &g) {
        iFFT(a);
        const FieldT gi = g.inverse();
        FieldT u = FieldT::one();
        for (auto &x : a) { x *= u; u *= gi; }
    }
    virtual std::vector<FieldT> evaluate_all_lagrange_polynomials(const FieldT &t) {
        std::vector<FieldT> u(m, FieldT::zero());
        if ((t ^ (unsigned long)m) == FieldT::one()) {
            FieldT w = FieldT::one();
            for (size_t i = 0; i < m; ++i) { if (w == t) { u[i] = FieldT::one(); return u; } w *= omega; }
        }
        const FieldT Z = (t ^ (unsigned long)m) - FieldT::one();
        FieldT l = Z * FieldT((unsigned long)m).inverse();
        FieldT r = FieldT::one();
        for (size_t i = 0; i < m; ++i) { u[i] = l * (t - r).inverse(); l *= omega; r *= omega; }
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
    size_t m = 1;
    while (m < min_size) m <<= 1;
    return std::make_shared<evaluation_domain<FieldT>>(m);
}

}  // namespace libfqfft
