// Evaluation domains with libfqfft's interface
// (libfqfft/evaluation_domain/{evaluation_domain,get_evaluation_domain}.hpp and domains/{basic,step}_radix2_domain),
// for /root/reference/src/prototools/interp.h:62-78 and src/gadgets/lipmaa.cc:46-175.  FFT / iFFT / cosetFFT / icosetFFT
// forward to the GPU (lsa_fr_ntt, lsa_fr_ntt_step: csrc/ntt.hip); the O(m) helpers
// (Lagrange coefficients, vanishing polynomial, division by Z on the coset) are inline host glue.
//   basic_radix2_domain   m = 2^k                    (the base class below)
//   step_radix2_domain    m = 2^b + 2^s, s < b       (every other size: get_evaluation_domain rounds the part above the
//                                                     top power of two up to a power of two)
// libfqfft's extended radix-2 domain (m = 2^29 for this field: 16 GiB of scalars) and its arithmetic / geometric sequence
// domains (reached only beyond the 2-adicity of Fr) are not provided: such sizes are refused with a message.
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
    evaluation_domain(const size_t m_, const FieldT &omega_) : m(m_), omega(omega_) {}
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
using basic_radix2_domain = evaluation_domain<FieldT>;

// The Lagrange coefficients of the basic radix-2 domain of size n at t (libfqfft _basic_radix2_evaluate_all_lagrange_polynomials)
template <typename FieldT>
std::vector<FieldT> lsa_radix2_lagrange(const size_t n, const FieldT &t) {
    if (n == 1) return std::vector<FieldT>(1, FieldT::one());
    evaluation_domain<FieldT> d(n);
    return d.evaluate_all_lagrange_polynomials(t);
}

// libfqfft step_radix2_domain [upstream, recalled: domains/step_radix2_domain.tcc]: m = big_m + small_m, big_m =
// 2^(ceil(log2 m) - 1), small_m = m - big_m a power of two.  Points: big_omega^k (k < big_m), then omega small_omega^j
// (j < small_m); omega = get_root_of_unity(2 big_m), big_omega = omega^2, small_omega = get_root_of_unity(small_m).
// Everything below follows from that choice of points: Z(x) = (x^big_m - 1)(x^small_m - omega^small_m) is their monic
// vanishing polynomial, and the Lagrange basis is the radix-2 basis of either part times the other part's vanishing
// polynomial, normalised at the point.
template <typename FieldT>
class step_radix2_domain : public evaluation_domain<FieldT> {
public:
    size_t big_m, small_m;
    FieldT big_omega, small_omega;
    static size_t big_part(const size_t m) {
        if (m <= 1) throw std::invalid_argument("step_radix2(): expected m > 1");
        return size_t(1) << (libff::log2(m) - 1);
    }
    static FieldT domain_omega(const size_t m) {
        const size_t big = big_part(m);
        if (big >= (size_t(1) << 28)) throw std::invalid_argument("step_radix2(): size exceeds the 2-adicity of Fr");
        return libff::get_root_of_unity<FieldT>(2 * big);
    }
    step_radix2_domain(const size_t m_) : evaluation_domain<FieldT>(m_, domain_omega(m_)) {
        big_m = big_part(m_);
        small_m = m_ - big_m;
        if (small_m == 0 || small_m != (size_t(1) << libff::log2(small_m))) throw std::invalid_argument("step_radix2(): expected small_m == 1ul<<log2(small_m)");
        big_omega = this->omega.squared();
        small_omega = libff::get_root_of_unity<FieldT>(small_m);
    }
    size_t big_log() const { return libff::log2(big_m); }
    size_t small_log() const { return libff::log2(small_m); }
    void step(std::vector<FieldT> &a, int inverse, const FieldT *g) {
        this->check(a);
        libff::lsa_shim::GpuLock lock;
        libff::lsa_require(lsa_fr_ntt_step(a.data(), big_log(), small_log(), &this->omega, inverse, g, 0), "step_radix2_domain FFT");
    }
    void FFT(std::vector<FieldT> &a) override { step(a, 0, nullptr); }
    void iFFT(std::vector<FieldT> &a) override { step(a, 1, nullptr); }
    void cosetFFT(std::vector<FieldT> &a, const FieldT &g) override { step(a, 0, &g); }
    void icosetFFT(std::vector<FieldT> &a, const FieldT &g) override { step(a, 1, &g); }
    std::vector<FieldT> evaluate_all_lagrange_polynomials(const FieldT &t) override {
        const std::vector<FieldT> inner_big = lsa_radix2_lagrange<FieldT>(big_m, t);
        const std::vector<FieldT> inner_small = lsa_radix2_lagrange<FieldT>(small_m, t * this->omega.inverse());
        std::vector<FieldT> result(this->m, FieldT::zero());
        // a point of the big part: its radix-2 coefficient times (t^small_m - omega^small_m) / (x^small_m - omega^small_m);
        // x^small_m runs through the powers of big_omega^small_m: big_m / small_m distinct values, ONE inversion for all
        const FieldT omega_to_small_m = this->omega ^ (unsigned long)small_m;
        const FieldT big_omega_to_small_m = big_omega ^ (unsigned long)small_m;
        const FieldT L0 = (t ^ (unsigned long)small_m) - omega_to_small_m;
        const size_t period = big_m / small_m;
        std::vector<FieldT> den(period), pre(period);
        FieldT elt = FieldT::one(), acc = FieldT::one();
        for (size_t i = 0; i < period; ++i) { den[i] = elt - omega_to_small_m; pre[i] = acc; acc *= den[i]; elt *= big_omega_to_small_m; }
        FieldT inv = acc.inverse();                          // (no denominator vanishes: omega^small_m is not a small_m-th power of big_omega)
        for (size_t i = period; i-- > 0;) { const FieldT d = den[i]; den[i] = L0 * (inv * pre[i]); inv *= d; }
        for (size_t i = 0; i < big_m; ++i) result[i] = inner_big[i] * den[i % period];
        // a point of the small part: (t^big_m - 1) / (x^big_m - 1) with x^big_m = omega^big_m = -1
        const FieldT L1 = ((t ^ (unsigned long)big_m) - FieldT::one()) * ((this->omega ^ (unsigned long)big_m) - FieldT::one()).inverse();
        for (size_t i = 0; i < small_m; ++i) result[big_m + i] = L1 * inner_small[i];
        return result;
    }
    FieldT get_domain_element(const size_t idx) override {
        if (idx < big_m) return big_omega ^ (unsigned long)idx;
        return this->omega * (small_omega ^ (unsigned long)(idx - big_m));
    }
    FieldT compute_vanishing_polynomial(const FieldT &t) override {
        return ((t ^ (unsigned long)big_m) - FieldT::one()) * ((t ^ (unsigned long)small_m) - (this->omega ^ (unsigned long)small_m));
    }
    void add_poly_Z(const FieldT &coeff, std::vector<FieldT> &H) override {
        if (H.size() != this->m + 1) throw std::invalid_argument("step_radix2: expected H.size() == this->m+1");
        const FieldT omega_to_small_m = this->omega ^ (unsigned long)small_m;
        H[this->m] += coeff;
        H[big_m] -= coeff * omega_to_small_m;
        H[small_m] -= coeff;
        H[0] += coeff * omega_to_small_m;
    }
    void divide_by_Z_on_coset(std::vector<FieldT> &P) override {
        // P[k] /= Z(g x_k), g = multiplicative_generator: (g^big_m - 1)(g^small_m x^small_m - omega^small_m) on the big part
        // (big_m / small_m distinct values), one constant on the small part
        const FieldT coset = FieldT::multiplicative_generator;
        const FieldT omega_to_small_m = this->omega ^ (unsigned long)small_m;
        const FieldT Z0 = (coset ^ (unsigned long)big_m) - FieldT::one();
        const FieldT coset_to_small_m = coset ^ (unsigned long)small_m;
        const FieldT step_ = big_omega ^ (unsigned long)small_m;
        const size_t period = big_m / small_m;
        std::vector<FieldT> zi(period);
        FieldT elt = FieldT::one();
        for (size_t i = 0; i < period; ++i) { zi[i] = (Z0 * (coset_to_small_m * elt - omega_to_small_m)).inverse(); elt *= step_; }
        for (size_t i = 0; i < big_m; ++i) P[i] *= zi[i % period];
        const FieldT cw = coset * this->omega;
        const FieldT Z1 = ((cw ^ (unsigned long)big_m) - FieldT::one()) * ((cw ^ (unsigned long)small_m) - omega_to_small_m);
        const FieldT Z1_inverse = Z1.inverse();
        for (size_t i = 0; i < small_m; ++i) P[big_m + i] *= Z1_inverse;
    }
};

// libfqfft get_evaluation_domain [upstream, recalled: get_evaluation_domain.tcc]: the first domain that accepts the size,
// tried in the order basic_radix2(min_size), extended_radix2(min_size), step_radix2(min_size), then the same three at
// big + rounded_small (big = 2^(ceil(log2 min_size) - 1), rounded_small = the rest rounded up to a power of two), then
// the geometric and arithmetic sequence domains.  For this field (2-adicity 28) that is: a power of two -> basic; 2^b +
// 2^s -> step of that very size; any other size -> step (or basic, when the rounding reaches the next power of two) of
// big + rounded_small.  The extended and sequence domains are not provided: refused, never silently replaced.
template <typename FieldT>
std::shared_ptr<evaluation_domain<FieldT>> get_evaluation_domain(const size_t min_size) {
    // (sizes 0 and 1: every libfqfft domain constructor requires m > 1, upstream ends in DomainSizeException)
    if (min_size <= 1) throw std::invalid_argument("get_evaluation_domain: no matching domain (libfqfft's domains need m > 1)");
    const auto pow2 = [](size_t v) { return v && (v & (v - 1)) == 0; };
    if (min_size > (size_t(1) << 28))
        throw std::invalid_argument("get_evaluation_domain: size exceeds the 2-adicity of Fr (2^28); libfqfft's extended radix-2 and sequence domains are not provided");
    if (pow2(min_size)) return std::make_shared<evaluation_domain<FieldT>>(min_size);
    const size_t big = size_t(1) << (libff::log2(min_size) - 1);
    const size_t small = min_size - big;
    const size_t rounded_small = size_t(1) << libff::log2(small);
    if (small == rounded_small) return std::make_shared<step_radix2_domain<FieldT>>(min_size);
    if (pow2(big + rounded_small)) return std::make_shared<evaluation_domain<FieldT>>(big + rounded_small);
    return std::make_shared<step_radix2_domain<FieldT>>(big + rounded_small);
}

}  // namespace libfqfft
