// libfqfft/polynomial_arithmetic/basic_operations.hpp stand-in (host only; see
// ../evaluation_domain/get_evaluation_domain.hpp): FFT-based polynomial product used by
// /root/reference/src/gadgets/lipmaa.cc:85,90.
#pragma once
#include <vector>

#include "../evaluation_domain/get_evaluation_domain.hpp"

namespace libfqfft {

template <typename FieldT>
void _condense(std::vector<FieldT> &a) {
    while (!a.empty() && a.back() == FieldT::zero()) a.pop_back();
}

template <typename FieldT>
void _polynomial_multiplication(std::vector<FieldT> &c, const std::vector<FieldT> &a, const std::vector<FieldT> &b) {
    if (a.empty() || b.empty()) { c.clear(); return; }
    const size_t need = a.size() + b.size() - 1;
    auto dom = get_evaluation_domain<FieldT>(need);
    std::vector<FieldT> u(a), v(b);
    u.resize(dom->m, FieldT::zero());
    v.resize(dom->m, FieldT::zero());
    dom->FFT(u);
    dom->FFT(v);
    for (size_t i = 0; i < dom->m; i++) u[i] *= v[i];
    dom->iFFT(u);
    u.resize(need);
    c = u;
    _condense(c);
}

}  // namespace libfqfft
