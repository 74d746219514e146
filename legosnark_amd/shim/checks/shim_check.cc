// legosnark_amd/shim/checks/shim_check.cc -- corners of the libff / libfqfft surface the reference's examples do not
// reach but a LegoSNARK user may: libff's window_table is a vector of vectors a caller can index
// (powers_of_g[outer][inner] = inner * 2^(outer * window) * g, the last row short), and libfqfft picks other domain
// types for sizes that are not powers of two (the step radix-2 domain; sizes beyond the field's 2-adicity are refused).
// Prints one line per check and a final count; exit status = number of failures.
#include <cstdio>
#include <stdexcept>
#include <string>
#include "globl.h"

int main() {
    default_ec_pp::init_public_params();
    int fails = 0;
    auto check = [&](bool ok, const char *what) { printf("%s %s\n", ok ? "ok  " : "FAIL", what); if (!ok) fails++; };
    {
        const size_t bits = LFr::size_in_bits(), w = 5;                 // a small window: 51 rows of 32, the last of 16
        const LG1 g = LFr(12345) * LG1::one();
        auto tab = libff::get_window_table<LG1>(bits, w, g);
        const size_t outerc = (bits + w - 1) / w, last = size_t(1) << (bits - (outerc - 1) * w);
        check(tab.size() == outerc && tab[0].size() == (size_t(1) << w), "window_table<G1>: rows and row length as libff");
        bool ok = tab[0][0].is_zero() && tab[0][1] == g && tab[0][7] == LFr(7) * g;
        LFr p2 = LFr::one();
        for (size_t b = 0; b < 3 * w; b++) p2 = p2 + p2;
        ok = ok && tab[3][9] == (LFr(9) * p2) * g;                      // inner * 2^(outer * window) * g
        check(ok, "window_table<G1>: entries are inner * 2^(outer * window) * g");
        ok = true;
        for (size_t i = last; i < (size_t(1) << w); i++) ok = ok && tab[outerc - 1][i].is_zero();
        check(ok && !tab[outerc - 1][last - 1].is_zero(), "window_table<G1>: the short last row ends in zeros");
        size_t rows = 0;
        for (const auto &row : tab) rows += row.size() ? 1 : 0;
        check(rows == outerc, "window_table<G1>: iterable");
        // the opaque use of the reference (src/utils/util.h:125-133): batch_exp over the same table
        std::vector<LFr> xs = {LFr(3), LFr::random_element(), LFr::zero()};
        auto out = libff::batch_exp<LG1, LFr>(bits, w, tab, xs);
        check(out.size() == 3 && out[0] == LFr(3) * g && out[1] == xs[1] * g && out[2].is_zero(), "batch_exp over an indexed table");
        auto tab2 = libff::get_window_table<LG2>(bits, 4, LG2::one());
        check(tab2[1][3] == LFr(3 * 16) * LG2::one(), "window_table<G2>: entries");
    }
    {
        // libfqfft's choice for sizes that are not powers of two (checks/domain_check.cc runs these domains)
        check(libfqfft::get_evaluation_domain<LFr>(1000)->m == 1024, "get_evaluation_domain(1000): 512 + 488 rounds to the basic domain of 1024");
        check(libfqfft::get_evaluation_domain<LFr>(768)->m == 768, "get_evaluation_domain(768): the step domain of 512 + 256");
        bool threw = false;
        try { (void)libfqfft::get_evaluation_domain<LFr>((size_t(1) << 28) + 5); } catch (const std::invalid_argument &e) { threw = true; }
        check(threw, "get_evaluation_domain(2^28 + 5): refused with a message (no extended / sequence domains)");
        auto d = libfqfft::get_evaluation_domain<LFr>(1024);
        check(d && d->m == 1024, "get_evaluation_domain(1024): basic radix-2 domain");
    }
    printf("shim_check: %d failure(s)\n", fails);
    return fails;
}
