// tests/cpp/test_fp_host.cc -- the host side of fp.h (four 64-bit limbs, straight-line carries, binary-Euclid inverse)
// against the device side (eight 32-bit limbs, CIOS, Fermat inverse): the program prints a digest line per operation
// over a fixed sequence of operands; tests/test_host_cpp.py builds it twice (default and -DLSA_FP_HOST32) and compares
// the two outputs byte for byte.  Also checks the field identities on its own.
#include <cstdio>
#include <cstdint>
#include "fp.h"
using namespace lsa;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rng() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
template <class F> static F rand_f() {
    uint32_t l[8];
    for (int i = 0; i < 8; i += 2) { uint64_t r = rng(); l[i] = (uint32_t)r; l[i + 1] = (uint32_t)(r >> 32); }
    l[7] &= 0x0fffffffu;                                     // < 2^252 < MOD
    F x;
    for (int i = 0; i < 8; i++) x.l[i] = l[i];
    return x * F::r2();                                      // any residue, in Montgomery form
}
static int fails = 0;
#define CHECK(c, msg) do { if (!(c)) { fprintf(stderr, "FAIL %s (line %d)\n", msg, __LINE__); fails++; } } while (0)

template <class F> static void digest(const char *name, const F &x, uint64_t &h) {
    for (int i = 0; i < 8; i++) h = (h ^ x.l[i]) * 0x100000001B3ull;
    (void)name;
}
template <class F> static F minus_one() { return F::zero() - F::one(); }

template <class F> static void run(const char *field) {
    uint64_t hm = 1469598103934665603ull, ha = hm, hs = hm, hi = hm, hn = hm;
    F edge[6] = {F::zero(), F::one(), minus_one<F>(), F::one() + F::one(), F::r2(), minus_one<F>() - F::one()};
    for (int i = 0; i < 4000; i++) {
        const F a = i < 36 ? edge[i / 6] : rand_f<F>(), b = i < 36 ? edge[i % 6] : rand_f<F>();
        const F m = a * b, s = a + b, d = a - b, n = a.neg();
        digest("mul", m, hm); digest("add", s, ha); digest("sub", d, hs); digest("neg", n, hn);
        CHECK(s - b == a, "a + b - b");
        CHECK(d + b == a, "a - b + b");
        CHECK(n + a == F::zero(), "-a + a");
        CHECK(a * (b + F::one()) == m + a, "distributive");
        CHECK(a.sqr() == a * a, "sqr");
        CHECK(a.dbl() == a + a, "dbl");
        // every result canonical (< MOD): adding zero must not change the limbs
        CHECK(m + F::zero() == m && s + F::zero() == s && d + F::zero() == d, "canonical");
        if (i < 600) {
            const F inv = a.inverse();
            digest("inv", inv, hi);
            if (a.is_zero()) CHECK(inv.is_zero(), "inverse(0) == 0");
            else CHECK(inv * a == F::one(), "a^-1 a");
        }
    }
    // powers of two and values around them (long runs of halvings in the binary Euclid)
    F p2 = F::one();
    for (int i = 0; i < 300; i++) {
        const F inv = p2.inverse(), inv1 = (p2 + F::one()).inverse(), invm = (p2 - F::one()).inverse();
        digest("inv", inv, hi); digest("inv", inv1, hi); digest("inv", invm, hi);
        CHECK(inv * p2 == F::one(), "2^-i");
        CHECK((p2 + F::one()).is_zero() || inv1 * (p2 + F::one()) == F::one(), "(2^i + 1)^-1");
        CHECK((p2 - F::one()).is_zero() || invm * (p2 - F::one()) == F::one(), "(2^i - 1)^-1");
        p2 = p2 + p2;
    }
    uint32_t can[8];
    F::r2().to_canonical(can);
    CHECK(F::from_canonical(can) == F::r2(), "canonical round trip");
    printf("%s mul %016llx add %016llx sub %016llx neg %016llx inv %016llx\n", field, (unsigned long long)hm, (unsigned long long)ha,
           (unsigned long long)hs, (unsigned long long)hn, (unsigned long long)hi);
}

int main() {
    run<Fq>("Fq");
    run<Fr>("Fr");
    if (fails) { printf("FAILED %d\n", fails); return 1; }
    printf("PASS\n");
    return 0;
}
