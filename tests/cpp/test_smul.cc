// Host unit test of csrc/smul.h (the per-lane body of k_smul_g1): k*P by GLV + signed 4-bit
// windows on 29-bit limbs against plain double-and-add with the generic Jacobian formulas of ec.h.
#include <cstdio>
#include <array>
#include <random>
#include <vector>
#include "ec.h"
#include "smul.h"
#include "tower.h"
#include "fixed_base.h"
#include "smul_host.h"
#include <chrono>
using namespace lsa;

// csrc/fixed_base.h (the shim's `Fr * generator` on the host: /root/reference/src/prototools/commit.h:43-44,162-163,
// src/gadgets/poly.h:117): 32 signed 8-bit digits over a table of the base's multiples must give the point
// double-and-add gives, for G1 and G2, on edge scalars (digit boundaries 127/128/129, carries through whole limbs, r - 1)
// and random ones; also reports the time per product next to double-and-add.
template <class F>
static Jac<F> plain_mul_any(const Jac<F> &P, const uint64_t k[4]) {
    Jac<F> acc = Jac<F>::inf();
    for (int i = 255; i >= 0; --i) {
        acc = jac_dbl(acc);
        if ((k[i >> 6] >> (i & 63)) & 1) acc = jac_add(acc, P);
    }
    return acc;
}
template <class F>
static int fixed_base_check(const Jac<F> &G, const char *name, std::mt19937_64 &rng) {
    FixedBaseTable<F> tab;
    auto t0 = std::chrono::steady_clock::now();
    tab.build(G);
    const double build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    int fails = 0;
    std::vector<std::array<uint64_t, 4>> ks;
    for (uint64_t v : {0ull, 1ull, 2ull, 127ull, 128ull, 129ull, 255ull, 256ull, 257ull, 0x8080ull, 0x7f7full, 0x8181ull, 0xffffull, 0x10000ull,
                       0xffffffffffffffffull, 0x8080808080808080ull, 0x7f807f807f807f80ull})
        ks.push_back({v, 0, 0, 0});
    ks.push_back({0xffffffffffffffffull, 0xffffffffffffffffull, 0xffffffffffffffffull, 0x3fffffffffffffffull});   // 2^254 - 1 (every digit carries)
    ks.push_back({0x8080808080808080ull, 0x8080808080808080ull, 0x8080808080808080ull, 0x2080808080808080ull});
    ks.push_back({0, 0, 0, 0x2000000000000000ull});                                                               // 2^253
    {
        std::array<uint64_t, 4> rm1;
        for (int i = 0; i < 4; i++) rm1[i] = (uint64_t)FrParams::MOD[2 * i] | ((uint64_t)FrParams::MOD[2 * i + 1] << 32);
        rm1[0] -= 1;
        ks.push_back(rm1);                                                                                         // r - 1: the result is -G
    }
    for (int t = 0; t < 300; t++) { std::array<uint64_t, 4> k = {rng(), rng(), rng(), rng() & 0x3fffffffffffffffull}; ks.push_back(k); }
    for (size_t i = 0; i < ks.size(); i++) {
        const Jac<F> got = tab.mul(ks[i].data()), want = plain_mul_any(G, ks[i].data());
        if (!jac_eq(got, want)) { if (fails < 5) printf("%s fixed-base mismatch at scalar %zu\n", name, i); fails++; }
    }
    if (!jac_eq(tab.mul(ks[ks.size() - 301].data()), jac_neg(G))) { printf("%s (r - 1) G != -G\n", name); fails++; }
    // table entries themselves: T[w][j] = (j + 1) 2^(8w) G on the curve's affine form
    {
        uint64_t k[4] = {0, 0, 0, 0};
        k[2] = (uint64_t)77 << 8;                                                                                  // 77 * 2^136 = T[17][76]
        const Aff<F> e = tab.t[(size_t)17 * FixedBaseTable<F>::HALF + 76];
        if (!jac_eq(Jac<F>{e.x, e.y, F::one()}, plain_mul_any(G, k))) { printf("%s table entry wrong\n", name); fails++; }
    }
    t0 = std::chrono::steady_clock::now();
    Jac<F> sink = Jac<F>::inf();
    for (size_t i = 24; i < 224; i++) sink = jac_add(sink, tab.mul(ks[i].data()));
    const double fb_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
    t0 = std::chrono::steady_clock::now();
    for (size_t i = 24; i < 44; i++) sink = jac_add(sink, plain_mul_any(G, ks[i].data()));
    const double da_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20;
    printf("%s fixed base: table %.2f ms, %.1f us per product (double-and-add %.1f us)%s\n", name, build_ms, fb_us, da_us, sink.is_inf() ? "." : "");
    return fails;
}

// csrc/smul_host.h (the shim's `Fr * P` on a base that is not a generator): GLV halves in width-5 NAF against
// double-and-add, on a walk of bases with Z != 1, edge scalars (0, 1, small, r - 1, lambda, lambda + 1, 2^127 +- 1) and
// random ones; infinity as a base; reports the time per product.
template <class F>
static int glv_host_check(const char *name, std::mt19937_64 &rng) {
    const Jac<F> G = GlvGenerator<F>::get();
    int fails = 0;
    std::vector<std::array<uint64_t, 4>> ks;
    for (uint64_t v : {0ull, 1ull, 2ull, 3ull, 15ull, 16ull, 17ull, 31ull, 32ull, 33ull, 0xffffull, 0x10000ull, 0xffffffffffffffffull}) ks.push_back({v, 0, 0, 0});
    ks.push_back({0xffffffffffffffffull, 0x7fffffffffffffffull, 0, 0});                                            // 2^127 - 1
    ks.push_back({0, 0x8000000000000000ull, 0, 0});                                                                // 2^127
    ks.push_back({1, 0x8000000000000000ull, 0, 0});
    {
        std::array<uint64_t, 4> rm1, lam;
        constexpr uint32_t L[8] = LSA_GLV_LAMBDA;
        for (int i = 0; i < 4; i++) {
            rm1[i] = (uint64_t)FrParams::MOD[2 * i] | ((uint64_t)FrParams::MOD[2 * i + 1] << 32);
            lam[i] = (uint64_t)L[2 * i] | ((uint64_t)L[2 * i + 1] << 32);
        }
        rm1[0] -= 1;
        ks.push_back(rm1);
        ks.push_back(lam);
        lam[0] += 1;
        ks.push_back(lam);
        lam[0] -= 2;
        ks.push_back(lam);
    }
    const size_t edge = ks.size();
    for (int t = 0; t < 400; t++) {
        std::array<uint64_t, 4> k = {rng(), rng(), rng(), rng() & 0x1fffffffffffffffull};                          // below 2^253 < r
        if (t % 7 == 3) { k[2] = k[3] = 0; }                                                                       // 128-bit scalars
        if (t % 7 == 5) { k[1] = k[2] = k[3] = 0; }
        ks.push_back(k);
    }
    Jac<F> P = G;
    for (size_t i = 0; i < ks.size(); i++) {
        const Jac<F> want = plain_mul_any(P, ks[i].data());
        if (!jac_eq(glv_mul_host(P, ks[i].data()), want)) { if (fails < 5) printf("%s GLV mismatch at scalar %zu\n", name, i); fails++; }
        if (i % 4 == 0 && !jac_eq(glv_mul_host(P, ks[i].data(), false), want)) { if (fails < 5) printf("%s GLV (Jacobian table) mismatch at scalar %zu\n", name, i); fails++; }
        P = jac_add(jac_dbl(P), G);                                                                                // next base: 2P + G (Z != 1)
    }
    if (!glv_mul_host(Jac<F>::inf(), ks[edge].data()).is_inf()) { printf("%s k * O != O\n", name); fails++; }
    auto t0 = std::chrono::steady_clock::now();
    Jac<F> sink = Jac<F>::inf();
    for (size_t i = edge; i < edge + 100; i++) sink = jac_add(sink, glv_mul_host(P, ks[i].data()));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 100;
    printf("%s any base: %.1f us per product (GLV, width-5 NAF)%s\n", name, us, sink.is_inf() ? "." : "");
    return fails;
}

// csrc/smul_host.h, G2's four-dimensional split: (1) the decomposition k = sum k_i mu^i (mod r) in the scalar field, with
// |k_i| < 2^66, on edge and random scalars; (2) gls4_mul_host against double-and-add on a walk of bases with Z != 1.
static Fr fr_from_i128(__int128 v) {
    const bool neg = v < 0;
    unsigned __int128 m = (unsigned __int128)(neg ? -v : v);
    uint32_t w[8] = {(uint32_t)m, (uint32_t)(m >> 32), (uint32_t)(m >> 64), (uint32_t)(m >> 96), 0, 0, 0, 0};
    const Fr x = Fr::from_canonical(w);
    return neg ? x.neg() : x;
}
static int gls4_check(std::mt19937_64 &rng) {
    int fails = 0;
    if (!gls4_ok()) { printf("psi(G) != mu G on the generator of G2\n"); return 1; }
    const uint64_t U = 4965661367192848881ull;
    const unsigned __int128 mu128 = (unsigned __int128)6 * U * U;
    const Fr mu = fr_from_i128((__int128)mu128);
    std::vector<std::array<uint64_t, 4>> ks;
    for (uint64_t v : {0ull, 1ull, 2ull, 31ull, 32ull, 0xffffffffffffffffull}) ks.push_back({v, 0, 0, 0});
    ks.push_back({(uint64_t)mu128, (uint64_t)(mu128 >> 64), 0, 0});
    ks.push_back({(uint64_t)mu128 + 1, (uint64_t)(mu128 >> 64), 0, 0});
    ks.push_back({0, 0, 0, 0x2000000000000000ull});
    {
        std::array<uint64_t, 4> rm1;
        for (int i = 0; i < 4; i++) rm1[i] = (uint64_t)FrParams::MOD[2 * i] | ((uint64_t)FrParams::MOD[2 * i + 1] << 32);
        rm1[0] -= 1;
        ks.push_back(rm1);
    }
    for (int t = 0; t < 3000; t++) ks.push_back({rng(), rng(), rng(), rng() & 0x1fffffffffffffffull});
    __int128 worst = 0;
    for (size_t i = 0; i < ks.size(); i++) {
        const Gls4 d = gls4_decompose(ks[i].data());
        Fr acc = Fr::zero(), pw = Fr::one();
        for (int t = 0; t < 4; t++) {
            acc = acc + fr_from_i128(d.k[t]) * pw;
            pw = pw * mu;
            const __int128 a = d.k[t] < 0 ? -d.k[t] : d.k[t];
            if (a > worst) worst = a;
        }
        uint32_t w[8];
        for (int j = 0; j < 4; j++) { w[2 * j] = (uint32_t)ks[i][j]; w[2 * j + 1] = (uint32_t)(ks[i][j] >> 32); }
        if (acc != Fr::from_canonical(w)) { if (fails < 5) printf("GLS decomposition wrong at scalar %zu\n", i); fails++; }
    }
    if (worst >> 66) { printf("GLS sub-scalar of more than 66 bits\n"); fails++; }
    const Jac<Fq2> G = GlvGenerator<Fq2>::get();
    Jac<Fq2> P = G;
    for (size_t i = 0; i < 400; i++) {
        if (!jac_eq(gls4_mul_host(P, ks[i].data()), plain_mul_any(P, ks[i].data()))) { if (fails < 5) printf("GLS product mismatch at scalar %zu\n", i); fails++; }
        P = jac_add(jac_dbl(P), G);
    }
    if (!gls4_mul_host(Jac<Fq2>::inf(), ks[20].data()).is_inf()) { printf("GLS k * O != O\n"); fails++; }
    auto t0 = std::chrono::steady_clock::now();
    Jac<Fq2> sink = Jac<Fq2>::inf();
    for (size_t i = 20; i < 120; i++) sink = jac_add(sink, gls4_mul_host(P, ks[i].data()));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 100;
    printf("G2 any base, four-dimensional: %.1f us per product%s\n", us, sink.is_inf() ? "." : "");
    return fails;
}

static Jac<Fq> plain_mul(const Jac<Fq> &P, const uint32_t k[8]) {
    Jac<Fq> acc = Jac<Fq>::inf();
    for (int i = 255; i >= 0; --i) {
        acc = jac_dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1) acc = jac_add(acc, P);
    }
    return acc;
}

int main() {
    std::mt19937_64 rng(11);
    Fq one = Fq::one();
    Jac<Fq> G = {one, one + one, one};
    int fails = 0;
    Jac<Fq> P = G;
    for (int t = 0; t < 1500; t++) {
        uint32_t k[8];
        for (;;) {
            for (int i = 0; i < 4; i++) { uint64_t x = rng(); k[2 * i] = (uint32_t)x; k[2 * i + 1] = (uint32_t)(x >> 32); }
            k[7] &= 0x3fffffffu;
            if (t < 20) { for (int i = 0; i < 8; i++) k[i] = 0; k[0] = t; }
            if (t == 20) for (int i = 0; i < 8; i++) k[i] = FrParams::MOD[i] - (i == 0);   // r - 1
            if (t == 21) { for (int i = 0; i < 8; i++) k[i] = 0; k[3] = 0x80000000u; }      // 2^127
            if (t == 22) { for (int i = 0; i < 8; i++) k[i] = 0; k[0] = 0x88888888u; }
            bool lt = false;
            for (int i = 7; i >= 0; --i) if (k[i] != FrParams::MOD[i]) { lt = k[i] < FrParams::MOD[i]; break; }
            if (lt) break;
        }
        Jac<Fq> Pn = jac_normalize(P);
        Aff29 A = {F29::from_mont256(Pn.X), F29::from_mont256(Pn.Y)};
        XYZZ29 T[SMUL_TBL];
        Jac<Fq> got = xyzz29_to_jac(smul_glv(A, k, T));
        Jac<Fq> want = plain_mul(P, k);
        if (!jac_eq(got, want)) { if (fails < 5) printf("mismatch at t=%d\n", t); fails++; }
        P = jac_add(jac_dbl(P), G);   // next base: 2P + G
    }
    {
        std::mt19937_64 rng2(12);
        fails += fixed_base_check<Fq>(G, "G1", rng2);
        const Jac<Fq2> G2 = {fq2_const(LSA_G2_GEN_X), fq2_const(LSA_G2_GEN_Y), Fq2::one()};
        fails += fixed_base_check<Fq2>(G2, "G2", rng2);
        fails += fixed_base_check<Fq>(jac_add(jac_dbl(G), G), "G1 (3G, Z != 1)", rng2);
        fails += glv_host_check<Fq>("G1", rng2);
        fails += glv_host_check<Fq2>("G2", rng2);
        fails += gls4_check(rng2);
    }
    printf(fails ? "FAILED (%d)\n" : "PASS\n", fails);
    return fails ? 1 : 0;
}
