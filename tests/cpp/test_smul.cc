// Host unit test of csrc/smul.h (the per-lane body of k_smul_g1): k*P by GLV + signed 4-bit
// windows on 29-bit limbs against plain double-and-add with the generic Jacobian formulas of ec.h.
#include <cstdio>
#include <random>
#include "ec.h"
#include "smul.h"
using namespace lsa;

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
    printf(fails ? "FAILED (%d)\n" : "PASS\n", fails);
    return fails ? 1 : 0;
}
