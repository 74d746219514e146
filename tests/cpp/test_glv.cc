// Host unit test of csrc/glv.h: k1 + k2*lambda == k (mod r), |k1|,|k2| < 2^127.
#include <cstdio>
#include <random>
#include "glv.h"
using namespace lsa;
int main() {
    std::mt19937_64 rng(7);
    const uint32_t LAM[8] = LSA_GLV_LAMBDA;
    Fr lam = Fr::from_canonical(LAM);
    int fails = 0;
    for (int t = 0; t < 200000; t++) {
        uint32_t k[8];
        for (;;) {
            for (int i = 0; i < 4; i++) { uint64_t x = rng(); k[2 * i] = (uint32_t)x; k[2 * i + 1] = (uint32_t)(x >> 32); }
            if (t < 8) { for (int i = 0; i < 8; i++) k[i] = 0; k[0] = t; }
            if (t == 8) for (int i = 0; i < 8; i++) k[i] = FrParams::MOD[i] - (i == 0);   // r - 1
            k[7] &= 0x3fffffffu;
            bool lt = false;
            for (int i = 7; i >= 0; --i) if (k[i] != FrParams::MOD[i]) { lt = k[i] < FrParams::MOD[i]; break; }
            if (lt) break;
        }
        GlvSplit s = glv_decompose(k);
        if ((s.k1[3] >> 31) || (s.k2[3] >> 31)) { printf("too big\n"); fails++; }
        uint32_t a[8] = {s.k1[0], s.k1[1], s.k1[2], s.k1[3], 0, 0, 0, 0}, b[8] = {s.k2[0], s.k2[1], s.k2[2], s.k2[3], 0, 0, 0, 0};
        Fr k1 = Fr::from_canonical(a), k2 = Fr::from_canonical(b);
        if (s.neg1) k1 = k1.neg();
        if (s.neg2) k2 = k2.neg();
        Fr got = k1 + k2 * lam, want = Fr::from_canonical(k);
        if (got != want) { if (fails < 5) printf("mismatch at t=%d\n", t); fails++; }
    }
    printf(fails ? "FAILED (%d)\n" : "PASS\n", fails);
    return fails ? 1 : 0;
}
