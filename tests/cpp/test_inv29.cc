// Host unit test of csrc/inv29.h: the constant-time binary-GCD inversion (Pornin's Algorithm 2: 17 x 31 branch-free steps on
// 64-bit approximations) against the defining identity a * a^-1 = 1 and against the Fermat power, on random and edge
// values (small, p - k, single bits in every word).
#include <cstdio>
#include <random>
#include "fs29.h"
#include "inv29.h"
using namespace lsa;
static std::mt19937_64 rng(11);
static Fq rand_fq() {
    for (;;) {
        Fq r;
        for (int i = 0; i < 4; i++) { uint64_t x = rng(); r.l[2 * i] = (uint32_t)x; r.l[2 * i + 1] = (uint32_t)(x >> 32); }
        r.l[7] &= 0x3fffffffu;
        bool lt = false;
        for (int i = 7; i >= 0; --i) if (r.l[i] != FqParams::MOD[i]) { lt = r.l[i] < FqParams::MOD[i]; break; }
        if (lt) return r;
    }
}
int main() {
    int fails = 0;
    for (int t = 0; t < 20000; t++) {
        Fq x = rand_fq();
        if (t < 64) { for (int i = 0; i < 8; i++) x.l[i] = 0; x.l[0] = t + 1; }            // small values
        if (t >= 64 && t < 128) { for (int i = 0; i < 8; i++) x.l[i] = FqParams::MOD[i]; x.l[0] -= (t - 63); }   // p - k
        if (t >= 128 && t < 160) { for (int i = 0; i < 8; i++) x.l[i] = 0; x.l[(t - 128) / 4] = 1u << ((t * 7) % 32); }
        Fs a = Fs::from_mont256(x);
        F29 got = f29_inverse(a.v);
        Fs prod = Fs{got} * a;
        if (!(prod == Fs::one())) { if (fails < 10) printf("FAIL t=%d\n", t); fails++; }
        if (t % 50 == 0 && !(Fs{got} == a.inverse_fermat())) { if (fails < 10) printf("FAIL vs Fermat t=%d\n", t); fails++; }
    }
    // zero -> zero
    F29 z = f29_inverse(F29::zero());
    if (!z.is_zero_mod_p()) { printf("FAIL zero\n"); fails++; }
    printf(fails ? "FAILED %d\n" : "PASS\n", fails);
    return fails != 0;
}
