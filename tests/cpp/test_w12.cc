// Host unit test of csrc/w12.h: the wavefront-spread Fq12 engine (36-lane products, 6-lane
// reductions, lane-0 Frobenius / inverse) executed phase by phase over emulated lane ids,
// against the tower code on the same field.
#include <cstdio>
#include <random>
#include <vector>
#include "miller.h"
using namespace lsa;
static std::mt19937_64 rng(5);
static int fails = 0;
#define CHECK(c, msg) do { if (!(c)) { if (fails < 20) printf("FAIL %s (line %d)\n", msg, __LINE__); fails++; } } while (0)
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
static Fq12S rand12() { Fq12S r; Fs *o = reinterpret_cast<Fs *>(&r); for (int i = 0; i < 12; i++) o[i] = Fs::from_mont256(rand_fq()); return r; }
struct LoopExec {
    template <class F> void par(F f) { for (unsigned l = 0; l < 64; l++) f(l); }
    unsigned nlanes() const { return 64; }
};
int main() {
    std::vector<Fq2S> lds(W12_LDS_FQ2);
    LoopExec ex;
    W12<LoopExec> w{ex, lds.data(), lds.data() + 6 * W12_SLOTS};
    for (int t = 0; t < 10; t++) {
        Fq12S a = rand12(), b = rand12();
        w.store_tower(1, a);
        w.store_tower(2, b);
        CHECK(w.load_tower(1) == a, "tower round trip");
        w.mul(3, 1, 2);
        CHECK(w.load_tower(3) == fq12_mul(a, b), "mul");
        w.mul(1, 1, 2);                                   // aliasing d == a
        CHECK(w.load_tower(1) == fq12_mul(a, b), "mul alias");
        w.store_tower(1, a);
        w.sqr(4, 1);
        CHECK(w.load_tower(4) == fq12_sqr(a), "sqr");
        w.conj(5, 1);
        CHECK(w.load_tower(5) == a.unitary_inverse(), "conj");
        w.frobenius<1>(6, 1);
        CHECK(w.load_tower(6) == fq12_frobenius<1>(a), "frob1");
        w.frobenius<3>(6, 6);
        CHECK(w.load_tower(6) == fq12_frobenius<3>(fq12_frobenius<1>(a)), "frob3 alias");
        if (t < 2) {
            w.inverse(7, 1);
            CHECK(w.load_tower(7) == fq12_inverse(a), "inverse");
            w.mul(7, 7, 1);
            CHECK(w.load_tower(7) == Fq12S::one(), "a * a^-1");
            w.pow_u64(8, 1, 0x1234567ull, 9);
            CHECK(w.load_tower(8) == fq12_pow_u64(a, 0x1234567ull), "pow");
        }
    }
    // Granger-Scott squaring on twelve lanes against the tower's cyclotomic and general squarings, on elements of the
    // cyclotomic subgroup (f^((p^6-1)(p^2+1))), chained (outputs < 2p feed the next squaring)
    for (int t = 0; t < 6; t++) {
        Fq12S a = rand12();
        Fq12S e = fq12_mul(a.unitary_inverse(), fq12_inverse(a));           // f^(p^6-1)
        e = fq12_mul(fq12_frobenius<2>(e), e);                              // ^(p^2+1)
        w.store_tower(1, e);
        Fq12S want = e;
        int cur = 1, nxt = 2;
        for (int r = 0; r < 5; r++) {
            w.csqr(nxt, cur);
            want = fq12_sqr(want);
            CHECK(w.load_tower(nxt) == want, "cyclotomic squaring vs general squaring");
            CHECK(fq12_cyclotomic_sqr(w.load_tower(cur)) == want, "tower cyclotomic squaring");
            std::swap(cur, nxt);
        }
    }
    for (int t = 0; t < 2; t++) {
        Fq12S a = rand12();
        w.store_tower(0, a);
        w.final_exponentiation();
        Fq12S want = fq12_final_exponentiation(a);
        CHECK(w.load_tower(0) == want, "final exponentiation");
        // the result is an r-th root of unity of the cyclotomic subgroup: unitary
        CHECK(fq12_mul(want, want.unitary_inverse()) == Fq12S::one(), "unitary result");
    }
    // wavefront Miller loop vs the one-lane loop (the formulas are plain field arithmetic, so
    // random coordinates exercise them; Z = 1 and Z != 1 inputs)
    {
        CHECK(Fs::from_mont256(rand_fq()).halve().dbl() != Fs::zero(), "halve nonzero");
        for (int t = 0; t < 50; t++) {
            Fq a = rand_fq();
            Fs h = Fs::from_mont256(a).halve();
            CHECK((h + h).to_mont256() == a, "halve");
        }
        std::vector<Fq2S> lds2(WM_LDS_FQ2);
        for (int t = 0; t < 3; t++) {
            Jac<Fq> Pp = {rand_fq(), rand_fq(), t == 0 ? Fq::one() : rand_fq()};
            Jac<Fq2> Qq = {{rand_fq(), rand_fq()}, {rand_fq(), rand_fq()}, t == 0 ? Fq2::one() : Fq2{rand_fq(), rand_fq()}};
            if (t == 2) Pp.Z = Fq::zero();
            W12<LoopExec> w2{ex, lds2.data(), lds2.data() + 6 * W12_SLOTS};
            WMiller<LoopExec> m{w2, lds2.data() + W12_LDS_FQ2, lds2.data() + W12_LDS_FQ2 + WM_NVARS};
            m.run(Pp, Qq);
            CHECK(w2.load_tower(0) == miller_one(Pp, Qq), "wave miller loop");
        }
    }
    printf(fails ? "FAILED (%d)\n" : "PASS\n", fails);
    return fails ? 1 : 0;
}
