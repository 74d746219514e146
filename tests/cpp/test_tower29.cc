// Host unit test: the extension tower instantiated on the 29-bit-limb base field (Fs)
// against the same tower on libff-layout Fq, incl. cyclotomic squaring vs plain squaring.
#include <cstdio>
#include <random>
#include "tower.h"
#include "fs29.h"
using namespace lsa;
static std::mt19937_64 rng(99);
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
using F12s = Fq12T<Fs>;
static Fq12 rand12() { Fq12 r; Fq *w = reinterpret_cast<Fq *>(&r); for (int i = 0; i < 12; i++) w[i] = rand_fq(); return r; }
static F12s to_s(const Fq12 &a) { F12s r; const Fq *w = reinterpret_cast<const Fq *>(&a); Fs *o = reinterpret_cast<Fs *>(&r); for (int i = 0; i < 12; i++) o[i] = Fs::from_mont256(w[i]); return r; }
static Fq12 from_s(const F12s &a) { Fq12 r; Fq *w = reinterpret_cast<Fq *>(&r); const Fs *o = reinterpret_cast<const Fs *>(&a); for (int i = 0; i < 12; i++) w[i] = o[i].to_mont256(); return r; }
int main() {
    for (int t = 0; t < 300; t++) {
        Fq a = rand_fq(), b = rand_fq();
        Fs A = Fs::from_mont256(a), B = Fs::from_mont256(b);
        CHECK((A * B).to_mont256() == a * b, "mul");
        CHECK((A + B).to_mont256() == a + b, "add");
        CHECK((A - B).to_mont256() == a - b, "sub");
        CHECK(A.neg().to_mont256() == a.neg(), "neg");
        CHECK((A - A).is_zero() && Fs::zero().neg().is_zero(), "zero");
        CHECK(((A + B) - B) == A, "eq");
    }
    CHECK(Fs::from_mont256(rand_fq()).inverse().to_mont256() != Fq::zero(), "inv nonzero");
    { Fq a = rand_fq(); CHECK(Fs::from_mont256(a).inverse().to_mont256() == a.inverse(), "inverse"); }
    {
        // Fs::inverse (29-bit limbs) against Fq::inverse (another implementation: binary Euclid on 64-bit limbs, or the
        // power a^(p-2) under -DLSA_FP_HOST32): random values, 0, +-1, powers of two and their neighbours, non-canonical inputs
        bool ok = true;
        for (int t = 0; t < 300; t++) { const Fq a = rand_fq(); ok = ok && Fs::from_mont256(a).inverse().to_mont256() == a.inverse(); }
        CHECK(ok, "inverse, random");
        CHECK(Fs::zero().inverse().to_mont256() == Fq::zero(), "inverse(0) == 0");
        CHECK(Fs::one().inverse().to_mont256() == Fq::one(), "inverse(1)");
        const Fq m1 = Fq::zero() - Fq::one();
        CHECK(Fs::from_mont256(m1).inverse().to_mont256() == m1, "inverse(-1)");
        Fq p2 = Fq::one();
        ok = true;
        for (int i = 0; i < 260; i++) {
            for (const Fq &x : {p2, p2 + Fq::one(), p2 - Fq::one()}) {
                if (x.is_zero()) continue;
                const Fs xi = Fs::from_mont256(x).inverse();
                ok = ok && xi.to_mont256() == x.inverse() && (xi * Fs::from_mont256(x)).to_mont256() == Fq::one();
            }
            p2 = p2 + p2;
        }
        CHECK(ok, "inverse, powers of two");
        const Fs loose = Fs::from_mont256(m1) + Fs::from_mont256(m1);          // a representative above p
        CHECK((loose.inverse() * loose).to_mont256() == Fq::one(), "inverse of a non-canonical representative");
    }
    for (int t = 0; t < 20; t++) {
        Fq12 a = rand12(), b = rand12();
        F12s A = to_s(a), B = to_s(b);
        CHECK(from_s(fq12_mul(A, B)) == fq12_mul(a, b), "fq12 mul");
        CHECK(from_s(fq12_sqr(A)) == fq12_sqr(a), "fq12 sqr");
        CHECK(from_s(fq12_frobenius<1>(A)) == fq12_frobenius<1>(a), "frob1");
        CHECK(from_s(fq12_frobenius<2>(A)) == fq12_frobenius<2>(a), "frob2");
        CHECK(from_s(fq12_frobenius<3>(A)) == fq12_frobenius<3>(a), "frob3");
        Fq2 e0 = {rand_fq(), rand_fq()}, e1 = {rand_fq(), rand_fq()}, e2 = {rand_fq(), rand_fq()};
        Fq2T<Fs> s0 = {Fs::from_mont256(e0.c0), Fs::from_mont256(e0.c1)}, s1 = {Fs::from_mont256(e1.c0), Fs::from_mont256(e1.c1)},
                 s2 = {Fs::from_mont256(e2.c0), Fs::from_mont256(e2.c1)};
        CHECK(from_s(fq12_mul_by_024(A, s0, s1, s2)) == fq12_mul_by_024(a, e0, e1, e2), "mul_by_024");
        if (t < 3) {
            CHECK(from_s(fq12_inverse(A)) == fq12_inverse(a), "fq12 inverse");
            // cyclotomic element: c = (conj(a)/a)^(q^2+1)
            Fq12 c = a.unitary_inverse() * fq12_inverse(a);
            c = fq12_frobenius<2>(c) * c;
            CHECK(fq12_cyclotomic_sqr(c) == fq12_sqr(c), "cyclotomic sqr (Fq)");
            CHECK(from_s(fq12_cyclotomic_sqr(to_s(c))) == fq12_sqr(c), "cyclotomic sqr (Fs)");
            CHECK(fq12_cyclotomic_pow_u64(c, 0x44e992b44a6909f1ull) == fq12_pow_u64(c, 0x44e992b44a6909f1ull), "cyclotomic pow");
            CHECK(c * c.unitary_inverse() == Fq12::one(), "unitary");
        }
    }
    printf(fails ? "FAILED (%d)\n" : "PASS\n", fails);
    return fails ? 1 : 0;
}
