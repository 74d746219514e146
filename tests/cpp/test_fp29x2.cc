// Host unit test of csrc/fp29x2.h (Fq2 on 29-bit limbs + G2 XYZZ formulas) against the
// canonical generic code (fp.h Fq2 + ec.h XYZZ<Fq2>).
#include <cstdio>
#include <random>
#include "ec.h"
#include "tower.h"
#include "fp29x2.h"
#include "fs29.h"
using namespace lsa;
static std::mt19937_64 rng(4242);
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
static Fq2 rand_fq2() { return {rand_fq(), rand_fq()}; }
static Aff<Fq2> gen() { return {fq2_const(LSA_G2_GEN_X), fq2_const(LSA_G2_GEN_Y)}; }
static Aff<Fq2> rand_point() {
    XYZZ<Fq2> acc = XYZZ<Fq2>::inf(), cur = XYZZ<Fq2>::from_affine(gen());
    uint64_t k = rng() | 1;
    for (int i = 0; i < 40; i++) { if ((k >> i) & 1) acc = xyzz_add(acc, cur); cur = xyzz_dbl(cur); }
    Fq2 zi = acc.ZZ.inverse(), zzi = acc.ZZZ.inverse();
    return {acc.X * zi, acc.Y * zzi};
}
static Aff29x2 to29(const Aff<Fq2> &p) { return unpack_affine(pack_affine_g2(p)); }
static XYZZ29x2 to29(const XYZZ<Fq2> &p) {
    if (p.is_inf()) return XYZZ29x2::inf();
    return {f29x2_from_mont256(p.X), f29x2_from_mont256(p.Y), f29x2_from_mont256(p.ZZ), f29x2_from_mont256(p.ZZZ)};
}
static bool same(const XYZZ29x2 &a, const XYZZ<Fq2> &b) {
    if (a.is_inf() || b.is_inf()) return a.is_inf() && b.is_inf();
    Fq2 ax = f29x2_to_mont256(a.X) * f29x2_to_mont256(a.ZZ).inverse(), ay = f29x2_to_mont256(a.Y) * f29x2_to_mont256(a.ZZZ).inverse();
    Fq2 bx = b.X * b.ZZ.inverse(), by = b.Y * b.ZZZ.inverse();
    return ax == bx && ay == by;
}
static bool comp_ok(const F29 &v, uint32_t mult) { for (int i = 0; i < 8; i++) if (v.l[i] >> 29) return false; return v.l[8] <= mult * 0x30644fu; }
static bool in_bounds(const XYZZ29x2 &a) {
    if (a.is_inf()) return true;
    return comp_ok(a.X.c0, 4) && comp_ok(a.X.c1, 4) && comp_ok(a.Y.c0, 4) && comp_ok(a.Y.c1, 4) &&
           comp_ok(a.ZZ.c0, 2) && comp_ok(a.ZZ.c1, 2) && comp_ok(a.ZZZ.c0, 2) && comp_ok(a.ZZZ.c1, 2);
}
int main() {
    for (int t = 0; t < 1000; t++) {
        Fq2 a = rand_fq2(), b = rand_fq2(), c = rand_fq2();
        F29x2 A = f29x2_from_mont256(a), B = f29x2_from_mont256(b), C = f29x2_from_mont256(c);
        CHECK(f29x2_to_mont256(mul<2>(A, B)) == a * b, "mul");
        CHECK(f29x2_to_mont256(sqr<2>(A)) == a.sqr(), "sqr");
        F29x2 X = sub_k<4>(A, B), Y = sub_k<4>(C, A);       // components < 6p
        CHECK(f29x2_to_mont256(mul<6>(X, Y)) == (a - b) * (c - a), "mul big");
        CHECK(f29x2_to_mont256(sqr<6>(X)) == (a - b).sqr(), "sqr big");
        CHECK(f29x2_to_mont256(condsub4(sub_k<6>(A, B))) == a - b, "condsub4");
        CHECK(sub_k<4>(A, A).is_zero_mod_p() && !X.is_zero_mod_p(), "zero test");
        // f29x2_inverse (fs29.h; the one inversion of k_prepare_g2 / k_cmp_build_table_g2): a * a^-1 = 1, and the value libff's
        // Fq2 inverse gives
        if (t < 200) {
            const F29x2 I = f29x2_inverse(mul<2>(A, F29x2::one()));      // a tight operand below 2p
            CHECK(f29x2_to_mont256(I) == a.inverse(), "inverse");
            CHECK(f29x2_to_mont256(mul<2>(I, A)) == Fq2::one(), "a * a^-1");
        }
    }
    {
        Fq2 pure_real = {rand_fq(), Fq::zero()}, pure_imag = {Fq::zero(), rand_fq()};
        CHECK(f29x2_to_mont256(f29x2_inverse(f29x2_from_mont256(pure_real))) == pure_real.inverse(), "inverse (c1 = 0)");
        CHECK(f29x2_to_mont256(f29x2_inverse(f29x2_from_mont256(pure_imag))) == pure_imag.inverse(), "inverse (c0 = 0)");
    }
    for (int t = 0; t < 12; t++) {
        XYZZ<Fq2> ref = XYZZ<Fq2>::inf();
        XYZZ29x2 acc = XYZZ29x2::inf();
        Aff<Fq2> last = Aff<Fq2>::inf();
        for (int s = 0; s < 50; s++) {
            Aff<Fq2> pt = rand_point();
            int mode = (int)(rng() % 8);
            if (mode == 0 && !last.is_inf()) pt = last;
            if (mode == 1 && !last.is_inf()) pt = last.neg();
            if (mode == 2) pt = Aff<Fq2>::inf();
            if (mode == 3 && !ref.is_inf()) { Fq2 zi = ref.ZZ.inverse(), zzi = ref.ZZZ.inverse(); pt = {ref.X * zi, ref.Y * zzi}; }
            if (mode == 4 && !ref.is_inf()) { Fq2 zi = ref.ZZ.inverse(), zzi = ref.ZZZ.inverse(); pt = {ref.X * zi, (ref.Y * zzi).neg()}; }
            ref = xyzz_madd(ref, pt);
            acc = g2_madd(acc, to29(pt));
            CHECK(same(acc, ref), "madd chain"); CHECK(in_bounds(acc), "madd bounds");
            if (mode == 5) { ref = xyzz_dbl(ref); acc = g2_dbl(acc); CHECK(same(acc, ref), "dbl"); CHECK(in_bounds(acc), "dbl bounds"); }
            if (mode == 6) { XYZZ<Fq2> o = xyzz_dbl(xyzz_madd(XYZZ<Fq2>::inf(), rand_point()));
                             ref = xyzz_add(ref, o); acc = g2_add(acc, to29(o)); CHECK(same(acc, ref), "add"); CHECK(in_bounds(acc), "add bounds"); }
            if (mode == 7) { ref = xyzz_add(ref, ref); acc = g2_add(acc, acc); CHECK(same(acc, ref), "add self"); CHECK(in_bounds(acc), "add self bounds"); }
            last = pt;
        }
        Jac<Fq2> j = g2_to_jac(acc), jr = xyzz_to_jac(ref);
        CHECK(jac_eq(j, jr), "to_jac");
        // negated base: p - y
        Aff<Fq2> pt = rand_point();
        Aff29x2 q = to29(pt);
        q.y = sub_k<1>(F29x2::zero(), q.y);
        CHECK(same(g2_madd(acc, q), xyzz_madd(ref, pt.neg())), "negated base");
    }
    printf(fails ? "FAILED (%d)\n" : "PASS\n", fails);
    return fails ? 1 : 0;
}
