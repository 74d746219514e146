// Host unit test of csrc/fp29.h (the 29-bit-limb field and its bound-aware point formulas)
// against csrc/fp.h + ec.h (saturated 32-bit limbs, canonical arithmetic), which in turn
// are pinned on the GPU against the oracle.  Build: g++ -std=c++17 -O2 -I legosnark_amd/csrc
#include <cstdio>
#include <cstdlib>
#include <random>
#include "ec.h"
#include "fp29.h"

using namespace lsa;
static std::mt19937_64 rng(12345);
static int fails = 0;
#define CHECK(c, msg) do { if (!(c)) { printf("FAIL %s (line %d)\n", msg, __LINE__); fails++; } } while (0)

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
static Aff<Fq> rand_point() {   // k*G by double-and-add from (1,2)
    Aff<Fq> g{Fq::from_u32(1), Fq::from_u32(2)};
    XYZZ<Fq> acc = XYZZ<Fq>::inf(), cur = XYZZ<Fq>::from_affine(g);
    uint64_t k = rng() | 1;
    for (int i = 0; i < 64; i++) { if ((k >> i) & 1) acc = xyzz_add(acc, cur); cur = xyzz_dbl(cur); }
    Fq zi = acc.ZZ.inverse(), zzi = acc.ZZZ.inverse();
    return {acc.X * zi, acc.Y * zzi};
}
static Aff29 to29(const Aff<Fq> &p) { if (p.is_inf()) return {F29::zero(), F29::zero()}; return {F29::from_mont256(p.x).canonical(), F29::from_mont256(p.y).canonical()}; }
static XYZZ29 to29(const XYZZ<Fq> &p) { if (p.is_inf()) return XYZZ29::inf(); return {F29::from_mont256(p.X), F29::from_mont256(p.Y), F29::from_mont256(p.ZZ), F29::from_mont256(p.ZZZ)}; }
static bool same(const XYZZ29 &a, const XYZZ<Fq> &b) {
    if (a.is_inf() || b.is_inf()) return a.is_inf() && b.is_inf();
    // compare affine: X/ZZ, Y/ZZZ
    Fq ax = a.X.to_mont256() * a.ZZ.to_mont256().inverse(), ay = a.Y.to_mont256() * a.ZZZ.to_mont256().inverse();
    Fq bx = b.X * b.ZZ.inverse(), by = b.Y * b.ZZZ.inverse();
    return ax == bx && ay == by;
}
static bool in_bounds(const XYZZ29 &a) {
    if (a.is_inf()) return true;
    for (int i = 0; i < 8; i++) if ((a.X.l[i] | a.Y.l[i] | a.ZZ.l[i] | a.ZZZ.l[i]) >> 29) return false;
    // top limb bounds: X<8p, Y<4p, ZZ,ZZZ<2p  (p >> 232 = 0x30644e)
    return a.X.l[8] <= 8 * 0x30644fu && a.Y.l[8] <= 4 * 0x30644fu && a.ZZ.l[8] <= 2 * 0x30644fu && a.ZZZ.l[8] <= 2 * 0x30644fu;
}

int main() {
    // field: roundtrip, mul, lazy add/sub chains
    for (int t = 0; t < 2000; t++) {
        Fq a = rand_fq(), b = rand_fq(), c = rand_fq();
        F29 A = F29::from_mont256(a), B = F29::from_mont256(b), C = F29::from_mont256(c);
        CHECK(A.to_mont256() == a, "roundtrip");
        CHECK(mul(A, B).to_mont256() == a * b, "mul");
        CHECK(sqr(A).to_mont256() == a.sqr(), "sqr");
        CHECK(add_lazy(A, B).norm().to_mont256() == a + b, "add");
        CHECK(sub_k<2>(A, B).to_mont256() == a - b, "sub2");
        CHECK(sub_k<8>(A, B).to_mont256() == a - b, "sub8");
        // operands near the documented limits: (A-B+8p) * (C - A + 4p) with loose limbs
        F29 X = sub_k<8>(A, B), Y = sub_k<4>(C, A);
        CHECK(mul(X, Y).to_mont256() == (a - b) * (c - a), "mul big operands");
        CHECK(mul(add_lazy(X, X), Y).to_mont256() == ((a - b) + (a - b)) * (c - a), "mul loose operand");
        CHECK(sub_k<2>(A, A).is_zero_mod_p(), "zero 2p");
        CHECK(sub_k<8>(A, A).is_zero_mod_p(), "zero 8p");
        CHECK(!sub_k<8>(A, B).is_zero_mod_p() || a == b, "nonzero");
        CHECK(A.canonical().to_mont256() == a, "canonical");
        uint32_t w[8]; F29 Ac = A.canonical(); Ac.pack256(w); F29 back = F29::unpack256(w);
        bool eq = true; for (int i = 0; i < 9; i++) eq = eq && back.l[i] == Ac.l[i];
        CHECK(eq, "pack/unpack");
    }
    CHECK(F29::one().to_mont256() == Fq::one(), "one");
    CHECK(F29::from_mont256(Fq::zero()).canonical().limbs_zero(), "zero");
    // points: long random chains of madd / add / dbl stay within the invariants and agree
    for (int t = 0; t < 20; t++) {
        XYZZ<Fq> ref = XYZZ<Fq>::inf();
        XYZZ29 acc = XYZZ29::inf();
        Aff<Fq> last = Aff<Fq>::inf();
        for (int s = 0; s < 60; s++) {
            Aff<Fq> pt = rand_point();
            int mode = (int)(rng() % 8);
            if (mode == 0 && !last.is_inf()) pt = last;                   // P + P (doubling inside madd)
            if (mode == 1 && !last.is_inf()) pt = last.neg();             // P + (-P) -> infinity or cancel
            if (mode == 2) pt = Aff<Fq>::inf();                           // infinity base
            if (mode == 3 && !ref.is_inf()) {                             // acc == pt exactly: doubling path
                Fq zi = ref.ZZ.inverse(), zzi = ref.ZZZ.inverse();
                pt = {ref.X * zi, ref.Y * zzi};
            }
            if (mode == 4 && !ref.is_inf()) {                             // acc == -pt: -> infinity
                Fq zi = ref.ZZ.inverse(), zzi = ref.ZZZ.inverse();
                pt = {ref.X * zi, (ref.Y * zzi).neg()};
            }
            ref = xyzz_madd(ref, pt);
            acc = xyzz29_madd(acc, to29(pt));
            CHECK(same(acc, ref), "madd chain");
            CHECK(in_bounds(acc), "madd bounds");
            if (mode == 5) { ref = xyzz_dbl(ref); acc = xyzz29_dbl(acc); CHECK(same(acc, ref), "dbl"); CHECK(in_bounds(acc), "dbl bounds"); }
            if (mode == 6) { XYZZ<Fq> o = xyzz_madd(XYZZ<Fq>::inf(), rand_point()); o = xyzz_dbl(o);
                             ref = xyzz_add(ref, o); acc = xyzz29_add(acc, to29(o)); CHECK(same(acc, ref), "add"); CHECK(in_bounds(acc), "add bounds"); }
            if (mode == 7) { ref = xyzz_add(ref, ref); acc = xyzz29_add(acc, acc); CHECK(same(acc, ref), "add self"); }
            last = pt;
        }
        // neg + add -> infinity; to_jac
        XYZZ29 n = xyzz29_neg(acc);
        CHECK(xyzz29_add(acc, n).is_inf(), "a + (-a)");
        Jac<Fq> j = xyzz29_to_jac(acc), jr = xyzz_to_jac(ref);
        CHECK(jac_eq(j, jr), "to_jac");
    }
    printf(fails ? "FAILED (%d)\n" : "PASS\n", fails);
    return fails ? 1 : 0;
}
