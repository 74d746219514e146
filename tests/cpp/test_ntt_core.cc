// Host unit test of csrc/ntt_core.h + csrc/fr29.h: the three-pass NTT of ntt.hip run tile by tile with the SAME
// per-element functions the kernel calls (load / butterfly / store), on 29-bit-limb Fr arithmetic, against
//   (1) the DFT definition X[k] = sum_i a[i] omega^(ik) with fp.h's canonical 32-bit-limb Fr (n <= 2^9), and
//   (2) a plain iterative radix-2 transform on fp.h's Fr (every size),
// for forward / inverse / coset / inverse-coset transforms (libfqfft FFT, iFFT, cosetFFT, icosetFFT as used by
// /root/reference/src/gadgets/lipmaa.cc:102-168), with the library's 2^10-element tiles and with smaller tiles so that
// small transforms exercise all three passes.  Also checks fr29.h's product / sum / difference against fp.h.
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "fp.h"
#include "ntt_core.h"
using namespace lsa;

static std::mt19937_64 rng(2026);
static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails < 20) { printf("FAIL "); printf(__VA_ARGS__); printf(" (line %d)\n", __LINE__); } fails++; } } while (0)

static Fr rand_fr() {
    for (;;) {
        Fr x;
        for (int i = 0; i < 4; i++) { uint64_t v = rng(); x.l[2 * i] = (uint32_t)v; x.l[2 * i + 1] = (uint32_t)(v >> 32); }
        x.l[7] &= 0x3fffffffu;
        bool lt = false;
        for (int i = 7; i >= 0; --i) if (x.l[i] != FrParams::MOD[i]) { lt = x.l[i] < FrParams::MOD[i]; break; }
        if (lt) return x;
    }
}
static Fr fr_pow(Fr b, uint64_t e) {
    Fr acc = Fr::one();
    for (int i = 63; i >= 0; --i) { acc = acc * acc; if ((e >> i) & 1) acc = acc * b; }
    return acc;
}
// primitive 2^L-th root of unity: 5^((r - 1) / 2^28) squared down
static Fr root_of_unity(unsigned L) {
    // (r - 1) >> 28 as 8 x 32-bit
    uint32_t e[8];
    uint64_t lo = 0;
    uint32_t rm1[8];
    for (int i = 0; i < 8; i++) rm1[i] = FrParams::MOD[i];
    rm1[0] -= 1;
    for (int i = 0; i < 8; i++) { lo = ((uint64_t)rm1[i] >> 28) | ((uint64_t)(i + 1 < 8 ? rm1[i + 1] : 0) << 4); e[i] = (uint32_t)lo; }
    Fr acc = Fr::one(), five = Fr::from_u32(5);
    for (int i = 255; i >= 0; --i) { acc = acc * acc; if ((e[i >> 5] >> (i & 31)) & 1) acc = acc * five; }
    for (unsigned i = 28; i > L; --i) acc = acc * acc;
    return acc;
}
// value * 32 in Montgomery form = the canonical words of the 2^261 form
static Fr to261(const Fr &x) { return x * Fr::from_u32(32); }

static void reference_radix2(std::vector<Fr> &a, unsigned L, Fr w) {
    const size_t n = (size_t)1 << L;
    for (size_t i = 0; i < n; i++) { size_t j = ntt_brev((unsigned)i, L); if (i < j) std::swap(a[i], a[j]); }
    for (unsigned s = 0; s < L; s++) {
        const size_t hl = (size_t)1 << s;
        const Fr wm = fr_pow(w, n >> (s + 1));
        for (size_t k = 0; k < n; k += 2 * hl) {
            Fr t = Fr::one();
            for (size_t j = 0; j < hl; j++) {
                const Fr u = a[k + j], v = a[k + j + hl] * t;
                a[k + j] = u + v;
                a[k + j + hl] = u - v;
                t = t * wm;
            }
        }
    }
}

struct Tables {
    std::vector<Fr> W, Tlo, Thi, T2, T1, Glo, Ghi, Gfull;
    Fr cst;
    unsigned gh;
};
static std::vector<Fr> pow_table(Fr base, Fr c0, size_t count) {
    std::vector<Fr> t(count);
    Fr x = c0;
    for (size_t i = 0; i < count; i++) { t[i] = x; x = x * base; }
    return t;
}
// the transform ntt.hip's host code sets up: w = omega or omega^-1; coset: g (forward, on load) / g^-1 with 1/n (inverse, on store)
static void run_three_pass(std::vector<Fr> &a, unsigned L, Fr omega, bool inverse, const Fr *coset, unsigned tile_log) {
    const size_t n = (size_t)1 << L;
    const NttPlan p = ntt_plan(L, tile_log);
    const Fr w = inverse ? omega.inverse() : omega;
    const Fr c32 = Fr::from_u32(32);
    Tables t;
    t.W = pow_table(fr_pow(w, n >> p.lmax), c32, (size_t)1 << (p.lmax - 1));
    t.Tlo = pow_table(w, c32, (size_t)1 << p.h);
    t.Thi = pow_table(fr_pow(w, (uint64_t)1 << p.h), c32, (size_t)1 << (p.L - p.h));
    Fr ninv = Fr::one();
    if (inverse) ninv = fr_pow(Fr::from_u32(2), L).inverse();
    t.cst = to261(ninv);
    if (p.l2 && (tile_log & 1)) {                     // odd tile sizes run pass 2 from its table, even ones through the two-level look-up
        const Fr w1 = fr_pow(w, (uint64_t)1 << p.l1);
        t.T2.resize((size_t)1 << (p.l2 + p.l3));
        for (uint64_t k2 = 0; k2 < (1ull << p.l2); k2++)
            for (uint64_t i3 = 0; i3 < (1ull << p.l3); i3++) t.T2[(k2 << p.l3) + i3] = to261(fr_pow(w1, i3 * k2));
    }
    if (p.l1 && (tile_log % 3) != 0) {                // two of three tile sizes run pass 1 from its full table
        t.T1.resize(n);
        const uint64_t m = 1ull << (p.L - p.l1);
        for (uint64_t k1 = 0; k1 < (1ull << p.l1); k1++)
            for (uint64_t col = 0; col < m; col++) t.T1[k1 * m + col] = to261(fr_pow(w, col * k1));
    }
    t.gh = p.h;
    if (coset) {
        const Fr g = inverse ? coset->inverse() : *coset;
        t.Glo = pow_table(g, c32, (size_t)1 << t.gh);
        t.Ghi = pow_table(fr_pow(g, (uint64_t)1 << t.gh), c32 * ninv, (size_t)1 << (p.L - t.gh));
        if ((tile_log % 3) == 1) t.Gfull = pow_table(g, c32 * ninv, n);      // one of three tile sizes takes the coset powers from the full table
    }
    std::vector<Fr> scratch(n);
    std::vector<unsigned> kinds;
    if (p.l1) kinds.push_back(1);
    if (p.l2) kinds.push_back(2);
    kinds.push_back(3);
    const Fr *src = a.data();
    for (size_t pi = 0; pi < kinds.size(); pi++) {
        NttArgs args = {};
        args.plan = p;
        args.pass = ntt_pass(p, kinds[pi]);
        // ping-pong as the library does: first pass a -> scratch, middle in place, last -> a (a single pass: a -> scratch, copied back)
        Fr *dst = (pi + 1 == kinds.size() && kinds.size() > 1) ? a.data() : scratch.data();
        args.src = src;
        args.dst = dst;
        std::vector<uint32_t> W9(t.W.size() * 9);
        for (size_t j = 0; j < t.W.size(); j++) ntt_lds_put(W9.data(), (unsigned)j, Fr29::from_words(t.W[j]));
        args.W = W9.data(); args.Tlo = t.Tlo.data(); args.Thi = t.Thi.data();
        args.T2 = t.T2.empty() ? nullptr : t.T2.data();
        args.T1 = t.T1.empty() ? nullptr : t.T1.data();
        args.Glo = coset ? t.Glo.data() : nullptr; args.Ghi = coset ? t.Ghi.data() : nullptr;
        args.Gfull = (coset && !t.Gfull.empty()) ? t.Gfull.data() : nullptr;
        args.gh = t.gh;
        args.pre_scale = (pi == 0 && coset && !inverse) ? 1 : 0;
        args.post_scale = (pi + 1 == kinds.size() && coset && inverse) ? 1 : 0;
        args.cst = t.cst;
        const unsigned T = 1u << (args.pass.l + args.pass.logC);
        std::vector<uint32_t> lds(ntt_tile_words(args.pass));
        for (unsigned w_ = 0; w_ < args.pass.tiles; w_++) {
            for (unsigned x = 0; x < T; x++) ntt_tile_load(args, w_, x, lds.data());
            // the kernel's schedule: an odd number of stages starts with a single one, then two stages per LDS round trip;
            // tiles of even size also run the plain one-stage-at-a-time schedule (the two must agree: checked through the result)
            if (tile_log & 1) {
                unsigned s = 0;
                if (args.pass.l & 1u) { for (unsigned b = 0; b < T / 2; b++) ntt_tile_stage(args, 0, b, lds.data()); s = 1; }
                for (; s < args.pass.l; s += 2)
                    for (unsigned g4 = 0; g4 < T / 4; g4++) ntt_tile_stage2(args, s, g4, lds.data());
            } else if (args.pass.l >= 2 && !(args.pass.l & 1u)) {
                for (unsigned s = 0; s < args.pass.l; s += 2)
                    for (unsigned g4 = 0; g4 < T / 4; g4++) ntt_tile_stage2(args, s, g4, lds.data());
            } else {
                for (unsigned s = 0; s < args.pass.l; s++)
                    for (unsigned b = 0; b < T / 2; b++) ntt_tile_stage(args, s, b, lds.data());
            }
            for (unsigned x = 0; x < T; x++) ntt_tile_store(args, w_, x, lds.data());
        }
        src = dst;
    }
    if (kinds.size() == 1) a = scratch;
}

int main() {
    // ---- fr29.h against fp.h
    for (int t = 0; t < 2000; t++) {
        const Fr a = rand_fr(), b = rand_fr(), c = rand_fr();
        const Fr29 A = Fr29::from_words(a), B = Fr29::from_words(to261(b)), C = Fr29::from_words(c);
        // (a viewed in the shifted form) * (b in 2^261 form) = a * b in the shifted form
        CHECK(memcmp(mul(A, B).canonical2().to_words().l, (a * b).l, 32) == 0, "fr29 mul");
        CHECK(memcmp(mul(add(A, C), B).canonical2().to_words().l, ((a + c) * b).l, 32) == 0, "fr29 add");
        const Fr29 D = sub2r(A, mul(C, Fr29::one()));                       // a - c + 2r, then reduced by a product with 1
        CHECK(memcmp(mul(D, Fr29::one()).canonical2().to_words().l, (a - c).l, 32) == 0, "fr29 sub2r");
        // growth: twelve lazy sums stay multipliable
        Fr29 S = A;
        Fr ref = a;
        for (int i = 0; i < 12; i++) { S = add(S, mul(C, Fr29::one())); ref = ref + c; }
        CHECK(memcmp(mul(S, B).canonical2().to_words().l, (ref * b).l, 32) == 0, "fr29 lazy sums");
    }
    // sums of up to four products with one reduction (fr29_wide_*): the same residue as four reduced products added up,
    // also on the largest operands the bounds allow (all limbs at their maximum: 2^29 - 1, the value 2^261 - 1 > r)
    for (int t = 0; t < 2000; t++) {
        Fr a[4], b[4];
        Fr want = Fr::zero();
        Fr29Wide w = fr29_wide_zero();
        const int terms = 1 + t % 4;
        for (int q = 0; q < terms; q++) {
            a[q] = rand_fr(); b[q] = rand_fr();
            fr29_wide_mac(w, Fr29::from_words(a[q]), Fr29::from_words(to261(b[q])));
            want = want + a[q] * b[q];
        }
        CHECK(memcmp(fr29_wide_reduce(w).canonical2().to_words().l, want.l, 32) == 0, "fr29 wide sum of products");
    }
    {
        Fr29 top;
        for (int i = 0; i < 9; i++) top.l[i] = Fr29::MASK;
        Fr29Wide w = fr29_wide_zero();
        for (int q = 0; q < 4; q++) fr29_wide_mac(w, top, Fr29::from_words(to261(Fr::one())));       // 4 * (2^261 - 1) * 1
        const Fr29 got = fr29_wide_reduce(w);
        const Fr29 one_term = mul(top, Fr29::from_words(to261(Fr::one())));
        const Fr29 four = add(add(one_term, one_term), add(one_term, one_term));
        CHECK(memcmp(mul(got, Fr29::one()).canonical2().to_words().l, mul(four, Fr29::one()).canonical2().to_words().l, 32) == 0, "fr29 wide on full limbs");
    }
    {
        Fr one_w;
        Fr29::one().pack256(one_w.l);
        CHECK(memcmp(one_w.l, to261(Fr::one()).l, 32) == 0, "fr29 one = 2^261 mod r");
    }
    // ---- plans
    for (unsigned L = 1; L <= 28; L++) {
        const NttPlan p = ntt_plan(L);
        CHECK(p.l1 + p.l2 + p.l3 == L && p.l3 >= 1 && p.lmax <= 10, "plan %u: %u %u %u", L, p.l1, p.l2, p.l3);
        for (unsigned kind = 1; kind <= 3; kind++) {
            if ((kind == 1 && !p.l1) || (kind == 2 && !p.l2)) continue;
            const NttPass q = ntt_pass(p, kind);
            CHECK(((uint64_t)q.tiles << (q.l + q.logC)) == ((uint64_t)1 << L), "pass %u of %u covers every element once", kind, L);
            if (L >= 17 && L <= 24) CHECK(q.l + q.logC == NTT_TILE_LOG && q.logC >= 2, "plan %u pass %u: runs of >= 128 bytes", L, kind);
        }
        if (L >= 17) CHECK(p.l1 && p.l2, "three passes from 2^17 on");
        if (L <= 10) CHECK(!p.l1 && !p.l2, "one pass up to 2^10");
    }
    // ---- transforms
    const Fr g = Fr::from_u32(5);                                           // FieldT::multiplicative_generator (lipmaa.cc:138)
    struct Case { unsigned L, tile_log; };
    const Case cases[] = {{1, 10}, {2, 10}, {3, 10}, {5, 10}, {8, 10}, {9, 10}, {10, 10}, {11, 10}, {12, 10}, {13, 10},
                          {3, 4}, {4, 4}, {5, 4}, {6, 4}, {7, 4}, {8, 4}, {9, 4}, {10, 4}, {11, 4}, {12, 4}, {7, 3}, {9, 3}, {13, 5}, {14, 6}};
    for (const Case &cs : cases) {
        const unsigned L = cs.L;
        const size_t n = (size_t)1 << L;
        const Fr omega = root_of_unity(L);
        CHECK(fr_pow(omega, n) == Fr::one() && (L == 0 || fr_pow(omega, n / 2) != Fr::one()), "root of unity 2^%u", L);
        std::vector<Fr> a(n);
        for (auto &x : a) x = rand_fr();
        a[0] = Fr::zero();
        if (n > 2) a[2] = Fr::one();
        for (int variant = 0; variant < 4; variant++) {
            const bool inverse = variant & 1, coset = variant & 2;
            std::vector<Fr> got = a, want = a;
            run_three_pass(got, L, omega, inverse, coset ? &g : nullptr, cs.tile_log);
            // reference: libfqfft's order of operations
            if (coset && !inverse) { Fr x = Fr::one(); for (size_t i = 0; i < n; i++) { want[i] = want[i] * x; x = x * g; } }
            reference_radix2(want, L, inverse ? omega.inverse() : omega);
            if (inverse) {
                const Fr ninv = fr_pow(Fr::from_u32(2), L).inverse(), gi = g.inverse();
                Fr x = ninv;
                for (size_t i = 0; i < n; i++) { want[i] = want[i] * x; if (coset) x = x * gi; }
            }
            bool same = true;
            for (size_t i = 0; i < n && same; i++) same = memcmp(got[i].l, want[i].l, 32) == 0;
            CHECK(same, "L=%u tile=%u variant=%d", L, cs.tile_log, variant);
            if (variant == 0 && L <= 9) {                                   // the definition itself
                bool ok = true;
                for (size_t k = 0; k < n && ok; k++) {
                    Fr acc = Fr::zero();
                    const Fr wk = fr_pow(omega, k);
                    Fr x = Fr::one();
                    for (size_t i = 0; i < n; i++) { acc = acc + a[i] * x; x = x * wk; }
                    ok = memcmp(acc.l, got[k].l, 32) == 0;
                }
                CHECK(ok, "L=%u tile=%u against the DFT definition", L, cs.tile_log);
            }
        }
        // round trip
        std::vector<Fr> rt = a;
        run_three_pass(rt, L, omega, false, &g, cs.tile_log);
        run_three_pass(rt, L, omega, true, &g, cs.tile_log);
        bool same = true;
        for (size_t i = 0; i < n && same; i++) same = memcmp(rt[i].l, a[i].l, 32) == 0;
        CHECK(same, "L=%u tile=%u coset round trip", L, cs.tile_log);
    }
    printf(fails ? "FAILED (%d)\n" : "PASS\n", fails);
    return fails ? 1 : 0;
}
