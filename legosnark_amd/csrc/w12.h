// legosnark_amd/csrc/w12.h -- Fq12 arithmetic with ONE element spread over a wavefront.
//
// A final exponentiation (and a Miller loop) is a chain of ~10^4 dependent base-field products:
// one lane per pairing (pairing.hip, k_miller / k_final_exp) leaves the chip idle unless tens of
// thousands of pairings are in flight, and a single pairing takes >20 ms.  Here an Fq12 value
// is a polynomial sum_k a_k w^k, a_k in Fq2, w^6 = xi, kept in LDS; a product
//   c_k = sum_{i+j=k} a_i b_j + xi * sum_{i+j=k+6} a_i b_j
// is computed from its 36 partial products a_i b_j in parallel -- one Fq component of one partial
// product per lane in W12::mul (72 lanes, two wavefronts), one whole Fq2 product per lane in the
// Miller engines of miller.h -- followed by a 12-lane reduction along the anti-diagonals: one
// product + a handful of lazy additions of latency instead of 18 Fq2 products.  The polynomial
// basis is a permutation of libff's tower basis (c0.c0, c1.c0, c0.c1, c1.c1, c0.c2, c1.c2), and
// field values are canonical, so every result is bit-identical to the tower code's.
//
// The code is written against an executor X with  template<class F> void par(F f)  that runs
// f(lane) for its lanes and then synchronises, and  nlanes() : on the device X is the workgroup
// itself (f(threadIdx.x); __syncthreads()), in tests/cpp/test_w12.cc it is a loop over 64 lane
// ids, so the host tests run the very same sequence of phases.
#pragma once
#include "fp29x2.h"
#include "fs29.h"
#include "tower.h"

namespace lsa {

using Fq2S = Fq2T<Fs>;
using Fq12S = Fq12T<Fs>;

static constexpr int W12_SLOTS = 12;                     // Fq12 registers in LDS
static constexpr int W12_LDS_FQ2 = W12_SLOTS * 6 + 36;   // + the 36 partial products

// Fq2 product with two fused reductions (fp29x2.h: c0 = a0*b0 + a1*(20p - b1),
// c1 = a0*b1 + a1*b0, each accumulated in one set of 64-bit columns): ~600 instructions
// instead of ~1100 for Karatsuba on reduced values.  Operand contract of the whole engine:
// tight limbs, a's components < 4p, b's components < 20p (2*4*20 = 160 < 169).  [< 2p; tight]
LSA_HD Fq2S w12_fq2_mul(const Fq2S &a, const Fq2S &b) {
    F29x2 r = mul<20>(F29x2{a.c0.v, a.c1.v}, F29x2{b.c0.v, b.c1.v});
    return {Fs{r.c0}, Fs{r.c1}};
}

LSA_HD uint32_t w12_mask(uint32_t m) { return lsa_mask(m); }      // (fp29.h)
// The engine's registers live in LDS but are reached through generic pointers (the executor
// abstraction, host tests): every dereference then carries the null check of the generic -> LDS address
// cast (v_cmp_ne_u64 + v_cndmask_b32, 46 pairs in one W12::mul).  On the device the hot accessors below
// go through the LDS offset directly -- the low 32 bits of a generic address inside the LDS aperture.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) uint32_t w12_lds_u32;
__device__ __forceinline__ w12_lds_u32 *w12_lds(const void *p) { return (w12_lds_u32 *)(uint32_t)(uintptr_t)p; }
__device__ __forceinline__ Fs w12_load(const Fs *p) {
    Fs r;
    w12_lds_u32 *w = w12_lds(p);
#pragma unroll
    for (int i = 0; i < 9; i++) r.v.l[i] = w[i];
    return r;
}
__device__ __forceinline__ Fq2S w12_load(const Fq2S *p) { return {w12_load(&p->c0), w12_load(&p->c1)}; }
__device__ __forceinline__ void w12_store(Fs *p, const Fs &v) {
    w12_lds_u32 *w = w12_lds(p);
#pragma unroll
    for (int i = 0; i < 9; i++) w[i] = v.v.l[i];
}
__device__ __forceinline__ void w12_store(Fq2S *p, const Fq2S &v) { w12_store(&p->c0, v.c0); w12_store(&p->c1, v.c1); }
#else
inline void w12_store(Fq2S *p, const Fq2S &v) { *p = v; }
inline Fs w12_load(const Fs *p) { return *p; }
inline Fq2S w12_load(const Fq2S *p) { return *p; }
inline void w12_store(Fs *p, const Fs &v) { *p = v; }
#endif
// component `part` (0: c0, 1: c1) of an Fq2 value in memory, by address: a select between the two
// components costs 9 v_cndmask_b32 (each ~5x a plain VALU op on gfx950), an offset costs nothing
LSA_HD const Fs &w12_comp(const Fq2S &v, unsigned part) { return (&v.c0)[part]; }
LSA_HD Fs &w12_comp(Fq2S &v, unsigned part) { return (&v.c0)[part]; }
static_assert(sizeof(Fq2S) == 2 * sizeof(Fs), "Fq2S is two consecutive Fs");

// carry pass for limbs that are unsigned sums up to 2^32 - 8 (F29::norm takes signed limbs)
LSA_HD F29 w12_norm_u(const F29 &a) {
    F29 r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t v = (uint64_t)a.l[i] + c;
        r.l[i] = (uint32_t)v & F29::MASK;
        c = v >> 29;
    }
    r.l[8] = (uint32_t)((uint64_t)a.l[8] + c);
    return r;
}

// The anti-diagonal sums of a product, one Fq component per lane (lane = 2k + part, 12 lanes):
//   c_k = lo + xi*hi,  lo = sum_{i+j=k} P_ij,  hi = sum_{i+j=k+6} P_ij,  xi = 9 + u
//   part 0:  lo.c0 + 9 hi.c0 - hi.c1        part 1:  lo.c1 + 9 hi.c1 + hi.c0
// accumulated lazily on the 29-bit limbs (every P component is tight and < 2p, so lo < 12p,
// hi.* < 10p and the total < 112p) and brought back below 2p by ONE Montgomery product with
// the Montgomery form of 1: ~500 instructions instead of ~2000 for reduced Fq2 additions.
LSA_HD void w12_reduce_lane12(unsigned lane, const Fq2S *Pp, Fq2S *D) {
    const int k = (int)(lane >> 1);
    const unsigned part = lane & 1;
    const uint32_t pm = w12_mask(0u - part);             // all ones for the c1 lanes
    F29 lo = F29::zero(), hm = F29::zero(), ho = F29::zero();
    for (int i = 0; i <= k; i++) lo = add_lazy(lo, w12_comp(Pp[i * 6 + (k - i)], part).v);
    for (int i = k + 1; i <= 5; i++) {
        const Fq2S &t = Pp[i * 6 + (k + 6 - i)];
        hm = add_lazy(hm, w12_comp(t, part).v);          // the component that takes the factor 9
        ho = add_lazy(ho, w12_comp(t, part ^ 1u).v);     // the other one: -hi.c1 (part 0) / +hi.c0 (part 1)
    }
    lo = w12_norm_u(lo);
    hm = w12_norm_u(hm);
    ho = w12_norm_u(ho);
    F29 h8;
#pragma unroll
    for (int i = 0; i < 9; i++) h8.l[i] = hm.l[i] << 3;
    h8 = w12_norm_u(h8);
    const F29 neg = sub_k<10>(F29::zero(), ho);          // 10p - ho
    F29 sel;
#pragma unroll
    for (int i = 0; i < 9; i++) sel.l[i] = (ho.l[i] & pm) | (neg.l[i] & ~pm);
    F29 sum = w12_norm_u(add_lazy(add_lazy(add_lazy(h8, hm), lo), sel));   // < 112p
    Fs res = {mul(sum, F29::one())};
    w12_comp(D[k], part) = res;
}

// xi * t for t < 2 (tight):  (9 t0 - t1 + 2p,  9 t1 + t0)   [< 20; tight]
LSA_HD F29x2 w12_xi_times(const F29x2 &t) {
    F29 a8, b8;
#pragma unroll
    for (int i = 0; i < 9; i++) { a8.l[i] = t.c0.l[i] << 3; b8.l[i] = t.c1.l[i] << 3; }
    a8 = w12_norm_u(a8);
    b8 = w12_norm_u(b8);
    return {sub_k<2>(add_lazy(a8, t.c0), t.c1), add_lazy(add_lazy(b8, t.c1), t.c0).norm()};
}
// component `part` of a*b: part 0: a0*b0 + a1*(KB p - b1), part 1: a0*b1 + a1*b0; b's components
// < KB p and 2 * bound(a) * KB < 169.  [< 2p; tight]
template <int KB>
LSA_HD F29 w12_comp_mul(unsigned part, const Fq2S &a, const Fq2S &b) {
    const uint32_t pm = w12_mask(0u - part);
    const F29 nb1 = sub_k<KB>(F29::zero(), b.c1.v);
    F29 y0, y1;
#pragma unroll
    for (int l = 0; l < 9; l++) {
        y0.l[l] = (b.c1.v.l[l] & pm) | (b.c0.v.l[l] & ~pm);
        y1.l[l] = (b.c0.v.l[l] & pm) | (nb1.l[l] & ~pm);
    }
    return dot2(a.c0.v, y0, a.c1.v, y1);
}

// tower <-> polynomial basis: poly index k -> (which Fq6 half, which coefficient)
LSA_HD Fq2S &w12_tower_ref(Fq12S &t, int k) {
    Fq6T<Fs> &h = (k & 1) ? t.c1 : t.c0;
    return (k >> 1) == 0 ? h.c0 : ((k >> 1) == 1 ? h.c1 : h.c2);
}

template <class X>
struct W12 {
    X &x;
    Fq2S *R;   // W12_SLOTS x 6 coefficients
    Fq2S *P;   // 36 partial products

    LSA_HD Fq2S *slot(int s) const { return R + 6 * s; }

    // d = a * b   (d may alias a or b; both operands < 2p, as every value of a final
    // exponentiation is).  One Fq COMPONENT of one partial product per lane -- 72 tasks, one
    // fused two-product reduction (dot2) each, spread over the executor's lanes (two wavefronts
    // on the device) -- with the wrapped terms (i + j >= 6) taking xi*a_i as their a-operand, so
    // that coefficient k is the plain sum of six partial products and one reduction by a
    // Montgomery product with 1.  ~430 + ~350 instructions per lane instead of ~600 + ~500 for
    // whole Fq2 products and a xi step in the reduction.
    LSA_HD_NOINLINE void mul(int d, int a, int b) {
        Fq2S *A = slot(a), *B = slot(b), *D = slot(d), *Pp = P;
        const unsigned nl = x.nlanes();
        x.par([=](unsigned lane) {
            for (unsigned t = lane; t < 72; t += nl) {
                const unsigned i = t / 12, j = (t >> 1) % 6, part = t & 1;
                const Fq2S ai = w12_load(&A[i]);
                const F29x2 xa = w12_xi_times(F29x2{ai.c0.v, ai.c1.v});        // [< 20]
                const uint32_t wrap = w12_mask(0u - (uint32_t)(i + j >= 6));
                Fq2S asel;
#pragma unroll
                for (int l = 0; l < 9; l++) {
                    asel.c0.v.l[l] = (xa.c0.l[l] & wrap) | (ai.c0.v.l[l] & ~wrap);
                    asel.c1.v.l[l] = (xa.c1.l[l] & wrap) | (ai.c1.v.l[l] & ~wrap);
                }
                const Fs r = {w12_comp_mul<2>(part, asel, w12_load(&B[j]))};
                w12_store(&w12_comp(Pp[i * 6 + j], part), r);
            }
        });
        x.par([=](unsigned lane) {
            if (lane < 12) {
                const unsigned k = lane >> 1, part = lane & 1;
                F29 sum = F29::zero();
                for (int i = 0; i < 6; i++) sum = add_lazy(sum, w12_load(&w12_comp(Pp[i * 6 + ((int)k - i + 6) % 6], part)).v);
                const Fs r = {f29_mul(w12_norm_u(sum), F29::one())};          // < 12p -> < 2p
                w12_store(&w12_comp(D[k], part), r);
            }
        });
    }
    LSA_HD void sqr(int d, int a) { mul(d, a, a); }
    LSA_HD void copy(int d, int a) {
        Fq2S *A = slot(a), *D = slot(d);
        x.par([=](unsigned lane) { if (lane < 6) D[lane] = A[lane]; });
    }
    // unitary inverse = conjugation over Fq6: negate the odd powers of w
    LSA_HD void conj(int d, int a) {
        Fq2S *A = slot(a), *D = slot(d);
        x.par([=](unsigned lane) { if (lane < 6) D[lane] = (lane & 1) ? A[lane].neg() : A[lane]; });
    }
    LSA_HD Fq12S load_tower(int a) const {
        Fq12S t;
        for (int k = 0; k < 6; k++) w12_tower_ref(t, k) = slot(a)[k];
        return t;
    }
    LSA_HD void store_tower(int d, Fq12S t) const {
        for (int k = 0; k < 6; k++) slot(d)[k] = w12_tower_ref(t, k);
    }
    // rare operations run on lane 0 with the tower code (1 inversion, 5 Frobenius maps per
    // final exponentiation)
    LSA_HD void inverse(int d, int a) {
        W12 self = *this;
        x.par([=](unsigned lane) { if (lane == 0) self.store_tower(d, fq12_inverse(self.load_tower(a))); });
    }
    // d = a^-1 for a in the subfield Fq6 = Fq2[w^2] (odd coefficients zero, whatever their residues'
    // representatives say): the tower's Fq6 inversion on lane 0 (15 Fq2 products and one Fermat
    // inversion in Fq).  The final exponentiation inverts f as conj(f) * (f * conj(f))^-1 with it:
    // two products of the parallel engine instead of four Fq6 products on one lane.
    LSA_HD void inverse6(int d, int a) {
        Fq2S *A = slot(a), *D = slot(d);
        x.par([=](unsigned lane) {
            if (lane == 0) {
                const Fq6T<Fs> r = fq6_inverse(Fq6T<Fs>{A[0], A[2], A[4]});
                D[0] = r.c0; D[2] = r.c1; D[4] = r.c2;
                D[1] = Fq2S::zero(); D[3] = Fq2S::zero(); D[5] = Fq2S::zero();
            }
        });
    }
    // Frobenius map, one coefficient per lane: slot k is tower position (half k & 1, coefficient k >> 1),
    // so it becomes frob(a_k) * FROB6_C{k>>1} * FROB12_C1 (the factors the tower code applies, in the same
    // order: the same canonical value) -- at most two Fq2 products per lane instead of eleven on lane 0.
    template <int POWER>
    LSA_HD void frobenius(int d, int a) {
        Fq2S *A = slot(a), *D = slot(d);
        x.par([=](unsigned lane) {
            if (lane < 6) {
                Fq2S v = fq2_frobenius<POWER>(A[lane]);
                const unsigned j = lane >> 1;
                if (j) v = v * fq2_constT<Fs>(j == 1 ? LSA_FROB6_C1[POWER % 6] : LSA_FROB6_C2[POWER % 6]);
                if (lane & 1) v = v * fq2_constT<Fs>(LSA_FROB12_C1[POWER % 12]);
                D[lane] = v;
            }
        });
    }
    // d = a^e for a in the cyclotomic subgroup (plain squarings: with 36 lanes a general
    // squaring has the latency of one Fq2 product, the Granger-Scott shortcut buys nothing)
    LSA_HD void pow_u64(int d, int a, uint64_t e, int tmp) {
        bool started = false;
        for (int i = 63; i >= 0; --i) {
            if (started) sqr(tmp, tmp);
            if ((e >> i) & 1) {
                if (started) mul(tmp, tmp, a);
                else { copy(tmp, a); started = true; }
            }
        }
        copy(d, tmp);
    }
    // libff alt_bn128_exp_by_neg_z: d = conj(a^z) for a in the cyclotomic subgroup, z =
    // 0x44e992b44a6909f1.  Same value as libff's square-and-multiply, shorter chain: width-3 NAF of
    // z (digits 0, +-1, +-3; 18 non-zero of 63) -- inverses are conjugations here, so a^z costs
    // 62 squarings + 17 products + 2 for a^3 instead of 62 + 27.  Uses slots tmp, tmp+1, tmp+2.
    LSA_HD void exp_by_neg_z(int d, int a, int tmp) {
        static constexpr int8_t NAF3[63] = {1, 0, 0, 0, -1, 0, 0, 0, 0, -3, 0, 0, 1, 0, 0, 0, 1, 0, 0, -3, 0, 0, 0, -3, 0, 0, 3, 0, 0, 0, 1,
                                            0, 0, 0, -3, 0, 0, 0, 3, 0, 0, 1, 0, 0, 1, 0, 0, 3, 0, 0, 0, -3, 0, 0, 0, 0, -3, 0, 0, 1, 0, 0, 1};
        const int acc = tmp, a3 = tmp + 1, neg = tmp + 2;
        sqr(neg, a);
        mul(a3, neg, a);                    // a^3
        copy(acc, a);                       // top digit (bit 62) is +1
        for (int i = 61; i >= 0; --i) {
            sqr(acc, acc);
            const int dg = NAF3[i];
            if (dg == 1) mul(acc, acc, a);
            else if (dg == 3) mul(acc, acc, a3);
            else if (dg == -1) { conj(neg, a); mul(acc, acc, neg); }
            else if (dg == -3) { conj(neg, a3); mul(acc, acc, neg); }
        }
        conj(d, acc);
    }

    // libff alt_bn128_final_exponentiation on slot 0 -> slot 0 (same chain as final_exp_one in
    // pairing.hip).  Uses every slot.
    LSA_HD void final_exponentiation() {
        enum { ELT = 0, FIRST, A, B, C, D, E, F, G, T0, T1, T2 };
        conj(A, ELT);             // conj(f): f^(p^6)
        mul(C, ELT, A);           // f * conj(f), an element of Fq6
        inverse6(D, C);
        mul(B, A, D);             // f^-1 = conj(f) / (f * conj(f))
        mul(C, A, B);             // f^(p^6 - 1)
        frobenius<2>(D, C);
        mul(FIRST, D, C);
        exp_by_neg_z(A, FIRST, T0);
        sqr(B, A);
        sqr(C, B);
        mul(D, C, B);
        exp_by_neg_z(E, D, T0);
        sqr(F, E);
        exp_by_neg_z(G, F, T0);
        conj(T2, D);              // H
        conj(G, G);               // I
        mul(G, G, E);             // J = I * E
        mul(G, G, T2);            // K = J * H
        mul(T2, G, B);            // L = K * B
        mul(T0, G, E);            // M = K * E
        mul(T0, T0, FIRST);       // N = M * first
        frobenius<1>(T1, T2);     // O = frob1(L)
        mul(T0, T1, T0);          // P = O * N
        frobenius<2>(T1, G);      // Q = frob2(K)
        mul(T0, T1, T0);          // R = Q * P
        conj(T1, FIRST);          // S
        mul(T1, T1, T2);          // T = S * L
        frobenius<3>(T1, T1);     // U
        mul(ELT, T1, T0);         // U * R
    }
};

}  // namespace lsa
