// legosnark_amd/csrc/w12.h -- Fq12 arithmetic with ONE element spread over a wavefront.
//
// A final exponentiation (and a Miller loop) is a chain of ~10^4 dependent base-field products:
// one lane per pairing (pairing.hip, k_miller / k_final_exp) leaves the chip idle unless tens of
// thousands of pairings are in flight, and a single pairing takes >20 ms.  Here an Fq12 value
// is a polynomial sum_k a_k w^k, a_k in Fq2, w^6 = xi, kept in LDS; a product is computed by 36
// lanes (lane 6i+j: a_i * b_j in Fq2) followed by a 6-lane reduction along the anti-diagonals
//   c_k = sum_{i+j=k} a_i b_j + xi * sum_{i+j=k+6} a_i b_j,
// i.e. one Fq2 product + ~7 Fq2 additions of latency instead of 18 Fq2 products.  The polynomial
// basis is a permutation of libff's tower basis (c0.c0, c1.c0, c0.c1, c1.c1, c0.c2, c1.c2), and
// field values are canonical, so every result is bit-identical to the tower code's.
//
// The code is written against an executor X with  template<class F> void par(F f)  that runs
// f(lane) for the 64 lanes and then synchronises: on the device X is the wavefront itself
// (f(threadIdx.x); __syncthreads()), in tests/cpp/test_w12.cc it is a loop over lane ids, so
// the host tests run the very same sequence of phases.
#pragma once
#include "fs29.h"
#include "tower.h"

namespace lsa {

using Fq2S = Fq2T<Fs>;
using Fq12S = Fq12T<Fs>;

static constexpr int W12_SLOTS = 12;                     // Fq12 registers in LDS
static constexpr int W12_LDS_FQ2 = W12_SLOTS * 6 + 36;   // + the 36 partial products

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

    // d = a * b   (d may alias a or b)
    LSA_HD_NOINLINE void mul(int d, int a, int b) {
        Fq2S *A = slot(a), *B = slot(b), *D = slot(d), *Pp = P;
        x.par([=](unsigned lane) {
            if (lane < 36) Pp[lane] = A[lane / 6] * B[lane % 6];
        });
        x.par([=](unsigned lane) {
            if (lane < 6) {
                const int k = (int)lane;
                Fq2S lo = Pp[k];                                  // i = 0, j = k
                for (int i = 1; i <= k; i++) lo = lo + Pp[i * 6 + (k - i)];
                if (k < 5) {
                    Fq2S hi = Pp[(k + 1) * 6 + 5];                // i + j = k + 6
                    for (int i = k + 2; i <= 5; i++) hi = hi + Pp[i * 6 + (k + 6 - i)];
                    lo = lo + hi.mul_xi();
                }
                D[k] = lo;
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
    template <int POWER>
    LSA_HD void frobenius(int d, int a) {
        W12 self = *this;
        x.par([=](unsigned lane) { if (lane == 0) self.store_tower(d, fq12_frobenius<POWER>(self.load_tower(a))); });
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
    // libff alt_bn128_exp_by_neg_z
    LSA_HD void exp_by_neg_z(int d, int a, int tmp) {
        pow_u64(tmp, a, LSA_FINAL_EXP_Z, tmp + 1);
        conj(d, tmp);
    }

    // libff alt_bn128_final_exponentiation on slot 0 -> slot 0 (same chain as final_exp_one in
    // pairing.hip).  Uses every slot.
    LSA_HD void final_exponentiation() {
        enum { ELT = 0, FIRST, A, B, C, D, E, F, G, T0, T1, T2 };
        conj(A, ELT);
        inverse(B, ELT);
        mul(C, A, B);
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
