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
#include "frob_rows.h"

namespace lsa {

using Fq2S = Fq2T<Fs>;
using Fq12S = Fq12T<Fs>;

static constexpr int W12_SLOTS = 15;                     // Fq12 registers in LDS
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
#if defined(LSA_F29_COLS)
    return f29_dot2_cols(a.c0.v, y0, a.c1.v, y1);
#endif
    return dot2(a.c0.v, y0, a.c1.v, y1);
}

// the same with the bound of b's components, K p, chosen at run time (tmiller.h: G2Pre keeps bounds per product)
LSA_HD F29 w12_comp_mul_k(unsigned part, const Fq2S &a, const Fq2S &b, int K) {
    const uint32_t pm = w12_mask(0u - part);
    const F29 nb1 = lin2(b.c1.v, -1, F29::zero(), 0, K);
    F29 y0, y1;
#pragma unroll
    for (int l = 0; l < 9; l++) {
        y0.l[l] = (b.c1.v.l[l] & pm) | (b.c0.v.l[l] & ~pm);
        y1.l[l] = (b.c0.v.l[l] & pm) | (nb1.l[l] & ~pm);
    }
    return dot2(a.c0.v, y0, a.c1.v, y1);
}

// The same with K p - b1 taken WITHOUT a carry chain (round 6, second session; lin2 above: three 64-bit multiply-adds, a mask and a
// 64-bit shift per limb, ~60 instructions; here 12): `lift` holds K p with its limbs 0..6 "lifted" -- limb 0 plus 2^29, limbs 1..6
// plus 2^29 - 1, limb 7 minus 1: the same integer, each limb lending the 2^29 of its lower neighbour -- so that for a tight b1
// limbs 0..6 of the difference are lift_i - b1_i >= 0 with no borrow between them (below 2^30: "loose"); only limbs 7 and 8 run
// a borrow.  The value is K p - b1 exactly, as before; dot2 accepts one loose factor per product (a column: 9 * 2^58 + 9 * 2^59 +
// 9 * 2^58 < 2^64).  lift: nine words (device: in LDS).
LSA_HD F29 w12_comp_mul_lift(unsigned part, const Fq2S &a, const Fq2S &b, const uint32_t *lift) {
    const uint32_t pm = w12_mask(0u - part);
    uint32_t lw[9];
#if defined(__HIP_DEVICE_COMPILE__)
    w12_lds_u32 *lp = w12_lds(lift);
#pragma unroll
    for (int l = 0; l < 9; l++) lw[l] = lp[l];
#else
    for (int l = 0; l < 9; l++) lw[l] = lift[l];
#endif
    F29 nb1;
#pragma unroll
    for (int l = 0; l < 7; l++) nb1.l[l] = lw[l] - b.c1.v.l[l];
    const int32_t v7 = (int32_t)lw[7] - (int32_t)b.c1.v.l[7];
    nb1.l[7] = (uint32_t)v7 & F29::MASK;
    nb1.l[8] = (uint32_t)((int32_t)lw[8] - (int32_t)b.c1.v.l[8] + (v7 >> 29));
    F29 y0, y1;
#pragma unroll
    for (int l = 0; l < 9; l++) {
        y0.l[l] = (b.c1.v.l[l] & pm) | (b.c0.v.l[l] & ~pm);
        y1.l[l] = (b.c0.v.l[l] & pm) | (nb1.l[l] & ~pm);
    }
    return dot2(a.c0.v, y0, a.c1.v, y1);
}
// the nine words of K p for w12_comp_mul_lift
LSA_HD void w12_lift_kp(int K, uint32_t out[9]) {
    uint64_t c = 0;
    for (int i = 0; i < 9; i++) {
        c += (uint64_t)F29::p(i) * (uint32_t)K;
        out[i] = i < 8 ? (uint32_t)c & F29::MASK : (uint32_t)c;
        c >>= 29;
    }
    out[0] += 1u << 29;
    for (int i = 1; i < 7; i++) out[i] += F29::MASK;
    out[7] -= 1u;                                        // (as a signed word: -1 when limb 7 of K p is 0)
}

// tower <-> polynomial basis: poly index k -> (which Fq6 half, which coefficient)
LSA_HD Fq2S &w12_tower_ref(Fq12S &t, int k) {
    Fq6T<Fs> &h = (k & 1) ? t.c1 : t.c0;
    return (k >> 1) == 0 ? h.c0 : ((k >> 1) == 1 ? h.c1 : h.c2);
}

// Granger-Scott squaring of a cyclotomic element in the polynomial basis, one Fq component of the result per lane as ONE
// fused dot product (fp29.h dot4) -- no second reduction phase, no barrier inside.  With f = sum a_k w^k (a_k = tower
// coefficient: a0, a2, a4 = c0.c0, c0.c1, c0.c2; a1, a3, a5 = c1.c0, c1.c1, c1.c2) libff's cyclotomic_squared reads
//   a0' = 3 (a0^2 + xi a3^2) - 2 a0     a3' = 6 a0 a3 + 2 a3          (pair x = a0, y = a3)
//   a2' = 3 (a1^2 + xi a4^2) - 2 a2     a5' = 6 a1 a4 + 2 a5          (pair x = a1, y = a4)
//   a4' = 3 (a2^2 + xi a5^2) - 2 a4     a1' = 6 xi a2 a5 + 2 a1       (pair x = a2, y = a5)
// and each Fq component is sum_i L_i * R_i + (+-2) * s with L_i one component of x or y (or x0 + x1) and R_i a small
// integer combination of two components plus a multiple of p that keeps it non-negative:
//   E.c0 = (x0 + x1)(3x0 - 3x1) + y0 (27y0 - 6y1) - 27 y1^2 - 2 s0        E.c1 = 6 x0 x1 + y0 (3y0 + 54y1) - 3 y1^2 - 2 s1
//   O.c0 = 6 x0 y0 - 6 x1 y1 + 2 s0                                        O.c1 = 6 x0 y1 + 6 x1 y0 + 2 s1
//   X.c0 = x0 (54y0 - 6y1) - x1 (6y0 + 54y1) + 2 s0                        X.c1 = x0 (6y0 + 54y1) + x1 (54y0 - 6y1) + 2 s1
// Inputs < 2p (tight): every R_i < 120p < 2^261, T = sum L_i R_i < 484 p^2, so the one reduction leaves < 3.87p and one
// conditional subtraction of 2p restores < 2p.  ~800 instructions on 12 lanes against ~430 + ~350 on 72 + 12 lanes and a
// workgroup barrier in between for the general product.
struct W12SqTerm { int8_t la, fa, lb, fb, ra, ca, rb, cb, K; };    // L = fa * comp[la] + fb * comp[lb];  R = ca * comp[ra] + cb * comp[rb] + K p
struct W12SqRow { W12SqTerm t[3]; int8_t lin; };                    // + lin * s   (comp: 0 x0, 1 x1, 2 y0, 3 y1; a zero factor drops a term)
LSA_HD const W12SqRow &w12_sq_row(unsigned type, unsigned part) {
    // type 0: E, 1: O, 2: X
    static constexpr W12SqRow ROWS[6] = {
        {{{0, 1, 1, 1, 0, 3, 1, -3, 6}, {2, 1, 2, 0, 2, 27, 3, -6, 12}, {3, 1, 3, 0, 3, -27, 3, 0, 54}}, -2},     // E.c0
        {{{0, 1, 0, 0, 1, 6, 1, 0, 0}, {2, 1, 2, 0, 2, 3, 3, 54, 0}, {3, 1, 3, 0, 3, -3, 3, 0, 6}}, -2},           // E.c1
        {{{0, 1, 0, 0, 2, 6, 2, 0, 0}, {1, 1, 1, 0, 3, -6, 3, 0, 12}, {0, 0, 0, 0, 0, 0, 0, 0, 0}}, 2},            // O.c0
        {{{0, 1, 0, 0, 3, 6, 3, 0, 0}, {1, 1, 1, 0, 2, 6, 2, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0}}, 2},              // O.c1
        {{{0, 1, 0, 0, 2, 54, 3, -6, 12}, {1, 1, 1, 0, 2, -6, 3, -54, 120}, {0, 0, 0, 0, 0, 0, 0, 0, 0}}, 2},      // X.c0
        {{{0, 1, 0, 0, 2, 6, 3, 54, 0}, {1, 1, 1, 0, 2, 54, 3, -6, 12}, {0, 0, 0, 0, 0, 0, 0, 0, 0}}, 2},          // X.c1
    };
    return ROWS[type * 2 + part];
}

#if defined(__HIP_DEVICE_COMPILE__)
// ------------------------------------------------------------------------------------------------------------------
// The Fq12 product in ONE phase (device only; W12::mul takes it when the workgroup has 192 lanes).  The two-phase
// product above costs its first wavefront ~430 + ~350 instructions and two workgroup barriers with an LDS round trip
// between them: ~3 us for a chain link whose arithmetic is one Fq product deep.  Here a ROW of 16 lanes owns one Fq
// component of the result (12 rows = three wavefronts): lane r < 12 of row (k, part) computes the plain integer
// product L * R of ONE pair of 29-bit-limb operands -- 81 multiply-adds into 17 column sums, no reduction -- where L is
// a component of a_i and R the matching combination of b_j's components (i = r / 2, j = k - i mod 6; terms with
// i + j >= 6 carry the factor xi = 9 + u on the b side):
//   part 0:  a_i0 * ( b0)        + a_i1 * (-b1)          wrapped:  a_i0 * (9 b0 - b1)  + a_i1 * (-b0 - 9 b1)
//   part 1:  a_i0 * ( b1)        + a_i1 * ( b0)          wrapped:  a_i0 * (b0 + 9 b1)  + a_i1 * (9 b0 - b1)
// (R made non-negative with a multiple of p: < 22p).  The lane carries its columns into 18 tight limbs, the row adds
// them up with four DPP butterfly steps (one carry pass in the middle keeps the sums in 32 bits), and every lane of the
// row runs the ONE Montgomery reduction of the sum: T < 12 * 44 p^2 gives < 4p, one conditional subtraction < 2p.
// ~550 instructions, one barrier (two when the destination aliases an operand).
// ------------------------------------------------------------------------------------------------------------------
// t.l[q] += t.l[q] of the lane the permutation names, all 18 limbs: one v_add_u32_dpp each, written as ONE asm block (the
// intrinsic leaves a v_mov per limb behind; inside the block consecutive instructions touch different registers, the
// leading s_nop covers the DPP read-after-VALU-write hazard against whatever the compiler scheduled before it)
#define W12_DPP18(T, CTRL)                                                                                                  \
    asm volatile("s_nop 1\n"                                                                                               \
                 "v_add_u32_dpp %0, %0, %0 " CTRL "\nv_add_u32_dpp %1, %1, %1 " CTRL "\nv_add_u32_dpp %2, %2, %2 " CTRL "\n"   \
                 "v_add_u32_dpp %3, %3, %3 " CTRL "\nv_add_u32_dpp %4, %4, %4 " CTRL "\nv_add_u32_dpp %5, %5, %5 " CTRL "\n"   \
                 "v_add_u32_dpp %6, %6, %6 " CTRL "\nv_add_u32_dpp %7, %7, %7 " CTRL "\nv_add_u32_dpp %8, %8, %8 " CTRL "\n"   \
                 "v_add_u32_dpp %9, %9, %9 " CTRL "\nv_add_u32_dpp %10, %10, %10 " CTRL "\nv_add_u32_dpp %11, %11, %11 " CTRL "\n" \
                 "v_add_u32_dpp %12, %12, %12 " CTRL "\nv_add_u32_dpp %13, %13, %13 " CTRL "\nv_add_u32_dpp %14, %14, %14 " CTRL "\n" \
                 "v_add_u32_dpp %15, %15, %15 " CTRL "\nv_add_u32_dpp %16, %16, %16 " CTRL "\nv_add_u32_dpp %17, %17, %17 " CTRL "\n" \
                 : "+v"((T).l[0]), "+v"((T).l[1]), "+v"((T).l[2]), "+v"((T).l[3]), "+v"((T).l[4]), "+v"((T).l[5]), "+v"((T).l[6]),     \
                   "+v"((T).l[7]), "+v"((T).l[8]), "+v"((T).l[9]), "+v"((T).l[10]), "+v"((T).l[11]), "+v"((T).l[12]), "+v"((T).l[13]), \
                   "+v"((T).l[14]), "+v"((T).l[15]), "+v"((T).l[16]), "+v"((T).l[17]))
struct W12Limbs18 { uint32_t l[18]; };
// plain product a * b of two tight 9-limb values as 18 tight limbs (a * b < 2^522)
__device__ __forceinline__ W12Limbs18 w12_wide_mul(const F29 &a, const F29 &b) {
    // the 17 column sums are independent of each other (< 9 * 2^58 each): accumulated apart and carried afterwards, so
    // that a lone wavefront can interleave their multiply-adds instead of waiting on one accumulator
    W12Limbs18 r;
    uint64_t col[17];
#pragma unroll
    for (int k = 0; k < 17; k++) {
        col[k] = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j >= 0 && j < 9) col[k] += (uint64_t)a.l[i] * b.l[j];
        }
    }
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        acc += col[k];
        r.l[k] = (uint32_t)acc & F29::MASK;
        acc >>= 29;
    }
    r.l[17] = (uint32_t)acc;
    return r;
}
// (T + U) / 2^261 mod p for T = sum t_k 2^(29k), U = sum u_k 2^(29k) with loose limbs (any 32-bit values): < (T + U) / 2^261 + p,
// limbs tight.  Written column-wise for a LONE wavefront: m_k goes into the eight later columns as soon as it is known --
// eight independent multiply-adds the scheduler can place between the dependent instructions of the one serial chain
// (column + carry -> m_k -> carry) -- instead of 81 multiply-adds accumulated one after the other into the same register
// pair (a dependent v_mad_u64_u32 issues every 11 cycles, an independent one every 6: tools/ubench_clock.hip).
__device__ __forceinline__ F29 w12_redc18(const W12Limbs18 &t, const W12Limbs18 &u) {
    uint64_t col[17];
#pragma unroll
    for (int k = 0; k < 17; k++) col[k] = (uint64_t)t.l[k] + u.l[k];
    uint64_t carry = 0;
    F29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        uint64_t acc = col[k] + carry;
        const uint32_t m = ((uint32_t)acc * F29::PINV) & F29::MASK;
        acc += (uint64_t)m * F29::p(0);
        carry = acc >> 29;
#pragma unroll
        for (int j = 1; j < 9; j++) col[k + j] += (uint64_t)m * F29::p(j);
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
        const uint64_t acc = col[k] + carry;
        r.l[k - 9] = (uint32_t)acc & F29::MASK;
        carry = acc >> 29;
    }
    r.l[8] = (uint32_t)carry + t.l[17] + u.l[17];
    return r;
}
// the same lanes' limbs as seen through a DPP permutation (v_mov_b32_dpp, all 18 limbs, one asm block as W12_DPP18)
#define W12_DPP18_MOV(D, T, CTRL)                                                                                          \
    asm volatile("s_nop 1\n"                                                                                               \
                 "v_mov_b32_dpp %0, %18 " CTRL "\nv_mov_b32_dpp %1, %19 " CTRL "\nv_mov_b32_dpp %2, %20 " CTRL "\n"          \
                 "v_mov_b32_dpp %3, %21 " CTRL "\nv_mov_b32_dpp %4, %22 " CTRL "\nv_mov_b32_dpp %5, %23 " CTRL "\n"          \
                 "v_mov_b32_dpp %6, %24 " CTRL "\nv_mov_b32_dpp %7, %25 " CTRL "\nv_mov_b32_dpp %8, %26 " CTRL "\n"          \
                 "v_mov_b32_dpp %9, %27 " CTRL "\nv_mov_b32_dpp %10, %28 " CTRL "\nv_mov_b32_dpp %11, %29 " CTRL "\n"        \
                 "v_mov_b32_dpp %12, %30 " CTRL "\nv_mov_b32_dpp %13, %31 " CTRL "\nv_mov_b32_dpp %14, %32 " CTRL "\n"       \
                 "v_mov_b32_dpp %15, %33 " CTRL "\nv_mov_b32_dpp %16, %34 " CTRL "\nv_mov_b32_dpp %17, %35 " CTRL "\n"       \
                 : "=&v"((D).l[0]), "=&v"((D).l[1]), "=&v"((D).l[2]), "=&v"((D).l[3]), "=&v"((D).l[4]), "=&v"((D).l[5]),       \
                   "=&v"((D).l[6]), "=&v"((D).l[7]), "=&v"((D).l[8]), "=&v"((D).l[9]), "=&v"((D).l[10]), "=&v"((D).l[11]),    \
                   "=&v"((D).l[12]), "=&v"((D).l[13]), "=&v"((D).l[14]), "=&v"((D).l[15]), "=&v"((D).l[16]), "=&v"((D).l[17]) \
                 : "v"((T).l[0]), "v"((T).l[1]), "v"((T).l[2]), "v"((T).l[3]), "v"((T).l[4]), "v"((T).l[5]), "v"((T).l[6]),   \
                   "v"((T).l[7]), "v"((T).l[8]), "v"((T).l[9]), "v"((T).l[10]), "v"((T).l[11]), "v"((T).l[12]), "v"((T).l[13]), \
                   "v"((T).l[14]), "v"((T).l[15]), "v"((T).l[16]), "v"((T).l[17]))
// Keep a value computed by EVERY lane: its only use is a store by one lane of the row, and left alone the compiler sinks
// the whole computation under that lane's EXEC mask.  A lone wavefront runs instructions with a sparse EXEC mask slower
// than with all lanes on -- 1.1x to 2.1x, depending on the CU it landed on (tools/ubench_placement.hip, "one active
// lane per wavefront"; every CU runs the dense version at the same speed) -- so the arithmetic stays dense and only
// the store is predicated.
__device__ __forceinline__ void w12_pin(F29 &v) {
    asm volatile("" : "+v"(v.l[0]), "+v"(v.l[1]), "+v"(v.l[2]), "+v"(v.l[3]), "+v"(v.l[4]), "+v"(v.l[5]), "+v"(v.l[6]), "+v"(v.l[7]), "+v"(v.l[8]));
}
enum { W12_MUL = 0, W12_FROB = 1, W12_LINE = 2, W12_SCALE = 3 };
// ------------------------------------------------------------------------------------------------------------------
// A FOURTH wavefront beside the chain (HLP; round 6, second session).  The lone final exponentiation inverted one Fq value on
// its critical path -- the binary GCD of inv29.h, ~20 000 instructions on every lane, 75 us of a 0.39-ms kernel whose other
// ~265 chain links are one row product each.  tools/gen_fe_scalar_exponent.py: leave the division by n1 (the norm of f f^(q^6)
// down to Fq) out and the chain ends in FE(f) * n1^(2K) for a constant K -- n1 lies in Fq, Frobenius maps and conjugations fix
// it -- so the correction n1^e, e = -2K mod (q - 1), is ONE exponentiation in Fq by a fixed 254-bit exponent that does not
// depend on the chain.  Rows 12 and 13 of a 256-lane workgroup (a wavefront of their own, on the CU's fourth SIMD) run it
// right to left in the same instruction stream as the chain's rows: per chain link row 12 squares the running power
// n1^(2^i), row 13 multiplies the accumulator by it where bit i of e is set; 254 of the 264 links that follow the norm, then
// one link (W12_SCALE) multiplies the twelve components by the accumulator.  Same field element as libff's, so the same
// bytes (tests: every final exponentiation of the GPU suite goes through it).  LSA_FE_HELPER=0: the 192-lane kernel.
// State, in the 36 partial-product slots the row engine does not use (LDS words at H):
//   0..8 the running power, 9..17 the accumulator, 18..26 zero, 28..35 the exponent's words, 36 the state -- 2 * (links done) + (the
//   exponentiation is running: row 12 squares the power; zero: both rows multiply by one) --, 37 the exponent's bits from this
//   link's on (bit 0: row 13 multiplies by the power in THIS link), refilled from 28..35 every 32 links.
// ------------------------------------------------------------------------------------------------------------------
// generated by tools/gen_fe_scalar_exponent.py: e = -2 K mod (q - 1), 254 bits
static constexpr uint32_t W12_FE_SCALAR_EXP[8] = {0x8d10a36eu, 0xfa9264cdu, 0xc9df5cd9u, 0x2f1120f5u, 0x4317591cu, 0x4bdc2634u, 0xe131a027u, 0x30644e72u};
static constexpr int W12_FE_SCALAR_EXP_BITS = 254;
enum { W12_H_PW = 0, W12_H_ACC = 9, W12_H_ZERO = 18, W12_H_EXP = 28, W12_H_STATE = 36, W12_H_BITS = 37 };
static constexpr uint32_t W12_H_IDLE = 0xffffu << 1;
// NOALIAS: D is neither A nor B -- the barrier between the operand loads and the result's stores (which only keeps a fast
// wavefront from overwriting what a slow one has not read yet) is not needed then: one barrier per chain link instead of two
template <int MODE, bool HLP = false, bool NOALIAS = false>
__device__ __forceinline__ void w12_rows(Fq2S *D, const Fq2S *A, const Fq2S *B, const uint32_t *frob, Fq2S *H = nullptr) {
    const unsigned lane = threadIdx.x, row = lane >> 4, r = lane & 15;
    const unsigned k = row >> 1, part = row & 1, i = r >> 1, h = r & 1;
    F29 L, Rv;
    Fs *dst = &w12_comp(D[k < 6 ? k : 0], part);
    bool store = r == 0;
    uint32_t hst2 = 0, hbits2 = 0;
    if (HLP && row >= 12) {                            // (the whole fourth wavefront)
        // one LDS round trip in front of the barrier, as the chain's rows have: the state words, the power (every lane: row 12's
        // second factor, row 13's when the bit is set) and lane 0's first factor by ADDRESS (row 12: the power, rows 13..15: the
        // accumulator, lanes 1..15 of a row: the zero kept beside them).  The next link's state is worked out HERE, where this
        // wavefront waits for the chain's rows anyway, and stored behind the barrier.
        w12_lds_u32 *hw = w12_lds(H);
        const uint32_t hst = hw[W12_H_STATE], hbits = hw[W12_H_BITS];
        const Fs *hs = reinterpret_cast<const Fs *>(H);
        const Fs pw = w12_load(hs);
        L = w12_load(hs + (r != 0 ? 2 : (row == 12 ? 0 : 1))).v;
        const F29 one = F29::one();
        const uint32_t act = w12_mask(0u - (hst & 1u)), sel = act & w12_mask(0u - (hbits & 1u)), sq = w12_mask(0u - (uint32_t)(row == 12));
        const uint32_t m = (act & sq) | (sel & ~sq);
#pragma unroll
        for (int q = 0; q < 9; q++) Rv.l[q] = (pw.v.l[q] & m) | (one.l[q] & ~m);
        const uint32_t c2 = (hst >> 1) + (hst & 1u);
        hbits2 = hbits >> 1;
        if ((c2 & 31u) == 0u) hbits2 = hw[W12_H_EXP + ((c2 >> 5) & 7u)];          // (every 32nd link)
        hst2 = (c2 << 1) | ((hst & 1u) & (uint32_t)(c2 < (uint32_t)W12_FE_SCALAR_EXP_BITS));
        dst = const_cast<Fs *>(hs) + (row == 12 ? 0 : 1);
        store = r == 0 && row <= 13;
    } else if (MODE == W12_SCALE) {                    // D_k <- A_k * (the helper's accumulator): lane 0 of a row, the others add nothing
        const uint32_t lm = w12_mask(0u - (uint32_t)(r == 0));
        L = w12_load(&w12_comp(A[k], part)).v;
        const Fs acc = w12_load(reinterpret_cast<const Fs *>(H) + 1);
#pragma unroll
        for (int q = 0; q < 9; q++) Rv.l[q] = acc.v.l[q] & lm;
    } else if (MODE == W12_FROB) {
        static constexpr uint32_t ZERO9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        L = w12_load(&w12_comp(A[k], h)).v;
        const uint32_t *c = r < 2 ? frob + ((k * 2 + part) * 2 + h) * 9 : ZERO9;      // lanes 2..15 of a row add nothing
#pragma unroll
        for (int q = 0; q < 9; q++) Rv.l[q] = c[q];
    } else {
        const unsigned iw = i < 6 ? i : 0;
        const bool wrapped = iw > k;
        const unsigned j = k - iw + (wrapped ? 6u : 0u);
        // R = c0 * b_j0 + c1 * b_j1 + K p, the three small integers looked up by (wrapped, part, h) in packed bytes:
        //   plain:    part 0: (1, 0, 0), (0, -1, 2)       part 1: (0, 1, 0), (1, 0, 0)
        //   wrapped:  part 0: (9, -1, 2), (-1, -9, 20)    part 1: (1, 9, 0), (9, -1, 2)
        // (all zero for lanes 12..15 of a row: they add nothing)
        const unsigned sh = 8u * ((wrapped ? 4u : 0u) | (part << 1) | h);
        int lv = r < 12 ? -1 : 0;
        unsigned jb = j;
        if (MODE == W12_LINE) {                          // B = {ell_0, ell_VW', ell_VV'}: coefficients 0, 3, 4 of a line (libff mul_by_024), the others are zero
            lv &= (j == 0 || j == 3 || j == 4) ? -1 : 0;
            jb = j == 0 ? 0u : (j == 3 ? 1u : 2u);
        }
        const int c0 = (int)(int8_t)(0x0901ff0901000001ull >> sh) & lv, c1 = (int)(int8_t)(0xff09f7ff0001ff00ull >> sh) & lv, K = (int)(int8_t)(0x0200140200000200ull >> sh) & lv;
        L = w12_load(&w12_comp(A[iw], h)).v;
        const Fq2S bj = w12_load(&B[jb]);
        Rv = lin2(bj.c0.v, c0, bj.c1.v, c1, K);
    }
    if (!NOALIAS) __syncthreads();                     // every lane holds its operands: D may alias A or B from here on
    if (HLP && lane == 192) { w12_lds(H)[W12_H_STATE] = hst2; w12_lds(H)[W12_H_BITS] = hbits2; }
    W12Limbs18 t = w12_wide_mul(L, Rv), u;
    W12_DPP18(t, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    W12_DPP18(t, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");         // sums of four (< 2^31)
    W12_DPP18(t, "row_half_mirror row_mask:0xf bank_mask:0xf");             // sums of eight (< 2^32: tight limbs)
    W12_DPP18_MOV(u, t, "row_mirror row_mask:0xf bank_mask:0xf");           // the other half of the row: added inside the reduction
    Fs res = {condsub2(w12_redc18(t, u))};
    w12_pin(res.v);                                    // (dense: see w12_pin)
    if (store) w12_store(dst, res);
    __syncthreads();
}
// LEAF functions (no calls inside, so no return address to park in a spilled VGPR: a product that calls a product
// function pays one scratch store and one scratch load with their waits -- a round trip to L2 -- per chain link)
__device__ __noinline__ void w12_mul_rows(Fq2S *D, const Fq2S *A, const Fq2S *B) { w12_rows<W12_MUL>(D, A, B, nullptr); }
__device__ __noinline__ void w12_frob_rows(Fq2S *D, const Fq2S *A, const uint32_t *frob) { w12_rows<W12_FROB>(D, A, nullptr, frob); }
// slot d <- conj(slot a ^ z) for a in the cyclotomic subgroup (W12::exp_by_neg_z, which explains the digits), 192 lanes:
// the whole loop as ONE block of code -- both row products inlined, the factor of a multiplication chosen by slot
// number, no calls in the 62 trips.  A free function of the register file's base R (slot s = R + 6 s), NOT a member:
// a member that is not inlined takes `this`, the W12 object then lives in scratch memory, and every slot address in the
// caller and here becomes a load from it -- a round trip to L2 in front of each chain link.
template <bool HLP>
__device__ __forceinline__ void w12_exp_by_neg_z_rows_t(Fq2S *R, int d, int a, int tmp) {
    Fq2S *const H = R + 6 * W12_SLOTS;                  // (the helper's state, HLP only)
    // (every product of this function writes a slot that is neither of its factors: the 256-lane form drops the first barrier of
    // a link -- w12_rows, NOALIAS; the 192-lane fallback stays as it was)
    constexpr uint64_t D_P1 = 0x4800120040011001ull, D_P3 = 0x0000804004000000ull, D_M1 = 0x0000000000000010ull, D_M3 = 0x0108000400880200ull;
    const unsigned lane = threadIdx.x;
    auto slot = [R](int s) { return R + 6 * s; };
    auto conj = [&](int dd, int aa) {
        if (lane < 6) { const Fq2S v = slot(aa)[lane]; slot(dd)[lane] = (lane & 1) ? v.neg() : v; }
        __syncthreads();
    };
    int acc = tmp, alt = tmp + 3;
    const int a3 = tmp + 1, na = tmp + 2, na3 = tmp + 4, sq = tmp + 5;
    w12_rows<W12_MUL, HLP, HLP>(slot(sq), slot(a), slot(a), nullptr, H);
    w12_rows<W12_MUL, HLP, HLP>(slot(a3), slot(sq), slot(a), nullptr, H);        // a^3
    conj(na, a);
    conj(na3, a3);
#pragma unroll 1
    for (int i = 61; i >= 0; --i) {
        w12_rows<W12_MUL, HLP, HLP>(slot(alt), slot(i == 61 ? a : acc), slot(i == 61 ? a : acc), nullptr, H);
        const unsigned p1 = (unsigned)(D_P1 >> i) & 1u, p3 = (unsigned)(D_P3 >> i) & 1u, m1 = (unsigned)(D_M1 >> i) & 1u, m3 = (unsigned)(D_M3 >> i) & 1u;
        if (p1 | p3 | m1 | m3) {
            const int f = p1 ? a : (p3 ? a3 : (m1 ? na : na3));
            w12_rows<W12_MUL, HLP, HLP>(slot(acc), slot(alt), slot(f), nullptr, H);
        } else { const int t = acc; acc = alt; alt = t; }
    }
    conj(d, acc);
}
__device__ __noinline__ void w12_exp_by_neg_z_rows(Fq2S *R, int d, int a, int tmp) { w12_exp_by_neg_z_rows_t<false>(R, d, a, tmp); }
// (the 256-lane forms: the chain's rows and the helper's, see w12_rows)
__device__ __noinline__ void w12_exp_by_neg_z_rows_h(Fq2S *R, int d, int a, int tmp) { w12_exp_by_neg_z_rows_t<true>(R, d, a, tmp); }
__device__ __noinline__ void w12_mul_rows_h(Fq2S *D, const Fq2S *A, const Fq2S *B, Fq2S *H) { w12_rows<W12_MUL, true>(D, A, B, nullptr, H); }
__device__ __noinline__ void w12_frob_rows_h(Fq2S *D, const Fq2S *A, const uint32_t *frob, Fq2S *H) { w12_rows<W12_FROB, true>(D, A, nullptr, frob, H); }
__device__ __noinline__ void w12_scale_rows_h(Fq2S *D, const Fq2S *A, Fq2S *H) { w12_rows<W12_SCALE, true>(D, A, nullptr, nullptr, H); }

// libff alt_bn128_final_exponentiation on slot 0 -> slot 0, 256 lanes: W12::final_exponentiation's chain with the one
// inversion replaced by the helper's exponentiation (see w12_rows).  R: the W12_SLOTS registers, followed by the 36 slots of
// the two-phase product that hold the helper's state here.
__device__ __forceinline__ void w12_final_exponentiation_h(Fq2S *R) {
    enum { ELT = 0, FIRST, A, B, C, D, E, F, G, T0, T1, T2, T3, T4, T5 };
    Fq2S *const H = R + 6 * W12_SLOTS;
    const unsigned lane = threadIdx.x;
    auto slot = [R](int s) { return R + 6 * s; };
    auto conj = [&](int d, int a) {
        if (lane < 6) { const Fq2S v = slot(a)[lane]; slot(d)[lane] = (lane & 1) ? v.neg() : v; }
        __syncthreads();
    };
    auto mul = [&](int d, int a, int b) { w12_mul_rows_h(slot(d), slot(a), slot(b), H); };
    auto frob = [&](int power, int d, int a) { w12_frob_rows_h(slot(d), slot(a), &LSA_FROB_ROWS[power - 1][0][0][0][0], H); };
    {                                                   // the helper idles (times one) until the norm is there
        w12_lds_u32 *hw = w12_lds(H);
        if (lane < 28) hw[lane] = 0;
        if (lane >= 28 && lane < 36) hw[lane] = W12_FE_SCALAR_EXP[lane - 28];
        if (lane == 36) { hw[W12_H_STATE] = W12_H_IDLE; hw[W12_H_BITS] = 0; }
    }
    conj(A, ELT);             // conj(f): f^(q^6)
    mul(C, ELT, A);           // f * conj(f), an element of Fq6
    // (f conj f)^-1 * n1: a^(q^2) a^(q^4) * conj(n2), n2 = a a^(q^2) a^(q^4) the norm down to Fq2, n1 = n2 conj(n2) the norm down to Fq
    frob(2, T0, C);
    frob(2, T1, T0);
    mul(T0, T0, T1);
    mul(T1, C, T0);           // n2 (odd and higher coefficients are 0 mod p)
    {
        const Fq2S n2 = w12_load(&slot(T1)[0]);
        Fs n1 = n2.c0.sqr() + n2.c1.sqr();              // every lane the same arithmetic: no sparse EXEC mask (w12_pin)
        Fs nc1 = n2.c1.neg();
        w12_pin(n1.v);
        w12_pin(nc1.v);
        __syncthreads();                                // (every lane has read n2)
        if (lane == 0) slot(T1)[0] = Fq2S{n2.c0, nc1};
        else if (lane < 6) slot(T1)[lane] = Fq2S::zero();
        if (lane == 64) {                               // the helper starts: power = n1, accumulator = 1, bit 0 of the exponent
            w12_store(reinterpret_cast<Fs *>(H), n1);
            w12_store(reinterpret_cast<Fs *>(H) + 1, Fs::one());
            w12_lds(H)[W12_H_STATE] = 1u;
            w12_lds(H)[W12_H_BITS] = W12_FE_SCALAR_EXP[0];
        }
        __syncthreads();
    }
    mul(D, T0, T1);
    mul(B, A, D);             // f^-1 * n1
    mul(C, A, B);             // f^(q^6 - 1) * n1
    frob(2, D, C);
    mul(FIRST, D, C);         // the easy part * n1^2
    w12_exp_by_neg_z_rows_h(R, A, FIRST, T0);
    mul(B, A, A);
    mul(C, B, B);
    mul(D, C, B);
    w12_exp_by_neg_z_rows_h(R, E, D, T0);
    mul(F, E, E);
    w12_exp_by_neg_z_rows_h(R, G, F, T0);
    conj(T2, D);              // H
    conj(G, G);               // I
    mul(G, G, E);             // J = I * E
    mul(G, G, T2);            // K = J * H
    mul(T2, G, B);            // L = K * B
    mul(T0, G, E);            // M = K * E
    mul(T0, T0, FIRST);       // N = M * first
    frob(1, T1, T2);          // O = frob1(L)
    mul(T0, T1, T0);          // P = O * N
    frob(2, T1, G);           // Q = frob2(K)
    mul(T0, T1, T0);          // R = Q * P
    conj(T1, FIRST);          // S
    mul(T1, T1, T2);          // T = S * L
    frob(3, T1, T1);          // U
    mul(ELT, T1, T0);         // U * R = FE(f) * n1^(2K)
    // (264 links since the norm, 254 needed; the loop is for a chain someone shortens)
    while (w12_lds(H)[W12_H_STATE] & 1u) mul(T5, T5, T5);
    w12_scale_rows_h(slot(ELT), slot(ELT), H);
}

#endif

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
    LSA_HD void mul(int d, int a, int b) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (x.nlanes() >= 192) { w12_mul_rows(slot(d), slot(a), slot(b)); return; }      // (a compile-time constant on the device)
#endif
        mul_lanes(d, a, b);
    }
    LSA_HD_NOINLINE void mul_lanes(int d, int a, int b) {
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
    // d = a^2 for a in the cyclotomic subgroup (every squaring of the hard part); d != a
    LSA_HD void csqr(int d, int a) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (x.nlanes() >= 192) { w12_mul_rows(slot(d), slot(a), slot(a)); return; }      // the row product is shorter than twelve fused lanes
#endif
        csqr_lanes(d, a);
    }
    LSA_HD_NOINLINE void csqr_lanes(int d, int a) {
        Fq2S *A = slot(a), *D = slot(d);
        x.par([=](unsigned lane) {
            if (lane < 12) {
                const unsigned k = lane >> 1, part = lane & 1;
                // output coefficient k: its pair (x, y) = (a_xi, a_(xi+3)) and its formula
                const unsigned xi = (0x120120u >> (4 * k)) & 15u;              // k = 0..5 -> 0, 2, 1, 0, 2, 1
                const unsigned type = (0x101020u >> (4 * k)) & 15u;            // k = 0..5 -> E, X, E, O, E, O = 0, 2, 0, 1, 0, 1
                const W12SqRow &row = w12_sq_row(type, part);
                // component c of the pair, by address (an index into a register array would live in scratch)
                auto comp = [&](int c) { return w12_load(&w12_comp(A[xi + 3 * (unsigned)(c >> 1)], (unsigned)(c & 1))).v; };
                F29 L[3], R[3];
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const W12SqTerm &t = row.t[i];
                    L[i] = lin2(comp(t.la), t.fa, comp(t.lb), t.fb, 0);
                    R[i] = lin2(comp(t.ra), t.ca, comp(t.rb), t.cb, t.K);
                }
                const F29 two = add_lazy(F29::one(), F29::one());              // Montgomery form of 2 (< 2p)
                const F29 lin = lin2(two, row.lin > 0 ? 1 : -1, F29::zero(), 0, row.lin > 0 ? 0 : 2);     // +-2
                const F29 sv = w12_load(&w12_comp(A[k], part)).v;
                const Fs r = {condsub2(dot4(L[0], R[0], L[1], R[1], L[2], R[2], sv, lin))};
                w12_store(&w12_comp(D[k], part), r);
            }
        });
    }
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
    LSA_HD void inverse6(int d, int a, int tmp = -1) {
        Fq2S *A = slot(a), *D = slot(d);
#if defined(__HIP_DEVICE_COMPILE__)
        if (x.nlanes() >= 192 && tmp >= 0) {
            // on the row engine: a^-1 = a^(q^2) a^(q^4) / N, N = a a^(q^2) a^(q^4) the norm down to Fq2 (coefficient 0) -- two
            // Frobenius maps and three row products around ONE Fq2 inversion on lane 0, instead of the tower's 33 Fq
            // products on that lane.  The same field element as the tower code's, so the same canonical bytes.  Uses
            // slots tmp, tmp + 1; d may be a.
            Fq2S *T = slot(tmp), *U = slot(tmp + 1);
            w12_frob_rows(T, A, &LSA_FROB_ROWS[1][0][0][0][0]);       // a^(q^2)
            w12_frob_rows(U, T, &LSA_FROB_ROWS[1][0][0][0][0]);       // a^(q^4)
            w12_mul_rows(T, T, U);
            w12_mul_rows(U, A, T);                                    // N (odd and higher coefficients are 0 mod p)
            x.par([=](unsigned lane) {
                Fq2S inv = w12_load(&U[0]).inverse();           // every lane the same inversion: no sparse EXEC mask (w12_pin)
                w12_pin(inv.c0.v);
                w12_pin(inv.c1.v);
                if (lane == 0) U[0] = inv;
                else if (lane < 6) U[lane] = Fq2S::zero();
            });
            w12_mul_rows(D, T, U);
            return;
        }
#endif
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
#if defined(__HIP_DEVICE_COMPILE__)
        static_assert(POWER >= 1 && POWER <= 3, "frob_rows.h holds the maps of a final exponentiation");
        if (x.nlanes() >= 192) { w12_frob_rows(D, A, &LSA_FROB_ROWS[POWER - 1][0][0][0][0]); return; }   // one row product (constants: frob_rows.h)
#endif
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
#if defined(__HIP_DEVICE_COMPILE__)
        if (x.nlanes() >= 192) { w12_exp_by_neg_z_rows(R, d, a, tmp); return; }
#endif
        // the digits as four bit masks (bit i set: digit i is +1 / +3 / -1 / -3): a table in memory costs the chain one
        // load round trip per squaring
        constexpr uint64_t D_P1 = 0x4800120040011001ull, D_P3 = 0x0000804004000000ull, D_M1 = 0x0000000000000010ull, D_M3 = 0x0108000400880200ull;
        static_assert(D_P1 + 3 * D_P3 - D_M1 - 3 * D_M3 == 0x44e992b44a6909f1ull && (D_P1 >> 62) == 1, "width-3 NAF of z");
        // every squaring is a cyclotomic one (csqr: d != a, so the accumulator alternates between two slots); the
        // inverses a^-1, a^-3 (conjugates) are made once.  Uses slots tmp .. tmp + 5.
        int acc = tmp, alt = tmp + 3;
        const int a3 = tmp + 1, na = tmp + 2, na3 = tmp + 4, sq = tmp + 5;
        csqr(sq, a);
        mul(a3, sq, a);                     // a^3
        conj(na, a);
        conj(na3, a3);
        csqr(alt, a);                       // top digit (bit 62) is +1: the accumulator starts as a, the first squaring reads it in place
        for (int i = 61; i >= 0; --i) {
            if (i != 61) csqr(alt, acc);
            if ((D_P1 >> i) & 1) mul(acc, alt, a);
            else if ((D_P3 >> i) & 1) mul(acc, alt, a3);
            else if ((D_M1 >> i) & 1) mul(acc, alt, na);
            else if ((D_M3 >> i) & 1) mul(acc, alt, na3);
            else { const int t = acc; acc = alt; alt = t; }
        }
        conj(d, acc);
    }

    // libff alt_bn128_final_exponentiation on slot 0 -> slot 0 (same chain as final_exp_one in
    // pairing.hip).  Uses every slot.
    LSA_HD void final_exponentiation() {
        enum { ELT = 0, FIRST, A, B, C, D, E, F, G, T0, T1, T2, T3, T4, T5 };      // (exp_by_neg_z uses T0 .. T5)
        conj(A, ELT);             // conj(f): f^(p^6)
        mul(C, ELT, A);           // f * conj(f), an element of Fq6
        inverse6(D, C, T0);
        mul(B, A, D);             // f^-1 = conj(f) / (f * conj(f))
        mul(C, A, B);             // f^(p^6 - 1)
        frobenius<2>(D, C);
        mul(FIRST, D, C);
        exp_by_neg_z(A, FIRST, T0);
        csqr(B, A);
        csqr(C, B);
        mul(D, C, B);
        exp_by_neg_z(E, D, T0);
        csqr(F, E);
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
