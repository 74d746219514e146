// legosnark_amd/csrc/pairing.hip -- batched optimal-ate Miller loops and final
// exponentiations for alt_bn128 on gfx950.
//
// Replaces libff alt_bn128_pp::{precompute_G1, precompute_G2, miller_loop,
// double_miller_loop, final_exponentiation, reduced_pairing} as called from
// /root/reference/src/utils/globl.h:94-105 (simple_pairing_check),
// /root/reference/src/gadgets/subspace.cc:88-170 (CPlink verify) and
// /root/reference/src/gadgets/poly.h:97-123 (CPpoly verify).
//
// One lane per pairing.  G2 "precomputation" is fused: each doubling / mixed-addition
// step of the flipped Miller loop produces its line coefficients (ell_0, ell_VW, ell_VV)
// and applies them to f immediately, so the ~20 KB coefficient table of libff's
// G2_precomp never exists.  The step formulas, the loop over the bits of 6u+2, the two
// Frobenius steps and the final-exponentiation addition chain are libff's, so both the
// Miller-loop value f and the GT value are the same canonical Fq12 elements.
// Independent pairings are embarrassingly parallel; products are folded 8-ary.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#include "ec.h"
#include "msm.h"
#include "tower.h"

namespace lsa {

struct G2Proj { Fq2 X, Y, Z; };
struct Line { Fq2 e0, eVW, eVV; };

// libff doubling_step_for_flipped_miller_loop
static __device__ __noinline__ Line doubling_step(G2Proj &c) {
    const Fq two_inv = []() { Fq t; for (int i = 0; i < 8; i++) t.l[i] = LSA_FQ_TWO_INV[i]; return t; }();
    const Fq2 twist_b = fq2_const(LSA_TWIST_B);
    Fq2 X = c.X, Y = c.Y, Z = c.Z;
    Fq2 A = (X * Y).mul_fq(two_inv);
    Fq2 B = Y.sqr();
    Fq2 C = Z.sqr();
    Fq2 D = C + C + C;
    Fq2 E = twist_b * D;
    Fq2 F = E + E + E;
    Fq2 G = (B + F).mul_fq(two_inv);
    Fq2 H = (Y + Z).sqr() - (B + C);
    Fq2 I = E - B;
    Fq2 J = X.sqr();
    Fq2 E2 = E.sqr();
    c.X = A * (B - F);
    c.Y = G.sqr() - (E2 + E2 + E2);
    c.Z = B * H;
    return {I.mul_xi(), H.neg(), J + J + J};
}

// libff mixed_addition_step_for_flipped_miller_loop
static __device__ __noinline__ Line addition_step(const Fq2 &x2, const Fq2 &y2, G2Proj &c) {
    Fq2 X1 = c.X, Y1 = c.Y, Z1 = c.Z;
    Fq2 D = X1 - x2 * Z1;
    Fq2 E = Y1 - y2 * Z1;
    Fq2 F = D.sqr();
    Fq2 G = E.sqr();
    Fq2 H = D * F;
    Fq2 I = X1 * F;
    Fq2 J = H + Z1 * G - (I + I);
    c.X = D * J;
    c.Y = E * (I - J) - (H * Y1);
    c.Z = Z1 * H;
    return {(E * x2 - D * y2).mul_xi(), D, E.neg()};
}

static __device__ __forceinline__ Fq12 apply_line(const Fq12 &f, const Line &l, const Fq &px, const Fq &py) {
    return fq12_mul_by_024(f, l.e0, l.eVW.mul_fq(py), l.eVV.mul_fq(px));
}

static __device__ __forceinline__ int ate_bit(int i) {
    if (i >= 64) return (int)((LSA_ATE_LOOP_COUNT_HI >> (i - 64)) & 1);
    return (int)((LSA_ATE_LOOP_COUNT_LO >> i) & 1);
}

// precompute_G1 + precompute_G2 + miller_loop for one pair (libff layout in, Fq12 out)
static __device__ __noinline__ Fq12 miller_one(const Jac<Fq> &P, const Jac<Fq2> &Q) {
    // to_affine_coordinates(): O -> (0, 1, 0)
    Fq px, py;
    if (P.Z.is_zero()) { px = Fq::zero(); py = Fq::one(); }
    else {
        Fq zi = P.Z.inverse(), zi2 = zi.sqr();
        px = P.X * zi2; py = P.Y * (zi2 * zi);
    }
    Fq2 qx, qy;
    if (Q.Z.is_zero()) { qx = Fq2::zero(); qy = Fq2::one(); }
    else {
        Fq2 zi = Q.Z.inverse(), zi2 = zi.sqr();
        qx = Q.X * zi2; qy = Q.Y * (zi2 * zi);
    }
    G2Proj R = {qx, qy, Fq2::one()};
    Fq12 f = Fq12::one();
    // bits of 6u+2 below the MSB (bit 64), MSB first
    for (int i = 63; i >= 0; --i) {
        Line l = doubling_step(R);
        f = fq12_sqr(f);
        f = apply_line(f, l, px, py);
        if (ate_bit(i)) {
            l = addition_step(qx, qy, R);
            f = apply_line(f, l, px, py);
        }
    }
    // Q1 = pi(Q), Q2 = -pi^2(Q)   (mul_by_q on affine points: Z stays 1)
    const Fq2 gx = fq2_const(LSA_TWIST_MUL_BY_Q_X), gy = fq2_const(LSA_TWIST_MUL_BY_Q_Y);
    Fq2 q1x = gx * qx.conj(), q1y = gy * qy.conj();
    Fq2 q2x = gx * q1x.conj(), q2y = (gy * q1y.conj()).neg();
    Line l = addition_step(q1x, q1y, R);
    f = apply_line(f, l, px, py);
    l = addition_step(q2x, q2y, R);
    f = apply_line(f, l, px, py);
    return f;
}

static __device__ __noinline__ Fq12 exp_by_neg_z(const Fq12 &a) {
    return fq12_pow_u64(a, LSA_FINAL_EXP_Z).unitary_inverse();
}

// libff alt_bn128_final_exponentiation: first chunk (q^6-1)(q^2+1), last chunk by the
// Fuentes-Castaneda et al. addition chain.
static __device__ __noinline__ Fq12 final_exp_one(const Fq12 &elt) {
    Fq12 A = elt.unitary_inverse();
    Fq12 B = fq12_inverse(elt);
    Fq12 C = A * B;
    Fq12 D = fq12_frobenius<2>(C);
    Fq12 first = D * C;
    A = exp_by_neg_z(first);
    B = fq12_sqr(A);
    C = fq12_sqr(B);
    D = C * B;
    Fq12 E = exp_by_neg_z(D);
    Fq12 F = fq12_sqr(E);
    Fq12 G = exp_by_neg_z(F);
    Fq12 H = D.unitary_inverse();
    Fq12 I = G.unitary_inverse();
    Fq12 J = I * E;
    Fq12 K = J * H;
    Fq12 L = K * B;
    Fq12 M = K * E;
    Fq12 N = M * first;
    Fq12 O = fq12_frobenius<1>(L);
    Fq12 Pp = O * N;
    Fq12 Qq = fq12_frobenius<2>(K);
    Fq12 Rr = Qq * Pp;
    Fq12 S = first.unitary_inverse();
    Fq12 T = S * L;
    Fq12 U = fq12_frobenius<3>(T);
    return U * Rr;
}

__global__ __launch_bounds__(64) void k_miller(const Jac<Fq> *__restrict__ g1, const Jac<Fq2> *__restrict__ g2, size_t n,
                                               Fq12 *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = miller_one(g1[i], g2[i]);
}

__global__ __launch_bounds__(64) void k_final_exp(const Fq12 *__restrict__ in, size_t n, Fq12 *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = final_exp_one(in[i]);
}

// out[i] = prod in[8i .. 8i+7]
__global__ __launch_bounds__(64) void k_fq12_prod8(const Fq12 *__restrict__ in, size_t n, Fq12 *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t lo = i * 8;
    if (lo >= n) return;
    Fq12 acc = in[lo];
    for (size_t j = lo + 1; j < lo + 8 && j < n; j++) acc = acc * in[j];
    out[i] = acc;
}

#define HIPCHK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return LSA_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

int miller_device(const void *d_g1, const void *d_g2, size_t n, void *d_out, hipStream_t st) {
    if (n == 0) return LSA_OK;
    hipLaunchKernelGGL(k_miller, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, (const Jac<Fq> *)d_g1, (const Jac<Fq2> *)d_g2, n,
                       (Fq12 *)d_out);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}

int final_exp_device(const void *d_in, size_t n, void *d_out, hipStream_t st) {
    if (n == 0) return LSA_OK;
    hipLaunchKernelGGL(k_final_exp, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, (const Fq12 *)d_in, n, (Fq12 *)d_out);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}

// d_buf holds n Fq12 values and is overwritten; scratch must hold ceil(n/8) values.
// On return *result points at the single product (inside d_buf or d_scratch).
int fq12_product_device(void *d_buf, void *d_scratch, size_t n, void **result, hipStream_t st) {
    Fq12 *a = (Fq12 *)d_buf, *b = (Fq12 *)d_scratch;
    while (n > 1) {
        size_t m = (n + 7) / 8;
        hipLaunchKernelGGL(k_fq12_prod8, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, st, a, n, b);
        Fq12 *t = a; a = b; b = t;
        n = m;
    }
    HIPCHK(hipGetLastError());
    *result = a;
    return LSA_OK;
}

size_t fq12_bytes() { return sizeof(Fq12); }

}  // namespace lsa
