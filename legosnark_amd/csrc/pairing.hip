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
#include "fs29.h"
#include "msm.h"
#include "tower.h"
#include "w12.h"

namespace lsa {

// The kernels compute on B = Fs (fs29.h: 9 x 29-bit limbs, carry-free Montgomery product)
// and convert from / to libff's layout (Fq, R = 2^256) when they load inputs and store
// results, so everything in memory stays in the C-ABI's byte layout.
using PB = Fs;
using P2 = Fq2T<PB>;
using P12 = Fq12T<PB>;

static __device__ __forceinline__ P2 load2(const Fq2 &v) { return {PB::from_mont256(v.c0), PB::from_mont256(v.c1)}; }
static __device__ __noinline__ P12 load12(const Fq12 &v) {
    P12 r;
    const Fq *w = reinterpret_cast<const Fq *>(&v);
    PB *o = reinterpret_cast<PB *>(&r);
    for (int i = 0; i < 12; i++) o[i] = PB::from_mont256(w[i]);
    return r;
}
static __device__ __noinline__ Fq12 store12(const P12 &v) {
    Fq12 r;
    Fq *w = reinterpret_cast<Fq *>(&r);
    const PB *o = reinterpret_cast<const PB *>(&v);
    for (int i = 0; i < 12; i++) w[i] = o[i].to_mont256();
    return r;
}

struct G2Proj { P2 X, Y, Z; };
struct Line { P2 e0, eVW, eVV; };

// libff doubling_step_for_flipped_miller_loop
static __device__ __noinline__ Line doubling_step(G2Proj &c, const PB &two_inv, const P2 &twist_b) {
    P2 X = c.X, Y = c.Y, Z = c.Z;
    P2 A = (X * Y).mul_fq(two_inv);
    P2 B = Y.sqr();
    P2 C = Z.sqr();
    P2 D = C + C + C;
    P2 E = twist_b * D;
    P2 F = E + E + E;
    P2 G = (B + F).mul_fq(two_inv);
    P2 H = (Y + Z).sqr() - (B + C);
    P2 I = E - B;
    P2 J = X.sqr();
    P2 E2 = E.sqr();
    c.X = A * (B - F);
    c.Y = G.sqr() - (E2 + E2 + E2);
    c.Z = B * H;
    return {I.mul_xi(), H.neg(), J + J + J};
}

// libff mixed_addition_step_for_flipped_miller_loop
static __device__ __noinline__ Line addition_step(const P2 &x2, const P2 &y2, G2Proj &c) {
    P2 X1 = c.X, Y1 = c.Y, Z1 = c.Z;
    P2 D = X1 - x2 * Z1;
    P2 E = Y1 - y2 * Z1;
    P2 F = D.sqr();
    P2 G = E.sqr();
    P2 H = D * F;
    P2 I = X1 * F;
    P2 J = H + Z1 * G - (I + I);
    c.X = D * J;
    c.Y = E * (I - J) - (H * Y1);
    c.Z = Z1 * H;
    return {(E * x2 - D * y2).mul_xi(), D, E.neg()};
}

static __device__ __forceinline__ P12 apply_line(const P12 &f, const Line &l, const PB &px, const PB &py) {
    return fq12_mul_by_024(f, l.e0, l.eVW.mul_fq(py), l.eVV.mul_fq(px));
}

static __device__ __forceinline__ int ate_bit(int i) {
    if (i >= 64) return (int)((LSA_ATE_LOOP_COUNT_HI >> (i - 64)) & 1);
    return (int)((LSA_ATE_LOOP_COUNT_LO >> i) & 1);
}

// precompute_G1 + precompute_G2 + miller_loop for one pair (libff layout in)
static __device__ __noinline__ P12 miller_one(const Jac<Fq> &P, const Jac<Fq2> &Q) {
    // to_affine_coordinates(): O -> (0, 1, 0)
    PB px, py;
    if (P.Z.is_zero()) { px = PB::zero(); py = PB::one(); }
    else {
        PB zi = PB::from_mont256(P.Z).inverse(), zi2 = zi.sqr();
        px = PB::from_mont256(P.X) * zi2; py = PB::from_mont256(P.Y) * (zi2 * zi);
    }
    P2 qx, qy;
    if (Q.Z.is_zero()) { qx = P2::zero(); qy = P2::one(); }
    else {
        P2 zi = load2(Q.Z).inverse(), zi2 = zi.sqr();
        qx = load2(Q.X) * zi2; qy = load2(Q.Y) * (zi2 * zi);
    }
    Fq ti;
#pragma unroll
    for (int i = 0; i < 8; i++) ti.l[i] = LSA_FQ_TWO_INV[i];
    const PB two_inv = PB::from_mont256(ti);
    const P2 twist_b = fq2_constT<PB>(LSA_TWIST_B);
    G2Proj R = {qx, qy, P2::one()};
    P12 f = P12::one();
    // bits of 6u+2 below the MSB (bit 64), MSB first
    for (int i = 63; i >= 0; --i) {
        Line l = doubling_step(R, two_inv, twist_b);
        f = fq12_sqr(f);
        f = apply_line(f, l, px, py);
        if (ate_bit(i)) {
            l = addition_step(qx, qy, R);
            f = apply_line(f, l, px, py);
        }
    }
    // Q1 = pi(Q), Q2 = -pi^2(Q)   (mul_by_q on affine points: Z stays 1)
    const P2 gx = fq2_constT<PB>(LSA_TWIST_MUL_BY_Q_X), gy = fq2_constT<PB>(LSA_TWIST_MUL_BY_Q_Y);
    P2 q1x = gx * qx.conj(), q1y = gy * qy.conj();
    P2 q2x = gx * q1x.conj(), q2y = (gy * q1y.conj()).neg();
    Line l = addition_step(q1x, q1y, R);
    f = apply_line(f, l, px, py);
    l = addition_step(q2x, q2y, R);
    f = apply_line(f, l, px, py);
    return f;
}

static __device__ __noinline__ P12 final_exp_one(const P12 &elt) { return fq12_final_exponentiation(elt); }

__global__ __launch_bounds__(64) void k_miller(const Jac<Fq> *__restrict__ g1, const Jac<Fq2> *__restrict__ g2, size_t n,
                                               Fq12 *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = store12(miller_one(g1[i], g2[i]));
}

__global__ __launch_bounds__(64) void k_final_exp(const Fq12 *__restrict__ in, size_t n, Fq12 *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = store12(final_exp_one(load12(in[i])));
}

// One wavefront per element (w12.h): 36-lane Fq12 products out of LDS.  ~10x shorter chain than
// k_final_exp; used whenever fewer elements than the chip has lanes are in flight.
struct WaveExec {
    template <class F>
    __device__ __forceinline__ void par(F f) {
        f(threadIdx.x);
        __syncthreads();
    }
};
// libff Fq12 layout (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2, each Fq2 = 2 Fq) <-> slot: lane
// l < 12 moves Fq number l, which is part l%2 of tower coefficient t = l/2, i.e. of w^(2*(t%3) + t/3).
static __device__ __forceinline__ Fs *w12_fq_ref(Fq2S *slot, unsigned l) {
    const unsigned t = l >> 1, k = 2 * (t % 3) + t / 3;
    return (l & 1) ? &slot[k].c1 : &slot[k].c0;
}
__global__ __launch_bounds__(64) void k_final_exp_wave(const Fq12 *__restrict__ in, size_t n, Fq12 *__restrict__ out) {
    __shared__ Fq2S lds[W12_LDS_FQ2];
    const size_t e = blockIdx.x;
    if (e >= n) return;
    const unsigned lane = threadIdx.x;
    WaveExec ex;
    W12<WaveExec> w{ex, lds, lds + 6 * W12_SLOTS};
    if (lane < 12) *w12_fq_ref(w.slot(0), lane) = Fs::from_mont256(reinterpret_cast<const Fq *>(&in[e])[lane]);
    __syncthreads();
    w.final_exponentiation();
    if (lane < 12) reinterpret_cast<Fq *>(&out[e])[lane] = w12_fq_ref(w.slot(0), lane)->to_mont256();
}

// out[i] = prod in[8i .. 8i+7]
__global__ __launch_bounds__(64) void k_fq12_prod8(const Fq12 *__restrict__ in, size_t n, Fq12 *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t lo = i * 8;
    if (lo >= n) return;
    P12 acc = load12(in[lo]);
    for (size_t j = lo + 1; j < lo + 8 && j < n; j++) acc = acc * load12(in[j]);
    out[i] = store12(acc);
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
    if (n < 16384)   // fewer elements than lanes to fill the chip: one wavefront per element
        hipLaunchKernelGGL(k_final_exp_wave, dim3((unsigned)n), dim3(64), 0, st, (const Fq12 *)d_in, n, (Fq12 *)d_out);
    else
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

size_t fq12_bytes() { return sizeof(Fq12); }   // libff layout in memory

}  // namespace lsa
