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
#include <stdlib.h>
#include <vector>

#include "ec.h"
#include "fs29.h"
#include "msm.h"
#include "tower.h"
#include "miller.h"
#include "tmiller.h"
#include "w12.h"

namespace lsa {

// The kernels compute on B = Fs (fs29.h: 9 x 29-bit limbs, carry-free Montgomery product)
// and convert from / to libff's layout (Fq, R = 2^256) when they load inputs and store
// results, so everything in memory stays in the C-ABI's byte layout.

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
// (the lane count is a compile-time constant: W12 asks for it in every product, and blockDim.x is a load from the
// dispatch packet -- a round trip to L2, 0.2-0.5 us depending on the CU, in front of each of the ~300 chain links)
template <unsigned LANES>
struct WaveExecN {
    template <class F>
    __device__ __forceinline__ void par(F f) {
        f(threadIdx.x);
        __syncthreads();
    }
    __device__ __forceinline__ unsigned nlanes() const { return LANES; }
};
using WaveExec = WaveExecN<64>;
using WaveExec192 = WaveExecN<192>;
// libff Fq12 layout (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2, each Fq2 = 2 Fq) <-> slot: lane
// l < 12 moves Fq number l, which is part l%2 of tower coefficient t = l/2, i.e. of w^(2*(t%3) + t/3).
static __device__ __forceinline__ Fs *w12_fq_ref(Fq2S *slot, unsigned l) {
    const unsigned t = l >> 1, k = 2 * (t % 3) + t / 3;
    return (l & 1) ? &slot[k].c1 : &slot[k].c0;
}
template <unsigned LANES>
__global__ __launch_bounds__(LANES) void k_final_exp_wave(const Fq12 *__restrict__ in, size_t n, Fq12 *__restrict__ out) {
    __shared__ Fq2S lds[W12_LDS_FQ2];
    const size_t e = blockIdx.x;
    if (e >= n) return;
    const unsigned lane = threadIdx.x;
    WaveExecN<LANES> ex;
    W12<WaveExecN<LANES>> w{ex, lds, lds + 6 * W12_SLOTS};
    if (lane < 12) *w12_fq_ref(w.slot(0), lane) = Fs::from_mont256(reinterpret_cast<const Fq *>(&in[e])[lane]);
    __syncthreads();
    w.final_exponentiation();
    if (lane < 12) reinterpret_cast<Fq *>(&out[e])[lane] = w12_fq_ref(w.slot(0), lane)->to_mont256();
}

// The same with a fourth wavefront that computes, beside the chain, the power of the norm that the chain's one inversion
// would have divided out (w12.h: w12_rows, HLP): no binary GCD on the critical path, 0.39 -> 0.34 ms for a lone element.
__global__ __launch_bounds__(256) void k_final_exp_wave_h(const Fq12 *__restrict__ in, size_t n, Fq12 *__restrict__ out) {
    __shared__ Fq2S lds[W12_LDS_FQ2];
    const size_t e = blockIdx.x;
    if (e >= n) return;
    const unsigned lane = threadIdx.x;
    if (lane < 12) *w12_fq_ref(lds, lane) = Fs::from_mont256(reinterpret_cast<const Fq *>(&in[e])[lane]);
    __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)                    // (w12.h's row engine is device code only)
    w12_final_exponentiation_h(lds);
#endif
    if (lane < 12) reinterpret_cast<Fq *>(&out[e])[lane] = w12_fq_ref(lds, lane)->to_mont256();
}

// out[b] = prod in[G b .. G b + G - 1] (G <= 8), one workgroup per group (W12 products)
__global__ __launch_bounds__(192) void k_fq12_prod8_wave(const Fq12 *__restrict__ in, size_t n, Fq12 *__restrict__ out, unsigned G) {
    __shared__ Fq2S lds[W12_LDS_FQ2];
    const size_t lo = (size_t)blockIdx.x * G;
    if (lo >= n) return;
    const unsigned lane = threadIdx.x;
    WaveExec192 ex;
    W12<WaveExec192> w{ex, lds, lds + 6 * W12_SLOTS};
    for (unsigned x = lane; x < 12 * G; x += 192) {  // G elements x 12 Fq
        const unsigned e = x / 12, l = x % 12;
        Fs v = (l == 0) ? Fs::one() : Fs::zero();       // missing inputs = 1
        if (lo + e < n) v = Fs::from_mont256(reinterpret_cast<const Fq *>(&in[lo + e])[l]);
        *w12_fq_ref(w.slot((int)e), l) = v;
    }
    __syncthreads();
    for (int s = 1; s < (int)G; s++) w.mul(0, 0, s);
    if (lane < 12) reinterpret_cast<Fq *>(&out[blockIdx.x])[lane] = w12_fq_ref(w.slot(0), lane)->to_mont256();
}

// out[j] = prod in[off[j] .. off[j+1]) (1 for an empty segment), one workgroup per segment: the
// verifiers' many short products (CPPoly::verify multiplies 2-3 Miller values per final
// exponentiation, /root/reference/src/gadgets/poly.h:105-122)
__global__ __launch_bounds__(192) void k_fq12_prod_seg_wave(const Fq12 *__restrict__ in, const uint64_t *__restrict__ off, Fq12 *__restrict__ out) {
    __shared__ Fq2S lds[W12_LDS_FQ2];
    const size_t lo = off[blockIdx.x], hi = off[blockIdx.x + 1];
    const unsigned lane = threadIdx.x;
    WaveExec192 ex;
    W12<WaveExec192> w{ex, lds, lds + 6 * W12_SLOTS};
    if (lane < 12) {
        Fs v = (lane == 0) ? Fs::one() : Fs::zero();
        if (lo < hi) v = Fs::from_mont256(reinterpret_cast<const Fq *>(&in[lo])[lane]);
        *w12_fq_ref(w.slot(0), lane) = v;
    }
    __syncthreads();
    for (size_t e = lo + 1; e < hi; e++) {
        if (lane < 12) *w12_fq_ref(w.slot(1), lane) = Fs::from_mont256(reinterpret_cast<const Fq *>(&in[e])[lane]);
        __syncthreads();
        w.mul(0, 0, 1);
    }
    if (lane < 12) reinterpret_cast<Fq *>(&out[blockIdx.x])[lane] = w12_fq_ref(w.slot(0), lane)->to_mont256();
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

// ---- G2 line tables and Miller loops over them (tmiller.h) --------------------------------------------
// libff alt_bn128_ate_precompute_G2 for five points per wavefront: tabs[i] (TM_TAB_WORDS words) <- table of g2[i]
__global__ __launch_bounds__(64) void k_g2_precomp(const Jac<Fq2> *__restrict__ g2, size_t n, uint32_t *const *__restrict__ tabs) {
    __shared__ Fq2S lds[GP_LDS_FQ2];
    __shared__ uint32_t *out[GP_GROUPS];
    __shared__ const Jac<Fq2> *qp[GP_GROUPS];
    const size_t lo = (size_t)blockIdx.x * GP_GROUPS;
    if (lo >= n) return;
    const unsigned count = (unsigned)(n - lo < (size_t)GP_GROUPS ? n - lo : (size_t)GP_GROUPS);
    if (threadIdx.x < (unsigned)GP_GROUPS) {
        out[threadIdx.x] = threadIdx.x < count ? tabs[lo + threadIdx.x] : nullptr;
        qp[threadIdx.x] = g2 + lo + (threadIdx.x < count ? threadIdx.x : 0);
    }
    __syncthreads();
    WaveExec ex;
    G2Pre<WaveExec> pre{ex, lds};
    pre.run(qp, count, out);
}

// Pairs the device has no table for, FUSED: a workgroup of two wavefronts owns four pairs -- wavefront 0 runs the G2
// point arithmetic (G2Pre, four groups) one table entry ahead and puts each line straight into the row ring of
// wavefront 1, which runs the Fq12 chain (TabMiller, four accumulators); one workgroup barrier per entry.  The G2 side
// (2.5 K / 3.4 K instructions per doubling / addition entry) and the f side (3.8 K / 1.9 K) overlap instead of adding
// up, no table ever goes through memory.  The G2 wavefront also scales each line by the pair's (px, py) -- on lanes that
// are free in the round after the one that fixes ell_VW and ell_VV -- so the Fq12 wavefront needs no helper lanes and no
// scaling rounds and serves FIVE accumulators (TabMillerP): 1640 wavefronts for 4096 pairs.
struct WaveLocalExec {
    // one wavefront of a larger workgroup: its phases are ordered by the wavefront's own in-order LDS queue; only the
    // compiler has to be kept from moving memory accesses across the phase boundary
    template <class F>
    __device__ __forceinline__ void par(F f) {
        f(threadIdx.x & 63u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ __forceinline__ unsigned nlanes() const { return 64; }
};
static constexpr int FU_PAIRS = 5;                                     // pairs per workgroup: 60 lanes of each wavefront
// tabs (optional): tabs[i] non-null -> the table of g2[i] is also written there (a point seen for the first time: its
// table goes into the cache while its first Miller loop runs).
__global__ __launch_bounds__(128) void k_miller_fused(const Jac<Fq> *__restrict__ g1, const Jac<Fq2> *__restrict__ g2, const uint8_t *__restrict__ flags,
                                                      uint32_t *const *__restrict__ tabs, size_t n, Fq12 *__restrict__ out, int naf, int decouple) {
    using TP = TabMillerP<WaveLocalExec, FU_PAIRS>;
    __shared__ Fq2S g2mem[FU_PAIRS * GP_STRIDE];
    __shared__ Fq2S tpmem[TP::LDS_FQ2];
    __shared__ const Jac<Fq2> *qp[FU_PAIRS];
    __shared__ const Jac<Fq> *pp[FU_PAIRS];
    __shared__ uint8_t ng[FU_PAIRS];
    __shared__ Fq2S *rows[TP_RING][FU_PAIRS];
    __shared__ int s_produced, s_consumed;
    __shared__ uint32_t *tout[FU_PAIRS];
    __shared__ uint8_t kinds[ATE_NUM_COEFFS];
    const size_t lo = (size_t)blockIdx.x * FU_PAIRS;
    if (lo >= n) return;
    const unsigned count = (unsigned)(n - lo < (size_t)FU_PAIRS ? n - lo : (size_t)FU_PAIRS);
    const unsigned tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    // naf: the signed-digit loop (88 entries instead of 102; miller.h: ate_naf_digit) -- values equal to libff's up to factors the
    // final exponentiation kills; never together with table output (the tables are libff's, row for row).  The identity is the
    // group law's: a workgroup that holds a G2 point at infinity -- libff feeds its step formulas the non-point (0, 1) then, and
    // what comes out is defined by those formulas alone -- keeps libff's binary loop ...
    // ... and the identity holds for points ON the curves only (modulo the curve equations the two loops' functions agree up to
    // vertical lines; off the curves they are different polynomials).  libff's Miller loop is defined for any coordinates -- it
    // just evaluates its formulas -- so a workgroup that holds a pair with P off E or Q off the twist keeps the binary loop too:
    // the values this library returns are then libff's for EVERY input, valid or not.  One lane per pair tests
    // Y^2 = X^3 + b Z^6 in Jacobian coordinates (a dozen products, once per workgroup).
    __shared__ int s_naf;
    __shared__ int s_off_curve;
    if (tid == 0) {
        s_off_curve = 0;
        s_produced = 0;
        s_consumed = 0;
    }
    __syncthreads();
    if (naf && tid < count) {
        const Jac<Fq2> &Q = g2[lo + tid];
        const Jac<Fq> &P = g1[lo + tid];
        bool bad = Q.Z.is_zero();
        if (!bad) {
            const P2 X = load2(Q.X), Y = load2(Q.Y), Z = load2(Q.Z);
            const P2 z2 = Z.sqr(), z6 = z2.sqr() * z2;
            bad = !(Y.sqr() == X.sqr() * X + fq2_constT<PB>(LSA_TWIST_B) * z6);
        }
        if (!bad && !P.Z.is_zero()) {                      // (P at infinity: both loops give an element of Fq4, which the final exponent kills)
            const PB x = PB::from_mont256(P.X), y = PB::from_mont256(P.Y), z = PB::from_mont256(P.Z);
            const PB z2 = z.sqr(), z6 = z2.sqr() * z2;
            const PB three = PB::one() + PB::one() + PB::one();
            bad = !(y.sqr() == x.sqr() * x + three * z6);
        }
        if (bad) atomicOr(&s_off_curve, 1);
    }
    __syncthreads();
    if (tid == 0) s_naf = naf && !s_off_curve;
    __syncthreads();
    const int use_naf = s_naf;
    const int entries = use_naf ? NAF_NUM_ENTRIES : ATE_NUM_COEFFS;
    if (tid < (unsigned)entries) kinds[tid] = (uint8_t)(use_naf ? tm_naf_entry_kind((int)tid) : tm_entry_kind((int)tid));
    if (tid < (unsigned)FU_PAIRS) {
        const bool have = tid < count;
        qp[tid] = g2 + lo + (have ? tid : 0);
        pp[tid] = g1 + lo + (have ? tid : 0);
        ng[tid] = have && flags ? (uint8_t)(flags[lo + tid] & 1) : (uint8_t)0;
        tout[tid] = have && tabs ? tabs[lo + tid] : nullptr;
        for (int s = 0; s < TP_RING; s++) rows[s][tid] = tpmem + tid * TP_STRIDE + TP_RAW + 3 * s;
    }
    __syncthreads();
    WaveLocalExec ex;
    G2Pre<WaveLocalExec, FU_PAIRS> pre{ex, g2mem};
    TP tp{ex, tpmem};
    if (wave == 0) {
        pre.setup(qp, count, tout);
        pre.setup_g1(pp, ng, count);
    } else {
        tp.setup();
    }
    __syncthreads();
    if (decouple) {
        // The two wavefronts as producer and consumer of a ring of TP_RING rows, ordered by two counters in LDS instead of a
        // workgroup barrier per entry (round 6).  In lockstep an entry costs the LONGER of the two sides -- the Fq12 chain on a
        // doubling entry (a squaring and a line product against three G2 rounds), the G2 side on an addition entry (four G2
        // rounds against one line product) -- so the kernel ran the sum of the maxima; decoupled it runs the maximum of the sums.
        if (wave == 0) {
#pragma unroll 1
            for (int e = 0; e < entries; e++) {
                while (__hip_atomic_load(&s_consumed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) + TP_RING <= e) __builtin_amdgcn_s_sleep(1);   // slot e % TP_RING is free
                pre.entry_rounds(kinds[e], e, tout, rows[e % TP_RING], true);
                __hip_atomic_store(&s_produced, e + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
#pragma unroll 1
            for (int e = 0; e < entries; e++) {
                while (__hip_atomic_load(&s_produced, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= e) __builtin_amdgcn_s_sleep(1);
                tp.entry(kinds[e], e);
                __hip_atomic_store(&s_consumed, e + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    } else {
    // step e: the G2 wavefront computes entry e + 1 (scaled by the pair's (px, py)) while the Fq12 wavefront consumes
    // entry e (one call site each)
#pragma unroll 1
    for (int e = -1; e < entries; e++) {
        if (wave == 0) {
            if (e + 1 < entries) pre.entry_rounds(kinds[e + 1], e + 1, tout, rows[(e + 1) % TP_RING], true);
        } else if (e >= 0) {
            tp.entry(kinds[e], e);
        }
        __syncthreads();
    }
    }
    if (wave == 1 && lane < 12u * FU_PAIRS) {
        const unsigned c = lane / 12, k = (lane % 12) >> 1, part = lane & 1;
        if (c < count) {
            const unsigned t = (k & 1) * 3 + (k >> 1);                  // tower position of w^k
            const Fq2S cf = tpmem[c * TP_STRIDE + TP_F + k];
            reinterpret_cast<Fq *>(&out[lo + c])[2 * t + part] = (part ? cf.c1 : cf.c0).to_mont256();
        }
    }
}


// internal table (x * 2^261 mod p, < 4p, 256-bit packed) -> libff's alt_bn128_ate_G2_precomp as bytes: QX, QY, then
// {ell_0, ell_VW, ell_VV} per step, canonical Montgomery Fq2 of 64 B.  One lane per Fq.
__global__ __launch_bounds__(256) void k_g2_tab_export(const uint32_t *const *__restrict__ tabs, size_t n, Fq *__restrict__ pub) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, per = 2 * (size_t)G2_PRECOMP_FQ2;
    if (i >= n * per) return;
    const size_t t = i / per, f = i % per;                       // public Fq number f of table t
    const size_t src = f < 4 ? 6 * (size_t)ATE_NUM_COEFFS + f : f - 4;   // the point sits behind the coefficients internally
    uint32_t w[8];
#pragma unroll
    for (int l = 0; l < 8; l++) w[l] = tabs[t][src * 8 + l];
    pub[i] = F29::unpack256(w).to_mont256();
}
__global__ __launch_bounds__(256) void k_g2_tab_import(const Fq *__restrict__ pub, size_t n, uint32_t *const *__restrict__ tabs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, per = 2 * (size_t)G2_PRECOMP_FQ2;
    if (i >= n * per) return;
    const size_t t = i / per, f = i % per;
    const size_t dst = f < 4 ? 6 * (size_t)ATE_NUM_COEFFS + f : f - 4;
    uint32_t w[8];
    F29::from_mont256(pub[i]).pack256(w);
#pragma unroll
    for (int l = 0; l < 8; l++) tabs[t][dst * 8 + l] = w[l];
}
// every row = the line (1, 0, 0): the table of a pair that is not there
__global__ void k_g2_tab_identity(uint32_t *tab) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (unsigned)TM_TAB_WORDS) return;
    uint32_t w[8];
    F29::one().pack256(w);
    const unsigned wi = i % TM_ROW_WORDS;
    uint32_t v = 0;
#pragma unroll
    for (int l = 0; l < 8; l++) if (wi == (unsigned)l) v = w[l];
    tab[i] = i < (unsigned)(ATE_NUM_COEFFS * TM_ROW_WORDS) ? v : 0u;
}

// Four accumulators per wavefront: accumulator a multiplies the Miller loops of pairs [acc_off[a], acc_off[a+1])
// (at most M <= TM_MAXM of them) over their tables.  flags bit 0: use -P (the conjugate value).
__global__ __launch_bounds__(64) void k_miller_tab(const Jac<Fq> *__restrict__ g1, const uint32_t *const *__restrict__ tabs,
                                                   const uint8_t *__restrict__ flags, const uint32_t *__restrict__ acc_off, size_t nacc,
                                                   unsigned M, const uint32_t *__restrict__ ident, Fq12 *__restrict__ out) {
    __shared__ Fq2S lds[TM_LDS_FQ2];
    __shared__ const uint32_t *tp[TM_CHUNKS * TM_MAXM];
    __shared__ const Jac<Fq> *pp[TM_CHUNKS * TM_MAXM];
    __shared__ uint8_t ng[TM_CHUNKS * TM_MAXM];
    __shared__ unsigned cnt[TM_CHUNKS];
    const size_t a0 = (size_t)blockIdx.x * TM_CHUNKS;
    if (a0 >= nacc) return;
    const unsigned lane = threadIdx.x;
    if (lane < (unsigned)(TM_CHUNKS * TM_MAXM)) {
        const unsigned c = lane / TM_MAXM, i = lane % TM_MAXM;
        unsigned lo = 0, len = 0;
        if (a0 + c < nacc) { lo = acc_off[a0 + c]; len = acc_off[a0 + c + 1] - lo; }
        const bool have = i < len;
        tp[lane] = have ? tabs[lo + i] : ident;
        pp[lane] = have ? g1 + lo + i : g1;
        ng[lane] = have ? (uint8_t)(flags[lo + i] & 1) : (uint8_t)0;
        if (i == 0) cnt[c] = len;
    }
    __syncthreads();
    WaveExec ex;
    TabMiller<WaveExec> tm{ex, lds, tp};
    tm.run(pp, ng, cnt, M);
    if (lane < 12u * TM_CHUNKS) {
        const unsigned c = lane / 12, k = (lane % 12) >> 1, part = lane & 1;
        if (a0 + c < nacc) {
            const unsigned t = (k & 1) * 3 + (k >> 1);                  // tower position of w^k
            const Fq2S cf = lds[c * TM_STRIDE + TM_F + k];
            reinterpret_cast<Fq *>(&out[a0 + c])[2 * t + part] = (part ? cf.c1 : cf.c0).to_mont256();
        }
    }
}

// One accumulator per wavefront (WTabMiller): the latency shape of the same job.
__global__ __launch_bounds__(64) void k_miller_wtab(const Jac<Fq> *__restrict__ g1, const uint32_t *const *__restrict__ tabs,
                                                    const uint8_t *__restrict__ flags, const uint32_t *__restrict__ acc_off, size_t nacc,
                                                    unsigned M, const uint32_t *__restrict__ ident, Fq12 *__restrict__ out) {
    __shared__ Fq2S lds[WT_LDS_FQ2];
    __shared__ uint32_t desc[256];
    __shared__ const uint32_t *tp[TM_MAXM];
    __shared__ const Jac<Fq> *pp[TM_MAXM];
    __shared__ uint8_t ng[TM_MAXM];
    const size_t a = blockIdx.x;
    if (a >= nacc) return;
    const unsigned lane = threadIdx.x;
    const unsigned lo = acc_off[a], len = acc_off[a + 1] - lo;
    if (lane < (unsigned)TM_MAXM) {
        const bool have = lane < len;
        tp[lane] = have ? tabs[lo + lane] : ident;
        pp[lane] = have ? g1 + lo + lane : g1;
        ng[lane] = have ? (uint8_t)(flags[lo + lane] & 1) : (uint8_t)0;
    }
    __syncthreads();
    WaveExec ex;
    WTabMiller<WaveExec> wt{ex, lds, tp, desc};
    wt.run(pp, ng, len, M);
    if (lane < 12) {
        const unsigned k = lane >> 1, part = lane & 1, t = (k & 1) * 3 + (k >> 1);
        const Fq2S cf = lds[WT_F + k];
        reinterpret_cast<Fq *>(&out[a])[2 * t + part] = (part ? cf.c1 : cf.c0).to_mont256();
    }
}

// One accumulator per workgroup of 192 lanes, at most RT_MAXM pairs each (tmiller.h, rt_miller_run): the row-engine version.
// K > 1: K workgroups per accumulator (blockIdx = a * K + r), out[a * K + r] = the factor of the steps r mod K
__global__ __launch_bounds__(192) void k_miller_rtab(const Jac<Fq> *__restrict__ g1, const uint32_t *const *__restrict__ tabs,
                                                     const uint8_t *__restrict__ flags, const uint32_t *__restrict__ acc_off, size_t nacc,
                                                     unsigned M, const uint32_t *__restrict__ ident, Fq12 *__restrict__ out, unsigned K, RtSplit split) {
    __shared__ Fq2S lds[RT_LDS_FQ2];
    __shared__ const uint32_t *tp[RT_MAXM];
    __shared__ const Jac<Fq> *pp[RT_MAXM];
    __shared__ uint8_t ng[RT_MAXM];
    __shared__ uint8_t own[ATE_NUM_COEFFS + 2];
    const size_t a = blockIdx.x / K;
    const unsigned r = blockIdx.x % K;
    if (a >= nacc) return;
    const unsigned lane = threadIdx.x;
    const unsigned lo = acc_off[a], len = acc_off[a + 1] - lo;
    if (lane < (unsigned)RT_MAXM) {
        const bool have = lane < len;
        tp[lane] = have ? tabs[lo + lane] : ident;
        pp[lane] = have ? g1 + lo + lane : g1;
        ng[lane] = have ? (uint8_t)(flags[lo + lane] & 1) : (uint8_t)0;
    }
    __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
    rt_miller_run(lds, tp, pp, ng, len, M, split.start[r], split.start[r + 1], own);
#endif
    if (lane < 12) {
        const unsigned k = lane >> 1, part = lane & 1, t = (k & 1) * 3 + (k >> 1);
        const Fq2S cf = lds[RT_F + k];
        reinterpret_cast<Fq *>(&out[blockIdx.x])[2 * t + part] = (part ? cf.c1 : cf.c0).to_mont256();
    }
}

#define HIPCHK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return LSA_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

// One lane per pairing (miller.h: miller_one): the least total work and no cooperation between lanes -- the fallback of
// the family (LSA_MILLER_KERNEL=3) beside the fused kernel (k_miller_fused), which every fresh pair takes by default.
// (Rounds 1-2 also had one pairing per wavefront, six and twelve lanes per pairing -- k_miller_wave / _g6 / _g12; they
// lost every shape to the fused kernel in rounds 3 and 4 and were removed in round 5.)
// the K ranges of rt_split per number of pairs per accumulator (K = 1, 2, 4, 8), made once
static RtSplit rt_split_for(unsigned M, unsigned K) {
    static const std::vector<RtSplit> tab = [] {
        std::vector<RtSplit> t;
        for (unsigned k = 0; k < 4; k++)
            for (unsigned m = 0; m <= (unsigned)RT_MAXM; m++) t.push_back(rt_split(m ? m : 1u, 1u << k));
        return t;
    }();
    const unsigned k = K >= 8 ? 3u : (K >= 4 ? 2u : (K >= 2 ? 1u : 0u));
    return tab[k * ((unsigned)RT_MAXM + 1) + (M <= (unsigned)RT_MAXM ? M : (unsigned)RT_MAXM)];
}

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
    {
        // three wavefronts per element: the one-phase row product of w12.h (the two-wavefront, two-phase engine it
        // replaced in round 4 -- LSA_FINAL_EXP_LANES=128 -- was removed in round 5); the one-lane kernel below is the fallback
        // (+ a fourth one that takes the chain's inversion off its critical path; LSA_FE_HELPER=0: without)
        static const bool helper = [] { const char *e = getenv("LSA_FE_HELPER"); return !(e && *e == '0'); }();
        if (helper) hipLaunchKernelGGL(k_final_exp_wave_h, dim3((unsigned)n), dim3(256), 0, st, (const Fq12 *)d_in, n, (Fq12 *)d_out);
        else hipLaunchKernelGGL(k_final_exp_wave<192>, dim3((unsigned)n), dim3(192), 0, st, (const Fq12 *)d_in, n, (Fq12 *)d_out);
    }
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
        if (m < 16384) hipLaunchKernelGGL(k_fq12_prod8_wave, dim3((unsigned)m), dim3(192), 0, st, a, n, b, 8u);      // three wavefronts: the one-phase row product (w12.h)
        else hipLaunchKernelGGL(k_fq12_prod8, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, st, a, n, b);
        Fq12 *t = a; a = b; b = t;
        n = m;
    }
    HIPCHK(hipGetLastError());
    *result = a;
    return LSA_OK;
}

// d_out[j] = prod d_in[off[j] .. off[j+1]), j < nseg; d_off: nseg + 1 offsets on the device
int fq12_segment_products_device(const void *d_in, const uint64_t *d_off, size_t nseg, void *d_out, hipStream_t st) {
    if (nseg == 0) return LSA_OK;
    hipLaunchKernelGGL(k_fq12_prod_seg_wave, dim3((unsigned)nseg), dim3(192), 0, st, (const Fq12 *)d_in, d_off, (Fq12 *)d_out);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}

// ---- tables
size_t g2_table_words() { return (size_t)TM_TAB_WORDS; }
size_t g2_precomp_public_bytes() { return (size_t)G2_PRECOMP_BYTES; }
int g2_precomp_device(const void *d_g2, size_t n, uint32_t *const *d_tabs, hipStream_t st) {
    if (n == 0) return LSA_OK;
    hipLaunchKernelGGL(k_g2_precomp, dim3((unsigned)((n + GP_GROUPS - 1) / GP_GROUPS)), dim3(64), 0, st, (const Jac<Fq2> *)d_g2, n, d_tabs);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}
int g2_table_export_device(const uint32_t *const *d_tabs, size_t n, void *d_public, hipStream_t st) {
    if (n == 0) return LSA_OK;
    const size_t total = n * 2 * (size_t)G2_PRECOMP_FQ2;
    hipLaunchKernelGGL(k_g2_tab_export, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_tabs, n, (Fq *)d_public);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}
int g2_table_import_device(const void *d_public, size_t n, uint32_t *const *d_tabs, hipStream_t st) {
    if (n == 0) return LSA_OK;
    const size_t total = n * 2 * (size_t)G2_PRECOMP_FQ2;
    hipLaunchKernelGGL(k_g2_tab_import, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const Fq *)d_public, n, d_tabs);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}
int g2_table_identity_device(uint32_t *d_tab, hipStream_t st) {
    hipLaunchKernelGGL(k_g2_tab_identity, dim3((TM_TAB_WORDS + 255) / 256), dim3(256), 0, st, d_tab);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}
int miller_fused_device(const void *d_g1, const void *d_g2, const uint8_t *d_flags, uint32_t *const *d_tabs, size_t n, void *d_out, hipStream_t st, bool gt_only) {
    if (n == 0) return LSA_OK;
    // values that only ever leave through a final exponentiation take the signed-digit loop (LSA_MILLER_NAF=0: libff's binary one)
    static const bool allow_naf = getenv("LSA_MILLER_NAF") == nullptr || getenv("LSA_MILLER_NAF")[0] != '0';
    const int naf = gt_only && allow_naf && d_tabs == nullptr ? 1 : 0;
    // LSA_FUSED_LOCKSTEP=1: one workgroup barrier per table entry, as before round 6
    static const int decouple = getenv("LSA_FUSED_LOCKSTEP") && getenv("LSA_FUSED_LOCKSTEP")[0] == '1' ? 0 : 1;
    hipLaunchKernelGGL(k_miller_fused, dim3((unsigned)((n + FU_PAIRS - 1) / FU_PAIRS)), dim3(128), 0, st, (const Jac<Fq> *)d_g1, (const Jac<Fq2> *)d_g2, d_flags,
                       d_tabs, n, (Fq12 *)d_out, naf, decouple);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}
unsigned miller_tab_max_pairs() { return (unsigned)TM_MAXM; }
static void *g_rt_parts = nullptr;         // the factors of split Miller loops (32 accumulators x 8 workgroups)
void miller_split_release() { if (g_rt_parts) { (void)hipFree(g_rt_parts); g_rt_parts = nullptr; } }
int miller_tab_device(const void *d_g1, const uint32_t *const *d_tabs, const uint8_t *d_flags, const uint32_t *d_acc_off, size_t nacc, unsigned M,
                      const uint32_t *d_ident, void *d_out, hipStream_t st) {
    if (nacc == 0) return LSA_OK;
    if (M == 0 || M > (unsigned)TM_MAXM) { set_error("miller_tab: %u pairs per accumulator (1..%d)", M, TM_MAXM); return LSA_ERR_INVALID; }
    // up to two wavefronts per SIMD: one accumulator per wavefront (~170 K instructions each); beyond, four per
    // wavefront (~340 K: half the instructions per accumulator).  LSA_MILLER_TAB = 1 / 4 forces a shape.
    static const int force = getenv("LSA_MILLER_TAB") ? atoi(getenv("LSA_MILLER_TAB")) : 0;
    const bool wave = force ? force == 1 : nacc <= 2048;
    // few accumulators of one or two pairs (the verifiers' lone checks): the row engine, 0.2 ms against 0.35 (LSA_MILLER_ROWS=0: off)
    static const bool rows = getenv("LSA_MILLER_ROWS") == nullptr || atoi(getenv("LSA_MILLER_ROWS")) != 0;
    if (wave && rows && M <= (unsigned)RT_MAXM && nacc <= 512) {
        // up to 128 accumulators: K = 8, 4 or 2 workgroups (at most one per CU in all) share each loop's line products (rt_miller_run:
        // contiguous ranges of steps, cut even by rt_split), k_fq12_prod8_wave multiplies their K factors -- 65 / 73 chain links
        // (one / two pairs per accumulator) with K = 8 instead of 63 + 102 M (LSA_MILLER_SPLIT=1: off)
        static const unsigned split = getenv("LSA_MILLER_SPLIT") ? (unsigned)atoi(getenv("LSA_MILLER_SPLIT")) : 8u;
        const unsigned K = split == 1 ? 1u : (nacc <= 32 ? 8u : (nacc <= 64 ? 4u : (nacc <= 128 ? 2u : 1u)));
        if (K > 1) {
            if (!g_rt_parts && hipMalloc(&g_rt_parts, 256 * sizeof(Fq12)) != hipSuccess) { g_rt_parts = nullptr; set_error("miller_tab: hipMalloc failed"); return LSA_ERR_NOMEM; }
            hipLaunchKernelGGL(k_miller_rtab, dim3((unsigned)nacc * K), dim3(192), 0, st, (const Jac<Fq> *)d_g1, d_tabs, d_flags, d_acc_off, nacc, M, d_ident,
                               (Fq12 *)g_rt_parts, K, rt_split_for((unsigned)M, K));
            hipLaunchKernelGGL(k_fq12_prod8_wave, dim3((unsigned)nacc), dim3(192), 0, st, (const Fq12 *)g_rt_parts, nacc * K, (Fq12 *)d_out, K);
        } else
            hipLaunchKernelGGL(k_miller_rtab, dim3((unsigned)nacc), dim3(192), 0, st, (const Jac<Fq> *)d_g1, d_tabs, d_flags, d_acc_off, nacc, M, d_ident,
                               (Fq12 *)d_out, 1u, rt_split_for((unsigned)M, 1u));
    } else if (wave)
        hipLaunchKernelGGL(k_miller_wtab, dim3((unsigned)nacc), dim3(64), 0, st, (const Jac<Fq> *)d_g1, d_tabs, d_flags, d_acc_off, nacc, M, d_ident,
                           (Fq12 *)d_out);
    else
        hipLaunchKernelGGL(k_miller_tab, dim3((unsigned)((nacc + TM_CHUNKS - 1) / TM_CHUNKS)), dim3(64), 0, st, (const Jac<Fq> *)d_g1, d_tabs, d_flags,
                           d_acc_off, nacc, M, d_ident, (Fq12 *)d_out);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}

size_t fq12_bytes() { return sizeof(Fq12); }   // libff layout in memory

}  // namespace lsa
