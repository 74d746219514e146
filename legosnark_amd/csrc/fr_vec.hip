// legosnark_amd/csrc/fr_vec.hip -- the O(2^d) Fr loops around the MSMs of the CPpoly / CPhad /
// CPsc provers, as streaming kernels on device-resident vectors.
//
//   k_fold_pairs   one round of CPPoly::prove's witness recursion
//                  (/root/reference/src/gadgets/poly.h:55-67):
//                      w[p]  = v[2p+1] - v[2p]
//                      v'[p] = -v[2p]*(r-1) + v[2p+1]*r
//                  each lane reads one adjacent pair (64 contiguous bytes) and writes 2 x 32 B.
//   k_fold_halves  one round of DPMle::pushRandomness
//                  (/root/reference/src/prototools/mle.h:199-210):
//                      cur[p] = old[p]*(1-r) + old[p+half]*r
//   evalMLE        MultiVPolyT::evalMLE (/root/reference/src/prototools/polytools.h:207-234): the reference builds
//                  the table of eq-monomials and takes a dot product; folding one variable after the other (here:
//                  from the bottom, the pairs recursion without its coefficients) gives the same element of Fr, and
//                  Fr values are canonical Montgomery residues, so the 32 output bytes are identical.
// Both are HBM streams.  Round 5: ONE product per output element instead of two -- v' = v0 + r (v1 - v0) is the same
// field element as -v0 (r - 1) + v1 r, and Fr values are canonical, so the bytes are the reference's -- on the 29-bit
// limbs of fr29.h (the round's r is put into 2^261 form once per workgroup, so data words go in and out without
// conversion: a fifth of the instructions of two 32-bit CIOS products).  The recursions run up to TWELVE rounds per pass
// over the data (k_fold_pairs_fused: two rounds in registers, the others as a tree in LDS) and their last rounds inside
// ONE workgroup (k_fold_pairs_tail: a barrier between rounds instead of a launch each): the witness recursion at d = 24
// reads v once and writes every coefficient once (1.07 GB), evalMLE reads its table once (0.54 GB).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "fp.h"
#include "fr29.h"
#include "msm.h"
#include "capi_internal.h"

namespace lsa {


// the round's scalar in 2^261 form (Montgomery value * 32: the canonical words of r * 2^261), as limbs
__device__ __forceinline__ Fr29 fr_to_261(const Fr &r) { return Fr29::from_words(r * Fr::from_u32(32)); }
// x * r as libff words: x's words are read as the 2^261 form of x / 32, the product with r in 2^261 form is x r 2^256
__device__ __forceinline__ Fr fr_mul_261(const Fr &x, const Fr29 &r261) { return mul(Fr29::from_words(x), r261).canonical2().to_words(); }

// Several consecutive rounds in ONE pass over the data: round j + 1 pairs up neighbouring outputs of round j, so a
// workgroup that owns a block of 2^bl neighbouring inputs (bl <= 12) runs tr <= bl rounds on its own -- two in registers (a
// lane takes four neighbouring inputs to one value), the others as a tree over the block's values in LDS -- and the
// intermediate vectors never reach memory; the block's 2^(bl - tr) results go to vout.  The rounds' scalars are put into
// 2^261 form once per workgroup and held in scalar registers.
//   m0: outputs of this launch's first round over ALL blocks (blocks << (bl - 1)); round j's witness coefficients go to
//   w0 + m0 + m0 / 2 + ... (j terms), a block's share of them contiguous.
static constexpr unsigned FUSE_BLOCK_LOG = 12;
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_fold_pairs_fused(const Fr *__restrict__ v, size_t blocks, unsigned bl, unsigned tr,
                                                                                                const Fr *__restrict__ r_ptr, Fr *__restrict__ w0, size_t m0,
                                                                                                Fr *__restrict__ vout) {
    __shared__ uint32_t s_r[FUSE_BLOCK_LOG][9];
    __shared__ Fr s_x[(size_t)1 << (FUSE_BLOCK_LOG - 2)];
    if (threadIdx.x < tr) {
        const Fr29 c = fr_to_261(r_ptr[threadIdx.x]);
#pragma unroll
        for (int k = 0; k < 9; k++) s_r[threadIdx.x][k] = c.l[k];
    }
    __syncthreads();
    auto scalar = [&](unsigned j) {
        Fr29 c;                                                // the same value in every lane: kept in scalar registers
#pragma unroll
        for (int k = 0; k < 9; k++) c.l[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_r[j][k]);
        return c;
    };
    const size_t vals = (size_t)1 << (bl - 2);               // values a block leaves in LDS after its first two rounds
    for (size_t blk = blockIdx.x; blk < blocks; blk += gridDim.x) {
        {
            const Fr29 r0 = scalar(0), r1 = scalar(1);
            const Fr *in = v + (blk << bl);
            Fr *wa = w0 + (blk << (bl - 1)), *wb = w0 + m0 + (blk << (bl - 2));
#pragma unroll 1
            for (size_t u = threadIdx.x; u < vals; u += 256) {
                const Fr x0 = in[4 * u], x1 = in[4 * u + 1], x2 = in[4 * u + 2], x3 = in[4 * u + 3];
                const Fr d0 = x1 - x0, d1 = x3 - x2;
                wa[2 * u] = d0;
                wa[2 * u + 1] = d1;
                const Fr y0 = x0 + fr_mul_261(d0, r0), y1 = x2 + fr_mul_261(d1, r0);
                const Fr d = y1 - y0;
                wb[u] = d;
                s_x[u] = y0 + fr_mul_261(d, r1);
            }
        }
        __syncthreads();
        size_t woff = m0 + (m0 >> 1);                        // where round 2's coefficients start
        for (unsigned j = 2; j < tr; j++) {
            const size_t cnt = (size_t)1 << (bl - 1 - j);    // this block's outputs in round j: at most 512
            const Fr29 rj = scalar(j);
            Fr *wj = w0 + woff + (blk << (bl - 1 - j));
            Fr y[2];
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const size_t u = threadIdx.x + (size_t)it * 256;
                if (u < cnt) {
                    const Fr a = s_x[2 * u], b = s_x[2 * u + 1];
                    const Fr d = b - a;
                    wj[u] = d;
                    y[it] = a + fr_mul_261(d, rj);
                }
            }
            __syncthreads();                                 // every pair has been read
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const size_t u = threadIdx.x + (size_t)it * 256;
                if (u < cnt) s_x[u] = y[it];
            }
            __syncthreads();
            woff += m0 >> j;
        }
        const size_t outs = (size_t)1 << (bl - tr);
        for (size_t u = threadIdx.x; u < outs; u += 256) vout[(blk << (bl - tr)) + u] = s_x[u];
        __syncthreads();                                     // s_x is free for the next block
    }
}

// evalMLE's passes: the same blocks and rounds, on LAZY 29-bit-limb values -- no coefficient leaves the kernel, so
// nothing between a block's inputs and its results has to be canonical: a round is
//   d = b - a + K_j r   (K_j = 2 (j + 1) >= the bound of a: non-negative),   y = a + d r_j / 2^261   (the product is < 2r)
// and a value that enters round j below (1 + 2j) r leaves it below (3 + 2j) r: twelve rounds stay far below the 121 r the
// product accepts.  Per product: a limb subtraction, the product, a limb addition -- no unpacking, canonical form or
// packing.  A block's results are brought back to canonical words by one more product (with 2^261: the shifted form's one).
// (Measured and not kept: taking variables 6 and 7 of a block in the register rounds, so that a wavefront's loads are 2 KB
// runs instead of 64 pieces of 128 B -- 211-232 us per pass either way -- and requesting the next slice's inputs before
// the current ones are used -- 262 us.  The pass is bound by the product: 16.8 M of them in 0.21 ms.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_eval_mle_fused(const Fr *__restrict__ v, size_t blocks, unsigned bl, unsigned tr,
                                                                                              const Fr *__restrict__ r_ptr, Fr *__restrict__ vout) {
    __shared__ uint32_t s_r[FUSE_BLOCK_LOG][9], s_off[FUSE_BLOCK_LOG][9];
    __shared__ uint32_t s_x[((size_t)1 << (FUSE_BLOCK_LOG - 2)) * 9];      // lazy values, 9 words apart: conflict-free
    if (threadIdx.x < tr) {
        const unsigned j = threadIdx.x;
        const Fr29 c = fr_to_261(r_ptr[j]);
        uint64_t carry = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            s_r[j][k] = c.l[k];
            carry += (uint64_t)Fr29::r(k) * (2u * (j + 1u));               // K_j r as tight limbs (the top limb takes the rest)
            s_off[j][k] = k < 8 ? (uint32_t)carry & Fr29::MASK : (uint32_t)carry;
            carry >>= 29;
        }
    }
    __syncthreads();
    auto uniform = [&](const uint32_t (*tab)[9], unsigned j) {
        Fr29 c;
#pragma unroll
        for (int k = 0; k < 9; k++) c.l[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)tab[j][k]);
        return c;
    };
    // y = a + (b - a + off) * rj
    auto round = [](const Fr29 &a, const Fr29 &b, const Fr29 &off, const Fr29 &rj) {
        Fr29 d;
        int32_t c = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int32_t t = (int32_t)b.l[k] - (int32_t)a.l[k] + (int32_t)off.l[k] + c;
            if (k < 8) { d.l[k] = (uint32_t)t & Fr29::MASK; c = t >> 29; }
            else d.l[k] = (uint32_t)t;
        }
        return add(a, mul(d, rj));
    };
    auto lds_get = [&](size_t u) {
        Fr29 x;
#pragma unroll
        for (int k = 0; k < 9; k++) x.l[k] = s_x[u * 9 + k];
        return x;
    };
    auto lds_put = [&](size_t u, const Fr29 &x) {
#pragma unroll
        for (int k = 0; k < 9; k++) s_x[u * 9 + k] = x.l[k];
    };
    const size_t vals = (size_t)1 << (bl - 2);
    for (size_t blk = blockIdx.x; blk < blocks; blk += gridDim.x) {
        {
            const Fr29 r0 = uniform(s_r, 0), r1 = uniform(s_r, 1), o0 = uniform(s_off, 0), o1 = uniform(s_off, 1);
            const Fr *in = v + (blk << bl);
#pragma unroll 1
            for (size_t u = threadIdx.x; u < vals; u += 256) {
                const Fr x0 = in[4 * u], x1 = in[4 * u + 1], x2 = in[4 * u + 2], x3 = in[4 * u + 3];
                const Fr29 y0 = round(Fr29::from_words(x0), Fr29::from_words(x1), o0, r0);
                const Fr29 y1 = round(Fr29::from_words(x2), Fr29::from_words(x3), o0, r0);
                lds_put(u, round(y0, y1, o1, r1));
            }
        }
        __syncthreads();
        for (unsigned j = 2; j < tr; j++) {
            const size_t cnt = (size_t)1 << (bl - 1 - j);
            const Fr29 rj = uniform(s_r, j), oj = uniform(s_off, j);
            Fr29 y[2];
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const size_t u = threadIdx.x + (size_t)it * 256;
                if (u < cnt) y[it] = round(lds_get(2 * u), lds_get(2 * u + 1), oj, rj);
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const size_t u = threadIdx.x + (size_t)it * 256;
                if (u < cnt) lds_put(u, y[it]);
            }
            __syncthreads();
        }
        const size_t outs = (size_t)1 << (bl - tr);
        for (size_t u = threadIdx.x; u < outs; u += 256) vout[(blk << (bl - tr)) + u] = mul(lds_get(u), Fr29::one()).canonical2().to_words();
        __syncthreads();
    }
}

// evalMLE from 2^16 entries on, as the reference states it (MultiVPolyT::evalMLE, polytools.h:207-234: the table of
// eq-monomials and a dot product) -- with the monomial factored over three groups of index bits,
//   eq(idx) = E6[idx & 63] * WL[(idx >> 6) & 511] * WH[idx >> 15],
// so that the n products are data x constant, independent of each other (no tree, no barrier between them), and a
// wavefront's loads are contiguous 2-KB runs: lane l of a wavefront sums v[(C 512 + j) 64 + l] * WL[j] over its j's (the
// constant is the same in every lane), multiplies the sum by WH[C] once per 128 elements, and by E6[l] once at the end.
// The three tables (64 + 512 + 2^(d-15) entries in 2^261 form) cost 2^(d-15) (d - 15) + 5 000 products to build.
// Four products share ONE Montgomery reduction (fr29.h: the column sums of four schoolbook products fit 64 bits), and the
// reduced partial sums stay lazy on fr29.h's limbs (at most 32 of < 2r each per unit).
static constexpr unsigned MLE_LO = 6, MLE_MID = 9, MLE_MIN_D = 16;
__global__ __launch_bounds__(256) void k_mle_eq_tables(const Fr *__restrict__ r, unsigned d, Fr *__restrict__ E6, Fr *__restrict__ WL, Fr *__restrict__ WH) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n6 = (size_t)1 << MLE_LO, nl = (size_t)1 << MLE_MID, nh = (size_t)1 << (d - MLE_LO - MLE_MID);
    unsigned first, bits;
    size_t idx;
    Fr *dst;
    if (t < n6) { first = 0; bits = MLE_LO; idx = t; dst = E6 + t; }
    else if (t < n6 + nl) { first = MLE_LO; bits = MLE_MID; idx = t - n6; dst = WL + idx; }
    else if (t < n6 + nl + nh) { first = MLE_LO + MLE_MID; bits = d - MLE_LO - MLE_MID; idx = t - n6 - nl; dst = WH + idx; }
    else return;
    Fr acc = Fr::from_u32(32);                                   // (value * 32: the words of the 2^261 form)
    for (unsigned i = 0; i < bits; i++) {
        const Fr ri = r[first + i];
        acc = acc * (((idx >> i) & 1) ? ri : Fr::one() - ri);
    }
    *dst = acc;
}
__global__ __launch_bounds__(256) void k_mle_dot(const Fr *__restrict__ v, unsigned d, unsigned ju_log, const Fr *__restrict__ E6, const Fr *__restrict__ WL,
                                                 const Fr *__restrict__ WH, Fr *__restrict__ partial) {
    __shared__ Fr s_red[256];
    // a unit: 2^ju_log values of j for one C; wavefront-uniform values are made so explicitly: the constants WL[j], WH[C]
    // then come through scalar loads into scalar registers
    const unsigned lane = threadIdx.x & 63, wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned ju = 1u << ju_log, per_c = (1u << MLE_MID) >> ju_log;
    const size_t waves = (size_t)gridDim.x * 4, units = ((size_t)per_c) << (d - MLE_LO - MLE_MID);
    const Fr29 one = Fr29::one();
    Fr29 total = Fr29::zero();
    unsigned since = 0;
    for (size_t u = (size_t)blockIdx.x * 4 + wave; u < units; u += waves) {
        const size_t C = u / per_c;
        const unsigned j0 = (unsigned)(u % per_c) << ju_log;
        const Fr *in = v + (((C << MLE_MID) + j0) << MLE_LO) + lane;
        const Fr *wl = WL + j0;
        Fr29 acc = Fr29::zero();
#pragma unroll 1
        for (unsigned jj = 0; jj < ju; jj += 4) {
            const Fr x0 = in[(size_t)(jj + 0) << MLE_LO], x1 = in[(size_t)(jj + 1) << MLE_LO], x2 = in[(size_t)(jj + 2) << MLE_LO], x3 = in[(size_t)(jj + 3) << MLE_LO];
            // four products, ONE reduction (fr29.h: fr29_wide_*): 4 * 81 + 81 multiply-adds instead of 4 * 162
            Fr29Wide w = fr29_wide_zero();
            fr29_wide_mac(w, Fr29::from_words(x0), Fr29::from_words(wl[jj]));
            fr29_wide_mac(w, Fr29::from_words(x1), Fr29::from_words(wl[jj + 1]));
            fr29_wide_mac(w, Fr29::from_words(x2), Fr29::from_words(wl[jj + 2]));
            fr29_wide_mac(w, Fr29::from_words(x3), Fr29::from_words(wl[jj + 3]));
            acc = add(acc, fr29_wide_reduce(w));                  // < 2r each: at most 32 of them per unit, 64r
        }
        total = add(total, mul(acc, Fr29::from_words(WH[C])));
        if (++since == 32) { total = mul(total, one); since = 0; }
    }
    s_red[threadIdx.x] = mul(total, Fr29::from_words(E6[lane])).canonical2().to_words();
    __syncthreads();
    for (unsigned sft = 128; sft >= 1; sft >>= 1) {
        if (threadIdx.x < sft) s_red[threadIdx.x] = s_red[threadIdx.x] + s_red[threadIdx.x + sft];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s_red[0];
}
// out = sum of n partials (one workgroup)
__global__ __launch_bounds__(256) void k_mle_finish(const Fr *__restrict__ partial, unsigned n, Fr *__restrict__ out) {
    __shared__ Fr lds[256];
    Fr acc = Fr::zero();
    for (unsigned b = threadIdx.x; b < n; b += 256) acc = acc + partial[b];
    lds[threadIdx.x] = acc;
    __syncthreads();
    for (unsigned sft = 128; sft >= 1; sft >>= 1) {
        if (threadIdx.x < sft) lds[threadIdx.x] = lds[threadIdx.x] + lds[threadIdx.x + sft];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = lds[0];
}

// the remaining rounds (m_first, m_first / 2, ... 1 output elements) by one workgroup; round j reads what round j - 1
// wrote (ping-pong between bufA and bufB, continuing the caller's parity), r_ptr / w advance per round.
// w == nullptr: evalMLE's last rounds -- no coefficients kept; out != nullptr: the last round's value goes there
__global__ __launch_bounds__(256) void k_fold_pairs_tail(const Fr *src, size_t m_first, const Fr *__restrict__ r_ptr, Fr *w, Fr *bufA, Fr *bufB,
                                                         unsigned parity, Fr *out) {
    for (size_t m = m_first; m >= 1; m >>= 1) {
        const Fr29 r261 = fr_to_261(*r_ptr);
        Fr *dst = (m == 1 && out) ? out : ((parity & 1u) ? bufB : bufA);
        for (size_t p = threadIdx.x; p < m; p += 256) {
            const Fr a = src[2 * p], b = src[2 * p + 1];
            const Fr d = b - a;
            if (w) w[p] = d;
            dst[p] = a + fr_mul_261(d, r261);
        }
        __syncthreads();                                 // the block's global writes are visible to the block
        src = dst;
        if (w) w += m;
        r_ptr++;
        parity++;
    }
}

// The one-product-per-element streams (pushRandomness, the suffix update, the eq table): a lane takes STREAM_UNROLL elements
// a grid stride apart and issues all their loads before the first product -- with one 32-byte load in flight per lane the
// chip holds too few bytes in flight to cover HBM's latency (the suffix update ran at 0.34 of the peak, the fold -- two loads
// per element -- at 0.58).
static constexpr int STREAM_UNROLL = 4;
// cur may alias old: lane p reads old[p], old[p+half] and writes cur[p] only
// the round's scalar in 2^261 form: converted by ONE lane of the workgroup (a 32-bit-limb Montgomery product: as many
// instructions as a whole element of the stream costs, paid by every lane before), handed round through LDS, kept in SGPRs
__device__ __forceinline__ Fr29 block_scalar_261(const Fr &k, uint32_t (&s_k)[9]) {
    if (threadIdx.x == 0) {
        const Fr29 c = fr_to_261(k);
#pragma unroll
        for (int i = 0; i < 9; i++) s_k[i] = c.l[i];
    }
    __syncthreads();
    Fr29 c;
#pragma unroll
    for (int i = 0; i < 9; i++) c.l[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_k[i]);
    return c;
}
// (stream kernels: their loads and stores are marked non-temporal -- every byte is touched once; round 6: pushRandomness at half = 2^23
// 0.181 -> 0.177 ms, the suffix update equal; -DLSA_STREAM_TEMPORAL: plain accesses; profiles/r06_w9_stream_kernels_nontemporal.txt)
typedef uint32_t lsa_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Fr fr_stream_load(const Fr *p) {
#if !defined(LSA_STREAM_TEMPORAL)
    const lsa_u32x4 lo = __builtin_nontemporal_load(reinterpret_cast<const lsa_u32x4 *>(p)), hi = __builtin_nontemporal_load(reinterpret_cast<const lsa_u32x4 *>(p) + 1);
    Fr x;
    x.l[0] = lo.x; x.l[1] = lo.y; x.l[2] = lo.z; x.l[3] = lo.w; x.l[4] = hi.x; x.l[5] = hi.y; x.l[6] = hi.z; x.l[7] = hi.w;
    return x;
#else
    return *p;
#endif
}
__device__ __forceinline__ void fr_stream_store(Fr *p, const Fr &x) {
#if !defined(LSA_STREAM_TEMPORAL)
    lsa_u32x4 lo, hi;
    lo.x = x.l[0]; lo.y = x.l[1]; lo.z = x.l[2]; lo.w = x.l[3]; hi.x = x.l[4]; hi.y = x.l[5]; hi.z = x.l[6]; hi.w = x.l[7];
    __builtin_nontemporal_store(lo, reinterpret_cast<lsa_u32x4 *>(p));
    __builtin_nontemporal_store(hi, reinterpret_cast<lsa_u32x4 *>(p) + 1);
#else
    *p = x;
#endif
}
__global__ __launch_bounds__(256) void k_fold_halves(const Fr *old, size_t half, const Fr *__restrict__ r_ptr, Fr *cur) {
    __shared__ uint32_t s_k[9];
    const Fr29 r261 = block_scalar_261(*r_ptr, s_k);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t p0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p0 < half; p0 += stride * STREAM_UNROLL) {
        Fr a[STREAM_UNROLL], b[STREAM_UNROLL];
#pragma unroll
        for (int u = 0; u < STREAM_UNROLL; u++) {
            const size_t p = p0 + u * stride;
            if (p < half) { a[u] = fr_stream_load(old + p); b[u] = fr_stream_load(old + p + half); }
        }
#pragma unroll
        for (int u = 0; u < STREAM_UNROLL; u++) {
            const size_t p = p0 + u * stride;
            if (p < half) fr_stream_store(cur + p, a[u] + fr_mul_261(b[u] - a[u], r261));   // = a (1 - r) + old[p + half] r
        }
    }
}
// ------------------------------------------------------------------------------------
// Sumcheck round polynomial (CPSumcheck::make_new_h_poly, /root/reference/src/gadgets/sumcheck.h:85-106):
//   h_j(X) = sum_{p < half} betaPoly(j,p)(X) * prod_t mlePoly_t(j,p)(X)
// with mlePoly_t(j,p) = V_t[p] (1-X) + V_t[p+half] X          (DPMle::getMLEPoly, mle.h:217-226)
// and  betaPoly(j,p)  = eqbit_poly(rho_j) * pre * suff[p]      (DPBeta::getBetaPoly, mle.h:74-82),
// i.e. h_j = eqbit_poly(rho_j) * pre * S(X),  S(X) = sum_p suff[p] prod_t (V_t[p] + (V_t[p+half] - V_t[p]) X).
// k_sumcheck_partial accumulates the m+1 coefficients of S per lane over a grid-stride loop and
// tree-sums them per block through LDS; k_sumcheck_finish adds the block partials and applies
// the linear factor.  Fr values are canonical, so the coefficients equal the reference's
// whatever the summation order.  Streams 32*(2m+1) bytes per p.
// ------------------------------------------------------------------------------------
static constexpr int SC_MAX_M = 4;
struct ScTables { const Fr *t[SC_MAX_M]; };
// x / 2 mod r on a canonical representative (the fixed factor 2^256 of libff's words commutes with the halving)
__device__ __forceinline__ Fr fr_half(const Fr &x) {
    const uint32_t odd = 0u - (x.l[0] & 1u);
    uint32_t t[9];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)x.l[i] + (FrParams::MOD[i] & odd);
        t[i] = (uint32_t)c;
        c >>= 32;
    }
    t[8] = (uint32_t)c;
    Fr o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.l[i] = (t[i] >> 1) | (t[i + 1] << 31);
    return o;
}

// The products run on fr29.h's limbs with LAZY sums (a coefficient of q is a sum of two products: < 4r; the lane's running
// sums are brought back below 2r every sixteen indices).  Both operands of a product are data here, read in the shifted form
// (fr29.h: libff's words of x are the 2^261 form of x / 32), so every product loses a factor 32 -- the same number of times
// in every term of every coefficient (each term is suff times one factor per table: M products, M - 1 without suff, whose
// place the 2^261 form of 1 takes) -- and the factor comes back with the constant that makes the lane's sums canonical.
// RED3 (M == 3 only): every product of an index reduced on its own -- nine reductions instead of five and four shared ones,
// ~ 15 % more instructions, but no 64-bit column accumulators (4 x 34 registers, which lived in AGPRs and were moved in
// and out around every multiply-add): the kernel fits three wavefronts per SIMD instead of two, and at two a SIMD runs
// at the pace of the products' dependent multiply-add chains (one every 11 cycles per wavefront, 0.64-0.72 of the issue
// slots by the SQ counters: profiles/r06_w2_sumcheck_sq_pmc.txt).
template <int M, bool RED3>
__device__ __forceinline__ void sumcheck_partial_body(const Fr *__restrict__ suff, const ScTables &tabs, size_t half, Fr *__restrict__ partial, Fr *lds) {
    Fr29 c[M + 1];
#pragma unroll
    for (int i = 0; i <= M; i++) c[i] = Fr29::zero();
    const Fr29 one = Fr29::one();
    unsigned since = 0, wide_n = 0;
    Fr29Wide w0 = fr29_wide_zero(), w2 = fr29_wide_zero(), wm = fr29_wide_zero(), wn = fr29_wide_zero();      // (M == 2, 3 only)
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < half; p += (size_t)gridDim.x * blockDim.x) {
        // (round 6, measured and withdrawn: one dword of every line the NEXT index reads, loaded an iteration ahead -- the kernel holds
        // two wavefronts per SIMD -- made it slower: 0.47 -> 0.56 ms for two tables, 0.85 -> 1.25 ms for three at half = 2^23;
        // profiles/r06_w1_sumcheck_touch_ahead.txt)
        Fr29 q[M + 1];
        q[0] = suff ? Fr29::from_words(suff[p]) : one;
        if constexpr (M == 2) {
            // two tables (the Hadamard prover's shape): five products instead of six -- s (a0 + da X) first, then its
            // product with (b0 + db X) by Karatsuba: the middle coefficient is (sa0 + sda)(b0 + db) - sa0 b0 - sda db --
            // and the three last products only ever enter sums over p, so four indices share ONE reduction each
            // (fr29.h: fr29_wide_*): 2 * 162 + 3 * 81 + 3 * 81 / 4 multiply-adds per index instead of 5 * 162
            const Fr29 a0 = Fr29::from_words(tabs.t[0][p]), da = sub3r_loose(Fr29::from_words(tabs.t[0][p + half]), a0);    // (loose: only ever times the tight q[0])
            const Fr29 b0 = Fr29::from_words(tabs.t[1][p]), db = sub2r(Fr29::from_words(tabs.t[1][p + half]), b0);
            const Fr29 sa0 = mul(q[0], a0), sda = mul(q[0], da);                      // < 2r each
            fr29_wide_mac(w0, sa0, b0);
            fr29_wide_mac(w2, sda, db);
            fr29_wide_mac(wm, add(sa0, sda), add(b0, db));                              // (< 4r)(< 4r), tight limbs
            if (++wide_n == 4) {
                const Fr29 p0 = fr29_wide_reduce(w0), p2 = fr29_wide_reduce(w2), pm = fr29_wide_reduce(wm);
                c[0] = add(c[0], p0);
                c[1] = add(c[1], sub2r(sub2r(pm, p0), p2));                            // pm - p0 - p2 + 4r < 6r
                c[2] = add(c[2], p2);
                w0 = fr29_wide_zero(); w2 = fr29_wide_zero(); wm = fr29_wide_zero();
                wide_n = 0;
                if (++since == 16) {
#pragma unroll
                    for (int i = 0; i <= M; i++) c[i] = mul(c[i], one);
                    since = 0;
                }
            }
            continue;
        } else if constexpr (M == 3) {
            // three tables: five reduced products and four shared reductions instead of twelve reduced products.
            // (s a0 + s da X)(b0 + db X) = e0 + e1 X + e2 X^2 by Karatsuba (e1 = em - e0 - e2, em = (s a1) b1); then
            // g(X) = E(X) (c0 + dc X) has four coefficients, each only ever summed over p: its values at 0, 1, infinity and
            // -1 -- e0 c0, em c1 (E(1) = em, C(1) = c1: the upper table entry itself), e2 dc, (2 e0 + 2 e2 - em)(2 c0 - c1) --
            // go into four wide accumulators (one reduction per four indices each) and are turned into coefficients once,
            // after the block's tree sum.  Without suff the two products by s fall away (s a0 = a0, s da = da).
            const Fr29 a0 = Fr29::from_words(tabs.t[0][p]), a1 = Fr29::from_words(tabs.t[0][p + half]);
            const Fr29 b0 = Fr29::from_words(tabs.t[1][p]), b1 = Fr29::from_words(tabs.t[1][p + half]);
            const Fr29 c0 = Fr29::from_words(tabs.t[2][p]), c1 = Fr29::from_words(tabs.t[2][p + half]);
            // (RED3: sums that only ever meet a TIGHT factor in a product are taken without carries -- fr29.h: add_loose, sub3r_loose)
            const Fr29 db = sub2r(b1, b0);                                                  // < 3r
            const Fr29 da = RED3 ? sub3r_loose(a1, a0) : sub2r(a1, a0), dc = RED3 ? sub3r_loose(c1, c0) : sub2r(c1, c0);   // < 3r (RED3: < 4r, loose)
            Fr29 sa0 = a0, sda = da, sa1 = a1;
            if (suff) { sa0 = mul(q[0], a0); sda = mul(q[0], da); sa1 = RED3 ? add_loose(sa0, sda) : add(sa0, sda); }    // < 2r, < 2r, < 4r
            if constexpr (RED3) {
                const Fr29 e0 = mul(sa0, b0), e2 = mul(sda, db), em = mul(sa1, b1);         // < 2r each (a loose factor meets b0 / db / b1: tight)
                const Fr29 t = add_loose(e0, e2), em1 = sub2r(add(t, t), em);               // E(-1) + 2r < 10r, tight
                const Fr29 cm = sub3r_loose(add_loose(c0, c0), c1);                         // C(-1) + 3r < 5r, limbs < 2^31
                c[0] = add(c[0], mul(e0, c0));                                              // every product < 2r; operands: 2 r^2, 8 r^2, 2 r^2, 50 r^2 < 121 r^2
                c[1] = add(c[1], mul(em, c1));
                c[2] = add(c[2], mul(em1, cm));
                c[3] = add(c[3], mul(e2, dc));
                if (++since == 16) {                                                    // 16 * 2r + 2r < 121 r
#pragma unroll
                    for (int i = 0; i <= M; i++) c[i] = mul(c[i], one);
                    since = 0;
                }
                continue;
            }
            const Fr29 e0 = mul(sa0, b0).canonical2(), e2 = mul(sda, db).canonical2(), em = mul(sa1, b1);   // < r, < r, < 2r
            const Fr29 t = add(e0, e2), em1 = sub2r(add(t, t), em);                         // E(-1) + 2r < 6r
            const Fr29 cm = sub2r(add(c0, c0), c1);                                         // C(-1) + 2r < 4r
            fr29_wide_mac(w0, e0, c0);                                                      // 4 * r^2
            fr29_wide_mac(w2, e2, dc);                                                      // 4 * 3 r^2
            fr29_wide_mac(wm, em, c1);                                                      // 4 * 2 r^2
            fr29_wide_mac(wn, em1, cm);                                                     // 4 * 24 r^2 < 121 r^2
            if (++wide_n == 4) {
                c[0] = add(c[0], fr29_wide_reduce(w0));
                c[1] = add(c[1], fr29_wide_reduce(wm));
                c[2] = add(c[2], fr29_wide_reduce(wn));
                c[3] = add(c[3], fr29_wide_reduce(w2));
                w0 = fr29_wide_zero(); w2 = fr29_wide_zero(); wm = fr29_wide_zero(); wn = fr29_wide_zero();
                wide_n = 0;
                if (++since == 16) {
#pragma unroll
                    for (int i = 0; i <= M; i++) c[i] = mul(c[i], one);
                    since = 0;
                }
            }
            continue;
        } else {
#pragma unroll
        for (int t = 0; t < M; t++) {
            const Fr29 v0 = Fr29::from_words(tabs.t[t][p]);
            const Fr29 dv = sub2r(Fr29::from_words(tabs.t[t][p + half]), v0);       // < 3r
            q[t + 1] = mul(q[t], dv);                                               // q <- q * (v0 + dv X)
#pragma unroll
            for (int i = t; i >= 1; i--) q[i] = add(mul(q[i], v0), mul(q[i - 1], dv));
            q[0] = mul(q[0], v0);
        }
        }
#pragma unroll
        for (int i = 0; i <= M; i++) c[i] = add(c[i], q[i]);
        if (++since == 16) {                                                        // 16 * 6r + 2r < 121 r
#pragma unroll
            for (int i = 0; i <= M; i++) c[i] = mul(c[i], one);
            since = 0;
        }
    }
    if constexpr (M == 2) {
        if (wide_n) {                                                               // the last one to three indices
            const Fr29 p0 = fr29_wide_reduce(w0), p2 = fr29_wide_reduce(w2), pm = fr29_wide_reduce(wm);
            c[0] = add(c[0], p0);
            c[1] = add(c[1], sub2r(sub2r(pm, p0), p2));
            c[2] = add(c[2], p2);
        }
    }
    if constexpr (M == 3) {
        if (wide_n) {
            c[0] = add(c[0], fr29_wide_reduce(w0));
            c[1] = add(c[1], fr29_wide_reduce(wm));
            c[2] = add(c[2], fr29_wide_reduce(wn));
            c[3] = add(c[3], fr29_wide_reduce(w2));
        }
    }
    const Fr29 fix = fr_to_261(Fr::from_u32(1u << (5 * (suff ? M : M - 1))));       // 32^(products per term), in 2^261 form
    Fr tot[M + 1];
#pragma unroll
    for (int i = 0; i <= M; i++) {
        lds[threadIdx.x] = mul(c[i], fix).canonical2().to_words();
        __syncthreads();
        for (unsigned s = 128; s >= 1; s >>= 1) {
            if (threadIdx.x < s) lds[threadIdx.x] = lds[threadIdx.x] + lds[threadIdx.x + s];
            __syncthreads();
        }
        tot[i] = lds[0];
        __syncthreads();
    }
    if constexpr (M == 3) {
        // the block's sums of g(0), g(1), g(-1), g(infinity) (sub2r's offsets are multiples of r: they vanish mod r) ->
        // coefficients: f0 = g(0), f3 = g(inf), f1 + f2 = g(1) - f0 - f3, f2 - f1 = g(-1) - f0 + f3
        const Fr s12 = tot[1] - tot[0] - tot[3], d21 = tot[2] - tot[0] + tot[3];
        const Fr f2 = fr_half(s12 + d21), f1 = fr_half(s12 - d21);
        tot[1] = f1;
        tot[2] = f2;
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i <= M; i++) partial[(size_t)blockIdx.x * (SC_MAX_M + 1) + i] = tot[i];
    }
}
template <int M>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sumcheck_partial(const Fr *__restrict__ suff, ScTables tabs, size_t half, Fr *__restrict__ partial) {
    __shared__ Fr lds[256];
    sumcheck_partial_body<M, false>(suff, tabs, half, partial, lds);
}
// (two tables at three wavefronts per SIMD -- 168 VGPRs + 112 B of scratch -- measured and withdrawn: 0.475 -> 0.51-0.52 ms at half =
// 2^23; profiles/r06_w6_sumcheck_two_tables_three_wavefronts.txt)
#ifndef LSA_SC3_WAVES
#define LSA_SC3_WAVES 3
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LSA_SC3_WAVES, LSA_SC3_WAVES))) void k_sumcheck_partial3r(const Fr *__restrict__ suff, ScTables tabs, size_t half, Fr *__restrict__ partial) {
    __shared__ Fr lds[256];
    sumcheck_partial_body<3, true>(suff, tabs, half, partial, lds);
}

// out[0..m(+1)] = (have_beta ? ((1-rho) + (2rho-1) X) * pre : 1) * sum of the block partials (one workgroup: the partials of a
// coefficient are summed by all its lanes, then a tree -- a lone lane walking 1024 partials took 0.4 ms)
__global__ __launch_bounds__(256) void k_sumcheck_finish(const Fr *__restrict__ partial, unsigned nblocks, unsigned m, int have_beta,
                                                         Fr pre, Fr rho, Fr *__restrict__ out) {
    __shared__ Fr S[SC_MAX_M + 1];
    __shared__ Fr lds[256];
    for (unsigned i = 0; i <= m; i++) {
        Fr acc = Fr::zero();
        for (unsigned b = threadIdx.x; b < nblocks; b += 256) acc = acc + partial[(size_t)b * (SC_MAX_M + 1) + i];
        lds[threadIdx.x] = acc;
        __syncthreads();
        for (unsigned s = 128; s >= 1; s >>= 1) {
            if (threadIdx.x < s) lds[threadIdx.x] = lds[threadIdx.x] + lds[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) S[i] = lds[0];
        __syncthreads();
    }
    const unsigned i = threadIdx.x;
    if (!have_beta) {
        if (i <= m) out[i] = S[i];
        return;
    }
    if (i <= m + 1) {
        const Fr e0 = (Fr::one() - rho) * pre, e1 = (rho + rho - Fr::one()) * pre;
        Fr v = Fr::zero();
        if (i <= m) v = v + S[i] * e0;
        if (i >= 1) v = v + S[i - 1] * e1;
        out[i] = v;
    }
}

// DPBeta::pushRandomness suffix update (/root/reference/src/prototools/mle.h:46-53):
//   cur[p] = old[half + p] * k,  p < half   (cur may alias old)
__global__ __launch_bounds__(256) void k_scale_upper(const Fr *old, size_t half, Fr k, Fr *cur) {
    __shared__ uint32_t s_k[9];
    const Fr29 k261 = block_scalar_261(k, s_k);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t p0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p0 < half; p0 += stride * STREAM_UNROLL) {
        Fr x[STREAM_UNROLL];
#pragma unroll
        for (int u = 0; u < STREAM_UNROLL; u++) {
            const size_t p = p0 + u * stride;
            if (p < half) x[u] = fr_stream_load(old + p + half);
        }
#pragma unroll
        for (int u = 0; u < STREAM_UNROLL; u++) {
            const size_t p = p0 + u * stride;
            if (p < half) fr_stream_store(cur + p, fr_mul_261(x[u], k261));
        }
    }
}

// DPBeta::compute_eq_tbl (/root/reference/src/prototools/mle.h:93-105).  Two tables:
//  * as the reference's loop computes it (variant 0): its doubling step is tmp[p] = eqbit(msb(p), r[j]) * dst[p >> 1] -- the old
//    index keeps the top bit of p, not the low bits -- so every factor is selected by the TOP bit of p:
//        dst[p] = prod_j (1 - r[j]) for p < 2^(d-1),   prod_j r[j] above.
//    That is the table DPBeta::precomputeAll (mle.h:121-137) consumes, hence what a drop-in returns.
//  * the table the comment above the loop describes (variant 1): dst[p] = prod_{j < d} eqbit(bit j of p, r[j]) -- for a
//    maintainer who repairs the index upstream.  The monomial is factored over the low and the high half of the index bits:
//    two small tables (2^lo + 2^(d - lo) entries, <= d / 2 products each), then ONE product per entry of a pure write stream.
// eqbit(true, r) = r, eqbit(false, r) = 1 - r (mle.cc:12-15).  Fr values are canonical: the association of the products
// does not matter.
__global__ __launch_bounds__(256) void k_eq_fill_literal(const Fr *__restrict__ r, unsigned d, size_t n, Fr *__restrict__ out) {
    Fr lo = Fr::one(), hi = Fr::one();
    for (unsigned j = 0; j < d; j++) {
        const Fr rj = r[j];
        lo = lo * (Fr::one() - rj);
        hi = hi * rj;
    }
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) out[p] = p >= n / 2 ? hi : lo;
}
__global__ __launch_bounds__(256) void k_eq_factor_tables(const Fr *__restrict__ r, unsigned d, unsigned lo, Fr *__restrict__ LO, Fr *__restrict__ HI) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nlo = (size_t)1 << lo, nhi = (size_t)1 << (d - lo);
    if (t >= nlo + nhi) return;
    const bool low = t < nlo;
    const size_t idx = low ? t : t - nlo;
    const unsigned first = low ? 0u : lo, bits = low ? lo : d - lo;
    Fr acc = Fr::one();
    for (unsigned i = 0; i < bits; i++) {
        const Fr ri = r[first + i];
        acc = acc * (((idx >> i) & 1) ? ri : Fr::one() - ri);
    }
    (low ? LO : HI)[idx] = acc;
}
__global__ __launch_bounds__(256) void k_eq_expand(const Fr *__restrict__ LO, const Fr *__restrict__ HI, unsigned lo, size_t n, Fr *__restrict__ out) {
    const size_t mask = ((size_t)1 << lo) - 1;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x)
        out[p] = fr_mul_261(LO[p & mask], fr_to_261(HI[p >> lo]));
}

#define HIPCHK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return LSA_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

static unsigned stream_blocks(size_t m) {
    size_t b = (m + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// ---- the launch sequences of the two recursions as hipGraphs.  A witness recursion at d = 24 is a memset and ~8 kernels of
// 5-400 us: issued one by one the host needs ~0.1 ms for them and the short ones wait for it.  The sequence only depends on
// (d, the four pointers), and callers pass the same buffers again and again (the library's staging buffers behind the
// host entry points, a prover's resident vectors), so the instantiated graph of the last few argument sets is kept and
// relaunched with ONE call.  LSA_FR_GRAPHS=0: plain launches.
namespace {
struct FoldGraph {
    int kind = 0;                       // 1: witness recursion, 2: evalMLE
    size_t d = 0;
    const void *a0 = nullptr, *a1 = nullptr, *a2 = nullptr, *a3 = nullptr, *a4 = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t tick = 0;
};
constexpr int FOLD_GRAPHS = 6;
FoldGraph g_fold_graphs[FOLD_GRAPHS];
uint64_t g_fold_tick = 0;
// A failed capture / instantiation / launch (a caller with a capture of its own open on this thread, a runtime that is out
// of graph memory) sends the next FOLD_GRAPH_RETRY_AFTER recursions through plain launches -- ~0.1 ms more host time each at
// d = 24 -- and is then tried again; the first fallback of a run of them is reported under LSA_TRACE.
constexpr unsigned FOLD_GRAPH_RETRY_AFTER = 64;
unsigned g_fold_graph_skip = 0;
bool g_fold_graph_reported = false;
void fold_graphs_give_up(const char *what) {
    (void)hipGetLastError();
    g_fold_graph_skip = FOLD_GRAPH_RETRY_AFTER;
    if (!g_fold_graph_reported && trace_on())
        fprintf(stderr, "[lsa]   fr recursion: %s failed, the next %u recursions are issued as plain launches (no hipGraph)\n", what, FOLD_GRAPH_RETRY_AFTER);
    g_fold_graph_reported = true;
}
bool fold_graphs_on() {
    static const bool on = !(getenv("LSA_FR_GRAPHS") && getenv("LSA_FR_GRAPHS")[0] == '0');
    if (!on) return false;
    if (g_fold_graph_skip) { if (--g_fold_graph_skip == 0) g_fold_graph_reported = false; return false; }
    return true;
}
// issue(st) queues the sequence on st; returns an LSA code.  The first call with a new argument set captures it.
template <class Issue>
int fold_run(int kind, size_t d, const void *a0, const void *a1, const void *a2, const void *a3, const void *a4, hipStream_t st, Issue issue) {
    if (!fold_graphs_on() || d < 8) return issue(st);
    FoldGraph *victim = &g_fold_graphs[0];
    for (auto &g : g_fold_graphs) {
        if (g.exec && g.kind == kind && g.d == d && g.a0 == a0 && g.a1 == a1 && g.a2 == a2 && g.a3 == a3 && g.a4 == a4) {
            g.tick = ++g_fold_tick;
            if (hipGraphLaunch(g.exec, st) == hipSuccess) return LSA_OK;
            fold_graphs_give_up("hipGraphLaunch");
            return issue(st);
        }
        if (g.tick < victim->tick) victim = &g;
    }
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) { fold_graphs_give_up("hipStreamBeginCapture"); return issue(st); }
    const int rc = issue(st);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc != LSA_OK || e != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        fold_graphs_give_up("hipStreamEndCapture");      // (nothing was executed during the capture: run it plainly)
        return rc != LSA_OK ? rc : issue(st);
    }
    hipGraphExec_t exec = nullptr;
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess || !exec) {
        (void)hipGraphDestroy(graph);
        fold_graphs_give_up("hipGraphInstantiate");
        return issue(st);
    }
    (void)hipGraphDestroy(graph);
    if (victim->exec) (void)hipGraphExecDestroy(victim->exec);
    *victim = FoldGraph{kind, d, a0, a1, a2, a3, a4, exec, ++g_fold_tick};
    if (hipGraphLaunch(exec, st) != hipSuccess) { fold_graphs_give_up("hipGraphLaunch"); return issue(st); }
    return LSA_OK;
}
}  // namespace
void fr_vec_release() {
    for (auto &g : g_fold_graphs) { if (g.exec) (void)hipGraphExecDestroy(g.exec); g = FoldGraph(); }
    g_fold_graph_skip = 0;
    g_fold_graph_reported = false;
}

// CPPoly::prove witness coefficients: d_v (2^d, untouched), d_r (d), d_w (2^d; the first
// 2^d - 1 entries are written, the last is zeroed like the reference's value-initialised
// vector), d_tmp: scratch of 2^(d-1) + 2^(d-2) elements (ping-pong).  Asynchronous on st.
// A pass over a vector with `avail` rounds left (avail >= 10): blocks of 2^12 inputs (the whole vector when it is
// shorter), `want` rounds, but never so many that fewer than eight rounds are left for the one-workgroup tail.
struct FusedPass { unsigned bl, tr; size_t blocks; };
static FusedPass fused_pass(size_t avail, unsigned want) {
    FusedPass p;
    p.bl = (unsigned)(avail < FUSE_BLOCK_LOG ? avail : FUSE_BLOCK_LOG);
    const size_t room = avail - 8;
    p.tr = (unsigned)(room < want ? room : want);
    if (p.tr > p.bl) p.tr = p.bl;
    p.blocks = (size_t)1 << (avail - p.bl);
    return p;
}
static int fr_cppoly_fold_issue(const Fr *d_v, size_t d, const Fr *d_r, Fr *d_w, Fr *d_tmp, hipStream_t st);
int fr_cppoly_fold_device(const Fr *d_v, size_t d, const Fr *d_r, Fr *d_w, Fr *d_tmp, hipStream_t st) {
    return fold_run(1, d, d_v, d_r, d_w, d_tmp, nullptr, st, [=](hipStream_t s) { return fr_cppoly_fold_issue(d_v, d, d_r, d_w, d_tmp, s); });
}
static int fr_cppoly_fold_issue(const Fr *d_v, size_t d, const Fr *d_r, Fr *d_w, Fr *d_tmp, hipStream_t st) {
    const size_t N = (size_t)1 << d;
    HIPCHK(hipMemsetAsync(d_w + (N - 1), 0, sizeof(Fr), st));
    if (d == 0) return LSA_OK;
    const Fr *src = d_v;
    Fr *bufA = d_tmp, *bufB = d_tmp + (N >> 1);
    size_t start = 0, i = 0;
    unsigned which = 0;                                  // the next launch writes bufA (0) or bufB (1); it never reads the one it writes
    while (d - i >= 10) {                                // passes of up to twelve rounds each (k_fold_pairs_fused)
        const FusedPass p = fused_pass(d - i, FUSE_BLOCK_LOG);
        const size_t m = (size_t)1 << (d - i - 1);
        Fr *dst = which ? bufB : bufA;                   // (a pass's output is at most a quarter of its input: both parts of d_tmp are large enough)
        hipLaunchKernelGGL(k_fold_pairs_fused, dim3((unsigned)(p.blocks < 65536 ? p.blocks : 65536)), dim3(256), 0, st, src, p.blocks, p.bl, p.tr,
                           d_r + i, d_w + start, m, dst);
        src = dst;
        which ^= 1u;
        for (unsigned j = 0; j < p.tr; j++) start += m >> j;
        i += p.tr;
    }
    // the remaining rounds (at most nine, at most 256 outputs in the first): one workgroup
    hipLaunchKernelGGL(k_fold_pairs_tail, dim3(1), dim3(256), 0, st, src, (size_t)1 << (d - i - 1), d_r + i, d_w + start, bufA, bufB, which, (Fr *)nullptr);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}

// One pushRandomness round: cur[p] = old[p]*(1-r) + old[p+half]*r, p < half (cur may alias old).
int fr_fold_halves_device(const Fr *d_old, size_t half, const Fr *d_r, Fr *d_cur, hipStream_t st) {
    if (half == 0) return LSA_OK;
    hipLaunchKernelGGL(k_fold_halves, dim3(stream_blocks(half)), dim3(256), 0, st, d_old, half, d_r, d_cur);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}

// evalMLE(v, r): d_v (2^d, untouched), d_r (d), d_tmp scratch of 2^(d-1) + 2^(d-2) elements; the value
// ends up in d_out (one Fr).  Bit i of the index pairs with r[i] (polytools.h:219-226), so the
// top variable is r[d-1].
static int fr_eval_mle_issue(const Fr *d_v, size_t d, const Fr *d_r, Fr *d_tmp, Fr *d_out, hipStream_t st);
int fr_eval_mle_device(const Fr *d_v, size_t d, const Fr *d_r, Fr *d_tmp, Fr *d_out, hipStream_t st) {
    return fold_run(2, d, d_v, d_r, d_tmp, d_out, nullptr, st, [=](hipStream_t s) { return fr_eval_mle_issue(d_v, d, d_r, d_tmp, d_out, s); });
}
static int fr_eval_mle_issue(const Fr *d_v, size_t d, const Fr *d_r, Fr *d_tmp, Fr *d_out, hipStream_t st) {
    if (d == 0) {
        HIPCHK(hipMemcpyAsync(d_out, d_v, sizeof(Fr), hipMemcpyDeviceToDevice, st));
        return LSA_OK;
    }
    static const bool dot_on = !(getenv("LSA_MLE_DOT") && getenv("LSA_MLE_DOT")[0] == '0');      // 0: the fold passes at every size (A/B)
    if (dot_on && d >= MLE_MIN_D && d <= 40) {
        // tables and partials live in the scratch: 64 + 512 + 2^(d-15) + (number of workgroups) elements
        const size_t nh = (size_t)1 << (d - MLE_LO - MLE_MID);
        Fr *E6 = d_tmp, *WL = E6 + 64, *WH = WL + 512, *partial = WH + nh;
        const size_t ntab = 64 + 512 + nh;
        hipLaunchKernelGGL(k_mle_eq_tables, dim3((unsigned)((ntab + 255) / 256)), dim3(256), 0, st, d_r, (unsigned)d, E6, WL, WH);
        // units of 2^ju_log values of j: as long as 128 where that still leaves 8192 of them, never shorter than 16
        unsigned ju_log = 7;
        while (ju_log > 4 && ((nh << MLE_MID) >> ju_log) < 8192) ju_log--;
        const size_t units = (nh << MLE_MID) >> ju_log;
        const unsigned blocks = (unsigned)(units / 4 < 2048 ? units / 4 : 2048);       // four wavefronts per workgroup, a unit each at least
        hipLaunchKernelGGL(k_mle_dot, dim3(blocks), dim3(256), 0, st, d_v, (unsigned)d, ju_log, (const Fr *)E6, (const Fr *)WL, (const Fr *)WH, partial);
        hipLaunchKernelGGL(k_mle_finish, dim3(1), dim3(256), 0, st, (const Fr *)partial, blocks, d_out);
        HIPCHK(hipGetLastError());
        return LSA_OK;
    }
    // The variables from the BOTTOM: round j pairs neighbours with r[j] -- the same multilinear value as folding the top
    // variable first, and a canonical residue either way -- so the passes are the witness recursion's without its
    // coefficients: the table is read ONCE, then the one-workgroup tail.
    const Fr *in = d_v;
    Fr *pp[2] = {d_tmp, d_tmp + ((size_t)1 << (d - 1))};
    unsigned which = 0;
    size_t j = 0;
    while (d - j >= 10) {
        const FusedPass p = fused_pass(d - j, FUSE_BLOCK_LOG);
        hipLaunchKernelGGL(k_eval_mle_fused, dim3((unsigned)(p.blocks < 65536 ? p.blocks : 65536)), dim3(256), 0, st, in, p.blocks, p.bl, p.tr, d_r + j, pp[which]);
        in = pp[which];
        which ^= 1u;
        j += p.tr;
    }
    hipLaunchKernelGGL(k_fold_pairs_tail, dim3(1), dim3(256), 0, st, in, (size_t)1 << (d - j - 1), d_r + j, (Fr *)nullptr, pp[0], pp[1], which, d_out);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}

// d_out: m+1 (no beta factor) or m+2 coefficients (device).  d_partial: scratch of
// 1024*(SC_MAX_M+1) elements.  Asynchronous on st.
int fr_sumcheck_round_device(const Fr *d_suff, const Fr *const *d_tables, size_t m, size_t half, const Fr *pre, const Fr *rho,
                             Fr *d_partial, Fr *d_out, hipStream_t st) {
    if (m == 0 || m > SC_MAX_M) { set_error("sumcheck_round: m = %zu not in 1..%d", m, SC_MAX_M); return LSA_ERR_INVALID; }
    ScTables tabs;
    for (int t = 0; t < SC_MAX_M; t++) tabs.t[t] = t < (int)m ? d_tables[t] : nullptr;
    size_t b = (half + 255) / 256;
    // three tables: LSA_SC3=wide keeps the shared-reduction kernel (two wavefronts per SIMD: 1024 workgroups are two full rounds); the
    // default reduces every product and holds three per SIMD (768 workgroups are one round)
    static const bool sc3_reduced = [] { const char *e = getenv("LSA_SC3"); return !(e && *e == 'w'); }();
    const size_t cap = (m == 3 && sc3_reduced) ? 256 * LSA_SC3_WAVES : 1024;
    const unsigned blocks = (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
    if (m == 1) hipLaunchKernelGGL((k_sumcheck_partial<1>), dim3(blocks), dim3(256), 0, st, d_suff, tabs, half, d_partial);
    else if (m == 2) hipLaunchKernelGGL((k_sumcheck_partial<2>), dim3(blocks), dim3(256), 0, st, d_suff, tabs, half, d_partial);
    else if (m == 3 && sc3_reduced) hipLaunchKernelGGL(k_sumcheck_partial3r, dim3(blocks), dim3(256), 0, st, d_suff, tabs, half, d_partial);
    else if (m == 3) hipLaunchKernelGGL((k_sumcheck_partial<3>), dim3(blocks), dim3(256), 0, st, d_suff, tabs, half, d_partial);
    else hipLaunchKernelGGL((k_sumcheck_partial<4>), dim3(blocks), dim3(256), 0, st, d_suff, tabs, half, d_partial);
    hipLaunchKernelGGL(k_sumcheck_finish, dim3(1), dim3(256), 0, st, d_partial, blocks, (unsigned)m, rho ? 1 : 0,
                       pre ? *pre : Fr::one(), rho ? *rho : Fr::zero(), d_out);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}
size_t fr_sumcheck_scratch_elems() { return (size_t)1024 * (SC_MAX_M + 1); }

// d_tmp: 2^(d / 2) + 2^(d - d / 2) elements
int fr_eq_table_device(const Fr *d_r, size_t d, int variant, Fr *d_tmp, Fr *d_out, hipStream_t st) {
    const unsigned lo = (unsigned)(d / 2);
    const size_t nlo = (size_t)1 << lo, nhi = (size_t)1 << (d - lo), n = (size_t)1 << d;
    if (variant == 0) {
        hipLaunchKernelGGL(k_eq_fill_literal, dim3(stream_blocks(n)), dim3(256), 0, st, d_r, (unsigned)d, n, d_out);
        HIPCHK(hipGetLastError());
        return LSA_OK;
    }
    hipLaunchKernelGGL(k_eq_factor_tables, dim3((unsigned)((nlo + nhi + 255) / 256)), dim3(256), 0, st, d_r, (unsigned)d, lo, d_tmp, d_tmp + nlo);
    hipLaunchKernelGGL(k_eq_expand, dim3(stream_blocks(n)), dim3(256), 0, st, (const Fr *)d_tmp, (const Fr *)(d_tmp + nlo), lo, n, d_out);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}
size_t fr_eq_table_scratch_elems(size_t d) { return ((size_t)1 << (d / 2)) + ((size_t)1 << (d - d / 2)); }

int fr_scale_upper_device(const Fr *d_old, size_t half, const Fr &k, Fr *d_cur, hipStream_t st) {
    if (half == 0) return LSA_OK;
    hipLaunchKernelGGL(k_scale_upper, dim3(stream_blocks(half)), dim3(256), 0, st, d_old, half, k, d_cur);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}

}  // namespace lsa
