// legosnark_amd/csrc/msm.hip -- variable-base multi-scalar multiplication on gfx950.
//
// Replaces libff::multi_exp_with_mixed_addition / multi_exp<BDLO12> as called by
// multiExpMA (/root/reference/src/utils/globl.h:63-78) and sparsemexp
// (/root/reference/src/utils/sparsemexp.h:58,89).  Same sum, different schedule:
//
//   1 digits     scalars (Montgomery Fr, 32 B, coalesced) -> canonical -> signed c-bit digits,
//                recomputed by both sort passes instead of stored.  Resident bases with
//                pre-shifted copies (the usual case, "wide" path below): per tile of 2048 scalars
//                a histogram of 768 population-balanced bucket segments in LDS; plain path: per
//                (tile of 32768 scalars, window) a bucket histogram in LDS that ranks every entry.
//                A prefix over the tiles gives populations and run starts.  No global atomics.
//   2/3 sort     wide path: k_partition stages a tile's records per segment in LDS and copies
//                them out as whole runs; k_fine_sort_part (one workgroup per segment) orders a
//                segment in LDS from registers and writes it out linearly, with the per-bucket
//                populations and offsets.  Plain path: exclusive scan over the bucket populations,
//                then entry (point index | sign) written to offs[bucket] + tile_base + rank.
//   3b order     buckets ordered by population (largest first) so a wavefront's 64 lanes
//                walk equally long lists.
//   4 accumulate one lane per bucket walks its entry list, gathers 64-B affine points
//                and accumulates in XYZZ (8M+2S mixed add, 29-bit limbs for G1) -- the
//                dominant kernel.  Buckets above a population threshold are split across
//                workgroups (skewed scalars, e.g. the u[i]=i inputs of
//                /root/reference/src/examples/hadamard.cc:130-135).
//   5 reduce     sum_b (b+1)*S_b per window: per-lane running sums over L buckets,
//                then wavefront suffix-scan + tree reductions with cross-lane shuffles.
//   6 fold       Horner over the windows (c doublings + 1 add each) -> one Jacobian point;
//                for G1 each point is shared by a quad of lanes (quad29.h).
//
// Signed digits halve the bucket count (2^(c-1)); the zero/one filter of libff's
// multi_exp_with_mixed_addition needs no special case (0 contributes nothing, 1 lands in
// bucket 1 of window 0) and gives the same group element.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <type_traits>
#include <vector>

#include "curves.h"
#include "fp29x2l.h"
#include "fs29.h"
#include "glv.h"
#include "msm.h"
#include "msm_plan.h"

namespace lsa {

// ------------------------------------------------------------------------------------
// window choice
// ------------------------------------------------------------------------------------
unsigned msm_window_bits(size_t n) {
    unsigned lg = 0;
    while ((size_t(1) << (lg + 1)) <= n) lg++;   // floor(log2 n), 0 for n <= 1
    int c = (int)lg - 4;
    if (c < 8) c = 8;        // few, wide windows keep the latency-bound fold short for tiny inputs
    if (c > 16) c = 16;
    return (unsigned)c;
}

static inline unsigned num_windows(unsigned c) { return (255 + c - 1) / c; }

// ------------------------------------------------------------------------------------
// kernel 0: Jacobian (libff layout) -> affine, per-lane Montgomery batch inversion
// ------------------------------------------------------------------------------------
template <class F, int K>
__global__ __launch_bounds__(256) void k_normalize(const Jac<F> *__restrict__ in, Aff<F> *__restrict__ out, size_t n) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t base = t * K;
    if (base >= n) return;
    F prod[K];
    F acc = F::one();
    const F one = F::one();
    bool need_inv = false;
#pragma unroll
    for (int i = 0; i < K; i++) {
        if (base + i < n) {
            F z = in[base + i].Z;
            if (!z.is_zero() && z != one) { acc = acc * z; need_inv = true; }
        }
        prod[i] = acc;
    }
    F inv = need_inv ? acc.inverse() : one;
#pragma unroll
    for (int i = K - 1; i >= 0; i--) {
        if (base + i < n) {
            Jac<F> p = in[base + i];
            Aff<F> a;
            if (p.Z.is_zero()) {
                a = Aff<F>::inf();
            } else if (p.Z == one) {
                a.x = p.X; a.y = p.Y;
            } else {
                F before = (i == 0) ? one : prod[i - 1];
                F zi = inv * before;          // 1 / Z_i
                inv = inv * p.Z;
                F zi2 = zi.sqr();
                a.x = p.X * zi2;
                a.y = p.Y * (zi2 * zi);
            }
            out[base + i] = a;
        }
    }
}

// affine (libff Montgomery limbs) -> the curve's device-resident base format
template <class C>
__global__ __launch_bounds__(256) void k_convert_bases(const Aff<typename C::Field> *__restrict__ in, typename C::Base *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = C::from_affine(in[i]);
}

// G1: Jacobian (libff layout) -> packed affine bases in ONE kernel, on the 29-bit-limb field: a third
// of the instructions per product of the 32-bit-limb Fq, no 64-byte staging array, and the same
// canonical coordinates as k_normalize + k_convert_bases (affine coordinates are unique).  Per lane K
// points share one Fermat inversion (Montgomery's trick).
template <int K>
__global__ __launch_bounds__(256) void k_prepare_g1(const Jac<Fq> *__restrict__ in, AffPacked *__restrict__ out, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, base = t * K;
    if (base >= n) return;
    const Fq one256 = Fq::one();
    F29 prod[K];
    F29 acc = F29::one();
    bool need_inv = false;
#pragma unroll
    for (int i = 0; i < K; i++) {
        if (base + i < n) {
            const Fq z = in[base + i].Z;
            if (!z.is_zero() && z != one256) { acc = mul(acc, F29::from_mont256(z)); need_inv = true; }
        }
        prod[i] = acc;
    }
    F29 inv = need_inv ? Fs{acc}.inverse().v : F29::one();
#pragma unroll
    for (int i = K - 1; i >= 0; i--) {
        if (base + i >= n) continue;
        const Jac<Fq> p = in[base + i];
        AffPacked r;
        if (p.Z.is_zero()) {
#pragma unroll
            for (int w = 0; w < 8; w++) { r.x[w] = 0; r.y[w] = 0; }
        } else {
            F29 x = F29::from_mont256(p.X), y = F29::from_mont256(p.Y);
            if (p.Z != one256) {
                const F29 zi = mul(inv, i == 0 ? F29::one() : prod[i - 1]);      // 1 / Z_i
                inv = mul(inv, F29::from_mont256(p.Z));
                const F29 zi2 = sqr(zi);
                x = mul(x, zi2);
                y = mul(y, mul(zi2, zi));
            }
            x.canonical().pack256(r.x);
            y.canonical().pack256(r.y);
        }
        out[base + i] = r;
    }
}

// G2: the same on Fq2 over the 29-bit limbs (fp29x2.h), K points per inversion (through the norm: one Fq inversion).
// k_normalize<Fq2, 8> + k_convert_bases, which this replaces, kept 656 bytes of scratch per lane: 344 MB for a full
// grid, far above what the runtime keeps allocated between dispatches, so EVERY launch paid a scratch allocation --
// 1-10 ms per launch depending on the box, twelve launches per streamed first-sight G2 vector.
template <int K>
__global__ __launch_bounds__(256) void k_prepare_g2(const Jac<Fq2> *__restrict__ in, AffPackedG2 *__restrict__ out, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, base = t * K;
    if (base >= n) return;
    const Fq2 one256 = Fq2::one();
    F29x2 prod[K];
    F29x2 acc = F29x2::one();
    bool need_inv = false;
#pragma unroll
    for (int i = 0; i < K; i++) {
        if (base + i < n) {
            const Fq2 z = in[base + i].Z;
            if (!z.is_zero() && z != one256) { acc = mul<2>(acc, f29x2_from_mont256(z)); need_inv = true; }   // [< 2]
        }
        prod[i] = acc;
    }
    F29x2 inv = need_inv ? f29x2_inverse(acc) : F29x2::one();
#pragma unroll
    for (int i = K - 1; i >= 0; i--) {
        if (base + i >= n) continue;
        const Jac<Fq2> p = in[base + i];
        AffPackedG2 r;
        if (p.Z.is_zero()) {
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int w = 0; w < 8; w++) r.w[c][w] = 0;
        } else {
            F29x2 x = f29x2_from_mont256(p.X), y = f29x2_from_mont256(p.Y);
            if (p.Z != one256) {
                const F29x2 zi = mul<2>(inv, i == 0 ? F29x2::one() : prod[i - 1]);      // 1 / Z_i
                inv = mul<2>(inv, f29x2_from_mont256(p.Z));
                const F29x2 zi2 = sqr<2>(zi);
                x = mul<2>(x, zi2);
                y = mul<2>(y, mul<2>(zi2, zi));
            }
            const F29x2 xc = x.canonical(), yc = y.canonical();
            xc.c0.pack256(r.w[0]);
            xc.c1.pack256(r.w[1]);
            yc.c0.pack256(r.w[2]);
            yc.c1.pack256(r.w[3]);
        }
        out[base + i] = r;
    }
}

// ------------------------------------------------------------------------------------
// kernels 1a-1c: digits, LDS ranking, tile prefix.  No global atomics.
//   1a k_digits     scalars (Montgomery Fr, read once, coalesced) -> canonical -> signed
//                   c-bit digits, digits[k][i] (int16).
//   1b k_rank       one workgroup per (tile of TILE scalars, window): a bucket histogram of
//                   the tile lives in LDS (u16 counters packed in pairs, <= 64 KiB); each
//                   non-zero digit takes its rank inside (tile, bucket) with ONE returning
//                   LDS atomic.  rank[k][i] (u16) and the tile histogram go to HBM.
//   1c k_tile_scan  per (window, bucket): exclusive prefix over the tiles (u32) and the
//                   bucket population hist[k][b].
// ------------------------------------------------------------------------------------
#define SORT_TILE 32768u
__device__ __forceinline__ void write_digits(const uint32_t *s, int nlimbs, bool negate, size_t col, size_t nv, unsigned c,
                                             unsigned nwin, int32_t *__restrict__ digits) {
    const uint32_t B = 1u << (c - 1);
    uint32_t carry = 0;
    for (unsigned k = 0; k < nwin; k++) {
        unsigned bit = k * c;
        int w = (int)(bit >> 5);
        unsigned sh = bit & 31;
        uint64_t two = (uint64_t)(w < nlimbs ? s[w] : 0) | ((uint64_t)(w + 1 < nlimbs ? s[w + 1] : 0) << 32);
        uint32_t d = (uint32_t)(two >> sh) & ((1u << c) - 1);
        d += carry;
        int32_t sd;
        if (k + 1 < nwin && d >= B) { sd = (int32_t)d - (int32_t)(1u << c); carry = 1; }
        else { sd = (int32_t)d; carry = 0; }
        digits[(size_t)k * nv + col] = negate ? -sd : sd;
    }
}
// GLV: scalar i yields two virtual scalars, columns i (k1, point i) and n+i (k2, phi(point i)).
template <bool GLV>
__global__ __launch_bounds__(256) void k_digits(const Fr *__restrict__ scalars, size_t n, unsigned c, unsigned nwin,
                                                int32_t *__restrict__ digits) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s[8];
    scalars[i].to_canonical(s);
    if (GLV) {
        GlvSplit g = glv_decompose(s);
        write_digits(g.k1, 4, g.neg1, i, 2 * n, c, nwin, digits);
        write_digits(g.k2, 4, g.neg2, n + i, 2 * n, c, nwin, digits);
    } else {
        write_digits(s, 8, false, i, n, c, nwin, digits);
    }
}

// `shift` != 0 (wide path): the histogram runs over the coarse bins b >> shift (B = number of bins).
__global__ __launch_bounds__(1024) void k_rank(const int32_t *__restrict__ digits, size_t n, uint32_t B, uint32_t ntiles,
                                               uint16_t *__restrict__ rank, uint16_t *__restrict__ tile_hist, uint32_t shift) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cnt2[];   // B/2 words: two u16 counters each
    const uint32_t t = blockIdx.x, k = blockIdx.y;
    for (uint32_t x = threadIdx.x; x < B / 2; x += 1024) cnt2[x] = 0;
    __syncthreads();
    const size_t lo = (size_t)t * SORT_TILE;
    const size_t hi = lo + SORT_TILE < n ? lo + SORT_TILE : n;
    const int32_t *dg = digits + (size_t)k * n;
    uint16_t *rk = rank + (size_t)k * n;
    for (size_t i = lo + threadIdx.x; i < hi; i += 1024) {
        int32_t sd = dg[i];
        if (sd != 0) {
            uint32_t b = ((uint32_t)(sd < 0 ? -sd : sd) - 1) >> shift;
            uint32_t sh = (b & 1) * 16;
            uint32_t old = atomicAdd(&cnt2[b >> 1], 1u << sh);      // ds_add_rtn_u32
            rk[i] = (uint16_t)(old >> sh);
        }
    }
    __syncthreads();
    uint32_t *th = reinterpret_cast<uint32_t *>(tile_hist + ((size_t)k * ntiles + t) * B);
    for (uint32_t x = threadIdx.x; x < B / 2; x += 1024) th[x] = cnt2[x];
}

// (wide path: called with ntiles = nwin * ntiles and nb = B, i.e. every (window, tile) pair is a
// row of ONE bin space -- the pre-shifted copies make the windows interchangeable)
__global__ __launch_bounds__(256) void k_tile_scan(const uint16_t *__restrict__ tile_hist, uint32_t B, uint32_t ntiles, uint32_t nb,
                                                   uint32_t *__restrict__ tile_base, uint32_t *__restrict__ hist) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;      // g = k*B + b
    if (g >= nb) return;
    uint32_t k = g / B, b = g - k * B;
    uint32_t run = 0;
    for (uint32_t t = 0; t < ntiles; t++) {
        size_t idx = ((size_t)k * ntiles + t) * B + b;
        tile_base[idx] = run;
        run += tile_hist[idx];
    }
    hist[g] = run;
}

// ------------------------------------------------------------------------------------
// kernel 2: exclusive scan of `count` counters in three small launches:
//   a) per-block sums (2048 counters per 256-lane block, LDS-free wave shuffles)
//   b) one block scans the <= 1024 block sums
//   c) per-block exclusive scan seeded with the block offset
// ------------------------------------------------------------------------------------
#define SCAN_PER_BLOCK 2048
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t *lds, uint32_t *total) {
    // 256 lanes; wave-level shuffles then 4 wave totals through LDS
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += t;
    }
    if (lane == 63) lds[wv] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (unsigned w = 0; w < 4; w++) { uint32_t x = lds[w]; if (w < wv) base += x; tot += x; }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(256) void k_scan_sums(const uint32_t *__restrict__ hist, uint32_t count, uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t lds[4];
    const uint32_t base = blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * 8;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) if (base + j < count) s += hist[base + j];
    uint32_t tot;
    (void)block_exclusive_scan_256(s, lds, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_scan_blocks(uint32_t *__restrict__ block_sums, uint32_t nblocks) {
    __shared__ uint32_t lds[4];
    // nblocks <= 1024: 4 per lane
    uint32_t v[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { uint32_t idx = threadIdx.x * 4 + j; v[j] = idx < nblocks ? block_sums[idx] : 0; s += v[j]; }
    uint32_t tot;
    uint32_t run = block_exclusive_scan_256(s, lds, &tot);
#pragma unroll
    for (int j = 0; j < 4; j++) { uint32_t idx = threadIdx.x * 4 + j; if (idx < nblocks) block_sums[idx] = run; run += v[j]; }
}

__global__ __launch_bounds__(256) void k_scan_final(const uint32_t *__restrict__ hist, const uint32_t *__restrict__ block_sums,
                                                    uint32_t count, uint32_t *__restrict__ offs) {
    __shared__ uint32_t lds[4];
    const uint32_t base = blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * 8;
    uint32_t v[8], s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { v[j] = (base + j < count) ? hist[base + j] : 0; s += v[j]; }
    uint32_t tot;
    uint32_t run = block_sums[blockIdx.x] + block_exclusive_scan_256(s, lds, &tot);
#pragma unroll
    for (int j = 0; j < 8; j++) { if (base + j < count) offs[base + j] = run; run += v[j]; }
}

// ------------------------------------------------------------------------------------
// kernel 3: scatter.  One workgroup per (tile, window) loads base[b] = offs[k][b] +
// tile_base[k][t][b] into LDS (<= 128 KiB) and writes entry (point index | sign<<31) at
// base[b] + rank.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_scatter(const int32_t *__restrict__ digits, const uint16_t *__restrict__ rank,
                                                  const uint32_t *__restrict__ offs, const uint32_t *__restrict__ tile_base,
                                                  size_t n, size_t npoints, uint32_t B, uint32_t ntiles, uint32_t *__restrict__ entries,
                                                  uint32_t win_stride) {
    extern __shared__ __attribute__((aligned(16))) uint32_t base[];   // B words
    const uint32_t t = blockIdx.x, k = blockIdx.y;
    const uint32_t *of = offs + (size_t)k * B;
    const uint32_t *tb = tile_base + ((size_t)k * ntiles + t) * B;
    for (uint32_t x = threadIdx.x; x < B; x += 1024) base[x] = of[x] + tb[x];
    __syncthreads();
    const size_t lo = (size_t)t * SORT_TILE;
    const size_t hi = lo + SORT_TILE < n ? lo + SORT_TILE : n;
    const int32_t *dg = digits + (size_t)k * n;
    const uint16_t *rk = rank + (size_t)k * n;
    for (size_t i = lo + threadIdx.x; i < hi; i += 1024) {
        int32_t sd = dg[i];
        if (sd != 0) {
            uint32_t b = (uint32_t)(sd < 0 ? -sd : sd) - 1;
            // entry = point index | endo << 30 | sign << 31   (virtual scalar i >= npoints: phi(point))
            // win_stride != 0: window k reads its own pre-shifted copy of the bases (2^(c*k) * P)
            uint32_t pt = i < npoints ? (uint32_t)i : ((uint32_t)(i - npoints) | 0x40000000u);
            pt += k * win_stride;
            entries[base[b] + rk[i]] = pt | (sd < 0 ? 0x80000000u : 0u);
        }
    }
}

// Wide path: R = nwin * ntiles rows share ONE bin space of Bc bins; R is a few hundred, so a
// workgroup of 16 wavefronts takes 64 bins (one per lane, coalesced row reads) and splits the rows
// 16 ways: partial sums per wavefront, exclusive prefix across the wavefronts through LDS, then a
// second pass writes every row's base.
// Rows are `pitch` elements apart, pitch = Bc + 96, so that the rows of one bin are not a power of
// two apart.  (The kernel takes ~80 us for 2 MB whatever its shape -- one thread per bin over all
// rows, 64 bins x 16 row groups, this version, padded or not: it inherits the write-back of the
// 27 MB of ranks the preceding kernel left dirty in the L2s.)
__global__ __launch_bounds__(1024) void k_tile_scan_rows(const uint16_t *__restrict__ tile_hist, uint32_t Bc, uint32_t R, uint32_t pitch,
                                                         uint32_t *__restrict__ tile_base, uint32_t *__restrict__ hist,
                                                         uint32_t *__restrict__ zero1, uint32_t *__restrict__ zeron, uint32_t nzero) {
    __builtin_amdgcn_s_setprio(3);      // the sort runs beside the previous call's tail: do not starve behind its older wavefronts
    if (blockIdx.x == 0) {               // counters of the ordering stage (heavy_count; bin_count | bin_start | bin_cursor)
        if (threadIdx.x == 0) *zero1 = 0;
        for (uint32_t x = threadIdx.x; x < nzero; x += 1024) zeron[x] = 0;
    }
    // 32 bins x 32 row groups per workgroup: lane -> (bin = lane & 31, group = 2 * wavefront + (lane >> 5))
    __shared__ uint32_t part[32][33];
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned bl = lane & 31, grp = 2 * wv + (lane >> 5);
    const uint32_t b = blockIdx.x * 32 + bl;
    const uint32_t per = (R + 31) / 32, r0 = grp * per, r1 = r0 + per < R ? r0 + per : R;
    constexpr uint32_t MAXPER = 16;                  // rows held in registers (R <= 512: n <= 2^20 at 13 windows)
    uint32_t v[MAXPER];
    uint32_t sum = 0;
    const bool in_regs = per <= MAXPER;
    if (in_regs) {
#pragma unroll
        for (uint32_t j = 0; j < MAXPER; j++) {
            const uint32_t r = r0 + j;
            v[j] = (b < Bc && r < r1) ? tile_hist[(size_t)r * pitch + b] : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < MAXPER; j++) sum += v[j];
    } else if (b < Bc) {
        for (uint32_t r = r0; r < r1; r++) sum += tile_hist[(size_t)r * pitch + b];
    }
    part[grp][bl] = sum;
    __syncthreads();
    uint32_t run = 0, tot = 0;
#pragma unroll
    for (unsigned g2 = 0; g2 < 32; g2++) { uint32_t x = part[g2][bl]; if (g2 < grp) run += x; tot += x; }
    if (b < Bc) {
        if (in_regs) {
#pragma unroll
            for (uint32_t j = 0; j < MAXPER; j++) {
                const uint32_t r = r0 + j;
                if (r < r1) { tile_base[(size_t)r * pitch + b] = run; run += v[j]; }
            }
        } else {
            for (uint32_t r = r0; r < r1; r++) {
                const size_t idx = (size_t)r * pitch + b;
                tile_base[idx] = run;
                run += tile_hist[idx];
            }
        }
        if (grp == 0) hist[b] = tot;
    }
}

// ------------------------------------------------------------------------------------
// Wide-window path (resident bases with pre-shifted copies, see "Wide windows" below).
// Every entry is (bin, point reference): bin = segment * B + bucket.  With B = 2^(c-1) up to 2^21
// the bin space can be too large for an LDS histogram; then the sort runs in two passes:
// (A) k_rank / k_tile_scan_rows / k_scatter_wide<true> partition all entries by the COARSE bin
// (bin >> 7; a tile's histogram fits LDS) into 64-bit records (fine bits | entry); (B)
// k_fine_sort, one workgroup per coarse bin, counts its 128 fine bins in LDS and places the
// 32-bit entries, writing the per-bin populations and offsets.  With at most 32768 bins (the
// narrow digits used for small inputs and for segmented calls) pass A alone sorts completely.
// ------------------------------------------------------------------------------------
#define WIDE_FINE_BITS 7u
#define MSM_MAX_SEGMENTS 64u
// scalar slices of a segmented call: segment j = scalars[off[j] .. off[j+1]) against bases[0 .. len_j)
struct SegList {
    uint32_t nseg;
    uint32_t off[MSM_MAX_SEGMENTS + 1];
};

// Scalars per workgroup of the wide path's ranking / scatter passes: a tile's entries (nwin per
// scalar) must fit the u16 counters even when every one of them lands in the same bin
// (all scalars equal, all their digits equal): 13 x 4096 or 26 x 2048 = 53248 < 65536.
// Small inputs get 256- or 1024-scalar tiles: a lone workgroup ranking
// 26 digits of 1024 scalars keeps ONE CU's LDS busy for 13 + 17 us (hist + scatter); with more
// tiles the LDS atomics of a call spread over several CUs.
static inline uint32_t wide_tile(uint32_t nwin, size_t n) {
    if (n <= 4096) return 256u;
    if (n <= 16384) return 1024u;
    return nwin > 15 ? 2048u : 4096u;
}
__device__ __forceinline__ uint32_t segment_of(const SegList &segs, uint32_t i) {
    uint32_t seg = 0;
    for (uint32_t j = 1; j < segs.nseg; j++) if (i >= segs.off[j]) seg = j;
    return seg;
}

// Workgroup -> tile of the two wide-path passes.  Consecutive workgroup ids land on different
// XCDs (round-robin over the 8 dies, each with its own L2); a coarse bin's region of the record
// array is the concatenation of the tiles' runs in tile order, so giving every XCD a CONTIGUOUS
// range of tiles makes the short runs (a few records per tile and bin) of neighbouring tiles meet
// in the same L2 and leave it as whole lines instead of partial-line writes from eight dies.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t bid, uint32_t ntiles) {
    const uint32_t q = ntiles >> 3, r = ntiles & 7u, x = bid & 7u, j = bid >> 3;
    return x * q + (x < r ? x : r) + j;
}

// Bucket -> coarse bin of the first sort pass.  Windows of the widest width c reach all 2^(c-1)
// buckets, windows one bit narrower (5 of the 13 at n = 2^20) only the lower half, so a bucket of
// the lower half holds more than twice as many entries as one of the upper half (36 against 16):
// bins of equal POPULATION take 2^sh_lo buckets below `half` and 2^sh_hi above.  half = 0 makes it a
// plain shift by sh_hi.
struct CoarseMap {
    uint32_t half, sh_lo, sh_hi, nlo;      // nlo = half >> sh_lo bins below `half`
    __device__ __forceinline__ uint32_t bin(uint32_t b) const { return b < half ? b >> sh_lo : nlo + ((b - half) >> sh_hi); }
    __device__ __forceinline__ uint32_t fine(uint32_t b) const { return b < half ? b & ((1u << sh_lo) - 1) : (b - half) & ((1u << sh_hi) - 1); }
    // bin -> its first bucket and its bucket count
    __device__ __forceinline__ uint32_t first(uint32_t bin_) const { return bin_ < nlo ? bin_ << sh_lo : half + ((bin_ - nlo) << sh_hi); }
    __device__ __forceinline__ uint32_t bits(uint32_t bin_) const { return bin_ < nlo ? sh_lo : sh_hi; }
};

// One workgroup per tile of WIDE_TILE scalars, ALL windows: the histogram of the (coarse) bins of
// the tile's entries in LDS (u16 pairs, one ds_add_u32 per entry) -> the tile's row of tile_hist.
__global__ __launch_bounds__(1024) void k_hist_wide(const Fr *__restrict__ scalars, size_t n, SegList segs, WidePlan pl, uint32_t B,
                                                    uint32_t Bc, CoarseMap cm, uint32_t pitch, uint32_t tile,
                                                    uint16_t *__restrict__ tile_hist) {
    __builtin_amdgcn_s_setprio(3);      // the sort runs beside the previous call's tail: do not starve behind its older wavefronts
    extern __shared__ __attribute__((aligned(16))) uint32_t cnt2[];   // Bc/2 words: two u16 counters each
    for (uint32_t x = threadIdx.x; x < (Bc + 1) / 2; x += blockDim.x) cnt2[x] = 0;
    __syncthreads();
    const uint32_t t = xcd_tile(blockIdx.x, gridDim.x);
    const size_t lo = (size_t)t * tile;
    for (uint32_t il = threadIdx.x; il < tile; il += blockDim.x) {
        const size_t i = lo + il;
        if (i >= n) break;
        const uint32_t seg = segment_of(segs, (uint32_t)i);
        wide_digits(scalars[i], pl, seg * B, [&](unsigned, int32_t sd) {
            if (sd != 0) {
                const uint32_t b = cm.bin((uint32_t)(sd < 0 ? -sd : sd) - 1);
                atomicAdd(&cnt2[b >> 1], 1u << ((b & 1) * 16));                 // ds_add_u32
            }
        });
    }
    __syncthreads();
    uint16_t *th = tile_hist + (size_t)t * pitch;
    const uint16_t *c16 = reinterpret_cast<const uint16_t *>(cnt2);
    for (uint32_t x = threadIdx.x; x < Bc; x += blockDim.x) th[x] = c16[x];
}

// The same tiles again.  Every workgroup first rebuilds the exclusive prefix of the Bc bin
// populations in LDS (a few KB out of L2: cheaper than three scan launches), adds its tile's row
// of tile_base, and then hands out positions with one returning LDS atomic per entry: the order
// of a (tile, bin) run's records is whatever the atomics give -- bucket sums do not depend on it
// -- so no rank array travels between the passes.
// REC 0: the final 32-bit entries (one pass sorts completely); REC 1: 64-bit
// records (7 fine bits << 32 | entry) ordered by coarse bin; REC 2: 32-bit records
// (sign | 6 fine bits | 25-bit point reference) when every point reference fits 25 bits (tables of
// up to 2^20 points): half the bytes through the scatter and the fine sort.
// Entry = index inside the segment + copy * win_stride | sign << 31.
template <int REC>
__global__ __launch_bounds__(1024) void k_scatter_wide(const Fr *__restrict__ scalars, size_t n, SegList segs, WidePlan pl, uint32_t B,
                                                       uint32_t Bc, uint32_t pitch, uint32_t tile, const uint32_t *__restrict__ hist_c,
                                                       uint32_t *__restrict__ offs, const uint32_t *__restrict__ tile_base,
                                                       void *__restrict__ out, uint32_t win_stride) {
    extern __shared__ __attribute__((aligned(16))) uint32_t base[];   // Bc words
    __shared__ uint32_t wsum[16];
    const uint32_t t = xcd_tile(blockIdx.x, gridDim.x);
    const uint32_t *tb = tile_base + (size_t)t * pitch;
    {
        // exclusive scan of hist_c[0 .. Bc): thread x owns `per` consecutive bins
        const uint32_t per = (Bc + 1023) / 1024, b0 = threadIdx.x * per;
        uint32_t sum = 0;
        for (uint32_t q = 0; q < per; q++) if (b0 + q < Bc) sum += hist_c[b0 + q];
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = __shfl_up(incl, d, 64);
            if ((int)(threadIdx.x & 63) >= d) incl += u;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) run += wsum[w];
        for (uint32_t q = 0; q < per; q++) {
            const uint32_t b = b0 + q;
            if (b < Bc) {
                if (blockIdx.x == 0) offs[b] = run;
                base[b] = run + tb[b];
                run += hist_c[b];
            }
        }
    }
    __syncthreads();
    const size_t lo = (size_t)t * tile;
    for (uint32_t il = threadIdx.x; il < tile; il += 1024) {
        const size_t i = lo + il;
        if (i >= n) break;
        const uint32_t seg = segment_of(segs, (uint32_t)i);
        const uint32_t local = (uint32_t)i - segs.off[seg];
        wide_digits(scalars[i], pl, seg * B, [&](unsigned k, int32_t sd) {
            if (sd != 0) {
                const uint32_t b = (uint32_t)(sd < 0 ? -sd : sd) - 1;
                const uint32_t ent = (local + k * win_stride) | (sd < 0 ? 0x80000000u : 0u);   // window k reads its own copy of the bases
                if (REC == 1) ((uint64_t *)out)[atomicAdd(&base[b >> 7], 1u)] = ((uint64_t)(b & 127u) << 32) | ent;
                else if (REC == 2) ((uint32_t *)out)[atomicAdd(&base[b >> 6], 1u)] = (ent & 0x81ffffffu) | ((b & 63u) << 25);
                else ((uint32_t *)out)[atomicAdd(&base[b], 1u)] = ent;
            }
        });
    }
}

// ------------------------------------------------------------------------------------
// Partitioned two-pass sort (the large single-MSM shape: 2^19 .. 2^21 buckets).  The scatter of
// k_scatter_wide writes 13.6 M four-byte records at n = 2^20 in runs of ~6 (one run per tile and
// coarse bin, 8192 bins): 183 us, against 35 us for the same tiles without the stores
// (k_hist_wide).  Here the first pass cuts the bin space into only PART_SEGS = 256 segments and
// stages a tile's records in LDS, so a tile leaves ~100-record runs behind (whole cache lines),
// and the second pass -- one workgroup per segment, 2^11 .. 2^13 fine buckets counted in LDS -- does
// its scattered stores inside the segment's own few hundred KB, which stay in one L2.
// A record is 32 bits: low bits of the tile number | fine bucket (10 .. 13 bits) << 16 | window << 12 | sign << 11 |
// index inside the tile; the rest of the tile number is implied by the record's position, because a
// segment's region is the concatenation of the tiles' runs in tile order (tile_base column of the
// segment): the second pass only has to find the block of 64 (.. 8) tiles a position falls into.
// ------------------------------------------------------------------------------------
// counter[idx]++ for every active lane, returning the lane's own old value.  When all active lanes of the
// wavefront name the SAME counter (few distinct scalar digits: every record of a segment in one
// bucket) the wavefront issues one atomic instead of 64 serialised ones.
__device__ __forceinline__ uint32_t lds_inc_rank(uint32_t *counters, uint32_t idx) {
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
    const uint64_t active = __ballot(1), same = __ballot(idx == first);
    if (same == active) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(active >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)active, 0u));
        uint32_t base = 0;
        if (rank == 0) base = atomicAdd(&counters[first], (uint32_t)__popcll(active));
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + rank;
    }
    return atomicAdd(&counters[idx], 1u);
}
#define PART_TILE 2048u
#define PART_SEGS 768u       // at most: 512 bins of 2^9 buckets below B/2 and 256 of 2^10 above, at 2^19 buckets
#define PART_STAGE 28672u     // records the second pass can stage in LDS (a segment holds 26624 +- 160 at n = 2^20)
__global__ __launch_bounds__(1024) void k_partition(const Fr *__restrict__ scalars, size_t n, SegList segs, WidePlan pl, uint32_t B, uint32_t Bc,
                                                    CoarseMap cm, uint32_t pitch, const uint16_t *__restrict__ tile_hist,
                                                    const uint32_t *__restrict__ hist_c, uint32_t *__restrict__ offs_c,
                                                    const uint32_t *__restrict__ tile_base, uint32_t *__restrict__ recs) {
    __builtin_amdgcn_s_setprio(3);      // the sort runs beside the previous call's tail: do not starve behind its older wavefronts
    extern __shared__ __attribute__((aligned(16))) uint32_t stage[];      // nwin * PART_TILE records
    __shared__ uint32_t lstart[PART_SEGS + 1], lcur[PART_SEGS], gdst[PART_SEGS], wtot[2][PART_SEGS / 64];
    const uint32_t t = xcd_tile(blockIdx.x, gridDim.x);
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    {
        // threads < Bc (<= 768: twelve wavefronts): exclusive prefix of the segment populations
        // (global: where a segment starts) and of this tile's row (where its run starts in LDS)
        uint32_t gv = 0, lv = 0;
        if (threadIdx.x < Bc) { gv = hist_c[threadIdx.x]; lv = tile_hist[(size_t)t * pitch + threadIdx.x]; }
        uint32_t gi = gv, li = lv;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t a = __shfl_up(gi, d, 64), b = __shfl_up(li, d, 64);
            if ((int)lane >= d) { gi += a; li += b; }
        }
        if (wv < PART_SEGS / 64 && lane == 63) { wtot[0][wv] = gi; wtot[1][wv] = li; }
        __syncthreads();
        if (threadIdx.x < Bc) {
            uint32_t gex = gi - gv, lex = li - lv;
            for (unsigned w = 0; w < wv; w++) { gex += wtot[0][w]; lex += wtot[1][w]; }
            lstart[threadIdx.x] = lex;
            lcur[threadIdx.x] = lex;
            gdst[threadIdx.x] = gex + tile_base[(size_t)t * pitch + threadIdx.x];
            if (blockIdx.x == 0) offs_c[threadIdx.x] = gex;
            if (threadIdx.x == Bc - 1) lstart[Bc] = lex + lv;
        }
    }
    __syncthreads();
    const size_t lo = (size_t)t * PART_TILE;
    const uint32_t fine_bits = cm.sh_hi;                               // (sh_lo <= sh_hi: the fine field is sh_hi bits wide)
    const uint32_t tlow = t & ((1u << (16 - fine_bits)) - 1);          // the record's spare high bits: low bits of the tile number
    for (uint32_t j = 0; j < PART_TILE / 1024; j++) {
        const uint32_t il = threadIdx.x + j * 1024;
        const size_t i = lo + il;
        if (i >= n) break;
        const uint32_t seg = segment_of(segs, (uint32_t)i);
        wide_digits(scalars[i], pl, seg * B, [&](unsigned k, int32_t sd) {
            if (sd != 0) {
                const uint32_t b = (uint32_t)(sd < 0 ? -sd : sd) - 1;
                const uint32_t pos = atomicAdd(&lcur[cm.bin(b)], 1u);
                stage[pos] = (tlow << (16 + fine_bits)) | (cm.fine(b) << 16) | (k << 12) | (sd < 0 ? 0x800u : 0u) | il;
            }
        });
    }
    __syncthreads();
    // half a wavefront moves one run (~50 records): consecutive lanes, consecutive words
    for (uint32_t sg = threadIdx.x >> 5; sg < Bc; sg += 32) {
        const uint32_t a = lstart[sg], e = lstart[sg + 1], g = gdst[sg];
        for (uint32_t x = a + (lane & 31); x < e; x += 32) recs[g + (x - a)] = stage[x];
    }
}

// Second pass: one workgroup per segment.  LDS: cnt[F] | cur[F] | col[T + 1] | stage[PART_STAGE] (F =
// 2^fine_bits fine buckets, T tiles; col[t] = start of tile t's run inside the segment).
__global__ __launch_bounds__(1024) void k_fine_sort_part(const uint32_t *__restrict__ recs, const uint32_t *__restrict__ offs_c,
                                                         const uint32_t *__restrict__ hist_c, const uint32_t *__restrict__ tile_base,
                                                         uint32_t pitch, uint32_t T, CoarseMap cm, SegList segs, uint32_t win_stride, uint32_t stage_cap,
                                                         uint32_t *__restrict__ entries, uint32_t *__restrict__ hist, uint32_t *__restrict__ offs) {
    __builtin_amdgcn_s_setprio(3);      // the sort runs beside the previous call's tail: do not starve behind its older wavefronts
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    __shared__ uint32_t wsum[16];
    const uint32_t sg = blockIdx.x, lo = offs_c[sg], ns = hist_c[sg];
    const uint32_t fine_bits = cm.sh_hi, FM = 1u << fine_bits, fmask = FM - 1, tbits = 16 - fine_bits;   // record layout (tbits low bits of the tile number ride in it)
    const uint32_t F = 1u << cm.bits(sg), bucket0 = cm.first(sg);      // this segment's buckets: [bucket0, bucket0 + F)
    uint32_t *cnt = sm, *cur = sm + FM, *col = sm + 2 * FM, *stage = col + T + 1;
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t x = threadIdx.x; x < F; x += 1024) cnt[x] = 0;
    for (uint32_t x = threadIdx.x; x < T; x += 1024) col[x] = tile_base[(size_t)x * pitch + sg];
    if (threadIdx.x == 0) col[T] = ns;
    // exclusive scan of the F counters -> cur[], and this segment's slice of hist / offs
    auto scan_counters = [&]() {
        const uint32_t per = (F + 1023) >> 10, f0 = threadIdx.x * per;
        uint32_t sum = 0;
        for (uint32_t q = 0; q < per; q++) if (f0 + q < F) sum += cnt[f0 + q];
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = __shfl_up(incl, d, 64);
            if ((int)lane >= d) incl += u;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (unsigned w = 0; w < wv; w++) run += wsum[w];
        for (uint32_t q = 0; q < per; q++) {
            if (f0 + q >= F) break;
            const uint32_t c = cnt[f0 + q];
            cur[f0 + q] = run;
            hist[(size_t)bucket0 + f0 + q] = c;
            offs[(size_t)bucket0 + f0 + q] = lo + run;
            run += c;
        }
    };
    // record at position p2 of the segment -> entry: it belongs to tile t with col[t] <= p2 < col[t + 1]
    // (runs may be empty): binary search for the last t with col[t] <= p2
    auto entry_of = [&](uint32_t r, uint32_t p2) {
        // block of 2^tbits tiles: the last j with col[j << tbits] <= p2 (few distinct addresses: cheap LDS reads)
        uint32_t a = 0, e = (T + (1u << tbits) - 1) >> tbits;   // invariant: col[a << tbits] <= p2, and p2 < col[e << tbits] (or e is the end)
        while (e - a > 1) {
            const uint32_t m = (a + e) >> 1;
            if (col[m << tbits] <= p2) a = m; else e = m;
        }
        const uint32_t i = ((a << tbits) | (r >> (16 + fine_bits))) * PART_TILE + (r & 0x7ffu);
        const uint32_t local = i - segs.off[segment_of(segs, i)];
        return (local + ((r >> 12) & 15u) * win_stride) | ((r & 0x800u) << 20);
    };
    if (ns <= stage_cap) {
        // The usual case (every segment, with uniformly distributed digits at n <= 2^20): the
        // segment's records are read ONCE, all loads of a thread in flight together, and stay in
        // registers across the count and the placement; the entries are ordered in LDS and leave
        // as one linear copy -- no scattered global stores at all.
        constexpr uint32_t NR = PART_STAGE / 1024;
        uint32_t r[NR];
#pragma unroll
        for (uint32_t u = 0; u < NR; u++) { const uint32_t p2 = threadIdx.x + u * 1024; r[u] = p2 < ns ? recs[lo + p2] : 0u; }
        __syncthreads();
        uint16_t rk[NR];                                     // rank inside the fine bucket, from the counting atomic
#pragma unroll
        for (uint32_t u = 0; u < NR; u++) {
            rk[u] = 0;
            if (threadIdx.x + u * 1024 < ns) rk[u] = (uint16_t)atomicAdd(&cnt[(r[u] >> 16) & fmask], 1u);
        }
        __syncthreads();
        scan_counters();
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < NR; u++) {
            const uint32_t p2 = threadIdx.x + u * 1024;
            if (p2 < ns) stage[cur[(r[u] >> 16) & fmask] + rk[u]] = entry_of(r[u], p2);
        }
        __syncthreads();
        for (uint32_t x = threadIdx.x; x < ns; x += 1024) entries[lo + x] = stage[x];
        return;
    }
    // A larger segment (skewed digits, or n > 2^20) is read twice and places its entries with
    // scattered stores inside its own region.
    __syncthreads();
    constexpr uint32_t UNR = 8;
    for (uint32_t p0 = threadIdx.x; p0 < ns; p0 += 1024 * UNR) {
        uint32_t r[UNR];
#pragma unroll
        for (uint32_t u = 0; u < UNR; u++) { const uint32_t p2 = p0 + u * 1024; r[u] = p2 < ns ? recs[lo + p2] : 0u; }
#pragma unroll
        for (uint32_t u = 0; u < UNR; u++) if (p0 + u * 1024 < ns) (void)lds_inc_rank(cnt, (r[u] >> 16) & fmask);
    }
    __syncthreads();
    scan_counters();
    __syncthreads();
    for (uint32_t p0 = threadIdx.x; p0 < ns; p0 += 1024 * UNR) {
        uint32_t r[UNR];
#pragma unroll
        for (uint32_t u = 0; u < UNR; u++) { const uint32_t p2 = p0 + u * 1024; r[u] = p2 < ns ? recs[lo + p2] : 0u; }
#pragma unroll
        for (uint32_t u = 0; u < UNR; u++) {
            const uint32_t p2 = p0 + u * 1024;
            if (p2 >= ns) break;
            entries[lo + lds_inc_rank(cur, (r[u] >> 16) & fmask)] = entry_of(r[u], p2);
        }
    }
}

template <class Rec>
struct RecOps;
template <>
struct RecOps<uint64_t> {          // 7 fine bits above a 32-bit entry
    static constexpr uint32_t NF = 128;
    static __device__ __forceinline__ uint32_t fine(uint64_t r) { return (uint32_t)(r >> 32); }
    static __device__ __forceinline__ uint32_t entry(uint64_t r) { return (uint32_t)r; }
};
template <>
struct RecOps<uint32_t> {          // sign | 6 fine bits | 25-bit point reference
    static constexpr uint32_t NF = 64;
    static __device__ __forceinline__ uint32_t fine(uint32_t r) { return (r >> 25) & 63u; }
    static __device__ __forceinline__ uint32_t entry(uint32_t r) { return r & 0x81ffffffu; }
};
template <class Rec>
__global__ __launch_bounds__(1024) void k_fine_sort(const Rec *__restrict__ recs, const uint32_t *__restrict__ offs_c,
                                                    const uint32_t *__restrict__ hist_c, uint32_t *__restrict__ entries,
                                                    uint32_t *__restrict__ hist, uint32_t *__restrict__ offs) {
    using O = RecOps<Rec>;
    constexpr uint32_t NF = O::NF;
    __shared__ uint32_t cnt[NF], cur[NF];
    const uint32_t bin = blockIdx.x, lo = offs_c[bin], n = hist_c[bin];
    if (threadIdx.x < NF) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < n; j += 1024) atomicAdd(&cnt[O::fine(recs[lo + j])], 1u);
    __syncthreads();
    if (threadIdx.x < 64) {          // exclusive scan of the NF counters by the first wavefront, NF/64 per lane
        constexpr uint32_t PER = NF / 64;
        uint32_t c[PER], tot = 0;
#pragma unroll
        for (uint32_t q = 0; q < PER; q++) { c[q] = cnt[PER * threadIdx.x + q]; tot += c[q]; }
        uint32_t incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(incl, d, 64);
            if ((int)threadIdx.x >= d) incl += t;
        }
        uint32_t ex = incl - tot;
#pragma unroll
        for (uint32_t q = 0; q < PER; q++) {
            cur[PER * threadIdx.x + q] = ex;
            hist[(size_t)bin * NF + PER * threadIdx.x + q] = c[q];
            offs[(size_t)bin * NF + PER * threadIdx.x + q] = lo + ex;
            ex += c[q];
        }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < n; j += 1024) {
        const Rec r = recs[lo + j];
        const uint32_t pos = atomicAdd(&cur[O::fine(r)], 1u);
        entries[lo + pos] = O::entry(r);
    }
}

// ------------------------------------------------------------------------------------
// kernel 3b: order buckets by population, largest first, so that the 64 lanes of a
// wavefront walk equally long entry lists (bucket sizes are ~Poisson(n/2^(c-1)); unsorted,
// a wavefront waits for its longest lane: ~1.45x the mean at n=2^20, c=16).
// Counting sort on min(count, SIZE_BINS-1) with LDS-aggregated histograms; empty buckets
// get their identity written here, over-threshold ones go to the heavy list.
// ------------------------------------------------------------------------------------
// lanes per bucket in k_accumulate: 2 on the plain path (GLV halves the bucket count; two lanes keep
// >= 2.5 wavefronts per SIMD slot in flight), 1 on the wide path (2^19 .. 2^21 buckets)
#define SIZE_BINS 1025          // bin 0 unused (total), bins 1..1024
// population -> bin: ((cnt - 1) >> bin_shift) + 1 in [1, SIZE_BINS - 1] for cnt in [1, thr]
__device__ __forceinline__ uint32_t size_bin(uint32_t cnt, uint32_t bin_shift) { return ((cnt - 1) >> bin_shift) + 1; }
template <class C>
__global__ __launch_bounds__(256) void k_size_hist(const uint32_t *__restrict__ hist, uint32_t nb, uint32_t thr, uint32_t *__restrict__ bin_count,
                                                   uint32_t group_size, uint32_t bin_shift) {
    // group_size != 0: buckets are ordered group by group (a group = one window's 2^(c-1)
    // buckets, a multiple of this block's 2048), by population inside each group
    bin_count += (group_size ? (blockIdx.x * 2048u) / group_size : 0u) * SIZE_BINS;
    __shared__ uint32_t lcnt[SIZE_BINS];
    for (uint32_t t = threadIdx.x; t < SIZE_BINS; t += 256) lcnt[t] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * 2048 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint32_t g = base + j * 256;
        if (g < nb) { uint32_t cnt = hist[g]; if (cnt > 0 && cnt <= thr) atomicAdd(&lcnt[size_bin(cnt, bin_shift)], 1u); }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < SIZE_BINS; t += 256) if (lcnt[t]) atomicAdd(&bin_count[t], lcnt[t]);
}
// bin_start[g][b] = number of buckets ordered before population b of group g (groups ascending,
// populations descending); the total goes to bin_start[0][0]
__global__ __launch_bounds__(256) void k_size_scan(const uint32_t *__restrict__ bin_count, uint32_t *__restrict__ bin_start, uint32_t ngroups) {
    __shared__ uint32_t lds[4];
    uint32_t carry = 0;
    for (uint32_t gidx = 0; gidx < ngroups; gidx++) {
        const uint32_t *bc = bin_count + gidx * SIZE_BINS;
        uint32_t *bs = bin_start + gidx * SIZE_BINS;
        // lane t owns bins [SIZE_BINS-1-4t-3 .. SIZE_BINS-1-4t], visited from large to small
        uint32_t v[4], s = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int bin = (int)SIZE_BINS - 1 - (int)(threadIdx.x * 4 + j);
            v[j] = (bin >= 1) ? bc[bin] : 0;
            s += v[j];
        }
        uint32_t tot;
        uint32_t run = carry + block_exclusive_scan_256(s, lds, &tot);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int bin = (int)SIZE_BINS - 1 - (int)(threadIdx.x * 4 + j);
            if (bin >= 1) bs[bin] = run;
            run += v[j];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) bin_start[0] = carry;     // number of buckets in the permutation
}
template <class C>
__global__ __launch_bounds__(256) void k_size_scatter(const uint32_t *__restrict__ hist, uint32_t nb, uint32_t thr, const uint32_t *__restrict__ bin_start,
                                                      uint32_t *__restrict__ bin_cursor, uint32_t *__restrict__ perm,
                                                      uint32_t *__restrict__ heavy_list, uint32_t *__restrict__ heavy_count,
                                                      typename C::Acc *__restrict__ buckets, uint32_t group_size, uint32_t bin_shift,
                                                      uint32_t split) {
    const uint32_t goff = (group_size ? (blockIdx.x * 2048u) / group_size : 0u) * SIZE_BINS;
    bin_start += goff;
    bin_cursor += goff;
    __shared__ uint32_t lcnt[SIZE_BINS];
    for (uint32_t t = threadIdx.x; t < SIZE_BINS; t += 256) lcnt[t] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * 2048 + threadIdx.x;
    uint32_t cnt[8], rank[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint32_t g = base + j * 256;
        cnt[j] = 0; rank[j] = 0;
        if (g < nb) {
            cnt[j] = hist[g];
            if (cnt[j] == 0) { for (uint32_t h = 0; h < split; h++) buckets[(size_t)g * split + h] = C::inf(); }
            else if (cnt[j] > thr) heavy_list[atomicAdd(heavy_count, 1u)] = g;
            else rank[j] = atomicAdd(&lcnt[size_bin(cnt[j], bin_shift)], 1u);
        }
    }
    __syncthreads();
    // reserve this block's slice of every bin it touched (reuse lcnt as the slice base)
    for (uint32_t t = threadIdx.x; t < SIZE_BINS; t += 256) {
        uint32_t c = lcnt[t];
        lcnt[t] = c ? atomicAdd(&bin_cursor[t], c) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint32_t g = base + j * 256;
        if (g < nb && cnt[j] > 0 && cnt[j] <= thr) {
            const uint32_t bin = size_bin(cnt[j], bin_shift);
            perm[bin_start[bin] + lcnt[bin] + rank[j]] = g;
        }
    }
}

// ------------------------------------------------------------------------------------
// kernel 4: bucket accumulation (one lane per bucket); heavy buckets deferred.
// C = curve traits (CurveG1: 29-bit limbs, 64-B packed bases; CurveG2: Fq2 on the same limbs, 128-B bases).
// The next entry's point is fetched before the current mixed add is issued, so the
// ~2 us gather latency hides under ~2300 VALU instructions of arithmetic.
// ------------------------------------------------------------------------------------
template <class C, uint32_t ACC_SPLIT>
__device__ __forceinline__ void accumulate_body(const typename C::Base *__restrict__ bases, const uint32_t *__restrict__ entries,
                                                const uint32_t *__restrict__ offs, const uint32_t *__restrict__ hist,
                                                const uint32_t *__restrict__ perm, const uint32_t *__restrict__ nperm,
                                                typename C::Acc *__restrict__ buckets) {
    // ACC_SPLIT lanes per bucket take interleaved parts of its entry list; the parts are summed
    // when the bucket reduction loads them.
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= *nperm * ACC_SPLIT) return;
    const uint32_t g = perm[t / ACC_SPLIT], part = t % ACC_SPLIT;
    const uint32_t cnt = hist[g];           // 1 .. heavy_threshold
    const uint32_t *e = entries + offs[g];
    typename C::Acc acc = C::inf();
    if (part < cnt) {
        uint32_t v = e[part];
        typename C::Base cur = bases[v & 0x3fffffffu];
        for (uint32_t j = part; j < cnt; j += ACC_SPLIT) {
            uint32_t vn = v;
            typename C::Base nxt = cur;
            if (j + ACC_SPLIT < cnt) { vn = e[j + ACC_SPLIT]; nxt = bases[vn & 0x3fffffffu]; }
            acc = C::madd(acc, cur, (v >> 31) != 0, ((v >> 30) & 1) != 0);
            v = vn;
            cur = nxt;
        }
    }
    buckets[(size_t)g * ACC_SPLIT + part] = acc;
}
template <class C, uint32_t ACC_SPLIT>
__global__ __launch_bounds__(256) void k_accumulate(const typename C::Base *__restrict__ bases, const uint32_t *__restrict__ entries,
                                                    const uint32_t *__restrict__ offs, const uint32_t *__restrict__ hist,
                                                    const uint32_t *__restrict__ perm, const uint32_t *__restrict__ nperm,
                                                    typename C::Acc *__restrict__ buckets) {
    accumulate_body<C, ACC_SPLIT>(bases, entries, offs, hist, perm, nperm, buckets);
}
// The G2 kernel capped to the registers of two wavefronts per SIMD: 256 VGPRs + 316 B of scratch per
// lane instead of 256 + 42 AGPRs at one wavefront per SIMD -- 3.35 -> 3.16 ms at n = 2^20 on the
// wide path (the second wavefront hides the gathers the single one stalled on)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_accumulate_g2_occ2(
    const CurveG2::Base *__restrict__ bases, const uint32_t *__restrict__ entries, const uint32_t *__restrict__ offs,
    const uint32_t *__restrict__ hist, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ nperm, CurveG2::Acc *__restrict__ buckets) {
    accumulate_body<CurveG2, 1u>(bases, entries, offs, hist, perm, nperm, buckets);
}

// G2 with a PAIR of lanes per bucket, each lane on one Fq component of every coordinate (fp29x2l.h): the same entry
// walk and the same formulas as above, half the registers per lane.  A lane fetches its own halves of the next base
// (x.c, y.c of its component: 2 x 32 B of the 128-B record, so a pair still reads one contiguous record).
__global__ __launch_bounds__(256) void k_accumulate_g2_pair(const AffPackedG2 *__restrict__ bases, const uint32_t *__restrict__ entries,
                                                            const uint32_t *__restrict__ offs, const uint32_t *__restrict__ hist,
                                                            const uint32_t *__restrict__ perm, const uint32_t *__restrict__ nperm,
                                                            XYZZ29x2 *__restrict__ buckets) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, pr = t >> 1, part = t & 1;
    if (pr >= *nperm) return;                               // (both lanes of a pair leave together)
    const uint32_t g = perm[pr];
    const uint32_t cnt = hist[g];                           // 1 .. heavy_threshold
    const uint32_t *e = entries + offs[g];
    struct Half { uint32_t x[8], y[8]; };
    auto fetch = [&](uint32_t v) {
        const AffPackedG2 &b = bases[v & 0x3fffffffu];
        Half h;
#pragma unroll
        for (int i = 0; i < 8; i++) { h.x[i] = b.w[part][i]; h.y[i] = b.w[2 + part][i]; }
        return h;
    };
    XyzzE<F29h> acc = XyzzE<F29h>::inf();
    uint32_t v = e[0];
    Half cur = fetch(v);
    for (uint32_t j = 0; j < cnt; j++) {
        uint32_t vn = v;
        Half nxt = cur;
        if (j + 1 < cnt) { vn = e[j + 1]; nxt = fetch(vn); }
        AffE<F29h> q = {{F29::unpack256(cur.x)}, {F29::unpack256(cur.y)}};
        if (!q.is_inf()) {
            if (v >> 31) q.y = sub_k<1>(F29h::zero(), q.y);  // p - y per component
            acc = g2_madd(acc, q);
        }
        v = vn;
        cur = nxt;
    }
    F29 *o = reinterpret_cast<F29 *>(&buckets[g]);          // X.c0, X.c1, Y.c0, Y.c1, ZZ.c0, ...
    o[part] = acc.X.v;
    o[2 + part] = acc.Y.v;
    o[4 + part] = acc.ZZ.v;
    o[6 + part] = acc.ZZZ.v;
}

// (G1 capped to 128 VGPRs / four wavefronts per SIMD spills 208 B per lane and measured 2 % slower.)
template <class A>
__device__ __forceinline__ A shfl_down_acc(const A &p, unsigned delta) {
    A r;
    constexpr int NW = sizeof(A) / 4;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&p);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&r);
#pragma unroll
    for (int i = 0; i < NW; i++) dst[i] = __shfl_down(src[i], delta, 64);
    return r;
}

// wavefront tree sum: result valid in lane 0
template <class C>
__device__ __forceinline__ typename C::Acc wave_sum(typename C::Acc v, unsigned lane) {
    for (unsigned d = 32; d >= 1; d >>= 1) {
        typename C::Acc t = shfl_down_acc(v, d);
        if (lane + d < 64) v = C::add(v, t);
    }
    return v;
}

// Heavy buckets (population > threshold): k_heavy_plan cuts every heavy bucket into chunks of
// HEAVY_CHUNK entries (exclusive scan of the chunk counts); k_accumulate_heavy gives each
// chunk one wavefront (8 sequential mixed adds per lane, then a shuffle tree);
// k_heavy_finish sums a bucket's chunk partials.  With uniformly random scalars and c | 128
// there are no heavy buckets and all three exit at once.
#define HEAVY_CHUNK 512u
// Chunk size of this call: 512 entries, doubled (up to 4096) while that still leaves >= 2048 chunks.
// A chunk costs 6 general additions per lane (the shuffle tree) on top of its mixed additions --
// 8 per lane at 512 entries, i.e. 40 % overhead, which is right for a handful of heavy buckets
// (many short chunks = parallelism) and wasteful when most of the input sits in heavy buckets
// (few distinct scalars: 13.6 M heavy entries at n = 2^20, k_accumulate_heavy 2.0 -> 1.3 ms).
// Stored behind the chunk offsets: chunk_off[nh + 1].
__global__ __launch_bounds__(256) void k_heavy_plan(const uint32_t *__restrict__ hist, const uint32_t *__restrict__ heavy_list,
                                                    const uint32_t *__restrict__ heavy_count, uint32_t *__restrict__ chunk_off) {
    __shared__ uint32_t lds[4];
    __shared__ uint32_t s_chunk;
    const uint32_t nh = *heavy_count;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nh; base += 256) {           // total number of heavy entries
        uint32_t h = base + threadIdx.x;
        uint32_t tot;
        (void)block_exclusive_scan_256(h < nh ? hist[heavy_list[h]] : 0, lds, &tot);
        carry += tot;
    }
    if (threadIdx.x == 0) {
        uint32_t chunk = HEAVY_CHUNK;
        while (chunk < 4096u && carry / (2 * chunk) >= 2048u) chunk *= 2;
        s_chunk = chunk;
    }
    __syncthreads();
    const uint32_t chunk = s_chunk;
    carry = 0;
    for (uint32_t base = 0; base < nh; base += 256) {
        uint32_t h = base + threadIdx.x;
        uint32_t v = h < nh ? (hist[heavy_list[h]] + chunk - 1) / chunk : 0;
        uint32_t tot;
        uint32_t ex = block_exclusive_scan_256(v, lds, &tot);
        if (h < nh) chunk_off[h] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) { chunk_off[nh] = carry; chunk_off[nh + 1] = chunk; }      // total number of chunks; chunk size
}

template <class C>
__global__ __launch_bounds__(64) void k_accumulate_heavy(const typename C::Base *__restrict__ bases, const uint32_t *__restrict__ entries,
                                                         const uint32_t *__restrict__ offs, const uint32_t *__restrict__ hist,
                                                         const uint32_t *__restrict__ heavy_list, const uint32_t *__restrict__ heavy_count,
                                                         const uint32_t *__restrict__ chunk_off, typename C::Acc *__restrict__ partials) {
    const uint32_t nh = *heavy_count;
    if (nh == 0) return;
    const uint32_t total = chunk_off[nh], chunk = chunk_off[nh + 1];
    const unsigned lane = threadIdx.x;
    for (uint32_t v = blockIdx.x; v < total; v += gridDim.x) {
        uint32_t lo = 0, hi = nh;                       // largest h with chunk_off[h] <= v
        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (chunk_off[mid] <= v) lo = mid; else hi = mid; }
        const uint32_t g = heavy_list[lo];
        const uint32_t cnt = hist[g];
        const uint32_t start = (v - chunk_off[lo]) * chunk;
        const uint32_t end = start + chunk < cnt ? start + chunk : cnt;
        const uint32_t *e = entries + offs[g];
        typename C::Acc acc = C::inf();
        for (uint32_t j = start + lane; j < end; j += 64) {
            uint32_t w = e[j];
            acc = C::madd(acc, bases[w & 0x3fffffffu], (w >> 31) != 0, ((w >> 30) & 1) != 0);
        }
        acc = wave_sum<C>(acc, lane);
        if (lane == 0) partials[v] = acc;
    }
}

template <class C>
__global__ __launch_bounds__(64) void k_heavy_finish(const uint32_t *__restrict__ heavy_list, const uint32_t *__restrict__ heavy_count,
                                                     const uint32_t *__restrict__ chunk_off, const typename C::Acc *__restrict__ partials,
                                                     typename C::Acc *__restrict__ buckets, uint32_t split) {
    const uint32_t nh = *heavy_count;
    unsigned lane = threadIdx.x;
    for (uint32_t h = blockIdx.x; h < nh; h += gridDim.x) {
        typename C::Acc acc = C::inf();
        for (uint32_t v = chunk_off[h] + lane; v < chunk_off[h + 1]; v += 64) acc = C::add(acc, partials[v]);
        acc = wave_sum<C>(acc, lane);
        if (lane == 0) {
            buckets[(size_t)heavy_list[h] * split] = acc;
            for (uint32_t x = 1; x < split; x++) buckets[(size_t)heavy_list[h] * split + x] = C::inf();
        }
    }
}

// ------------------------------------------------------------------------------------
// kernel 5: bucket reduction  sum_b (b+1) * S_b  per window.  Pure latency (a few thousand
// point additions on an otherwise idle chip), so every point is shared by a QUAD of lanes
// (quad29.h: one field product per lane per dependency level, ~2.7x shorter per addition than a
// lane-private one) and a wavefront holds 16 points.
//
// For quads j = 0..15 of a wavefront holding (acc_j, run_j) the helper returns in quad 0
//   ACC = sum_j acc_j + 2^log_mult * sum_j j*run_j      RUN = sum_j run_j
// using an inclusive suffix scan of run over the quads (4 shuffle steps), then two tree sums.
// ------------------------------------------------------------------------------------
template <class A>
__device__ __forceinline__ A select_acc(uint32_t mask, const A &a, const A &b) {   // mask all ones: a, zero: b
    A r;
    const uint32_t *pa = reinterpret_cast<const uint32_t *>(&a), *pb = reinterpret_cast<const uint32_t *>(&b);
    uint32_t *pr = reinterpret_cast<uint32_t *>(&r);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(A) / 4); i++) pr[i] = (pa[i] & mask) | (pb[i] & ~mask);
    return r;
}
// One call site of quad_add and one of quad_dbl for the whole helper (a loop of nine trips whose
// operands are selected with opaque masks, as in k_reduce1_lane below): the kernels built on it
// run ONE wavefront through their code once, so an unrolled version (13 inlined additions, 122 KB
// for k_reduce2) spent more time fetching instructions than executing them.  Order of work: the
// suffix scan of run (4 additions), then every quad scales its own suffix sum (log_mult
// doublings, all quads side by side) and adds it to its acc, then ONE tree sum (4 additions):
// 9 + log_mult sequential operations instead of 13 + log_mult.
template <class A>
__device__ __forceinline__ void quadwave_weighted(A &acc, A &run, unsigned log_mult, unsigned lane) {
    const unsigned q = lane & 3, qi = lane >> 2;
    A s = A::inf();
#pragma unroll 1
    for (unsigned step = 0; step < 9; step++) {
        // steps 0-3: run_j <- run_j + run_(j+d), d = 1, 2, 4, 8 (suffix scan); step 4: acc_j <- acc_j +
        // 2^log_mult * Suf_j (j >= 1); steps 5-8: acc_j <- acc_j + acc_(j+d), d = 8, 4, 2, 1
        if (step == 4) {
            s = (qi == 0) ? A::inf() : run;
#pragma unroll 1
            for (unsigned i = 0; i < log_mult; i++) s = quad_dbl(s, q);
        }
        uint32_t m_run = step < 4 ? 0xffffffffu : 0u, m_s = step == 4 ? 0xffffffffu : 0u;
        asm volatile("" : "+v"(m_run), "+v"(m_s));
        const unsigned d = step < 4 ? (1u << step) : (step == 4 ? 0u : (8u >> (step - 5)));
        const A lhs = select_acc(m_run, run, acc);
        const A rhs = select_acc(m_s, s, shfl_down_acc(lhs, 4 * d));
        A r = lhs;
        if (qi + d < 16) r = quad_add(lhs, rhs, q);
        run = select_acc(m_run, r, run);
        acc = select_acc(m_run, acc, r);
    }
}

// Wide path, level 1: ONE bucket space of B = 2^19 .. 2^21 buckets (every window gathers from its
// own pre-shifted copy of the bases, so bucket b has weight b+1 whatever the window).  That many
// buckets are throughput, not latency: one LANE per segment of L buckets with lane-private
// additions, writing the (ACC, RUN) pair the quad levels (k_reduce2) continue from.
// The kernel holds ONE inlined copy of the general addition (~40 KB of code with its doubling
// and infinity paths): written with two call sites (RUN += bucket; ACC += RUN) the loop body was
// 83 KB for G1 -- more than the 64-KB instruction cache two CUs share -- and every kernel that
// ran beside it on another stream (the next call's first sort kernels) spent its time on
// instruction fetches: k_hist_wide 35 -> 124 us, k_tile_scan_rows 6 -> 80 us.  The two additions of
// a bucket are therefore two trips through the same call site; which operands a trip takes is
// selected word by word with a mask the compiler cannot see through (it would unswitch the loop
// into two copies again).
template <class C>
__global__ __launch_bounds__(64) void k_reduce1_lane(const typename C::Acc *__restrict__ buckets, uint32_t B, uint32_t L, uint32_t split,
                                                     typename C::Acc *__restrict__ out) {
    using A = typename C::Acc;
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if ((uint64_t)t * L >= B) return;
    A acc = A::inf(), run = A::inf();
    const A *bk = buckets + (size_t)t * L * split;
    // trips per bucket: `split` of "RUN += part h", then one of "ACC += RUN"
    const uint32_t per = split + 1;
#pragma unroll 1
    for (int i = (int)L - 1; i >= 0; i--) {
#pragma unroll 1
        for (uint32_t h = 0; h < per; h++) {
            uint32_t m = h < split ? 0xffffffffu : 0u;       // all ones: RUN += bucket part
            asm volatile("" : "+v"(m));
            A rhs = run;
            if (h < split) rhs = bk[(size_t)i * split + h];
            const A lhs = select_acc(m, run, acc);
            const A r = C::add(lhs, rhs);
            run = select_acc(m, r, run);
            acc = select_acc(m, acc, r);
        }
    }
    out[2 * (size_t)t] = acc;
    out[2 * (size_t)t + 1] = run;
}

// The first 16-ary level over the 65536 (ACC, RUN) pairs of k_reduce1_lane, ONE LANE per group of sixteen (wide path, calls
// whose tail hides under the next call's front).  The quad version of this level (k_reduce2) is the shortest chain --
// 9 + log_mult quad operations -- but every quad operation keeps four lanes busy for one addition and carries ~40 % of
// selects and broadcasts: 4096 wavefronts x ~20 000 instructions = 82 M wavefront-instructions per MSM, as much as a
// sixth of the accumulate kernel, on a chip whose issue slots are the bottleneck of a pipelined step.  A lane-private
// running sum does the same arithmetic in 48 general additions per lane: 64 wavefronts x ~190 000 instructions = 12 M.
// Seven times less work, 0.2 ms more latency -- the right trade exactly when nobody waits for this call alone.
//   ACC = sum_j acc_j + 2^log_mult * sum_j j * run_j,  RUN = sum_j run_j   (j = 0 .. 15, as quadwave_weighted)
template <class C>
__global__ __launch_bounds__(64) void k_reduce2_lane(const typename C::Acc *__restrict__ in, uint32_t m_in, uint32_t m_out, uint32_t log_mult,
                                                     typename C::Acc *__restrict__ out) {
    using A = typename C::Acc;
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t >= m_out) return;
    A acc = A::inf(), run = A::inf(), wsum = A::inf();
    // trips per input pair through ONE addition site (see k_reduce1_lane): acc += acc_j; run += run_j; wsum += run (j >= 1)
#pragma unroll 1
    for (int j = 15; j >= 0; j--) {
        const uint32_t idx = 16 * t + (uint32_t)j;
        const bool have = idx < m_in;
        const uint32_t trips = j ? 3u : 2u;
#pragma unroll 1
        for (uint32_t h = 0; h < trips; h++) {
            uint32_t m0 = h == 0 ? 0xffffffffu : 0u, m1 = h == 1 ? 0xffffffffu : 0u;
            asm volatile("" : "+v"(m0), "+v"(m1));
            A rhs = run;                                                    // h == 2: wsum += run
            if (h < 2) rhs = have ? in[2 * (size_t)idx + h] : A::inf();
            const A lhs = select_acc(m0, acc, select_acc(m1, run, wsum));
            const A r = C::add(lhs, rhs);
            acc = select_acc(m0, r, acc);
            run = select_acc(m1, r, run);
            wsum = select_acc(~(m0 | m1), r, wsum);
        }
    }
#pragma unroll 1
    for (uint32_t i = 0; i < log_mult; i++) wsum = C::dbl(wsum);
    out[2 * (size_t)t] = C::add(acc, wsum);
    out[2 * (size_t)t + 1] = run;
}

// Level 1: T = B/L quads per window; quad t owns buckets [t*L, (t+1)*L) (their `split` partial
// sums are folded in here).  Writes one (ACC,RUN) pair per wavefront (16 quads = 16*L buckets).
template <class C>
__global__ __launch_bounds__(64) void k_reduce1(const typename C::Acc *__restrict__ buckets, uint32_t B, uint32_t L, uint32_t logL,
                                                uint32_t waves_per_window, uint32_t split, typename C::Acc *__restrict__ wave_out) {
    using A = typename C::Acc;
    uint32_t wave = blockIdx.x;                 // global wave id = k*waves_per_window + w
    uint32_t k = wave / waves_per_window, w = wave % waves_per_window;
    const unsigned lane = threadIdx.x, q = lane & 3;
    uint32_t t = w * 16 + (lane >> 2);
    A acc = A::inf(), run = A::inf();
    if ((uint64_t)t * L < B) {
        const A *bk = buckets + ((size_t)k * B + (size_t)t * L) * split;
        const uint32_t per = split + 1;              // trips per bucket through ONE addition site (see k_reduce1_lane)
#pragma unroll 1
        for (int i = (int)L - 1; i >= 0; i--) {
#pragma unroll 1
            for (uint32_t h = 0; h < per; h++) {
                uint32_t m = h < split ? 0xffffffffu : 0u;   // all ones: RUN += bucket part h; zero: ACC += RUN
                asm volatile("" : "+v"(m));
                A rhs = run;
                if (h < split) rhs = bk[(size_t)i * split + h];
                const A lhs = select_acc(m, run, acc);
                const A r = quad_add(lhs, rhs, q);
                run = select_acc(m, r, run);
                acc = select_acc(m, acc, r);
            }
        }
    }
    quadwave_weighted(acc, run, logL, lane);
    if (lane == 0) {
        wave_out[2 * (size_t)wave] = acc;
        wave_out[2 * (size_t)wave + 1] = run;
    }
}

// Level >= 2: per window, every wavefront folds 16 consecutive (ACC,RUN) pairs of the previous
// level (each covering 2^log_mult buckets) into one; the last level (m_out == 1) leaves the
// window sum in out[2*k].
template <class C>
__global__ __launch_bounds__(64) void k_reduce2(const typename C::Acc *__restrict__ in, uint32_t m_in, uint32_t m_out, uint32_t log_mult,
                                                typename C::Acc *__restrict__ out) {
    using A = typename C::Acc;
    const uint32_t k = blockIdx.x / m_out, w = blockIdx.x % m_out;
    const unsigned lane = threadIdx.x;
    const uint32_t j = w * 16 + (lane >> 2);
    A acc = A::inf(), run = A::inf();
    if (j < m_in) {
        acc = in[2 * ((size_t)k * m_in + j)];
        run = in[2 * ((size_t)k * m_in + j) + 1];
    }
    quadwave_weighted(acc, run, log_mult, lane);
    if (lane == 0) {
        out[2 * ((size_t)k * m_out + w)] = acc;
        out[2 * ((size_t)k * m_out + w) + 1] = run;
    }
}

// ------------------------------------------------------------------------------------
// kernel 6: Horner fold over windows -> Jacobian (libff layout)
// ------------------------------------------------------------------------------------
// Keep a value in VGPRs: without this the compiler proves the single active lane's data
// wave-uniform and moves the whole fold onto the scalar ALU (3x slower).
template <class A>
__device__ __forceinline__ void pin_vgpr(A &a) {
    uint32_t *w = reinterpret_cast<uint32_t *>(&a);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(A) / 4); i++) asm volatile("" : "+v"(w[i]));
}

// ------------------------------------------------------------------------------------
// kernel 6b (G1): the same Horner fold with each point shared by a QUAD of lanes.  The 240
// doublings are inherently sequential, so the lever is latency per doubling: the 9 field
// products of an XYZZ doubling have only 3 dependency levels, and the 14 of a general add
// have 4.  Lane q of the quad computes the q-th product of each level; results are
// replicated with DPP quad_perm broadcasts (v_mov_b32_dpp, no LDS).  ~2.7x shorter chain.
// ------------------------------------------------------------------------------------
template <class A, class J>
__device__ __forceinline__ void fold_quad_body(const A *__restrict__ window_sums, unsigned nwin, unsigned c, J *__restrict__ out) {
    // EVERY quad of the wavefront runs the same fold (one of them would do): with four lanes on, a lone wavefront executes
    // the chain 1.1-2.1x slower, depending on the CU it landed on (tools/ubench_exec_mask.hip); lane 0 stores
    if (blockIdx.x != 0) return;
    const unsigned q = threadIdx.x & 3;
    // window_sums holds (ACC,RUN) pairs of the last reduction level: window k at index 2k
    A r = window_sums[2 * (nwin - 1)];
    pin_vgpr(r);
    for (int k = (int)nwin - 2; k >= 0; k--) {
        for (unsigned i = 0; i < c; i++) r = quad_dbl(r, q);
        A w = window_sums[2 * k];
        pin_vgpr(w);
        r = quad_add(r, w, q);
    }
    J res;
    if constexpr (std::is_same<A, XYZZ29>::value) res = xyzz29_to_jac(r);
    else res = g2_to_jac(r);
    pin_vgpr(res);
    if (threadIdx.x == 0) *out = res;
}
__global__ __launch_bounds__(64) void k_fold_quad(const XYZZ29 *__restrict__ window_sums, unsigned nwin, unsigned c, Jac<Fq> *__restrict__ out) {
    fold_quad_body(window_sums, nwin, c, out);
}
__global__ __launch_bounds__(64) void k_fold_quad_g2(const XYZZ29x2 *__restrict__ window_sums, unsigned nwin, unsigned c, Jac<Fq2> *__restrict__ out) {
    fold_quad_body(window_sums, nwin, c, out);
}

// segmented calls: window j's sum (slot 2j of the last reduction level) -> result j
template <class A, class J>
__device__ __forceinline__ void emit_body(const A *__restrict__ window_sums, J *__restrict__ out) {
    if (threadIdx.x != 0) return;
    A r = window_sums[2 * (size_t)blockIdx.x];
    if constexpr (std::is_same<A, XYZZ29>::value) out[blockIdx.x] = xyzz29_to_jac(r);
    else out[blockIdx.x] = g2_to_jac(r);
}
__global__ __launch_bounds__(64) void k_emit_g1(const XYZZ29 *__restrict__ window_sums, Jac<Fq> *__restrict__ out) { emit_body(window_sums, out); }
__global__ __launch_bounds__(64) void k_emit_g2(const XYZZ29x2 *__restrict__ window_sums, Jac<Fq2> *__restrict__ out) { emit_body(window_sums, out); }

// ------------------------------------------------------------------------------------
// The upper reduction levels as bit trees (wide path, one bucket space).  After the first k_reduce2 level m = 4096
// (ACC, RUN) pairs are left, pair t covering 2^lm buckets:  total = sum_t ACC_t + 2^lm * sum_t t * RUN_t.  Three more
// 16-ary levels are three dependent chains of 9 + log_mult quad operations each (53 + 61 + 67 us of a lone call's
// 1.55 ms); written as  sum_t t * RUN_t = sum_j 2^j * T_j,  T_j = sum of the RUN_t whose index has bit j,  the twelve
// T_j and the sum of the ACC_t are thirteen INDEPENDENT tree sums (k_reduce_bits_a: eight wavefronts per 512 pairs,
// 7 + 3 quad additions), each weighted by its own lm + j doublings (k_reduce_bits_b: one wavefront per tree) and added
// up by one wavefront that also converts the result (k_reduce_bits_c): ~75 us.  Same quad arithmetic (quad29.h); 13 x
// the reads of the level it replaces, but that level is 4096 pairs: 832 short wavefronts against the 4096 of the level
// before it.
// ------------------------------------------------------------------------------------
template <class A>
__device__ __forceinline__ A bits_quad_tree(A a, unsigned lane, unsigned first_d) {
    const unsigned sub = lane & 3, qd = lane >> 2;
#pragma unroll 1
    for (unsigned d = first_d; d >= 1; d >>= 1) {
        A o = shfl_down_acc(a, 4 * d);
        if (qd + d >= 16) o = A::inf();
        a = quad_add(a, o, sub);
    }
    return a;
}
template <class C>
__global__ __launch_bounds__(512) void k_reduce_bits_a(const typename C::Acc *__restrict__ pairs, uint32_t m, uint32_t nbits, uint32_t G,
                                                       typename C::Acc *__restrict__ part) {
    using A = typename C::Acc;
    constexpr uint32_t AW = sizeof(A) / 4;
    __shared__ uint32_t wave_sum[8][AW];
    const uint32_t bit = blockIdx.x, g = blockIdx.y;
    const unsigned tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, sub = lane & 3, qd = lane >> 2;
    A a = A::inf();
#pragma unroll 1
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t t = g * 512 + wv * 64 + qd + 16 * j;
        A o = A::inf();
        if (t < m) {
            if (bit == nbits) o = pairs[2 * (size_t)t];                              // the ACC tree
            else if ((t >> bit) & 1u) o = pairs[2 * (size_t)t + 1];                  // RUN_t into T_bit
        }
        a = quad_add(a, o, sub);
    }
    a = bits_quad_tree(a, lane, 8);
    if (lane == 0) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&a);
#pragma unroll
        for (uint32_t w = 0; w < AW; w++) wave_sum[wv][w] = src[w];
    }
    __syncthreads();
    if (wv != 0) return;
    a = A::inf();
    if (qd < 8) a = *reinterpret_cast<const A *>(&wave_sum[qd][0]);
    a = bits_quad_tree(a, lane, 4);
    if (lane == 0) part[bit * G + g] = a;
}
// one wavefront per tree: the G <= 16 partials of tree `bit`, then its weight 2^(lm + bit) (the ACC tree: none)
template <class C>
__global__ __launch_bounds__(64) void k_reduce_bits_b(const typename C::Acc *__restrict__ part, uint32_t nbits, uint32_t G, uint32_t lm,
                                                      typename C::Acc *__restrict__ weighted) {
    using A = typename C::Acc;
    const uint32_t bit = blockIdx.x;
    const unsigned lane = threadIdx.x, sub = lane & 3, qd = lane >> 2;
    A a = A::inf();
    if (qd < G) a = part[bit * G + qd];
    a = bits_quad_tree(a, lane, 8);
    const uint32_t dbl = bit == nbits ? 0u : lm + bit;
#pragma unroll 1
    for (uint32_t i = 0; i < dbl; i++) a = quad_dbl(a, sub);
    if (lane == 0) weighted[bit] = a;
}
// the sum of the nbits + 1 <= 16 weighted trees, as libff's Jacobian point
template <class C>
__global__ __launch_bounds__(64) void k_reduce_bits_c(const typename C::Acc *__restrict__ weighted, uint32_t count, Jac<typename C::Field> *__restrict__ out) {
    using A = typename C::Acc;
    const unsigned lane = threadIdx.x, qd = lane >> 2;
    A a = A::inf();
    if (qd < count) a = weighted[qd];
    a = bits_quad_tree(a, lane, 8);
    if (lane == 0) *out = C::to_jac(a);
}

// publishes a slot's result(s) into the caller's buffer (a 96/192-byte hipMemcpyAsync costs ~30 us
// as a runtime copy kernel; this is one small workgroup)
__global__ __launch_bounds__(256) void k_publish(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, unsigned words) {
    for (unsigned i = threadIdx.x; i < words; i += 256) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------
#define HIPCHK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return LSA_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

struct Workspace {
    void *ptr = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        // LSA_TRACE=2: every growth of a workspace (a hipFree + hipMalloc pair inside somebody's call: 0.5 ms on one box, 10+ on another)
        static const bool noisy = getenv("LSA_TRACE") && getenv("LSA_TRACE")[0] == '2';
        if (noisy) fprintf(stderr, "[lsa]   workspace_grow              %zu -> %zu bytes\n", cap, bytes + bytes / 8);
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr; cap = 0;
        size_t want = bytes + bytes / 8;
        if (hipMalloc(&ptr, want) != hipSuccess) {
            if (hipMalloc(&ptr, bytes) != hipSuccess) { ptr = nullptr; return -1; }
            want = bytes;
        }
        cap = want;
        return 0;
    }
    void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; cap = 0; }
};

static Workspace g_ws;        // front phase (digits .. accumulate): reused by every call, ordered on the caller's stream
static Workspace g_prep_ws;   // prepare_bases staging

// Pipelining of back-to-back MSMs: the tail (bucket reduction + fold, 0.4-0.8 ms of pure latency
// on a nearly idle chip) runs on an internal stream, so it overlaps the next calls'
// throughput-bound fronts (sort + accumulate).  The buffers a tail reads (buckets, wave
// partials, window sums) belong to its slot and are guarded by events; results become ordered
// on the caller's stream again at msm_join() (every synchronous entry point and
// lsa_stream_join() do that).  LSA_NO_OVERLAP=1 runs everything on the caller's stream.
// NTAIL tail slots, each with its own stream and buffers, used round-robin: the tails of
// consecutive calls are independent, and for small MSMs (CPpoly's ladder) a tail is longer than
// a front, so several run at once.  Each tail computes its result into the slot and publishes
// it to the caller's buffer with a 96/192-byte copy that waits for the previous call's copy:
// results appear in call order even when a later tail finishes first.
static constexpr int NTAIL = 8;      // slots that exist; tail_slots() of them are used (LSA_TAIL_SLOTS)
struct TailBuf {
    Workspace ws;
    hipEvent_t done = nullptr;     // recorded after the slot's result has been published
    hipEvent_t front_done = nullptr;   // recorded on the caller's stream after the slot's accumulate stage
    hipStream_t stream = nullptr;
    bool pending = false;          // a tail has been issued on this slot
    bool unjoined = false;         // ... and the caller's stream has not waited for it yet
    void *aux = nullptr;           // 4 KiB of msm_compact_device's (zero between calls)
    const void *out = nullptr;     // where the slot's last tail writes its result
};
static TailBuf g_tail[NTAIL];
static unsigned g_slot = 0;
static unsigned tail_slots() {
    static const unsigned v = [] {
        const char *e = getenv("LSA_TAIL_SLOTS");
        const int want = e ? atoi(e) : NTAIL;
        return (unsigned)(want < 2 ? 2 : (want > NTAIL ? NTAIL : want));
    }();
    return v;
}
static int g_overlap = -1;

// (Measured and rejected: keeping the tail of call i back until call i+1 has issued its sort, so
// that it runs beside the next accumulate kernel instead of the next sort.  The first sort kernels
// are hit hard by a tail that starts beside them -- k_hist_wide 35 -> 110 us, k_tile_scan_rows
// 6 -> 52 us next to k_reduce1_lane's 1024 long single-wavefront workgroups -- but the tail's ~0.12 ms
// of multiplier work has to run somewhere: beside the accumulate kernel it stretches that one from
// 0.97 to 1.14 ms and itself to 1.8 ms, and the step stays at 1.61 ms either way.  Stream
// priorities and CU masks for the tail streams make every kernel slower: 1.95 - 3.7 ms per step.)
int msm_join(hipStream_t st) {
    for (auto &t : g_tail) {
        if (!t.unjoined) continue;
        if (hipStreamWaitEvent(st, t.done, 0) != hipSuccess) {
            set_error("msm_join: stream wait failed");
            return LSA_ERR_HIP;
        }
        t.unjoined = false;
    }
    return LSA_OK;
}
// `other` waits for every tail issued so far; the caller's own stream is left alone
int msm_join_to(hipStream_t other) {
    for (auto &t : g_tail) {
        if (!t.pending || !t.stream) continue;
        if (hipStreamWaitEvent(other, t.done, 0) != hipSuccess) {
            set_error("msm_join_to: stream wait failed");
            return LSA_ERR_HIP;
        }
    }
    return LSA_OK;
}
// Per-stage HIP events on the library stream.  A ring of EV_POOL call slots so that
// profiling never synchronises inside the timed loop; msm_profile_last() harvests.
static constexpr int EV_POOL = 64;
static constexpr int EV_MARKS = 8;   // 0..3 sort stage (caller's stream), 4..5 accumulate kernels (internal stream), 6..7 tail
static hipEvent_t g_ev[EV_POOL][EV_MARKS];
static bool g_ev_ready = false;
static bool g_profile = false;
static int g_ev_calls = 0;

void msm_release_workspace() {
    for (auto &t : g_tail) {
        if (t.stream) { (void)hipStreamSynchronize(t.stream); (void)hipStreamDestroy(t.stream); t.stream = nullptr; }
        t.ws.release();
        if (t.aux) (void)hipFree(t.aux);
        t.aux = nullptr; t.out = nullptr;
        if (t.done) (void)hipEventDestroy(t.done);
        if (t.front_done) (void)hipEventDestroy(t.front_done);
        t.done = nullptr; t.front_done = nullptr; t.pending = false; t.unjoined = false;
    }
    g_ws.release();
    g_prep_ws.release();
    if (g_ev_ready) { for (auto &row : g_ev) for (auto &e : row) (void)hipEventDestroy(e); g_ev_ready = false; }
}
void msm_profile_enable(bool on) { g_profile = on; g_ev_calls = 0; }
// Average per-stage milliseconds over the (up to EV_POOL) MSM calls recorded since
// profiling was enabled; returns the number of calls averaged.
int msm_profile_last(float ms[LSA_MSM_STAGES]) {
    for (int s = 0; s < LSA_MSM_STAGES; s++) ms[s] = 0.f;
    int cnt = g_ev_calls < EV_POOL ? g_ev_calls : EV_POOL;
    if (!g_ev_ready || cnt == 0) return 0;
    for (int c = 0; c < cnt; c++) {
        if (hipEventSynchronize(g_ev[c][EV_MARKS - 1]) != hipSuccess) return 0;
        // stage -> (from mark, to mark); stage 6 = hand-over from the sort to the accumulate kernel
        // (waiting for the previous call's accumulate + ordering the buckets by population)
        static const int span[7][2] = {{0, 1}, {1, 2}, {2, 3}, {4, 5}, {5, 6}, {6, 7}, {3, 4}};
        for (int s = 0; s < 7; s++) { float t = 0.f; (void)hipEventElapsedTime(&t, g_ev[c][span[s][0]], g_ev[c][span[s][1]]); ms[s] += t; }
        float t = 0.f; (void)hipEventElapsedTime(&t, g_ev[c][0], g_ev[c][EV_MARKS - 1]); ms[7] += t;
    }
    for (int s = 0; s < LSA_MSM_STAGES; s++) ms[s] /= (float)cnt;
    return cnt;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static int tail_slots_ready() {
    if (g_overlap < 0) g_overlap = getenv("LSA_NO_OVERLAP") ? 0 : 1;
    if (!g_tail[0].done) {
        for (auto &t : g_tail) {
            HIPCHK(hipEventCreateWithFlags(&t.done, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&t.front_done, hipEventDisableTiming));
            if (g_overlap) HIPCHK(hipStreamCreateWithFlags(&t.stream, hipStreamNonBlocking));
            HIPCHK(hipMalloc(&t.aux, 4096));
            HIPCHK(hipMemset(t.aux, 0, 4096));
        }
    }
    return LSA_OK;
}

static int msm_func_attrs() {
    static bool lds_attr_set = false;
    if (!lds_attr_set) {   // > 64 KiB of dynamic LDS needs an explicit opt-in
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter_wide<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter_wide<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter_wide<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_rank), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_hist_wide), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_partition), hipFuncAttributeMaxDynamicSharedMemorySize, 13 * PART_TILE * 4));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fine_sort_part), hipFuncAttributeMaxDynamicSharedMemorySize, 155648));
        lds_attr_set = true;
    }
    return LSA_OK;
}

// What the first MSM of a process used to pay inside its own call (17-20 ms whatever its size: the library's code object
// loaded on the first kernel launch, eight streams and their events, the function attributes, the first workspace
// allocations) is paid by lsa_init instead -- a prover's init_public_params(), not its first multiExpMA.  The workspaces
// are sized for the first G1 and G2 MSMs of 2^20 pairs: 400 MB of front workspace + 360 MB for tail slot 0 here, 34 + 202 MB of
// staging in capi.hip (warm_stage_buffers) -- about 1 GB of the 288 GB per process; LSA_WARM=0 skips all of it (everything
// is then allocated by the first call that needs it), LSA_WARM_MB sets the size for processes that share a GPU or never see
// 2^20 pairs (a verifier, a check program: LSA_WARM_MB=16).
int msm_warmup(hipStream_t st) {
    const char *w = getenv("LSA_WARM");
    if (w && w[0] == '0') return LSA_OK;
    int rc = tail_slots_ready();
    if (rc) return rc;
    rc = msm_func_attrs();
    if (rc) return rc;
    // Workspaces for the first calls of a process at the BASELINE scale (2^20 pairs, G1 and G2, plain layout and copies):
    // the front workspace and tail slot 0 as large as those calls need (377 MB / 343 MB, LSA_TRACE=2 prints every growth),
    // so that a prover's first multiExpMA does not start with hipFree + hipMalloc pairs -- 0.5 ms each on one box, 10 ms and
    // more on another (the first G2 MSM of a process: 11.6 ms on a good box with them, 26 ms of them on a slow one).
    // One tail slot: a caller that only ever blocks -- the reference's provers and verifiers -- never leaves slot 0; queued
    // callers grow the other slots on first use.  LSA_WARM_MB=m: front = m MB, tail = 0.9 m (0: nothing).
    const char *mb = getenv("LSA_WARM_MB");
    const size_t front = (mb ? (size_t)atoll(mb) : 400) << 20, tailb = front / 10 * 9;
    if (front) {
        // (no memory: the first call that needs the workspace will say so; LSA_TRACE reports it here)
        if (g_ws.ensure(front) != 0) { (void)hipGetLastError(); if (getenv("LSA_TRACE")) fprintf(stderr, "[lsa]   warm-up: front workspace of %zu MB not allocated\n", front >> 20); }
        if (g_tail[0].ws.ensure(tailb) != 0) { (void)hipGetLastError(); if (getenv("LSA_TRACE")) fprintf(stderr, "[lsa]   warm-up: tail workspace of %zu MB not allocated\n", tailb >> 20); }
    }
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, (const uint32_t *)nullptr, (uint32_t *)nullptr, 0u);      // loads the code object
    for (auto &t : g_tail) if (t.stream) hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, t.stream, (const uint32_t *)nullptr, (uint32_t *)nullptr, 0u);
    HIPCHK(hipDeviceSynchronize());
    return LSA_OK;
}

// ---- the slot interface other pipelines (msm_compact.hip) run their tails through
int msm_slot_begin(hipStream_t st, size_t ws_bytes, const void *d_out, MsmSlot *slot, bool inline_tail) {
    int rc = tail_slots_ready();
    if (rc) return rc;
    TailBuf &tb = g_tail[g_slot];
    if (ws_bytes > tb.ws.cap) {
        if (tb.pending) HIPCHK(hipEventSynchronize(tb.done));          // about to reallocate: the old tail must be finished
        if (tb.ws.ensure(ws_bytes) != 0) { set_error("msm: tail workspace allocation of %zu bytes failed", ws_bytes); return LSA_ERR_NOMEM; }
    }
    if (tb.pending) HIPCHK(hipStreamWaitEvent(st, tb.done, 0));        // the front may not overwrite what the slot's last tail still reads
    slot->tail = (g_overlap && !inline_tail) ? tb.stream : st;
    slot->ws = tb.ws.ptr;
    slot->aux = tb.aux;
    slot->index = (int)g_slot;
    tb.out = d_out;
    return LSA_OK;
}
int msm_slot_handover(MsmSlot *slot, hipStream_t st) {
    TailBuf &tb = g_tail[slot->index];
    if (slot->tail != st) {
        HIPCHK(hipEventRecord(tb.front_done, st));
        HIPCHK(hipStreamWaitEvent(slot->tail, tb.front_done, 0));
    }
    // results appear in call order wherever two calls write the same place: behind every earlier tail with this destination
    for (int i = 0; i < NTAIL; i++) {
        TailBuf &o = g_tail[i];
        if (i == slot->index || !o.pending || o.out != tb.out) continue;
        // (a tail that finished long ago needs no wait command: a blocking small call otherwise issues up to seven of them)
        if (!o.unjoined && hipEventQuery(o.done) == hipSuccess) { o.pending = false; continue; }
        HIPCHK(hipStreamWaitEvent(slot->tail, o.done, 0));
    }
    return LSA_OK;
}
int msm_slot_end(MsmSlot *slot, hipStream_t st) {
    TailBuf &tb = g_tail[slot->index];
    HIPCHK(hipEventRecord(tb.done, slot->tail));
    tb.pending = true;
    tb.unjoined = slot->tail != st;
    if (slot->tail != st || g_overlap == false) g_slot = (g_slot + 1) % tail_slots();      // (an inline tail = a blocking call: the slot is free again when it returns)
    return LSA_OK;
}

size_t msm_base_bytes(int group) { return group == 1 ? sizeof(CurveG1::Base) : sizeof(CurveG2::Base); }

// Jacobian (libff layout, device) -> device-resident bases of the curve's pipeline:
// batch-normalise to affine, then convert to the curve's base format.
template <class F>
int prepare_bases(const Jac<F> *d_in, void *d_out, size_t n, hipStream_t st) {
    using C = typename CurveOf<F>::type;
    if (n == 0) return LSA_OK;
    constexpr int K = 8;
    size_t threads = (n + K - 1) / K;
    unsigned blocks = (unsigned)((threads + 255) / 256);
    if (std::is_same<typename C::Base, Aff<F>>::value) {
        hipLaunchKernelGGL((k_normalize<F, K>), dim3(blocks), dim3(256), 0, st, d_in, (Aff<F> *)d_out, n);
        HIPCHK(hipGetLastError());
        return LSA_OK;
    }
    if constexpr (std::is_same<F, Fq>::value) {
        constexpr int KG = 16;                               // points per inversion (16 x 9 registers of prefix products)
        const unsigned gblocks = (unsigned)(((n + KG - 1) / KG + 255) / 256);
        hipLaunchKernelGGL((k_prepare_g1<KG>), dim3(gblocks), dim3(256), 0, st, d_in, (AffPacked *)d_out, n);
        HIPCHK(hipGetLastError());
        return LSA_OK;
    }
    if constexpr (std::is_same<F, Fq2>::value) {
        static const bool old_path = getenv("LSA_G2_PREPARE_OLD") != nullptr;      // A/B: k_normalize + k_convert_bases
        if (!old_path) {
            constexpr int KG = 4;
            const unsigned gblocks = (unsigned)(((n + KG - 1) / KG + 255) / 256);
            hipLaunchKernelGGL((k_prepare_g2<KG>), dim3(gblocks), dim3(256), 0, st, d_in, (AffPackedG2 *)d_out, n);
            HIPCHK(hipGetLastError());
            return LSA_OK;
        }
    }
    // staging buffer for the normalised affine points: grow-only, reused across calls
    // (every user is ordered on the same stream)
    if (g_prep_ws.ensure(n * sizeof(Aff<F>)) != 0) { set_error("prepare_bases: staging allocation failed"); return LSA_ERR_NOMEM; }
    Aff<F> *tmp = (Aff<F> *)g_prep_ws.ptr;
    hipLaunchKernelGGL((k_normalize<F, K>), dim3(blocks), dim3(256), 0, st, d_in, tmp, n);
    hipLaunchKernelGGL((k_convert_bases<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tmp, (typename C::Base *)d_out, n);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}
template int prepare_bases<Fq>(const Jac<Fq> *, void *, size_t, hipStream_t);
template int prepare_bases<Fq2>(const Jac<Fq2> *, void *, size_t, hipStream_t);

template <class F>
int normalize_to_affine(const Jac<F> *d_in, Aff<F> *d_out, size_t n, hipStream_t st) {
    if (n == 0) return LSA_OK;
    constexpr int K = 8;
    size_t threads = (n + K - 1) / K;
    unsigned blocks = (unsigned)((threads + 255) / 256);
    hipLaunchKernelGGL((k_normalize<F, K>), dim3(blocks), dim3(256), 0, st, d_in, d_out, n);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}
template int normalize_to_affine<Fq>(const Jac<Fq> *, Aff<Fq> *, size_t, hipStream_t);
template int normalize_to_affine<Fq2>(const Jac<Fq2> *, Aff<Fq2> *, size_t, hipStream_t);

// ------------------------------------------------------------------------------------
// Wide windows over pre-shifted bases.  A resident CRS can carry, next to P_i, the multiples
// 2^(pos_j) * P_i for 26 (24) bit positions pos_j (window-major: entry j*N + i; table_grid()).
// A digit that starts at pos_j gathers from copy j, so a digit d of ANY window is "add d * (its
// copy's point)": all windows share ONE bucket space in which bucket b has weight b+1.
// Consequences:
//   * the digit width is no longer tied to the per-window bucket count.  Large inputs
//     (n >= 2^16) use every other position: 13 (12) digits of 20/19 (22/21) bits per scalar instead
//     of 16 -- 13 mixed additions per pair -- and no GLV (its only gain, fewer windows to reduce
//     and fold, is moot), so no beta-multiplications either: 130 field products per pair instead
//     of 168.  Smaller inputs use every position: 26 digits, but only 512 (1024) buckets to reduce.
//   * one reduction over 2^(c-1) buckets instead of nwin reductions, and no Horner fold
//     (c*(nwin-1) sequential doublings: 0.3 ms G1 / 1.8 ms G2 of pure latency).
//   * several independent MSMs over prefixes of the same bases (CPPoly::prove's ladder,
//     /root/reference/src/gadgets/poly.h:77-86) run as ONE pass: segment j owns the bins
//     [j*B, (j+1)*B) of one sorted entry array (msm_segments_device).
// Price: 26 (24) x the base memory (G1 64 B, G2 128 B per point and copy) and one pass of 255
// doublings + 25 batch normalisations per key.
// ------------------------------------------------------------------------------------
static unsigned table_copies(size_t n_table) { return table_grid(n_table).ncopies; }
static size_t wide_big_min() {
    static const size_t v = getenv("LSA_WIDE_BIG_MIN") ? (size_t)atoll(getenv("LSA_WIDE_BIG_MIN")) : (size_t)1 << 16;
    return v;
}
static WidePlan wide_plan(size_t n_table, size_t n_call, unsigned nseg) {
    return wide_plan_for(n_table, nseg == 1 && n_call >= wide_big_min());
}
unsigned msm_table_windows(int /*group*/, size_t n) { return table_copies(n); }

static size_t g_merge_min = 0;
static bool g_merge_min_explicit = false;
// n != 0: vectors of at least n points get the copies and MSMs of at least n pairs use them;
// 0: defaults (copies from LSA_PRECOMPUTE_MIN / 2^19 points on, used by MSMs of every size)
void msm_set_merge_min(size_t n) { g_merge_min = n; g_merge_min_explicit = n != 0; }
size_t msm_merge_min() {
    if (g_merge_min == 0) {
        const char *e = getenv("LSA_PRECOMPUTE_MIN");
        g_merge_min = e && atoll(e) > 0 ? (size_t)atoll(e) : (size_t)1 << 19;
    }
    return g_merge_min;
}
bool msm_merge_min_is_explicit() { return g_merge_min_explicit; }
static size_t table_use_min() { return g_merge_min_explicit ? msm_merge_min() : 1; }
// whether an MSM of n pairs on a handle that carries the copies runs over them (the wide-window pipeline)
bool msm_uses_table(size_t n) { return n >= table_use_min(); }

template <class C>
__global__ __launch_bounds__(256) void k_shift_window(const typename C::Base *__restrict__ prev, Jac<typename C::Field> *__restrict__ out,
                                                      size_t n, unsigned c) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typename C::Acc p = C::madd(C::inf(), prev[i], false, false);   // infinity stays infinity
    for (unsigned j = 0; j < c; j++) p = C::dbl(p);
    out[i] = C::to_jac(p);
}

// d_table: msm_table_windows(group, n) * n entries, copy 0 (= the bases) already filled.
template <class F>
int precompute_windows(void *d_table, size_t n, hipStream_t st) {
    using C = typename CurveOf<F>::type;
    if (n == 0) return LSA_OK;
    const TableGrid grid = table_grid(n);
    Jac<F> *tmp = nullptr;
    if (hipMalloc(&tmp, n * sizeof(Jac<F>)) != hipSuccess) { set_error("precompute_windows: hipMalloc failed"); return LSA_ERR_NOMEM; }
    typename C::Base *tbl = (typename C::Base *)d_table;
    int rc = LSA_OK;
    for (unsigned k = 1; k < grid.ncopies && !rc; k++) {
        hipLaunchKernelGGL((k_shift_window<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tbl + (size_t)(k - 1) * n, tmp, n, grid.pos[k] - grid.pos[k - 1]);
        rc = prepare_bases<F>(tmp, tbl + (size_t)k * n, n, st);
    }
    hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    if (rc) return rc;
    if (e != hipSuccess) { set_error("precompute_windows: %s", hipGetErrorString(e)); return LSA_ERR_HIP; }
    return LSA_OK;
}
// One copy of the same: copy k (1 <= k < msm_table_windows) from copy k - 1, no allocation, no synchronisation; d_tmp
// holds n Jacobian points.  The background builder of the CRS cache issues the copies one at a time.
template <class F>
int precompute_window_step(void *d_table, size_t n, void *d_tmp, unsigned k, hipStream_t st) {
    using C = typename CurveOf<F>::type;
    const TableGrid grid = table_grid(n);
    if (n == 0 || k == 0 || k >= grid.ncopies) return LSA_OK;
    typename C::Base *tbl = (typename C::Base *)d_table;
    hipLaunchKernelGGL((k_shift_window<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tbl + (size_t)(k - 1) * n, (Jac<F> *)d_tmp, n, grid.pos[k] - grid.pos[k - 1]);
    return prepare_bases<F>((const Jac<F> *)d_tmp, tbl + (size_t)k * n, n, st);
}
template int precompute_window_step<Fq>(void *, size_t, void *, unsigned, hipStream_t);
template int precompute_window_step<Fq2>(void *, size_t, void *, unsigned, hipStream_t);
template int precompute_windows<Fq>(void *, size_t, hipStream_t);
template int precompute_windows<Fq2>(void *, size_t, hipStream_t);

// field multiplications per point-scalar pair in the accumulate kernel (bench.py's VALU roofline)
unsigned msm_field_mults_per_pair(size_t n, size_t table_n) {
    if (table_n && n >= table_use_min()) return wide_plan(table_n, n, 1).nwin * 10;   // mixed XYZZ addition: 8M + 2S
    return 16 * 10 + 8;                                                                // 16 additions + 8 beta-multiplications (GLV)
}

// The whole pipeline.  segs.nseg == 1: one MSM over n = segs.off[1] pairs.  segs.nseg > 1 (wide path
// only): segment j multiplies scalars[off[j] .. off[j+1]) with bases[first .. first + len_j) and
// d_out receives nseg points.
template <class F>
static int msm_pipeline(const void *d_bases_v, size_t first, const Fr *d_scalars, const SegList &segs, Jac<F> *d_out, hipStream_t st,
                        size_t table_stride, bool reuse_sort = false, bool blocking = false) {
    using C = typename CurveOf<F>::type;
    using A = typename C::Acc;
    const typename C::Base *d_bases = (const typename C::Base *)d_bases_v + first;
    const uint32_t nseg = segs.nseg;
    const size_t n = segs.off[nseg];
    if (n >= (size_t(1) << 27)) { set_error("msm: n too large (%zu)", n); return LSA_ERR_INVALID; }
    const bool wide = table_stride != 0 && (nseg > 1 || n >= table_use_min());
    if (nseg > 1 && !wide) { set_error("msm: segmented calls need bases with pre-shifted copies"); return LSA_ERR_INVALID; }
    WidePlan pl = {};
    if (wide) {
        pl = wide_plan(table_stride, n, nseg);
        if ((uint64_t)table_stride * table_copies(table_stride) >= (1u << 30)) { set_error("msm: base table too large for 30-bit entries"); return LSA_ERR_INVALID; }
    }
    const bool glv = C::GLV && !wide;
    const unsigned c = wide ? pl.c : msm_window_bits(glv ? 2 * n : n);       // plain path: sized by the virtual scalars
    const unsigned nwin = wide ? pl.nwin : (glv ? (128 + c - 1) / c : num_windows(c));   // |k1|,|k2| < 2^127 (glv.h)
    const size_t nv = glv ? 2 * n : n;                                       // virtual scalars
    const uint32_t B = 1u << (c - 1);
    const uint32_t nb = wide ? nseg * B : nwin * B;                          // wide: one bucket space per segment, shared by all windows
    if (wide && (uint64_t)nseg * B > (1u << 21)) { set_error("msm: %u segments of %u buckets exceed the bin space", nseg, B); return LSA_ERR_INVALID; }
    const bool fine = wide && nb > 32768;                                    // two-pass sort
    // 32-bit records (6 fine bits) when every point reference fits 25 bits, else 64-bit ones (7 fine bits)
    static const bool allow_rec32 = getenv("LSA_NO_REC32") == nullptr;
    const bool rec32 = fine && allow_rec32 && (uint64_t)table_stride * table_copies(table_stride) < (1u << 25) && (nb >> 6) <= 32768;
    // partitioned sort (k_partition / k_fine_sort_part): one large MSM whose tile count fits the second pass's LDS
    static const bool allow_part = getenv("LSA_NO_PART") == nullptr;
    const bool part = fine && allow_part && nseg == 1 && nwin <= 13 && n <= ((size_t)1 << 25) && (nb & (nb - 1)) == 0 && nb >= (1u << 19);
    const uint32_t fine_bits = rec32 ? 6u : WIDE_FINE_BITS;                  // (of the k_scatter_wide / k_fine_sort path)
    CoarseMap cm = {0u, 0u, fine ? fine_bits : 0u, 0u};
    uint32_t Bc = !wide ? B : (fine ? nb >> fine_bits : nb);                 // bins of the first sort pass
    if (part) {
        // 512 bins below B/2 and 256 above when some windows are a bit narrower than the widest
        // (they only reach the lower half of the buckets), 512 equal bins otherwise
        bool narrower = false;
        for (unsigned k = 0; k < nwin; k++) narrower |= pl.width[k] < c;
        unsigned lg = 0;
        while ((1u << lg) < B) lg++;
        if (narrower) { cm.half = B >> 1; cm.sh_lo = lg - 1 - 9; cm.sh_hi = lg - 1 - 8; cm.nlo = 512; Bc = 768; }
        else { cm.half = 0; cm.sh_lo = cm.sh_hi = lg - 9; cm.nlo = 0; Bc = 512; }
    }
    const size_t ne = nv * nwin;
    const bool big = wide && B > 4096;                                       // throughput-shaped reduction
    static const uint32_t wide_split = getenv("LSA_WIDE_SPLIT") ? (uint32_t)atoi(getenv("LSA_WIDE_SPLIT")) : 1u;
    const uint32_t split = wide ? (wide_split == 2 ? 2u : 1u) : 2u;        // lanes per bucket in k_accumulate
    // first reduction level: quads over L buckets (latency) or, for 2^19+ buckets, lanes over L buckets (throughput)
    // (a quad-shared first level over 2^19 buckets was measured too: 0.69 - 1.27 ms against 0.63 ms)
    // (G2, measured in round 6 with 32768 / 16384 / 8192 first-level lanes instead of 65536 -- longer lane-private chains, a half
    // to an eighth of the quad level's work behind them: pipelined 2^20-pair G2 MSMs 4.75-5.07 -> 4.80 / 5.41 / 6.78 ms.  65536 stays.)
    const uint32_t L = big ? std::max<uint32_t>(1, B / 65536) : (B > 4096 ? B / 4096 : 1);
    uint32_t logL = 0;
    while ((1u << logL) < L) logL++;
    const uint32_t T = B / L;                        // first-level segments per window
    const uint32_t wpw = big ? T : (T + 15) / 16;    // (ACC,RUN) pairs per window leaving level 1
    const uint32_t kw = wide ? nseg : nwin;          // bucket spaces ("windows") entering the reduction
    // Buckets far above the average population (skewed scalars; the partly filled top window)
    // are split across workgroups instead of being walked by their owner lanes.  With narrow
    // digits every bucket is long (26*n/512 entries), and the point of that path is latency:
    // anything above 4 entries is cut into chunks summed by a wavefront each.
    const uint32_t avg_pop = (uint32_t)(ne / nb + 1);
    // wide digits: the fullest buckets are those of the lower half, which every window reaches -- n / B
    // entries from each window of the widest width, twice that from each narrower one (36 at n = 2^20, 168
    // at n = 2^24); twice that expectation (6 sigma and more) separates them from skewed inputs, whose
    // long single-lane lists would otherwise bound the accumulate kernel (runs of equal scalars: 3.9 -> 3.2 ms)
    uint32_t pop_lo = avg_pop;
    if (wide && big) {
        unsigned nfull = 0;
        for (unsigned k = 0; k < nwin; k++) nfull += pl.width[k] == c;
        pop_lo = (uint32_t)(((uint64_t)n * (nfull + 2 * (nwin - nfull))) / B + 1);
    }
    const uint32_t heavy_threshold = (wide && !big) ? 4u : (wide ? std::max<uint32_t>(64, 2 * pop_lo + 32) : std::max<uint32_t>(64, 2 * avg_pop + 32));
    uint32_t bin_shift = 0;                          // populations above 1024 share bins (the order only balances wavefronts)
    while (((heavy_threshold - 1) >> bin_shift) + 1 > SIZE_BINS - 1) bin_shift++;
    const uint32_t max_heavy = (uint32_t)std::min<size_t>(nb, ne / heavy_threshold + 1);
    const size_t max_chunks = ne / HEAVY_CHUNK + max_heavy + 1;

    // workspace carve-up
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    size_t o_hist = carve((size_t)nb * 4 + 256);     // + heavy_count word
    size_t o_offs = carve((size_t)nb * 4);
    const uint32_t nscan = wide ? Bc : nb;           // counters the generic scan runs over
    const uint32_t scan_blocks = (nscan + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;   // <= 1024 since nscan <= 2^20
    size_t o_bsum = carve((size_t)scan_blocks * 4);
    const uint32_t ngroups = 1;
    size_t o_bins = carve((size_t)3 * ngroups * SIZE_BINS * 4);   // bin_count | bin_start | bin_cursor
    size_t o_perm = carve((size_t)nb * 4);
    const uint32_t ntiles = (uint32_t)((nv + SORT_TILE - 1) / SORT_TILE);
    const uint32_t wtile = part ? PART_TILE : wide_tile(nwin, n);
    const uint32_t wtiles = (uint32_t)((n + wtile - 1) / wtile);             // wide path: one row per tile (all windows)
    const size_t rows = wide ? wtiles : (size_t)nwin * ntiles;
    size_t o_digits = carve(wide ? 0 : ne * 4);
    size_t o_rank = carve(wide ? 0 : ne * 2);     // plain path only: the wide passes hand out positions with LDS atomics
    const uint32_t pitch = wide ? Bc + 96 : Bc;      // elements between rows of the tile arrays (see k_tile_scan_rows)
    size_t o_thist = carve(rows * pitch * 2);
    size_t o_tbase = carve(rows * pitch * 4);
    size_t o_entries = carve(ne * 4);
    size_t o_heavy = carve((size_t)max_heavy * 4);
    size_t o_choff = carve((size_t)(max_heavy + 2) * 4);    // chunk offsets, the chunk count, the chunk size
    size_t o_hpart = carve(max_chunks * sizeof(A));
    size_t o_recs = carve(fine ? ne * (part || rec32 ? 4 : 8) : 0);   // coarse-sorted records
    size_t o_chist = carve(fine ? (size_t)Bc * 4 : 0);
    size_t o_coffs = carve(fine ? (size_t)Bc * 4 : 0);
    if (g_ws.ensure(off) != 0) { set_error("msm: workspace allocation of %zu bytes failed", off); return LSA_ERR_NOMEM; }
    // tail buffers of this call parity
    { int rcs = tail_slots_ready(); if (rcs) return rcs; }
    // (Running the sort of call i+1 beside the accumulate of call i on a third stream was measured
    // and rejected: with enough hardware queues for real concurrency both kernels slow each other
    // down by more than the overlap gains -- 1.87 ms per step against 1.70 -- because the
    // accumulate kernel alone already fills every SIMD's issue slots and register file.)
    TailBuf &tb = g_tail[g_slot];
    TailBuf &prev = g_tail[(g_slot + tail_slots() - 1) % tail_slots()];
    hipStream_t tail = (g_overlap && !blocking) ? tb.stream : st;
    size_t toff = 0;
    auto tcarve = [&](size_t bytes) { size_t o = toff; toff = align_up(toff + bytes, 256); return o; };
    size_t o_buckets = tcarve((size_t)nb * split * sizeof(A));
    size_t o_wave = tcarve((size_t)kw * wpw * 2 * sizeof(A));
    size_t o_win = tcarve((size_t)kw * ((wpw + 15) / 16) * 2 * sizeof(A));   // reduction levels ping-pong between the two
    size_t o_res = tcarve((size_t)nseg * sizeof(Jac<F>));                      // this call's result(s) before they are published
    // A BLOCKING call grows only the slot it uses (round 5; before, all eight grew together: the first G2 MSM of a process --
    // the reference's provers issue a handful, each blocking -- paid nine hipFree + hipMalloc pairs, 4 ms on a good box
    // and tens of ms on a slow one).  Blocking calls keep re-using one slot (below), so a prover that only ever blocks
    // grows one slot per problem size; queued callers grow a slot the first time it sees the size.
    // A QUEUED call (the integrator's pipelined form) still grows every slot at once: its successors take the other slots
    // within microseconds, and a new problem size then pays its allocations in one call instead of in eight.
    for (auto &t : g_tail) {
        if (toff <= t.ws.cap || (blocking && &t != &tb)) continue;
        if (t.pending) HIPCHK(hipEventSynchronize(t.done));            // about to reallocate: the old tail must be finished
        if (t.ws.ensure(toff) != 0) { set_error("msm: tail workspace allocation of %zu bytes failed", toff); return LSA_ERR_NOMEM; }
    }
    if (tb.pending) HIPCHK(hipStreamWaitEvent(st, tb.done, 0));        // the front may not overwrite buckets a tail still reads
    tb.out = d_out;
    char *tws = (char *)tb.ws.ptr;
    char *ws = (char *)g_ws.ptr;
    uint32_t *hist = (uint32_t *)(ws + o_hist);
    uint32_t *heavy_count = hist + nb;
    uint32_t *offs = (uint32_t *)(ws + o_offs);
    uint32_t *bsum = (uint32_t *)(ws + o_bsum);
    uint32_t *bin_count = (uint32_t *)(ws + o_bins), *bin_start = bin_count + ngroups * SIZE_BINS, *bin_cursor = bin_start + ngroups * SIZE_BINS;
    uint32_t *perm = (uint32_t *)(ws + o_perm);
    int32_t *digits = (int32_t *)(ws + o_digits);
    uint16_t *rank = (uint16_t *)(ws + o_rank);
    uint16_t *tile_hist = (uint16_t *)(ws + o_thist);
    uint32_t *tile_base = (uint32_t *)(ws + o_tbase);
    uint32_t *entries = (uint32_t *)(ws + o_entries);
    A *buckets = (A *)(tws + o_buckets);
    uint32_t *heavy_list = (uint32_t *)(ws + o_heavy);
    uint32_t *chunk_off = (uint32_t *)(ws + o_choff);
    A *hpart = (A *)(ws + o_hpart);
    uint64_t *recs = (uint64_t *)(ws + o_recs);
    uint32_t *hist_c = fine ? (uint32_t *)(ws + o_chist) : hist;       // one pass: the bins ARE the buckets
    uint32_t *offs_c = fine ? (uint32_t *)(ws + o_coffs) : offs;
    A *wave_out = (A *)(tws + o_wave);
    A *window_sums = (A *)(tws + o_win);

    if (g_profile && !g_ev_ready) {
        for (auto &row : g_ev) for (auto &e : row) HIPCHK(hipEventCreate(&e));
        g_ev_ready = true;
    }
    int evi = 0;
    const int evslot = g_ev_calls % EV_POOL;
    auto mark = [&](hipStream_t s_) { if (g_profile) (void)hipEventRecord(g_ev[evslot][evi++], s_); };

    mark(st);  // 0
    { int rca = msm_func_attrs(); if (rca) return rca; }
    if (reuse_sort) {
        // second MSM of a commitment pair: the entries, populations and offsets of the call just
        // issued on this stream (same scalars, same digit plan, same table stride) are still in the
        // workspace -- the sort does not depend on the bases
        if (!wide) { set_error("msm: a shared sort needs bases with pre-shifted copies"); return LSA_ERR_INVALID; }
        mark(st); mark(st); mark(st);  // 1..3
    } else if (wide) {
        const uint32_t win_stride = (uint32_t)(table_stride * pl.copy_step);
        hipLaunchKernelGGL(k_hist_wide, dim3(wtiles), dim3(1024), (size_t)((Bc + 1) / 2) * 4, st, d_scalars, n, segs, pl, B, Bc, cm, pitch, wtile, tile_hist);
        mark(st);  // 1
        // (also clears the counters of the ordering stage: two memset launches less)
        hipLaunchKernelGGL(k_tile_scan_rows, dim3((Bc + 31) / 32), dim3(1024), 0, st, tile_hist, Bc, wtiles, pitch, tile_base, hist_c, heavy_count, bin_count,
                           (uint32_t)(3 * ngroups * SIZE_BINS));
        mark(st);  // 2
        if (part) {
            hipLaunchKernelGGL(k_partition, dim3(wtiles), dim3(1024), (size_t)nwin * PART_TILE * 4, st, d_scalars, n, segs, pl, B, Bc, cm, pitch,
                               tile_hist, hist_c, offs_c, tile_base, (uint32_t *)recs);
            const uint32_t fixed_words = 2 * (1u << cm.sh_hi) + wtiles + 1, budget_words = 155648 / 4;
            const uint32_t stage_cap = fixed_words + PART_STAGE <= budget_words ? PART_STAGE : 0u;   // larger problems have larger segments anyway
            hipLaunchKernelGGL(k_fine_sort_part, dim3(Bc), dim3(1024), (size_t)(fixed_words + stage_cap) * 4, st, (const uint32_t *)recs, offs_c, hist_c,
                               tile_base, pitch, wtiles, cm, segs, win_stride, stage_cap, entries, hist, offs);
        } else if (rec32) {
            hipLaunchKernelGGL(k_scatter_wide<2>, dim3(wtiles), dim3(1024), (size_t)Bc * 4, st, d_scalars, n, segs, pl, B, Bc, pitch, wtile, hist_c, offs_c, tile_base, (void *)recs, win_stride);
            hipLaunchKernelGGL(k_fine_sort<uint32_t>, dim3(Bc), dim3(1024), 0, st, (const uint32_t *)recs, offs_c, hist_c, entries, hist, offs);
        } else if (fine) {
            hipLaunchKernelGGL(k_scatter_wide<1>, dim3(wtiles), dim3(1024), (size_t)Bc * 4, st, d_scalars, n, segs, pl, B, Bc, pitch, wtile, hist_c, offs_c, tile_base, (void *)recs, win_stride);
            hipLaunchKernelGGL(k_fine_sort<uint64_t>, dim3(Bc), dim3(1024), 0, st, (const uint64_t *)recs, offs_c, hist_c, entries, hist, offs);
        } else {
            hipLaunchKernelGGL(k_scatter_wide<0>, dim3(wtiles), dim3(1024), (size_t)Bc * 4, st, d_scalars, n, segs, pl, B, Bc, pitch, wtile, hist_c, offs_c, tile_base, (void *)entries, win_stride);
        }
        mark(st);  // 3
    } else {
        hipLaunchKernelGGL((k_digits<C::GLV>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, n, c, nwin, digits);
        hipLaunchKernelGGL(k_rank, dim3(ntiles, nwin), dim3(1024), (size_t)B * 2, st, digits, nv, B, ntiles, rank, tile_hist, 0u);
        hipLaunchKernelGGL(k_tile_scan, dim3((nb + 255) / 256), dim3(256), 0, st, tile_hist, B, ntiles, nb, tile_base, hist);
        mark(st);  // 1
        hipLaunchKernelGGL(k_scan_sums, dim3(scan_blocks), dim3(256), 0, st, hist, nb, bsum);
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, st, bsum, scan_blocks);
        hipLaunchKernelGGL(k_scan_final, dim3(scan_blocks), dim3(256), 0, st, hist, bsum, nb, offs);
        mark(st);  // 2
        hipLaunchKernelGGL(k_scatter, dim3(ntiles, nwin), dim3(1024), (size_t)B * 4, st, digits, rank, offs, tile_base, nv, n, B, ntiles, entries, 0u);
        mark(st);  // 3
    }
    // ---- accumulate stage: bucket order, accumulation, heavy buckets
    if (!wide || reuse_sort) {         // (the wide path's k_tile_scan_rows has cleared them)
        HIPCHK(hipMemsetAsync(heavy_count, 0, 4, st));
        HIPCHK(hipMemsetAsync(bin_count, 0, (size_t)3 * ngroups * SIZE_BINS * 4, st));
    }
    {
        const unsigned sb = (nb + 2047) / 2048;
        const uint32_t gsz = 0u;
        hipLaunchKernelGGL((k_size_hist<C>), dim3(sb), dim3(256), 0, st, hist, nb, heavy_threshold, bin_count, gsz, bin_shift);
        hipLaunchKernelGGL(k_size_scan, dim3(1), dim3(256), 0, st, bin_count, bin_start, ngroups);
        hipLaunchKernelGGL((k_size_scatter<C>), dim3(sb), dim3(256), 0, st, hist, nb, heavy_threshold, bin_start, bin_cursor, perm, heavy_list, heavy_count, buckets, gsz, bin_shift, split);
    }
    mark(st);  // 4
    // (measured equal: 3.18 ms for the pair kernel -- 160 VGPRs, no scratch, three wavefronts per SIMD, ~18 % more
    // instructions per addition -- against 3.09 ms for the one-lane kernel at 2^20 pairs: both are instruction-issue
    // bound, the 316 B of scratch were never the cost.  LSA_G2_PAIR=1 selects the pair kernel.)
    static const bool g2_pair = getenv("LSA_G2_PAIR") && getenv("LSA_G2_PAIR")[0] == '1';
    if constexpr (std::is_same<C, CurveG2>::value) {
        if (split == 1 && g2_pair) {
            hipLaunchKernelGGL(k_accumulate_g2_pair, dim3((nb * 2 + 255) / 256), dim3(256), 0, st, d_bases, entries, offs, hist, perm, bin_start, buckets);
            goto acc_done;
        }
        if (split == 1) {       // (the generic one-lane kernel at full occupancy -- LSA_G2_OCC1 -- lost to this one in rounds 2-4; removed)
            hipLaunchKernelGGL(k_accumulate_g2_occ2, dim3((nb + 255) / 256), dim3(256), 0, st, d_bases, entries, offs, hist, perm, bin_start, buckets);
            goto acc_done;
        }
    }
    if (split == 1)
        hipLaunchKernelGGL((k_accumulate<C, 1u>), dim3((nb + 255) / 256), dim3(256), 0, st, d_bases, entries, offs, hist, perm, bin_start, buckets);
    else
        hipLaunchKernelGGL((k_accumulate<C, 2u>), dim3((nb * 2 + 255) / 256), dim3(256), 0, st, d_bases, entries, offs, hist, perm, bin_start, buckets);
acc_done:
    hipLaunchKernelGGL(k_heavy_plan, dim3(1), dim3(256), 0, st, hist, heavy_list, heavy_count, chunk_off);
    hipLaunchKernelGGL((k_accumulate_heavy<C>), dim3(4096), dim3(64), 0, st, d_bases, entries, offs, hist,
                       heavy_list, heavy_count, chunk_off, hpart);
    hipLaunchKernelGGL((k_heavy_finish<C>), dim3(256), dim3(64), 0, st, heavy_list, heavy_count, chunk_off, hpart, buckets, split);
    mark(st);  // 5
    if (tail != st) HIPCHK(hipEventRecord(tb.front_done, st));
    const bool profile = g_profile;
    hipEvent_t ev6 = profile ? g_ev[evslot][6] : nullptr, ev7 = profile ? g_ev[evslot][7] : nullptr;
    const size_t res_off = o_res;
    {
        if (tail != st) HIPCHK(hipStreamWaitEvent(tail, tb.front_done, 0));
        if (big)
            hipLaunchKernelGGL((k_reduce1_lane<C>), dim3(kw * ((T + 63) / 64)), dim3(64), 0, tail, buckets, B, L, split, wave_out);
        else
            hipLaunchKernelGGL((k_reduce1<C>), dim3(kw * wpw), dim3(64), 0, tail, buckets, B, L, logL, wpw, split, wave_out);
        A *lvl_in = wave_out, *lvl_out = window_sums;
        uint32_t m = wpw, lm = big ? logL : logL + 4;    // m pairs per window, each covering 2^lm buckets
        // one bucket space of 2^19 and more buckets: after the first 16-ary level the rest are bit trees (above)
        static const bool allow_bits = getenv("LSA_NO_REDUCE_BITS") == nullptr;
        // (for blocking calls only: the trees are ~45 us shorter in latency and ~1 % more work than the levels they replace,
        // which is the wrong trade for calls whose tails hide under the next call's front)
        const bool bits_tail = allow_bits && blocking && big && kw == 1 && nseg == 1 && m >= 4096 && (m / 16) <= 8192;
        bool converted = false;
        Jac<F> *res = tail != st ? (Jac<F> *)(tws + res_off) : d_out;
        // pipelined wide calls: the first 16-ary level lane-private (k_reduce2_lane: a seventh of the quad level's work)
        static const bool allow_lane_l1 = getenv("LSA_NO_LANE_L1") == nullptr;
        // (G1 only: 48 sequential G2 additions are 1.2 ms of latency; and only when calls are queued back to back: the
        // previous call's tail has not been joined by the caller since it was issued.  Decided from the CALL SEQUENCE alone
        // -- not from whether that tail happens to be running still -- so that the same calls always add in the same order
        // and return the same Jacobian bytes.  The second half of a commitment pair (reuse_sort) is waited for right away
        // and would pay the lane kernel's 0.2 ms of extra latency: 6.0 -> 6.25 ms.)
        // (G2, measured in round 6: lane-private, this level is 64 wavefronts at 256 VGPRs + 228 B of scratch and runs longer than
        // the step it should hide under -- pipelined 2^20-pair G2 MSMs 4.76 -> 5.03 ms.  LSA_G2_LANE_L1=1 selects it all the same.)
        static const bool g2_lane_l1 = getenv("LSA_G2_LANE_L1") != nullptr && getenv("LSA_G2_LANE_L1")[0] == '1';
        bool lane_l1 = allow_lane_l1 && (std::is_same<C, CurveG1>::value || g2_lane_l1) && !blocking && !reuse_sort && big && kw == 1 && nseg == 1 && m >= 16384 &&
                       tail != st && prev.pending && prev.unjoined && &prev != &tb;
        do {                                             // at least one k_reduce2 level (it leaves the sum in slot 0)
            const uint32_t m_out = (m + 15) / 16;
            if (lane_l1) {
                hipLaunchKernelGGL((k_reduce2_lane<C>), dim3((m_out + 63) / 64), dim3(64), 0, tail, lvl_in, m, m_out, lm, lvl_out);
                lane_l1 = false;
            } else
            hipLaunchKernelGGL((k_reduce2<C>), dim3(kw * m_out), dim3(64), 0, tail, lvl_in, m, m_out, lm, lvl_out);
            std::swap(lvl_in, lvl_out);
            m = m_out;
            lm += 4;
            if (bits_tail && m > 1) {
                uint32_t nbits = 0;
                while ((1u << nbits) < m) nbits++;        // indices 0 .. m - 1
                const uint32_t G = (m + 511) / 512;       // <= 16
                A *part = lvl_out;                        // (the level just consumed: (nbits + 1) * G + 16 values fit its m_in * 2)
                A *weighted = part + (size_t)(nbits + 1) * G;
                hipLaunchKernelGGL((k_reduce_bits_a<C>), dim3(nbits + 1, G), dim3(512), 0, tail, lvl_in, m, nbits, G, part);
                hipLaunchKernelGGL((k_reduce_bits_b<C>), dim3(nbits + 1), dim3(64), 0, tail, part, nbits, G, lm, weighted);
                if (profile) (void)hipEventRecord(ev6, tail);  // 6
                hipLaunchKernelGGL((k_reduce_bits_c<C>), dim3(1), dim3(64), 0, tail, weighted, nbits + 1, res);
                converted = true;
                break;
            }
        } while (m > 1);
        if (!converted && profile) (void)hipEventRecord(ev6, tail);  // 6
        // lvl_in[2*k] = sum of window k (pairs of (ACC,RUN): stride 2)
        if (converted) {
            // (the bit trees end in the Jacobian point)
        } else if (wide && nseg > 1) {   // every segment's bucket space is a finished sum: convert
            if constexpr (std::is_same<C, CurveG1>::value) hipLaunchKernelGGL(k_emit_g1, dim3(nseg), dim3(64), 0, tail, lvl_in, res);
            else hipLaunchKernelGGL(k_emit_g2, dim3(nseg), dim3(64), 0, tail, lvl_in, res);
        } else if constexpr (std::is_same<C, CurveG1>::value) {   // (wide, one segment: kw = 1, the fold only converts -- with a quad of lanes)
            hipLaunchKernelGGL(k_fold_quad, dim3(1), dim3(64), 0, tail, lvl_in, kw, c, res);
        } else {
            hipLaunchKernelGGL(k_fold_quad_g2, dim3(1), dim3(64), 0, tail, lvl_in, kw, c, res);
        }
        if (tail != st) {
            if (prev.pending && &prev != &tb) HIPCHK(hipStreamWaitEvent(tail, prev.done, 0));   // publish in call order
            // ... also behind every earlier tail that writes THIS destination and is not part of that chain: the compact
            // pipeline's tails (msm_compact.hip) write d_out themselves and only wait for tails with the same destination
            for (int i = 0; i < NTAIL; i++) {
                TailBuf &o = g_tail[i];
                if (&o != &tb && &o != &prev && o.pending && o.out == (const void *)d_out) HIPCHK(hipStreamWaitEvent(tail, o.done, 0));
            }
            hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, tail, (const uint32_t *)res, (uint32_t *)d_out, (unsigned)(nseg * sizeof(Jac<F>) / 4));
        }
        if (profile) (void)hipEventRecord(ev7, tail);  // 7
        HIPCHK(hipEventRecord(tb.done, tail));
    }
    tb.pending = true;
    tb.unjoined = (tail != st);
    // a blocking caller waits for this tail before it calls again: its next call can take the same slot (and its workspace)
    if (!blocking) g_slot = (g_slot + 1) % tail_slots();
    HIPCHK(hipGetLastError());
    if (g_profile) g_ev_calls++;
    return LSA_OK;
}

template <class F>
int msm_device(const void *d_bases_v, size_t first, const Fr *d_scalars, size_t n, Jac<F> *d_out, hipStream_t st, size_t table_stride, bool blocking) {
    if (n == 0) {
        int jr = msm_join(st);
        if (jr) return jr;
        Jac<F> inf = Jac<F>::inf();
        HIPCHK(hipMemcpyAsync(d_out, &inf, sizeof inf, hipMemcpyDefault, st));       // (d_out may be pinned host memory: blocking callers)
        HIPCHK(hipStreamSynchronize(st));
        return LSA_OK;
    }
    {
        // small and medium calls over a table: the four-launch pipeline of msm_compact.hip (LSA_NO_COMPACT=1: never;
        // LSA_NO_COMPACT_G2=1: G1 only, the round-4 state)
        static const bool allow_compact = getenv("LSA_NO_COMPACT") == nullptr;
        static const bool allow_g2 = getenv("LSA_NO_COMPACT_G2") == nullptr;
        const size_t cmax = std::is_same<F, Fq>::value ? msm_compact_max() : msm_compact_max_g2();
        if (allow_compact && (std::is_same<F, Fq>::value || allow_g2) && table_stride != 0 && n >= table_use_min() && n <= cmax)
            return msm_compact_device<F>(d_bases_v, first, d_scalars, n, d_out, st, table_stride, blocking);
    }
    SegList segs;
    segs.nseg = 1;
    segs.off[0] = 0;
    segs.off[1] = (uint32_t)n;
    return msm_pipeline<F>(d_bases_v, first, d_scalars, segs, d_out, st, table_stride, false, blocking);
}

// nseg independent MSMs in one pass: result j = sum_i scalars[off[j] + i] * bases[first + i],
// i < off[j+1] - off[j] (every segment multiplies a PREFIX of the same bases).  Needs the
// pre-shifted copies; at most MSM_MAX_SEGMENTS segments; empty segments give infinity.
template <class F>
int msm_segments_device(const void *d_bases_v, size_t first, const Fr *d_scalars, const uint64_t *seg_off, size_t nseg, Jac<F> *d_out,
                        hipStream_t st, size_t table_stride) {
    if (nseg == 0) return LSA_OK;
    if (nseg > MSM_MAX_SEGMENTS) { set_error("msm_segments: at most %u segments per call", MSM_MAX_SEGMENTS); return LSA_ERR_INVALID; }
    SegList segs;
    segs.nseg = (uint32_t)nseg;
    for (size_t j = 0; j <= nseg; j++) {
        if (seg_off[j] >= (uint64_t(1) << 27) || (j && seg_off[j] < seg_off[j - 1])) { set_error("msm_segments: bad offsets"); return LSA_ERR_INVALID; }
        segs.off[j] = (uint32_t)(seg_off[j] - seg_off[0]);
    }
    if (segs.off[nseg] == 0) {          // nothing but empty segments
        int jr = msm_join(st);
        if (jr) return jr;
        std::vector<Jac<F>> inf(nseg, Jac<F>::inf());
        HIPCHK(hipMemcpyAsync(d_out, inf.data(), nseg * sizeof(Jac<F>), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        return LSA_OK;
    }
    return msm_pipeline<F>(d_bases_v, first, d_scalars + seg_off[0], segs, d_out, st, table_stride);
}
// CommScheme::commit (/root/reference/src/prototools/commit.h:154-155): a G1 and a G2 MSM over the
// SAME scalar vector.  The sort (digits, ranks, scatter, fine sort: a third of a G1 call) depends on
// the scalars and the digit plan only, so it runs once: the G2 pipeline goes first (its workspace
// is the larger one), the G1 pipeline re-uses the sorted entries.  Both tables must have the same
// number of points (the same plan and copy stride); the caller checks that.
int msm_commit_pair_device(const void *d_g1_bases, const void *d_g2_bases, const Fr *d_scalars, size_t n, Jac<Fq> *d_out1,
                           Jac<Fq2> *d_out2, hipStream_t st, size_t table_stride) {
    SegList segs;
    segs.nseg = 1;
    segs.off[0] = 0;
    segs.off[1] = (uint32_t)n;
    int rc = msm_pipeline<Fq2>(d_g2_bases, 0, d_scalars, segs, d_out2, st, table_stride, false);
    if (rc) return rc;
    return msm_pipeline<Fq>(d_g1_bases, 0, d_scalars, segs, d_out1, st, table_stride, true);
}

template int msm_segments_device<Fq>(const void *, size_t, const Fr *, const uint64_t *, size_t, Jac<Fq> *, hipStream_t, size_t);
template int msm_segments_device<Fq2>(const void *, size_t, const Fr *, const uint64_t *, size_t, Jac<Fq2> *, hipStream_t, size_t);

template int msm_device<Fq>(const void *, size_t, const Fr *, size_t, Jac<Fq> *, hipStream_t, size_t, bool);
template int msm_device<Fq2>(const void *, size_t, const Fr *, size_t, Jac<Fq2> *, hipStream_t, size_t, bool);

}  // namespace lsa
