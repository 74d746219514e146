// legosnark_amd/csrc/msm_compact.hip -- small and medium MSMs over a resident base table in FOUR launches.
//
// The callers this is for: CPPoly::prove's ladder of 2d - 1 MSMs over prefixes 2^(d-1) .. 1 of its key vector
// (/root/reference/src/gadgets/poly.h:76-88), sparsemexp's handfuls (src/utils/sparsemexp.h:58,89) -- calls whose
// cost on the general pipeline (msm.hip: ~15 launches, three sort passes, population ordering, heavy-bucket
// plan, three reduction levels) is its dependency chain, not its work.  Same sum as multiExpMA
// (src/utils/globl.h:63-78), same digit plan as the wide path (msm_plan.h: every window gathers from its own
// pre-shifted copy of the bases, all windows share ONE bucket space of 512 (1024) signed-digit buckets):
//
//   1 k_cmp_sort        one workgroup per tile of 256 scalars: digits, a counting sort of the tile's 26 x 256 entries
//                       by bucket entirely in LDS (one returning LDS atomic per entry), the sorted run and its 513
//                       bucket offsets written out linearly; bucket totals by 512 global atomics per tile.
//   2 k_cmp_accumulate  work items = (bucket, chunk of `chunk` entries), laid out from the bucket totals by every
//                       workgroup for itself; one wavefront per item: each lane finds its entries through the
//                       per-tile offsets (a bucket's list is the concatenation of the tiles' runs), gathers the
//                       64-byte points, <= chunk/64 lane-private mixed additions, then the 64 partial sums go
//                       through LDS into 16 quads (quad29.h: one point per quad of lanes, a third of the latency
//                       of a lane-private addition) and a shuffle tree.  Skewed scalars only change the number of
//                       items of a bucket: no heavy-bucket path.
//   3 k_cmp_bits        sum_b (b+1) S_b = sum_j 2^j T_j with T_j = sum of the items whose weight has bit j: ten
//                       (eleven) independent tree sums, eight wavefronts per 512 items each; T_j's partial is
//                       doubled j times before it leaves (every wavefront does its own doublings side by side).
//   4 k_cmp_final       one wavefront sums the <= 64 pre-weighted partials, converts to libff's Jacobian form, and
//                       clears the bucket totals for the slot's next call.
//
// Chain of one call at n = 4096: ~9 (sort) + ~40 (2 mixed additions + 7 quad additions) + ~35 + ~25 us, against
// 0.26-0.31 ms on the general pipeline.  Kernel 1 runs on the caller's stream (it is the only reader of the scalars),
// kernels 2-4 on the call's tail slot (msm.h: msm_slot_*), so consecutive calls overlap like the general pipeline's.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

#include "curves.h"
#include "fs29.h"
#include "msm.h"
#include "msm_plan.h"

namespace lsa {

#define HIPCHK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return LSA_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

static constexpr uint32_t CMP_TILE = 256;          // scalars per sort workgroup
static constexpr uint32_t CMP_MAXB = 1024;         // buckets: 2^(c-1), c = 10 (11 for tables of 6*2^20 points and more)
static constexpr uint32_t CMP_MAXWIN = 26;
static constexpr uint32_t CMP_MAXTILES = 1024;     // n <= 262144
static constexpr uint32_t CMP_TILES_PER_LANE = CMP_MAXTILES / 64;
static constexpr uint32_t CMP_BITS_BLOCK = 512;    // items per workgroup of k_cmp_bits

// ------------------------------------------------------------------------------------ 1: tile sort
// entry = index of the scalar inside the call (20 bits) | window << 20 | sign << 31
__global__ __launch_bounds__(CMP_TILE) void k_cmp_sort(const Fr *__restrict__ scalars, uint32_t n, WidePlan pl, uint32_t B, uint32_t ent_stride,
                                                       uint32_t *__restrict__ ent, uint16_t *__restrict__ toff, uint32_t *__restrict__ ghist) {
    __shared__ uint32_t cnt[CMP_MAXB], off[CMP_MAXB + 1], wsum[4];
    __shared__ uint32_t code[CMP_MAXWIN * CMP_TILE];      // bucket | sign << 15 | rank inside (tile, bucket) << 16; ~0: no entry
    __shared__ uint32_t sorted[CMP_MAXWIN * CMP_TILE];
    const uint32_t t = blockIdx.x, tid = threadIdx.x, i = t * CMP_TILE + tid;
    for (uint32_t x = tid; x < B; x += CMP_TILE) cnt[x] = 0;
    __syncthreads();
    if (i < n) {
        wide_digits(scalars[i], pl, 0u, [&](unsigned k, int32_t sd) {
            uint32_t c = 0xffffffffu;
            if (sd != 0) {
                const uint32_t b = (uint32_t)(sd < 0 ? -sd : sd) - 1;
                const uint32_t r = atomicAdd(&cnt[b], 1u);                 // ds_add_rtn_u32
                c = b | (sd < 0 ? 0x8000u : 0u) | (r << 16);
            }
            code[k * CMP_TILE + tid] = c;
        });
    } else {
        for (uint32_t k = 0; k < pl.nwin; k++) code[k * CMP_TILE + tid] = 0xffffffffu;
    }
    __syncthreads();
    {   // exclusive prefix of the B counters: thread x owns B / 256 consecutive buckets
        const uint32_t per = B / CMP_TILE, b0 = tid * per;     // B is 512 or 1024
        uint32_t sum = 0;
        for (uint32_t q = 0; q < per; q++) sum += cnt[b0 + q];
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = __shfl_up(incl, d, 64);
            if ((int)(tid & 63) >= d) incl += u;
        }
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (uint32_t w = 0; w < (tid >> 6); w++) run += wsum[w];
        for (uint32_t q = 0; q < per; q++) {
            const uint32_t c = cnt[b0 + q];
            off[b0 + q] = run;
            if (c) atomicAdd(&ghist[b0 + q], c);
            run += c;
        }
        if (tid == CMP_TILE - 1) off[B] = run;
    }
    __syncthreads();
    for (uint32_t k = 0; k < pl.nwin; k++) {
        const uint32_t c = code[k * CMP_TILE + tid];
        if (c != 0xffffffffu) sorted[off[c & 0x7fffu] + (c >> 16)] = i | (k << 20) | ((c & 0x8000u) << 16);
    }
    __syncthreads();
    const uint32_t total = off[B];
    uint32_t *e = ent + (size_t)t * ent_stride;
    for (uint32_t x = tid; x < total; x += CMP_TILE) e[x] = sorted[x];
    uint16_t *o = toff + (size_t)t * (B + 1);
    for (uint32_t x = tid; x <= B; x += CMP_TILE) o[x] = (uint16_t)off[x];
}

// ------------------------------------------------------------------------------------ quad helpers
template <class A>
__device__ __forceinline__ A cmp_shfl_down(const A &p, unsigned delta) {
    A r;
    constexpr int NW = sizeof(A) / 4;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&p);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&r);
#pragma unroll
    for (int w = 0; w < NW; w++) dst[w] = __shfl_down(src[w], delta, 64);
    return r;
}
// the sum of the 16 quads' points, valid in quad 0 (four dependent quad additions; one call site)
template <class A>
__device__ __forceinline__ A cmp_quad_tree(A a, unsigned lane, unsigned first_d = 8) {
    const unsigned sub = lane & 3, qd = lane >> 2;
#pragma unroll 1
    for (unsigned d = first_d; d >= 1; d >>= 1) {
        A o = cmp_shfl_down(a, 4 * d);
        if (qd + d >= 16) o = A::inf();
        a = quad_add(a, o, sub);
    }
    return a;
}
template <class A>
__device__ __forceinline__ A cmp_load(const A *p) {
    A r;
    constexpr int NW = sizeof(A) / 4;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(p);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&r);
#pragma unroll
    for (int w = 0; w < NW; w++) dst[w] = src[w];
    return r;
}

// ------------------------------------------------------------------------------------ 2: items
// Work items from the bucket totals: bucket b has ceil(ghist[b] / chunk) of them; item (b, c) sums entries
// [c * chunk, min((c + 1) * chunk, ghist[b])) of bucket b's list = the tiles' runs in tile order.
template <class C>
__global__ __launch_bounds__(256) void k_cmp_accumulate(const typename C::Base *__restrict__ table, uint32_t win_stride, const uint32_t *__restrict__ ent,
                                                        uint32_t ent_stride, const uint16_t *__restrict__ toff, const uint32_t *__restrict__ ghist,
                                                        uint32_t B, uint32_t ntiles, uint32_t chunk, typename C::Acc *__restrict__ items,
                                                        uint32_t *__restrict__ item_w, uint32_t *__restrict__ nitems) {
    using A = typename C::Acc;
    constexpr uint32_t AW = sizeof(A) / 4, PITCH = AW + 1;          // (+1 word: the 64 lanes' rows start in different banks)
    __shared__ uint32_t istart[CMP_MAXB + 1], gcnt[CMP_MAXB], wsum[4];
    // per wavefront, one region used twice: first the exclusive prefix of its bucket's run lengths over the tiles (tpre) and
    // where the run starts inside each tile (tstart), then -- all entries fetched -- the 64 lane sums of the tree (a
    // wavefront's LDS operations execute in order, and the region is its own)
    constexpr uint32_t WREG = (64 * PITCH > CMP_MAXTILES + 1 + CMP_MAXTILES / 2 ? 64 * PITCH : CMP_MAXTILES + 1 + CMP_MAXTILES / 2);
    __shared__ uint32_t wave_region[4][WREG];
#define tpre_(w) (wave_region[w])
#define tstart_(w) (reinterpret_cast<uint16_t *>(wave_region[w] + CMP_MAXTILES + 1))
#define tree_(w) (wave_region[w])
    const unsigned tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    {
        const uint32_t per = B / 256, b0 = tid * per;
        uint32_t c[4], s[4], sum = 0;
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            c[q] = q < per ? ghist[b0 + q] : 0u;
            s[q] = (c[q] + chunk - 1) / chunk;
            sum += s[q];
        }
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
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            if (q < per) { istart[b0 + q] = run; gcnt[b0 + q] = c[q]; run += s[q]; }
        }
        if (tid == 255) { istart[B] = run; if (blockIdx.x == 0) *nitems = run; }
    }
    __syncthreads();
    const uint32_t W = istart[B], item = blockIdx.x * 4 + wv;
    const bool active = item < W;                                   // (wavefront-uniform)
    uint32_t b = 0, e0 = 0, e1 = 0;
    if (active) {
        uint32_t lo = 0, hi = B;                                    // the last b with istart[b] <= item
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (istart[mid] <= item) lo = mid; else hi = mid; }
        b = lo;
        e0 = (item - istart[b]) * chunk;
        e1 = gcnt[b] < e0 + chunk ? gcnt[b] : e0 + chunk;
        // run lengths of bucket b per tile: lane owns the tiles [TPL * lane, TPL * lane + TPL)
        constexpr uint32_t TPL = CMP_TILES_PER_LANE;
        uint32_t sum = 0;
#pragma unroll 4
        for (uint32_t q = 0; q < TPL; q++) {
            const uint32_t t = TPL * lane + q;
            uint32_t st = 0, len = 0;
            if (t < ntiles) {
                const uint16_t *o = toff + (size_t)t * (B + 1) + b;
                st = o[0];
                len = (uint32_t)o[1] - st;
            }
            tstart_(wv)[t] = (uint16_t)st;
            tpre_(wv)[t] = len;                        // (turned into the exclusive prefix below)
            sum += len;
        }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = __shfl_up(incl, d, 64);
            if ((int)lane >= d) incl += u;
        }
        uint32_t run = incl - sum;
        for (uint32_t q = 0; q < TPL; q++) {
            const uint32_t len = tpre_(wv)[TPL * lane + q];
            tpre_(wv)[TPL * lane + q] = run;
            run += len;
        }
        if (lane == 63) tpre_(wv)[CMP_MAXTILES] = run;
    }
    __syncthreads();
    A acc = C::inf();
    if (active) {
        // entry e of the bucket's list -> (tile t with tpre[t] <= e < tpre[t + 1]) -> the tile's sorted run
        auto fetch_entry = [&](uint32_t e) {
            uint32_t lo = 0, hi = CMP_MAXTILES;                     // the last t with tpre[t] <= e (empty runs repeat a value: the last one wins)
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (tpre_(wv)[mid] <= e) lo = mid; else hi = mid; }
            return ent[(size_t)lo * ent_stride + tstart_(wv)[lo] + (e - tpre_(wv)[lo])];
        };
        // one mixed-addition site; the next entry's two dependent loads (entry, then its 64-byte point) are in flight
        // while the current addition runs
        auto base_of = [&](uint32_t v) { return table[(size_t)((v >> 20) & 31u) * win_stride + (v & 0xfffffu)]; };
        uint32_t e = e0 + lane;
        if (e < e1) {
            uint32_t v = fetch_entry(e);
            typename C::Base cur = base_of(v);
#pragma unroll 1
            for (; e < e1; e += 64) {
                uint32_t vn = v;
                typename C::Base nxt = cur;
                if (e + 64 < e1) { vn = fetch_entry(e + 64); nxt = base_of(vn); }
                acc = C::madd(acc, cur, (v >> 31) != 0, false);
                v = vn;
                cur = nxt;
            }
        }
        uint32_t *row = &tree_(wv)[lane * PITCH];
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&acc);
#pragma unroll
        for (uint32_t w = 0; w < AW; w++) row[w] = src[w];
    }
    __syncthreads();
    if (!active) return;
    // 64 lane sums -> 16 quads of 4 -> one
    const unsigned sub = lane & 3, qd = lane >> 2;
    const uint32_t used = e1 - e0 < 64 ? e1 - e0 : 64;              // lanes that hold a sum
    A a = A::inf();
#pragma unroll 1
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t l = qd + 16 * j;                             // strided: with few sums the later trips add nothing anywhere
        A o = A::inf();
        if (l < used) o = cmp_load(reinterpret_cast<const A *>(&tree_(wv)[l * PITCH]));
        a = quad_add(a, o, sub);
    }
    a = cmp_quad_tree(a, lane);
    if (lane == 0) {
        items[item] = a;
        item_w[item] = b + 1;
    }
}

// ------------------------------------------------------------------------------------ 3: bit trees
template <class C>
__global__ __launch_bounds__(512) void k_cmp_bits(const typename C::Acc *__restrict__ items, const uint32_t *__restrict__ item_w,
                                                  const uint32_t *__restrict__ nitems, uint32_t G, typename C::Acc *__restrict__ part) {
    using A = typename C::Acc;
    constexpr uint32_t AW = sizeof(A) / 4;
    __shared__ uint32_t wave_sum[8][AW];
    const uint32_t bit = blockIdx.x, g = blockIdx.y, W = *nitems;
    const unsigned tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, sub = lane & 3, qd = lane >> 2;
    if (g * CMP_BITS_BLOCK >= W) {                                  // (block-uniform)
        if (tid == 0) part[bit * G + g] = A::inf();
        return;
    }
    A a = A::inf();
#pragma unroll 1
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t idx = g * CMP_BITS_BLOCK + wv * 64 + qd + 16 * j;
        A o = A::inf();
        if (idx < W && ((item_w[idx] >> bit) & 1u)) o = cmp_load(&items[idx]);
        a = quad_add(a, o, sub);
    }
    a = cmp_quad_tree(a, lane);
    if (lane == 0) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&a);
#pragma unroll
        for (uint32_t w = 0; w < AW; w++) wave_sum[wv][w] = src[w];
    }
    __syncthreads();
    if (wv != 0) return;
    a = A::inf();
    if (qd < 8) a = cmp_load(reinterpret_cast<const A *>(&wave_sum[qd][0]));
    a = cmp_quad_tree(a, lane, 4);
#pragma unroll 1
    for (uint32_t i = 0; i < bit; i++) a = quad_dbl(a, sub);        // weight 2^bit
    if (lane == 0) part[bit * G + g] = a;
}

// ------------------------------------------------------------------------------------ 4: final
template <class C>
__global__ __launch_bounds__(64) void k_cmp_final(const typename C::Acc *__restrict__ part, uint32_t nparts, uint32_t *__restrict__ ghist, uint32_t B,
                                                  Jac<typename C::Field> *__restrict__ out) {
    using A = typename C::Acc;
    const unsigned lane = threadIdx.x, sub = lane & 3, qd = lane >> 2;
    for (uint32_t x = lane; x < B; x += 64) ghist[x] = 0;           // the slot's next call counts from zero
    A a = A::inf();
#pragma unroll 1
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t idx = qd + 16 * j;
        A o = A::inf();
        if (idx < nparts) o = cmp_load(&part[idx]);
        a = quad_add(a, o, sub);
    }
    a = cmp_quad_tree(a, lane);
    if (lane == 0) *out = C::to_jac(a);
}

// ------------------------------------------------------------------------------------ host
static size_t cmp_max() {
    static const size_t v = getenv("LSA_COMPACT_MAX") ? (size_t)atoll(getenv("LSA_COMPACT_MAX")) : (size_t)1 << 17;   // 2^15: 0.31 ms against 0.45 on the general pipeline, 2^16: 0.44 / 0.67
    return v;
}
size_t msm_compact_max() { return cmp_max() < CMP_TILE * CMP_MAXTILES ? cmp_max() : CMP_TILE * CMP_MAXTILES; }
// G2: up to 2^16 pairs (from there on the wide digits of the general pipeline do half the additions)
size_t msm_compact_max_g2() {
    static const size_t v = getenv("LSA_COMPACT_MAX_G2") ? (size_t)atoll(getenv("LSA_COMPACT_MAX_G2")) : (size_t)1 << 16;
    return v < CMP_TILE * CMP_MAXTILES ? v : CMP_TILE * CMP_MAXTILES;
}

static inline size_t cmp_align(size_t x) { return (x + 255) & ~(size_t)255; }

template <class F>
int msm_compact_device(const void *d_table, size_t first, const Fr *d_scalars, size_t n, Jac<F> *d_out, hipStream_t st, size_t table_stride, bool blocking) {
    using C = typename CurveOf<F>::type;
    using A = typename C::Acc;
    if (n == 0 || n > (std::is_same<F, Fq>::value ? msm_compact_max() : msm_compact_max_g2()) || table_stride == 0) { set_error("msm_compact: not applicable (n = %zu)", n); return LSA_ERR_INVALID; }
    const WidePlan pl = wide_plan_for(table_stride, false);
    const uint32_t B = 1u << (pl.c - 1);
    if (B > CMP_MAXB || B < 256 || pl.nwin > CMP_MAXWIN) { set_error("msm_compact: digit plan out of range"); return LSA_ERR_INVALID; }
    const uint32_t ntiles = (uint32_t)((n + CMP_TILE - 1) / CMP_TILE);
    const size_t ne = (size_t)pl.nwin * n;
    // about 1024 items (one wavefront per SIMD) whatever n: chunk = entries per item, a multiple of 64
    uint32_t chunk = (uint32_t)(((ne + 1023) / 1024 + 63) / 64 * 64);
    if (chunk < 64) chunk = 64;
    const uint32_t wmax = (uint32_t)(ne / chunk) + B;                              // sum_b ceil(cnt_b / chunk) <= ne / chunk + B
    unsigned nbits = 0;
    while ((1u << nbits) <= B) nbits++;                                            // weights 1 .. B
    const uint32_t G = (wmax + CMP_BITS_BLOCK - 1) / CMP_BITS_BLOCK;
    if (nbits * G > 64) { set_error("msm_compact: internal (too many partial sums)"); return LSA_ERR_INVALID; }
    const uint32_t ent_stride = pl.nwin * CMP_TILE;

    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = cmp_align(off + bytes); return o; };
    const size_t o_ent = carve((size_t)ntiles * ent_stride * 4);
    const size_t o_toff = carve((size_t)ntiles * (B + 1) * 2);
    const size_t o_items = carve((size_t)wmax * sizeof(A));
    const size_t o_w = carve((size_t)wmax * 4);
    const size_t o_n = carve(4);
    const size_t o_part = carve((size_t)64 * sizeof(A));
    MsmSlot slot;
    int rc = msm_slot_begin(st, off, d_out, &slot, blocking);
    if (rc) return rc;
    char *ws = (char *)slot.ws;
    uint32_t *ghist = (uint32_t *)slot.aux;                     // zero between the calls of a slot (k_cmp_final clears it)
    uint32_t *ent = (uint32_t *)(ws + o_ent);
    uint16_t *toff = (uint16_t *)(ws + o_toff);
    A *items = (A *)(ws + o_items);
    uint32_t *item_w = (uint32_t *)(ws + o_w);
    uint32_t *nitems = (uint32_t *)(ws + o_n);
    A *part = (A *)(ws + o_part);
    const typename C::Base *table = (const typename C::Base *)d_table + first;
    const uint32_t win_stride = (uint32_t)(table_stride * pl.copy_step);

    hipLaunchKernelGGL(k_cmp_sort, dim3(ntiles), dim3(CMP_TILE), 0, st, d_scalars, (uint32_t)n, pl, B, ent_stride, ent, toff, ghist);
    rc = msm_slot_handover(&slot, st);
    if (rc) return rc;
    hipStream_t tail = slot.tail;
    hipLaunchKernelGGL((k_cmp_accumulate<C>), dim3((wmax + 3) / 4), dim3(256), 0, tail, table, win_stride, ent, ent_stride, toff, ghist, B, ntiles, chunk, items,
                       item_w, nitems);
    hipLaunchKernelGGL((k_cmp_bits<C>), dim3(nbits, G), dim3(512), 0, tail, items, item_w, nitems, G, part);
    hipLaunchKernelGGL((k_cmp_final<C>), dim3(1), dim3(64), 0, tail, part, nbits * G, ghist, B, d_out);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) {
        // a launch that did not happen may leave the slot's bucket totals behind: the slot's next call must count from zero
        (void)hipMemsetAsync(slot.aux, 0, CMP_MAXB * 4, tail);
        (void)msm_slot_end(&slot, st);
        set_error("msm_compact: kernel launch failed: %s", hipGetErrorString(e));
        return LSA_ERR_HIP;
    }
    return msm_slot_end(&slot, st);
}
template int msm_compact_device<Fq>(const void *, size_t, const Fr *, size_t, Jac<Fq> *, hipStream_t, size_t, bool);
// G2 (CommScheme::commit's second half, /root/reference/src/prototools/commit.h:155; InterpCommScheme::commit,
// src/gadgets/lipmaa.cc:27): the same four launches over 128-byte bases and 288-byte accumulators
template int msm_compact_device<Fq2>(const void *, size_t, const Fr *, size_t, Jac<Fq2> *, hipStream_t, size_t, bool);

// ------------------------------------------------------------------------------------ table builder
// All pre-shifted copies of a point in ONE kernel: lane i doubles P_i 255 times, parks the 25 (23) multiples it
// passes (XYZZ + the running product of their ZZZ) in scratch, inverts the product once (Montgomery's trick over the
// copies of ONE point: no cross-lane traffic) and writes the packed affine copies.  The 25-step builder of msm.hip
// (k_shift_window + k_prepare_g1 per copy: 25 Fermat inversions per point, 50 launches) takes 8.4 ms for 4096
// points and 24 ms for 2^20; this one ~1.2 ms up to 65536 points (latency: one wavefront per SIMD) and is also less
// work per point (one inversion instead of 25).
struct CmpBuildRec {
    XYZZ29 p;
    F29 pre;       // product of the ZZZ of the copies before this one
};
__global__ __launch_bounds__(64) void k_cmp_build_table(AffPacked *__restrict__ table, uint32_t n, uint32_t stride, TableGrid grid, CmpBuildRec *__restrict__ scratch) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const Aff29 p0 = unpack_affine(table[i]);
    if (p0.is_inf()) {
        AffPacked z;
#pragma unroll
        for (int w = 0; w < 8; w++) { z.x[w] = 0; z.y[w] = 0; }
        for (unsigned j = 1; j < grid.ncopies; j++) table[(size_t)j * stride + i] = z;
        return;
    }
    XYZZ29 cur = {p0.x, p0.y, F29::one(), F29::one()};
    F29 prod = F29::one();
#pragma unroll 1
    for (unsigned j = 1; j < grid.ncopies; j++) {
        const unsigned nd = grid.pos[j] - grid.pos[j - 1];
#pragma unroll 1
        for (unsigned d = 0; d < nd; d++) cur = xyzz29_dbl(cur);
        CmpBuildRec r;
        r.p = cur;
        r.pre = prod;
        scratch[(size_t)(j - 1) * n + i] = r;
        prod = mul(prod, cur.ZZZ);
    }
    F29 inv = Fs{prod}.inverse().v;
#pragma unroll 1
    for (unsigned j = grid.ncopies - 1; j >= 1; j--) {
        const CmpBuildRec r = scratch[(size_t)(j - 1) * n + i];
        const F29 i3 = mul(inv, r.pre);                 // 1 / ZZZ_j
        inv = mul(inv, r.p.ZZZ);
        const F29 izz = sqr(mul(r.p.ZZ, i3));           // ZZ^3 = ZZZ^2  =>  1 / ZZ = (ZZ / ZZZ)^2
        AffPacked o;
        mul(r.p.X, izz).canonical().pack256(o.x);
        mul(r.p.Y, i3).canonical().pack256(o.y);
        table[(size_t)j * stride + i] = o;
    }
}
// The same for G2: Fq2 doublings (fp29x2.h), the one inversion through the norm (1 / (a + b u) = (a - b u) / (a^2 + b^2)).
struct CmpBuildRecG2 {
    XYZZ29x2 p;
    F29x2 pre;
};
__global__ __launch_bounds__(64) void k_cmp_build_table_g2(AffPackedG2 *__restrict__ table, uint32_t n, uint32_t stride, TableGrid grid,
                                                           CmpBuildRecG2 *__restrict__ scratch) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const Aff29x2 p0 = unpack_affine(table[i]);
    if (p0.is_inf()) {
        AffPackedG2 z;
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int w = 0; w < 8; w++) z.w[c][w] = 0;
        for (unsigned j = 1; j < grid.ncopies; j++) table[(size_t)j * stride + i] = z;
        return;
    }
    XYZZ29x2 cur = {p0.x, p0.y, F29x2::one(), F29x2::one()};
    F29x2 prod = F29x2::one();
#pragma unroll 1
    for (unsigned j = 1; j < grid.ncopies; j++) {
        const unsigned nd = grid.pos[j] - grid.pos[j - 1];
#pragma unroll 1
        for (unsigned d = 0; d < nd; d++) cur = g2_dbl(cur);
        CmpBuildRecG2 r;
        r.p = cur;
        r.pre = prod;
        scratch[(size_t)(j - 1) * n + i] = r;
        prod = mul<2>(prod, cur.ZZZ);                      // [< 2]
    }
    F29x2 inv = f29x2_inverse(prod);                       // [< 2]
#pragma unroll 1
    for (unsigned j = grid.ncopies - 1; j >= 1; j--) {
        const CmpBuildRecG2 r = scratch[(size_t)(j - 1) * n + i];
        const F29x2 i3 = mul<2>(inv, r.pre);               // 1 / ZZZ_j
        inv = mul<2>(inv, r.p.ZZZ);
        const F29x2 izz = sqr<2>(mul<2>(r.p.ZZ, i3));      // ZZ^3 = ZZZ^2  =>  1 / ZZ = (ZZ / ZZZ)^2
        const F29x2 x = mul<2>(r.p.X, izz).canonical(), y = mul<2>(r.p.Y, i3).canonical();     // X, Y < 4: 2*4*2 = 16
        AffPackedG2 o;
        x.c0.pack256(o.w[0]);
        x.c1.pack256(o.w[1]);
        y.c0.pack256(o.w[2]);
        y.c1.pack256(o.w[3]);
        table[(size_t)j * stride + i] = o;
    }
}
size_t table_build_scratch_bytes(size_t n, int group) {
    return (size_t)(table_grid(n).ncopies - 1) * n * (group == 1 ? sizeof(CmpBuildRec) : sizeof(CmpBuildRecG2));
}
int table_build_g2_device(void *d_table, size_t n, size_t stride, void *d_scratch, hipStream_t st) {
    if (n == 0) return LSA_OK;
    hipLaunchKernelGGL(k_cmp_build_table_g2, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, (AffPackedG2 *)d_table, (uint32_t)n, (uint32_t)stride, table_grid(stride),
                       (CmpBuildRecG2 *)d_scratch);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}
// d_table: ncopies * stride packed points, copy 0 (the first n of them) filled; fills copies 1.. for points [0, n)
int table_build_g1_device(void *d_table, size_t n, size_t stride, void *d_scratch, hipStream_t st) {
    if (n == 0) return LSA_OK;
    hipLaunchKernelGGL(k_cmp_build_table, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, (AffPacked *)d_table, (uint32_t)n, (uint32_t)stride, table_grid(stride),
                       (CmpBuildRec *)d_scratch);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}

}  // namespace lsa
