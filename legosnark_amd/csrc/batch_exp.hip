// legosnark_amd/csrc/batch_exp.hip -- fixed-base batch scalar multiplication on gfx950.
//
// Replaces libff get_window_table + batch_exp as used by cputil::simpleBatchExp
// (/root/reference/src/utils/util.h:119-134) and Interpolator::mkG1Exp/mkG2Exp
// (/root/reference/src/prototools/interp.h:36-59):  out[i] = scalars[i] * base.
//
//   1 powers   pw[j] = 2^(w*j) * base                       (one quad of lanes, nwin*w doublings)
//   2 table    T[j][d] = d * pw[j], d < 2^w                 (one lane per entry, w steps)
//   3 affine   batch-normalise + pack the table (prepare_bases, msm.hip) -> 64/128 B entries, L2-resident
//   4 main     one lane per scalar: Montgomery -> canonical, nwin table lookups,
//              XYZZ mixed adds on 29-bit limbs, Jacobian result (96/192 B) written coalesced.
// HBM traffic per scalar: 32 B in + 96 B (G1) / 192 B (G2) out; table reads hit L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "curves.h"
#include "msm.h"

namespace lsa {

// pw[j] = 2^(w*j) * base as XYZZ; the doubling chain is sequential, so the point is shared by
// a quad of lanes (quad29.h: 3 dependency levels per doubling instead of 9 products).
template <class C>
__global__ __launch_bounds__(64) void k_bexp_powers(Jac<typename C::Field> base, typename C::Acc *__restrict__ pw, unsigned w, unsigned nwin) {
    // (every quad of the wavefront runs the chain, lane 0 stores: four lanes alone run it 1.1-2.1x slower -- tools/ubench_exec_mask.hip)
    if (blockIdx.x != 0) return;
    const unsigned q = threadIdx.x & 3;
    typename C::Acc cur = C::from_jac(base);
    for (unsigned j = 0; j < nwin; j++) {
        if (threadIdx.x == 0) pw[j] = cur;
        for (unsigned i = 0; i < w; i++) cur = quad_dbl(cur, q);
    }
}

// T[j][d] = d * pw[j] (double-and-add, w steps), written as libff Jacobian for the
// batch normalisation.
template <class C>
__global__ __launch_bounds__(256) void k_bexp_table(const typename C::Acc *__restrict__ pw, Jac<typename C::Field> *__restrict__ tbl,
                                                    unsigned w, unsigned nwin) {
    unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned per = 1u << w;
    if (g >= nwin * per) return;
    unsigned j = g >> w, d = g & (per - 1);
    typename C::Acc p = pw[j];
    typename C::Acc r = C::inf();
    for (int b = (int)w - 1; b >= 0; b--) {
        r = C::dbl(r);
        if ((d >> b) & 1) r = C::add(r, p);
    }
    tbl[g] = C::to_jac(r);
}

template <class C>
__global__ __launch_bounds__(256) void k_bexp_main(const typename C::Base *__restrict__ tbl, const Fr *__restrict__ scalars, size_t n,
                                                   unsigned w, unsigned nwin, Jac<typename C::Field> *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s[8];
    scalars[i].to_canonical(s);
    typename C::Acc acc = C::inf();
    for (unsigned j = 0; j < nwin; j++) {
        unsigned bit = j * w;
        unsigned wd = bit >> 5, sh = bit & 31;
        uint64_t two = (uint64_t)(wd < 8 ? s[wd] : 0) | ((uint64_t)(wd + 1 < 8 ? s[wd + 1] : 0) << 32);
        uint32_t d = (uint32_t)(two >> sh) & ((1u << w) - 1);
        if (d) acc = C::madd(acc, tbl[((size_t)j << w) + d], false, false);
    }
    out[i] = C::to_jac(acc);
}

// A handful of scalars (cputil::simpleBatchExp with two of them, /root/reference/src/utils/util.h:119-134): one
// WAVEFRONT per scalar -- lane j fetches window j's table entry, and the <= 64 points are added up by a shuffle tree
// (6 general additions deep) instead of 32 dependent mixed additions on one lane.
template <class C>
__global__ __launch_bounds__(64) void k_bexp_small(const typename C::Base *__restrict__ tbl, const Fr *__restrict__ scalars, size_t n,
                                                   unsigned w, unsigned nwin, Jac<typename C::Field> *__restrict__ out) {
    using A = typename C::Acc;
    const size_t i = blockIdx.x;
    if (i >= n) return;
    const unsigned lane = threadIdx.x;
    uint32_t s[8];
    scalars[i].to_canonical(s);
    A acc = C::inf();
    if (lane < nwin) {
        const unsigned bit = lane * w, wd = bit >> 5, sh = bit & 31;
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) { lo = wd == (unsigned)q ? s[q] : lo; hi = wd + 1 == (unsigned)q ? s[q] : hi; }
        const uint64_t two = (uint64_t)lo | ((uint64_t)hi << 32);
        const uint32_t d = (uint32_t)(two >> sh) & ((1u << w) - 1);
        if (d) acc = C::madd(acc, tbl[((size_t)lane << w) + d], false, false);
    }
    constexpr int NW = sizeof(A) / 4;
#pragma unroll 1
    for (unsigned dlt = 32; dlt >= 1; dlt >>= 1) {
        A t;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&acc);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&t);
#pragma unroll
        for (int q = 0; q < NW; q++) dst[q] = __shfl_down(src[q], dlt, 64);
        if (lane + dlt < 64) acc = C::add(acc, t);
    }
    if (lane == 0) out[i] = C::to_jac(acc);
}

#define HIPCHK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return LSA_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

unsigned batch_exp_window_bits(size_t n) { return n >= (size_t(1) << 16) ? 12 : 8; }

static void *g_bexp_ws = nullptr;
static size_t g_bexp_cap = 0;
// libff builds a window table ONCE per base (get_window_table) and runs many batch_exp calls over it
// (/root/reference/src/prototools/interp.h:36-59: setupExp, then mkG1Exp / mkG2Exp per key vector).  The C-ABI takes the
// base with every call, so the finished device tables of the last few bases are kept, keyed by the base's bytes (exact
// comparison) and the window width: a call on a base seen before runs the main kernel only -- no chain of 254 dependent
// doublings (0.56 ms G1, 1.3 ms G2), no table, no normalisation.  LSA_BEXP_TABLES=n entries (default 4; 0: off).
struct BexpTable {
    int group = 0;
    unsigned w = 0;
    unsigned char base[192];
    void *d_tab = nullptr;
    size_t bytes = 0;
    uint64_t tick = 0;
};
static BexpTable g_bexp_tabs[8];
static uint64_t g_bexp_tick = 0;
static unsigned bexp_cache_size() {
    static const unsigned v = [] {
        const char *e = getenv("LSA_BEXP_TABLES");
        const int want = e ? atoi(e) : 4;
        return (unsigned)(want < 0 ? 0 : (want > 8 ? 8 : want));
    }();
    return v;
}
void batch_exp_release() {
    if (g_bexp_ws) (void)hipFree(g_bexp_ws);
    g_bexp_ws = nullptr; g_bexp_cap = 0;
    for (auto &t : g_bexp_tabs) { if (t.d_tab) (void)hipFree(t.d_tab); t = BexpTable(); }
}

// base: host value; d_scalars / d_out: device.  Asynchronous on `st` except for the
// temporary table, which is freed after a stream sync.
template <class F>
int batch_exp_device(const Jac<F> &base, const Fr *d_scalars, size_t n, Jac<F> *d_out, hipStream_t st) {
    using C = typename CurveOf<F>::type;
    if (n == 0) return LSA_OK;
    const unsigned w = batch_exp_window_bits(n);
    const unsigned nwin = (254 + w - 1) / w;
    const size_t entries = (size_t)nwin << w;
    constexpr int group = std::is_same<F, Fq>::value ? 1 : 2;
    // ---- a table of this base from an earlier call?
    BexpTable *hit = nullptr, *victim = nullptr;
    for (unsigned i = 0; i < bexp_cache_size(); i++) {
        BexpTable &t = g_bexp_tabs[i];
        if (t.d_tab && t.group == group && t.w == w && memcmp(t.base, &base, sizeof base) == 0) hit = &t;
        if (!victim || !t.d_tab || (victim->d_tab && t.tick < victim->tick)) victim = &t;
    }
    if (hit) {
        hit->tick = ++g_bexp_tick;
        if (n <= 512 && nwin <= 64)
            hipLaunchKernelGGL((k_bexp_small<C>), dim3((unsigned)n), dim3(64), 0, st, (const typename C::Base *)hit->d_tab, d_scalars, n, w, nwin, d_out);
        else
            hipLaunchKernelGGL((k_bexp_main<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const typename C::Base *)hit->d_tab, d_scalars, n, w, nwin, d_out);
        const hipError_t e2 = hipStreamSynchronize(st);
        if (e2 != hipSuccess) { set_error("batch_exp: %s", hipGetErrorString(e2)); return LSA_ERR_HIP; }
        return LSA_OK;
    }
    // powers | table (Jacobian) | table (packed affine): one grow-only workspace (three hipMalloc / hipFree pairs per call
    // cost more than the kernels of a short call; every user is ordered on the same stream)
    const size_t b_pw = (nwin * sizeof(typename C::Acc) + 255) & ~(size_t)255, b_tbl = (entries * sizeof(Jac<F>) + 255) & ~(size_t)255,
                 b_base = entries * sizeof(typename C::Base), need = b_pw + b_tbl + b_base;
    if (need > g_bexp_cap) {
        if (g_bexp_ws) { (void)hipStreamSynchronize(st); (void)hipFree(g_bexp_ws); }
        g_bexp_ws = nullptr; g_bexp_cap = 0;
        if (hipMalloc(&g_bexp_ws, need) != hipSuccess) { (void)hipGetLastError(); g_bexp_ws = nullptr; set_error("batch_exp: table allocation failed"); return LSA_ERR_NOMEM; }
        g_bexp_cap = need;
    }
    typename C::Acc *d_pw = (typename C::Acc *)g_bexp_ws;
    Jac<F> *d_tbl = (Jac<F> *)((char *)g_bexp_ws + b_pw);
    void *d_base = (char *)g_bexp_ws + b_pw + b_tbl;
    hipLaunchKernelGGL((k_bexp_powers<C>), dim3(1), dim3(64), 0, st, base, d_pw, w, nwin);
    hipLaunchKernelGGL((k_bexp_table<C>), dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, st, d_pw, d_tbl, w, nwin);
    int rc = prepare_bases<F>(d_tbl, d_base, entries, st);
    if (!rc) {
        if (n <= 512 && nwin <= 64)
            hipLaunchKernelGGL((k_bexp_small<C>), dim3((unsigned)n), dim3(64), 0, st, (const typename C::Base *)d_base, d_scalars, n, w, nwin, d_out);
        else
            hipLaunchKernelGGL((k_bexp_main<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const typename C::Base *)d_base, d_scalars, n, w, nwin, d_out);
    }
    hipError_t e = hipStreamSynchronize(st);      // (the entry points promise a finished result)
    if (!rc && e == hipSuccess && victim) {
        // keep the finished table for the next call on this base (its own allocation: the workspace is reused)
        if (victim->d_tab) { (void)hipFree(victim->d_tab); *victim = BexpTable(); }
        void *keep = nullptr;
        if (hipMalloc(&keep, b_base) == hipSuccess && hipMemcpyAsync(keep, d_base, b_base, hipMemcpyDeviceToDevice, st) == hipSuccess &&
            hipStreamSynchronize(st) == hipSuccess) {
            victim->group = group; victim->w = w; victim->d_tab = keep; victim->bytes = b_base; victim->tick = ++g_bexp_tick;
            memcpy(victim->base, &base, sizeof base);
        } else {
            (void)hipGetLastError();
            if (keep) (void)hipFree(keep);
        }
    }
    if (rc) return rc;
    if (e != hipSuccess) { set_error("batch_exp: %s", hipGetErrorString(e)); return LSA_ERR_HIP; }
    return LSA_OK;
}
template int batch_exp_device<Fq>(const Jac<Fq> &, const Fr *, size_t, Jac<Fq> *, hipStream_t);
template int batch_exp_device<Fq2>(const Jac<Fq2> &, const Fr *, size_t, Jac<Fq2> *, hipStream_t);

// ------------------------------------------------------------------------------------
// sum of n Jacobian points (device) -> one Jacobian point: used to fold the per-GPU MSM
// partials after the RCCL all-gather and small result vectors.
// ------------------------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(64) void k_sum_points(const Jac<F> *__restrict__ in, size_t n, Jac<F> *__restrict__ out) {
    unsigned lane = threadIdx.x;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (size_t i = lane; i < n; i += 64) acc = xyzz_add(acc, jac_to_xyzz(in[i]));
    for (unsigned d = 32; d >= 1; d >>= 1) {
        XYZZ<F> t;
        constexpr int NW = sizeof(XYZZ<F>) / 4;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&acc);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&t);
#pragma unroll
        for (int i = 0; i < NW; i++) dst[i] = __shfl_down(src[i], d, 64);
        if (lane + d < 64) acc = xyzz_add(acc, t);
    }
    if (lane == 0) *out = xyzz_to_jac(acc);
}

template <class F>
int sum_points_device(const Jac<F> *d_in, size_t n, Jac<F> *d_out, hipStream_t st) {
    hipLaunchKernelGGL((k_sum_points<F>), dim3(1), dim3(64), 0, st, d_in, n, d_out);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}
template int sum_points_device<Fq>(const Jac<Fq> *, size_t, Jac<Fq> *, hipStream_t);
template int sum_points_device<Fq2>(const Jac<Fq2> *, size_t, Jac<Fq2> *, hipStream_t);

}  // namespace lsa
