// legosnark_amd/csrc/ntt.hip -- radix-2 number-theoretic transform over Fr on gfx950 in at most THREE global passes.
//
// Replaces libfqfft's basic_radix2_domain<Fr>::{FFT, iFFT, cosetFFT, icosetFFT}
// (_basic_radix2_FFT / _multiply_by_coset) as called by the Lipmaa Hadamard gadget,
// /root/reference/src/gadgets/lipmaa.cc:68-81,102-175:
//   FFT:   a[k] <- sum_i a[i] omega^(ik), natural order in and out; iFFT with omega^-1 and a
//   final factor 1/n; the coset variants scale a[i] by g^i before / by g^-i after.
// Fr values are canonical Montgomery residues, so any correct schedule gives libfqfft's bytes.
//
// Schedule (ntt_core.h): n = 2^(l1 + l2 + l3); every pass is one launch of k_ntt_pass, one workgroup per tile of 2^10
// elements held in LDS as 29-bit limbs (36 KB): pass 1 transforms 2^10 / 2^l1 neighbouring stride-m columns and applies
// the inter-pass twiddles, pass 2 the same inside each block of m, pass 3 the contiguous rows, and writes the result
// transposed to its natural position with the final scale.  Bit reversal happens on the way into LDS, coset shifts on
// the first load, 1/n and the inverse coset on the last store: no separate scale / bit-reversal launches.
//   n <= 2^10: one launch;  n <= 2^16: two;  n <= 2^28: three.   HBM traffic: 64 B per element and pass.
// Constants (the butterflies' twiddles W, the two-level tables of the inter-pass twiddles and of the coset powers) are
// computed once per domain (log n, omega, direction[, coset generator]) and cached (k_ntt_pow_table: one launch each).
// Work per element: (log n) / 2 butterfly products, less the trivial first stage of every pass, + 2 per inter-pass
// twiddle + 1 final product: 13.5 at n = 2^20 against 10 for an ideal radix-2 -- on fr29.h's product, 1.75x the rate of
// the 8 x 32-bit CIOS the previous version used.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "fp.h"
#include "msm.h"
#include "ntt_core.h"

namespace lsa {

#define HIPCHK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return LSA_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

__device__ __forceinline__ Fr fr_pow_dev(Fr base, uint64_t e) {
    Fr acc = Fr::one();
    bool started = false;
    for (int i = 63; i >= 0; --i) {
        if (started) acc = acc * acc;
        if ((e >> i) & 1) { acc = started ? acc * base : base; started = true; }
    }
    return acc;
}

// pass 2's twiddles: out[k2 n3 + i3] = c0 * base^(i3 k2), base = omega^n1
__global__ __launch_bounds__(256) void k_ntt_t2_table(Fr base, Fr c0, unsigned l2, unsigned l3, Fr *__restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >> (l2 + l3)) return;
    const uint64_t k2 = t >> l3, i3 = t & (((uint64_t)1 << l3) - 1);
    out[t] = c0 * fr_pow_dev(base, i3 * k2);
}

// pass 1's twiddles as one table: out[k1 m + col] = c0 * base^(col k1), m = 2^lm columns (each lane: one power by
// square-and-multiply, then a run of successive products with base^k1 along its row)
__global__ __launch_bounds__(256) void k_ntt_t1_table(Fr base, Fr c0, unsigned l1, unsigned lm, Fr *__restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t lo = t * 16;                                                       // 16 consecutive columns of one row (m >= 16: lm >= 4)
    if (lo >> (l1 + lm)) return;
    const uint64_t k1 = lo >> lm, col = lo & (((uint64_t)1 << lm) - 1);
    const Fr step = fr_pow_dev(base, k1);
    Fr x = c0 * fr_pow_dev(step, col);
    for (unsigned j = 0; j < 16; j++) { out[lo + j] = x; x = x * step; }
}

// out[k] = c0 * base^k, k < count (each lane: one power by square-and-multiply, then a run of successive products)
static constexpr unsigned NTT_RUN = 16;
__global__ __launch_bounds__(256) void k_ntt_pow_table(Fr base, Fr c0, size_t count, Fr *__restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t lo = t * NTT_RUN;
    if (lo >= count) return;
    Fr x = c0 * fr_pow_dev(base, lo);
    for (size_t k = lo; k < lo + NTT_RUN && k < count; k++) { out[k] = x; x = x * base; }
}

// 256-bit words -> limbs (the W table: read once per butterfly)
__global__ __launch_bounds__(256) void k_ntt_unpack_table(const Fr *__restrict__ in, size_t count, uint32_t *__restrict__ out9) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const Fr29 v = Fr29::from_words(in[t]);
#pragma unroll
    for (int i = 0; i < 9; i++) out9[t * 9 + i] = v.l[i];
}

// one pass over one tile (ntt_core.h); dynamic LDS: the tile's elements (9 words each: 36.9 KB, four workgroups per CU)
__global__ __launch_bounds__(256) void k_ntt_pass(NttArgs a) {
    extern __shared__ uint32_t lds[];
    const unsigned tid = threadIdx.x, w = blockIdx.x;
    const unsigned T = 1u << (a.pass.l + a.pass.logC);
    for (unsigned x = tid; x < T; x += 256) ntt_tile_load(a, w, x, lds);
    __syncthreads();
    // two stages per LDS round trip (ntt_tile_stage2); an odd number of stages starts with a single one
    unsigned s = 0;
    if (a.pass.l & 1u) {
        for (unsigned b = tid; b < T / 2; b += 256) ntt_tile_stage(a, 0, b, lds);
        __syncthreads();
        s = 1;
    }
#pragma unroll 1
    for (; s < a.pass.l; s += 2) {
        for (unsigned g = tid; g < T / 4; g += 256) ntt_tile_stage2(a, s, g, lds);
        __syncthreads();
    }
    for (unsigned x = tid; x < T; x += 256) ntt_tile_store(a, w, x, lds);
}

// (fp.h's operator* is a different function in the device pass: host code reaches it through __host__ __device__ helpers)
static __host__ __device__ Fr fr_mul_hd(const Fr &a, const Fr &b) { return a * b; }
static __host__ __device__ Fr host_pow(Fr base, uint64_t e) {
    Fr acc = Fr::one();
    for (int i = 63; i >= 0; --i) {
        acc = acc * acc;
        if ((e >> i) & 1) acc = acc * base;
    }
    return acc;
}

// ---- per-domain constants, cached on the device
namespace {
struct DomainTables {
    unsigned L = 0;
    Fr omega;                      // as the caller passed it (the key), with `inverse`
    bool inverse = false;
    void *mem = nullptr;
    Fr *Tlo = nullptr, *Thi = nullptr, *T2 = nullptr, *T1 = nullptr;
    uint32_t *W9 = nullptr;        // W as limbs
    uint64_t tick = 0;
    size_t bytes = 0;
};
struct CosetTables {
    unsigned L = 0;
    Fr g;
    bool inverse = false;
    void *mem = nullptr;
    Fr *Glo = nullptr, *Ghi = nullptr, *Gfull = nullptr;
    uint64_t tick = 0;
    size_t bytes = 0;
};
constexpr int NTT_CACHE = 8;          // a step domain keeps two sub-domains, each in both directions
DomainTables g_dom[NTT_CACHE];
CosetTables g_cos[NTT_CACHE];
uint64_t g_ntt_tick = 0;
// The tables are kept by BYTES as well as by count: a cached domain can be as large as its data (T1: 512 MB at 2^24), so eight
// of them -- a prover cycling through sizes, step domains in both directions -- could pin 4-5 GB.  Before a table is built the
// least recently used ones (of either kind, older than the transform in progress) go until the total fits LSA_NTT_CACHE_MB
// (default 2048).  hipFree waits for the device: a table a queued transform still reads is safe.
size_t g_ntt_bytes = 0;
size_t ntt_cache_budget() {
    static const size_t b = (size_t)(getenv("LSA_NTT_CACHE_MB") ? atol(getenv("LSA_NTT_CACHE_MB")) : 2048) << 20;
    return b;
}
template <class T> void ntt_drop(T &d) {
    if (d.mem) { (void)hipFree(d.mem); g_ntt_bytes -= d.bytes; }
    d = T();
}
void ntt_make_room(size_t need, uint64_t protect) {
    while (g_ntt_bytes + need > ntt_cache_budget()) {
        DomainTables *vd = nullptr;
        CosetTables *vc = nullptr;
        uint64_t oldest = ~0ull;
        for (auto &d : g_dom) if (d.mem && d.tick <= protect && d.tick < oldest) { oldest = d.tick; vd = &d; vc = nullptr; }
        for (auto &d : g_cos) if (d.mem && d.tick <= protect && d.tick < oldest) { oldest = d.tick; vc = &d; vd = nullptr; }
        if (vd) ntt_drop(*vd);
        else if (vc) ntt_drop(*vc);
        else return;                  // only this transform's own tables are left: they stay, whatever the budget
    }
}

int launch_pow_table(const Fr &base, const Fr &c0, size_t count, Fr *out, hipStream_t st) {
    const unsigned blocks = (unsigned)(((count + NTT_RUN - 1) / NTT_RUN + 255) / 256);
    hipLaunchKernelGGL(k_ntt_pow_table, dim3(blocks), dim3(256), 0, st, base, c0, count, out);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}
// a table about to be replaced may still be read by a transform queued on `st`: everything here is ordered on `st`,
// and hipFree waits for the device
int domain_tables(const NttPlan &p, const Fr &omega, bool inverse, hipStream_t st, uint64_t protect, DomainTables **out) {
    DomainTables *victim = &g_dom[0];
    for (auto &d : g_dom) {
        if (d.mem && d.L == p.L && d.inverse == inverse && memcmp(&d.omega, &omega, sizeof(Fr)) == 0) { d.tick = ++g_ntt_tick; *out = &d; return LSA_OK; }
        if (d.tick < victim->tick) victim = &d;
    }
    DomainTables &d = *victim;
    ntt_drop(d);
    const size_t nW = (size_t)1 << (p.lmax - 1), nlo = (size_t)1 << p.h, nhi = (size_t)1 << (p.L - p.h);
    const size_t n2t = p.l2 ? (size_t)1 << (p.l2 + p.l3) : 0;
    // pass 1's twiddles as one table of n entries (a product per element less) while n * 32 B stays within LSA_NTT_T1_MB
    // (default 512: up to 2^24 elements); beyond, and for rows shorter than a lane's run, the two-level look-up
    static const size_t t1_budget = (size_t)(getenv("LSA_NTT_T1_MB") ? atol(getenv("LSA_NTT_T1_MB")) : 512) << 20;
    const size_t n1t = (p.l1 && p.L - p.l1 >= 4 && (((size_t)1 << p.L) * sizeof(Fr)) <= t1_budget) ? (size_t)1 << p.L : 0;
    const size_t dbytes = (nW + nlo + nhi + n2t + n1t) * sizeof(Fr) + nW * 36;
    ntt_make_room(dbytes, protect);
    if (hipMalloc(&d.mem, dbytes) != hipSuccess) { (void)hipGetLastError(); d.mem = nullptr; set_error("fr_ntt: twiddle allocation failed"); return LSA_ERR_NOMEM; }
    d.bytes = dbytes;
    g_ntt_bytes += dbytes;
    Fr *Wp = (Fr *)d.mem;           // W as words: the source of W9
    d.Tlo = Wp + nW;
    d.Thi = d.Tlo + nlo;
    d.T2 = n2t ? d.Thi + nhi : nullptr;
    d.T1 = n1t ? d.Thi + nhi + n2t : nullptr;
    d.W9 = (uint32_t *)(d.Thi + nhi + n2t + n1t);
    d.L = p.L; d.omega = omega; d.inverse = inverse; d.tick = ++g_ntt_tick;
    const Fr w = inverse ? omega.inverse() : omega;
    const Fr c32 = Fr::from_u32(32);                       // Montgomery value * 32 = the canonical words of the 2^261 form
    int rc = launch_pow_table(host_pow(w, ((uint64_t)1 << p.L) >> p.lmax), c32, nW, Wp, st);
    if (!rc) {
        hipLaunchKernelGGL(k_ntt_unpack_table, dim3((unsigned)((nW + 255) / 256)), dim3(256), 0, st, (const Fr *)Wp, nW, d.W9);
        if (hipGetLastError() != hipSuccess) { set_error("fr_ntt: table kernel launch failed"); rc = LSA_ERR_HIP; }
    }
    if (!rc) rc = launch_pow_table(w, c32, nlo, d.Tlo, st);
    if (!rc) rc = launch_pow_table(host_pow(w, (uint64_t)1 << p.h), c32, nhi, d.Thi, st);
    if (!rc && n2t) {
        hipLaunchKernelGGL(k_ntt_t2_table, dim3((unsigned)((n2t + 255) / 256)), dim3(256), 0, st, host_pow(w, (uint64_t)1 << p.l1), c32, p.l2, p.l3, d.T2);
        if (hipGetLastError() != hipSuccess) { set_error("fr_ntt: table kernel launch failed"); rc = LSA_ERR_HIP; }
    }
    if (!rc && n1t) {
        hipLaunchKernelGGL(k_ntt_t1_table, dim3((unsigned)((n1t / 16 + 255) / 256)), dim3(256), 0, st, w, c32, p.l1, p.L - p.l1, d.T1);
        if (hipGetLastError() != hipSuccess) { set_error("fr_ntt: table kernel launch failed"); rc = LSA_ERR_HIP; }
    }
    if (rc) { ntt_drop(d); return rc; }
    *out = &d;
    return LSA_OK;
}
int coset_tables(const NttPlan &p, const Fr &g, bool inverse, hipStream_t st, uint64_t protect, CosetTables **out) {
    CosetTables *victim = &g_cos[0];
    for (auto &d : g_cos) {
        if (d.mem && d.L == p.L && d.inverse == inverse && memcmp(&d.g, &g, sizeof(Fr)) == 0) { d.tick = ++g_ntt_tick; *out = &d; return LSA_OK; }
        if (d.tick < victim->tick) victim = &d;
    }
    CosetTables &d = *victim;
    ntt_drop(d);
    const size_t nlo = (size_t)1 << p.h, nhi = (size_t)1 << (p.L - p.h);
    // the powers as one table of n entries (a product per element less) up to a quarter of the budget of pass 1's twiddles:
    // 128 MB, 2^22 elements -- at 2^24 the second 512-MB stream costs what the product saved (icosetFFT 2.52 -> 2.54 ms;
    // 2^20: 0.187 -> 0.172)
    static const size_t full_budget = ((size_t)(getenv("LSA_NTT_T1_MB") ? atol(getenv("LSA_NTT_T1_MB")) : 512) << 20) / 4;
    const size_t nfull = (p.L > NTT_TILE_LOG && (((size_t)1 << p.L) * sizeof(Fr)) <= full_budget) ? (size_t)1 << p.L : 0;
    const size_t cbytes = (nlo + nhi + nfull) * sizeof(Fr);
    ntt_make_room(cbytes, protect);
    if (hipMalloc(&d.mem, cbytes) != hipSuccess) { (void)hipGetLastError(); d.mem = nullptr; set_error("fr_ntt: coset table allocation failed"); return LSA_ERR_NOMEM; }
    d.bytes = cbytes;
    g_ntt_bytes += cbytes;
    d.Glo = (Fr *)d.mem;
    d.Ghi = d.Glo + nlo;
    d.Gfull = nfull ? d.Ghi + nhi : nullptr;
    d.L = p.L; d.g = g; d.inverse = inverse; d.tick = ++g_ntt_tick;
    const Fr c32 = Fr::from_u32(32);
    // forward: g^i on load; inverse: (1/n) g^-k on store (the 1/n rides in the high table)
    const Fr base = inverse ? g.inverse() : g;
    const Fr hi0 = inverse ? fr_mul_hd(c32, host_pow(Fr::from_u32(2), p.L).inverse()) : c32;
    int rc = launch_pow_table(base, c32, nlo, d.Glo, st);
    if (!rc) rc = launch_pow_table(host_pow(base, (uint64_t)1 << p.h), hi0, nhi, d.Ghi, st);
    if (!rc && nfull) rc = launch_pow_table(base, hi0, nfull, d.Gfull, st);
    if (rc) { ntt_drop(d); return rc; }
    *out = &d;
    return LSA_OK;
}
}  // namespace

void ntt_release() {
    for (auto &d : g_dom) ntt_drop(d);
    for (auto &d : g_cos) ntt_drop(d);
}

// d_a: 2^log_n elements, transformed in place (the result is in d_a when the call returns to the stream); d_tmp: scratch
// of the same size (untouched for log_n <= NTT_TILE_LOG).  omega: primitive 2^log_n-th root of unity.  coset: g or
// nullptr.  Asynchronous on st.
int fr_ntt_device(Fr *d_a, unsigned log_n, const Fr &omega, bool inverse, const Fr *coset, Fr *d_tmp, hipStream_t st) {
    if (log_n == 0) return LSA_OK;                        // the 1-point transform is the identity (1/n = 1, g^0 = 1)
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt_pass), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipGetLastError();
        return true;
    }();
    (void)attr_set;
    const NttPlan p = ntt_plan(log_n);
    DomainTables *dt = nullptr;
    const uint64_t protect = g_ntt_tick;                  // tables touched from here on belong to this transform
    int rc = domain_tables(p, omega, inverse, st, protect, &dt);
    if (rc) return rc;
    CosetTables *ct = nullptr;
    if (coset) {
        rc = coset_tables(p, *coset, inverse, st, protect, &ct);
        if (rc) return rc;
    }
    unsigned kinds[3], nk = 0;
    if (p.l1) kinds[nk++] = 1;
    if (p.l2) kinds[nk++] = 2;
    kinds[nk++] = 3;
    const Fr *src = d_a;
    for (unsigned i = 0; i < nk; i++) {
        NttArgs a = {};
        a.plan = p;
        a.pass = ntt_pass(p, kinds[i]);
        // first pass a -> tmp, the middle one in place on tmp, the last one tmp -> a; a single pass is one workgroup
        // working in place (it reads its whole tile before it writes)
        Fr *dst = (nk == 1 || i + 1 == nk) ? d_a : d_tmp;
        a.src = src;
        a.dst = dst;
        a.W = dt->W9; a.Tlo = dt->Tlo; a.Thi = dt->Thi; a.T2 = dt->T2; a.T1 = dt->T1;
        a.Glo = ct ? ct->Glo : nullptr;
        a.Ghi = ct ? ct->Ghi : nullptr;
        a.Gfull = ct ? ct->Gfull : nullptr;
        a.gh = p.h;
        a.pre_scale = (i == 0 && coset && !inverse) ? 1 : 0;
        a.post_scale = (i + 1 == nk && coset && inverse) ? 1 : 0;
        const Fr c32 = Fr::from_u32(32);
        a.cst = inverse ? fr_mul_hd(c32, host_pow(Fr::from_u32(2), p.L).inverse()) : c32;
        const size_t lds_bytes = (size_t)ntt_tile_words(a.pass) * 4;
        hipLaunchKernelGGL(k_ntt_pass, dim3(a.pass.tiles), dim3(256), lds_bytes, st, a);
        src = dst;
    }
    HIPCHK(hipGetLastError());
    return LSA_OK;
}

// ---------------------------------------------------------------- libfqfft's step radix-2 domain
// step_radix2_domain<Fr> (m = 2^b + 2^s, s < b: the domain get_evaluation_domain picks for every size that is not a
// power of two, /root/reference/src/prototools/interp.h:62, src/gadgets/lipmaa.cc:102): the evaluation points are
//   x_k = omega^(2k), k < 2^b   and   x_(2^b + j) = omega sigma^j, j < 2^s,
// omega a primitive 2^(b+1)-th root of unity, sigma = omega^(2^(b+1-s)).  FFT: a polynomial of degree < m reduced modulo
// x^big - 1 (c_i = a_i + a_(i+big)) goes through the big radix-2 transform with omega^2; reduced modulo x^big + 1
// (d_i = a_i - a_(i+big)), scaled by omega^i and wrapped modulo y^small - 1 (e_i = sum_j d_(i + j small)) through the small
// one with sigma.  iFFT undoes the two transforms and solves the 2 x 2 system for the two halves of the prefix.  The
// elementwise steps below run on fp.h's Fr (a handful of products per element against the transforms' 13);
// the two transforms are fr_ntt_device.  Grid-stride loops: lane t meets i = t, t + T, ... and steps its powers by base^T.
constexpr unsigned STEP_BLOCK = 256;
struct StepPow { Fr base, stride; };                      // base, base^T (T: the launch's lane count)

// The products of these kernels are data x constant: the constant in 2^261 form on fr29.h's limbs, the data word read as
// the shifted form (fr29.h), so the result's words are libff's -- a third of the instructions of the 8 x 32-bit product.
__device__ __forceinline__ Fr29 step_to261(const Fr &x) { return Fr29::from_words(x * Fr::from_u32(32)); }
__device__ __forceinline__ Fr step_mul(const Fr &x, const Fr29 &c) { return mul(Fr29::from_words(x), c).canonical2().to_words(); }
__device__ __forceinline__ Fr29 step_pow261(const Fr29 &base, uint64_t e) {          // base^e in 2^261 form (tight, < 2r)
    Fr29 acc = Fr29::one();
    bool started = false;
    for (int i = 63; i >= 0; --i) {
        if (started) acc = mul(acc, acc);
        if ((e >> i) & 1) { acc = started ? mul(acc, base) : base; started = true; }
    }
    return acc;
}

// forward, before the transforms: a[0 .. big) <- c, D[0 .. big) <- omega^i d_i (with the coset shift g^i folded in)
__global__ __launch_bounds__(256) void k_step_fwd_pre(Fr *__restrict__ a, Fr *__restrict__ D, size_t big, size_t small, StepPow w, StepPow g,
                                                      Fr g_big, int coset) {
    const size_t T = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= big) return;
    const Fr29 ws = step_to261(w.stride), gs = step_to261(g.stride), gb = step_to261(g_big);
    Fr29 wi = step_pow261(step_to261(w.base), t), gi = coset ? step_pow261(step_to261(g.base), t) : Fr29::one();
    for (size_t i = t; i < big; i += T) {
        Fr x0 = a[i];
        if (coset) x0 = step_mul(x0, gi);
        Fr c = x0, d = x0;
        if (i < small) {
            Fr x1 = a[i + big];
            if (coset) x1 = step_mul(x1, mul(gi, gb));
            c = x0 + x1;
            d = x0 - x1;
        }
        a[i] = c;
        D[i] = step_mul(d, wi);
        wi = mul(wi, ws);
        if (coset) gi = mul(gi, gs);
    }
}
// D[k] <- sum_(j < R) D[k + j q], k < q: R-fold wrap of a vector of R q elements
__global__ __launch_bounds__(256) void k_step_wrap(Fr *__restrict__ D, size_t q, unsigned R) {
    const size_t T = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < q; k += T) {
        Fr s = D[k];
        for (unsigned j = 1; j < R; j++) s = s + D[k + (size_t)j * q];
        D[k] = s;
    }
}
// inverse, after the transforms: D[i] <- omega^i U0[i] for small <= i < big, 0 below (the j = 0 term is not part of the sum)
__global__ __launch_bounds__(256) void k_step_inv_mid(const Fr *__restrict__ a, Fr *__restrict__ D, size_t big, size_t small, StepPow w) {
    const size_t T = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= big) return;
    const Fr29 ws = step_to261(w.stride);
    Fr29 wi = step_pow261(step_to261(w.base), t);
    for (size_t i = t; i < big; i += T) {
        D[i] = i < small ? Fr::zero() : step_mul(a[i], wi);
        wi = mul(wi, ws);
    }
}
// a[i], a[big + i] <- (U0[i] +- omega^-i (U1[i] - S[i])) / 2 for i < small; the inverse coset shift on every entry
__global__ __launch_bounds__(256) void k_step_inv_post(Fr *__restrict__ a, const Fr *__restrict__ S, size_t big, size_t small, StepPow wi_, StepPow gi_,
                                                       Fr gi_big, Fr half, int coset) {
    const size_t T = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= big) return;
    const Fr29 ws = step_to261(wi_.stride), gs = step_to261(gi_.stride), gb = step_to261(gi_big), h = step_to261(half);
    Fr29 wi = step_pow261(step_to261(wi_.base), t), gi = coset ? step_pow261(step_to261(gi_.base), t) : Fr29::one();
    for (size_t i = t; i < big; i += T) {
        if (i < small) {
            const Fr u0 = a[i], u1 = step_mul(a[big + i] - S[i], wi);
            // with the coset shift the halving and the shift are one constant each
            a[i] = step_mul(u0 + u1, coset ? mul(h, gi) : h);
            a[big + i] = step_mul(u0 - u1, coset ? mul(h, mul(gi, gb)) : h);
        } else if (coset) {
            a[i] = step_mul(a[i], gi);
        }
        wi = mul(wi, ws);
        if (coset) gi = mul(gi, gs);
    }
}

namespace {
unsigned step_blocks(size_t n) {
    // eight elements per lane where there are enough: a lane's first power costs ~30 products (square-and-multiply), the
    // later ones one each
    const size_t b = (n + 8 * STEP_BLOCK - 1) / (8 * STEP_BLOCK);
    return (unsigned)(b < 1 ? 1 : b > 2048 ? 2048 : b);          // at most 2048 workgroups of 256: eight per CU
}
StepPow step_pow(const Fr &base, unsigned blocks) { return {base, host_pow(base, (uint64_t)blocks * STEP_BLOCK)}; }
int step_wrap(Fr *D, size_t from, size_t to, hipStream_t st) {
    while (from > to) {
        const unsigned R = (unsigned)((from / to) < 8 ? (from / to) : 8);
        const size_t q = from / R;
        hipLaunchKernelGGL(k_step_wrap, dim3(step_blocks(q)), dim3(STEP_BLOCK), 0, st, D, q, R);
        from = q;
    }
    HIPCHK(hipGetLastError());
    return LSA_OK;
}
}  // namespace

// d_a: 2^big_log + 2^small_log elements, transformed in place; d_scratch: 2^big_log elements.  omega: primitive
// 2^(big_log + 1)-th root of unity (step_radix2_domain::omega).  Asynchronous on st.
int fr_ntt_step_device(Fr *d_a, unsigned big_log, unsigned small_log, const Fr &omega, bool inverse, const Fr *coset, Fr *d_scratch, hipStream_t st) {
    const size_t big = (size_t)1 << big_log, small = (size_t)1 << small_log;
    const Fr big_omega = fr_mul_hd(omega, omega), small_omega = host_pow(omega, (uint64_t)1 << (big_log + 1 - small_log));
    const unsigned blocks = step_blocks(big);
    Fr *D = d_scratch;
    int rc;
    if (!inverse) {
        const Fr g = coset ? *coset : Fr::one();
        hipLaunchKernelGGL(k_step_fwd_pre, dim3(blocks), dim3(STEP_BLOCK), 0, st, d_a, D, big, small, step_pow(omega, blocks), step_pow(g, blocks),
                           host_pow(g, big), coset ? 1 : 0);
        HIPCHK(hipGetLastError());
        rc = step_wrap(D, big, small, st);
        if (rc) return rc;
        // e lives in D[0 .. small), its transform's scratch right behind it (2 small <= big); then D is free for the big one
        rc = fr_ntt_device(D, small_log, small_omega, false, nullptr, D + small, st);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(d_a + big, D, small * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        return fr_ntt_device(d_a, big_log, big_omega, false, nullptr, D, st);
    }
    rc = fr_ntt_device(d_a, big_log, big_omega, true, nullptr, D, st);
    if (rc) return rc;
    rc = fr_ntt_device(d_a + big, small_log, small_omega, true, nullptr, D, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_step_inv_mid, dim3(blocks), dim3(STEP_BLOCK), 0, st, (const Fr *)d_a, D, big, small, step_pow(omega, blocks));
    HIPCHK(hipGetLastError());
    rc = step_wrap(D, big, small, st);
    if (rc) return rc;
    const Fr omega_inv = omega.inverse(), g_inv = coset ? coset->inverse() : Fr::one();
    hipLaunchKernelGGL(k_step_inv_post, dim3(blocks), dim3(STEP_BLOCK), 0, st, d_a, (const Fr *)D, big, small, step_pow(omega_inv, blocks),
                       step_pow(g_inv, blocks), host_pow(g_inv, big), Fr::from_u32(2).inverse(), coset ? 1 : 0);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}

}  // namespace lsa
