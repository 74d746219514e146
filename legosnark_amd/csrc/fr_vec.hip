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
//                  also the whole of MultiVPolyT::evalMLE
//                  (/root/reference/src/prototools/polytools.h:207-234): the reference builds the
//                  table of eq-monomials and takes a dot product; folding the top variable d
//                  times gives the same element of Fr, and Fr values are canonical Montgomery
//                  residues, so the 32 output bytes are identical.
// Both are HBM streams.  Round 5: ONE product per output element instead of two -- v' = v0 + r (v1 - v0) is the same
// field element as -v0 (r - 1) + v1 r, and Fr values are canonical, so the bytes are the reference's -- on the 29-bit
// limbs of fr29.h (the round's r is put into 2^261 form once per thread, so data words go in and out without
// conversion: a fifth of the instructions of two 32-bit CIOS products), and the rounds of at most FOLD_TAIL output
// elements run inside ONE workgroup (a barrier between rounds instead of a launch each).  Algorithmic bytes per round
// over m output elements: pairs 64 B in + 64 B out, halves 64 B in + 32 B out.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "fp.h"
#include "fr29.h"
#include "msm.h"

namespace lsa {

static constexpr size_t FOLD_TAIL = 2048;          // rounds of at most this many output elements: one workgroup, no launches

// the round's scalar in 2^261 form (Montgomery value * 32: the canonical words of r * 2^261), as limbs
__device__ __forceinline__ Fr29 fr_to_261(const Fr &r) { return Fr29::from_words(r * Fr::from_u32(32)); }
// x * r as libff words: x's words are read as the 2^261 form of x / 32, the product with r in 2^261 form is x r 2^256
__device__ __forceinline__ Fr fr_mul_261(const Fr &x, const Fr29 &r261) { return mul(Fr29::from_words(x), r261).canonical2().to_words(); }

__global__ __launch_bounds__(256) void k_fold_pairs(const Fr *__restrict__ v, size_t m, const Fr *__restrict__ r_ptr,
                                                    Fr *__restrict__ w, Fr *__restrict__ vout) {
    const Fr29 r261 = fr_to_261(*r_ptr);
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < m; p += (size_t)gridDim.x * blockDim.x) {
        const Fr a = v[2 * p], b = v[2 * p + 1];
        const Fr d = b - a;
        w[p] = d;
        vout[p] = a + fr_mul_261(d, r261);               // = -a (r - 1) + b r
    }
}
// R consecutive rounds in ONE pass: round j + 1 pairs up neighbouring outputs of round j, so a lane that owns 2^R
// neighbouring inputs runs all R rounds in registers -- the intermediate vectors never reach memory (3 rounds: 16
// elements moved per 8 inputs instead of 28) and R rounds cost one launch.  m: outputs of the LAST of the R rounds;
// w0: where the first round's witness coefficients go, the later rounds' follow (2^(R-1) m, 2^(R-2) m, ... entries).
template <int R>
__global__ __launch_bounds__(256) void k_fold_pairs_multi(const Fr *__restrict__ v, size_t m, const Fr *__restrict__ r_ptr, Fr *__restrict__ w0,
                                                          Fr *__restrict__ vout) {
    constexpr int IN = 1 << R;
    Fr29 r261[R];
#pragma unroll
    for (int j = 0; j < R; j++) r261[j] = fr_to_261(r_ptr[j]);
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < m; t += (size_t)gridDim.x * blockDim.x) {
        Fr x[IN];
#pragma unroll
        for (int i = 0; i < IN; i++) x[i] = v[(size_t)IN * t + i];
        Fr *w = w0;
        size_t outs = m << (R - 1);                      // outputs of the current round
#pragma unroll
        for (int j = 0; j < R; j++) {
            const int cnt = IN >> (j + 1);               // this lane's outputs in round j
#pragma unroll
            for (int i = 0; i < cnt; i++) {
                const Fr d = x[2 * i + 1] - x[2 * i];
                w[(size_t)cnt * t + i] = d;
                x[i] = x[2 * i] + fr_mul_261(d, r261[j]);
            }
            w += outs;
            outs >>= 1;
        }
        vout[t] = x[0];
    }
}
// the remaining rounds (m_first, m_first / 2, ... 1 output elements) by one workgroup; round j reads what round j - 1
// wrote (ping-pong between bufA and bufB, continuing the caller's parity), r_ptr / w advance per round
__global__ __launch_bounds__(256) void k_fold_pairs_tail(const Fr *src, size_t m_first, const Fr *__restrict__ r_ptr, Fr *w, Fr *bufA, Fr *bufB,
                                                         unsigned parity) {
    for (size_t m = m_first; m >= 1; m >>= 1) {
        const Fr29 r261 = fr_to_261(*r_ptr);
        Fr *dst = (parity & 1u) ? bufB : bufA;
        for (size_t p = threadIdx.x; p < m; p += 256) {
            const Fr a = src[2 * p], b = src[2 * p + 1];
            const Fr d = b - a;
            w[p] = d;
            dst[p] = a + fr_mul_261(d, r261);
        }
        __syncthreads();                                 // the block's global writes are visible to the block
        src = dst;
        w += m;
        r_ptr++;
        parity++;
    }
}

// cur may alias old: lane p reads old[p], old[p+half] and writes cur[p] only
__global__ __launch_bounds__(256) void k_fold_halves(const Fr *old, size_t half, const Fr *__restrict__ r_ptr, Fr *cur) {
    const Fr29 r261 = fr_to_261(*r_ptr);
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < half; p += (size_t)gridDim.x * blockDim.x) {
        const Fr a = old[p];
        cur[p] = a + fr_mul_261(old[p + half] - a, r261);   // = a (1 - r) + old[p + half] r
    }
}
// R consecutive rounds of the top-variable fold in ONE pass: with q outputs after the R rounds, lane p < q owns the 2^R
// inputs old[p + j q]; round 1 pairs (j, j + 2^(R-1)) with r_hi[0], round 2 (j, j + 2^(R-2)) with r_hi[-1], ... -- the
// intermediate vectors never reach memory (evalMLE reads its table ONCE).  cur must not alias old (lanes read q apart).
template <int R>
__global__ __launch_bounds__(256) void k_fold_halves_multi(const Fr *__restrict__ old, size_t q, const Fr *__restrict__ r_hi, Fr *__restrict__ cur) {
    constexpr int IN = 1 << R;
    Fr29 r261[R];
#pragma unroll
    for (int j = 0; j < R; j++) r261[j] = fr_to_261(*(r_hi - j));
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < q; p += (size_t)gridDim.x * blockDim.x) {
        Fr x[IN];
#pragma unroll
        for (int i = 0; i < IN; i++) x[i] = old[p + (size_t)i * q];
#pragma unroll
        for (int j = 0; j < R; j++) {
            const int cnt = IN >> (j + 1);
#pragma unroll
            for (int i = 0; i < cnt; i++) x[i] = x[i] + fr_mul_261(x[i + cnt] - x[i], r261[j]);
        }
        cur[p] = x[0];
    }
}
// evalMLE's last rounds (half = half_first, half_first / 2, ... 1; the scalar of the round with half = 2^i is r[i]) by one
// workgroup, in place on buf; the last round writes *out
__global__ __launch_bounds__(256) void k_fold_halves_tail(const Fr *src, size_t half_first, const Fr *__restrict__ r, Fr *buf, Fr *out) {
    unsigned i = 0;
    while (((size_t)1 << i) < half_first) i++;
    for (size_t half = half_first; half >= 1; half >>= 1, i--) {
        const Fr29 r261 = fr_to_261(r[i]);
        Fr *dst = half == 1 ? out : buf;
        for (size_t p = threadIdx.x; p < half; p += 256) {
            const Fr a = src[p];
            dst[p] = a + fr_mul_261(src[p + half] - a, r261);
        }
        __syncthreads();
        src = buf;
        if (half == 1) break;
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

__global__ __launch_bounds__(256) void k_sumcheck_partial(const Fr *__restrict__ suff, ScTables tabs, unsigned m, size_t half,
                                                         Fr *__restrict__ partial) {
    __shared__ Fr lds[256];
    Fr c[SC_MAX_M + 1];
    for (unsigned i = 0; i <= m; i++) c[i] = Fr::zero();
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < half; p += (size_t)gridDim.x * blockDim.x) {
        Fr q[SC_MAX_M + 1];
        q[0] = suff ? suff[p] : Fr::one();
        unsigned deg = 0;
        for (unsigned t = 0; t < m; t++) {
            const Fr v0 = tabs.t[t][p], dv = tabs.t[t][p + half] - v0;
            q[deg + 1] = q[deg] * dv;                    // q <- q * (v0 + dv X)
            for (unsigned i = deg; i >= 1; i--) q[i] = q[i] * v0 + q[i - 1] * dv;
            q[0] = q[0] * v0;
            deg++;
        }
        for (unsigned i = 0; i <= m; i++) c[i] = c[i] + q[i];
    }
    for (unsigned i = 0; i <= m; i++) {
        lds[threadIdx.x] = c[i];
        __syncthreads();
        for (unsigned s = 128; s >= 1; s >>= 1) {
            if (threadIdx.x < s) lds[threadIdx.x] = lds[threadIdx.x] + lds[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[(size_t)blockIdx.x * (SC_MAX_M + 1) + i] = lds[0];
        __syncthreads();
    }
}

// out[0..m(+1)] = (have_beta ? ((1-rho) + (2rho-1) X) * pre : 1) * sum of the block partials
__global__ __launch_bounds__(64) void k_sumcheck_finish(const Fr *__restrict__ partial, unsigned nblocks, unsigned m, int have_beta,
                                                        Fr pre, Fr rho, Fr *__restrict__ out) {
    __shared__ Fr S[SC_MAX_M + 1];
    const unsigned i = threadIdx.x;
    if (i <= m) {
        Fr acc = Fr::zero();
        for (unsigned b = 0; b < nblocks; b++) acc = acc + partial[(size_t)b * (SC_MAX_M + 1) + i];
        S[i] = acc;
    }
    __syncthreads();
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
    const Fr29 k261 = fr_to_261(k);
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < half; p += (size_t)gridDim.x * blockDim.x)
        cur[p] = fr_mul_261(old[p + half], k261);
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
bool g_fold_graphs_broken = false;
bool fold_graphs_on() {
    static const bool on = !(getenv("LSA_FR_GRAPHS") && getenv("LSA_FR_GRAPHS")[0] == '0');
    return on && !g_fold_graphs_broken;
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
            (void)hipGetLastError();
            g_fold_graphs_broken = true;
            return issue(st);
        }
        if (g.tick < victim->tick) victim = &g;
    }
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); g_fold_graphs_broken = true; return issue(st); }
    const int rc = issue(st);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc != LSA_OK || e != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        g_fold_graphs_broken = true;                     // (nothing was executed during the capture: run it plainly)
        return rc != LSA_OK ? rc : issue(st);
    }
    hipGraphExec_t exec = nullptr;
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess || !exec) {
        (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        g_fold_graphs_broken = true;
        return issue(st);
    }
    (void)hipGraphDestroy(graph);
    if (victim->exec) (void)hipGraphExecDestroy(victim->exec);
    *victim = FoldGraph{kind, d, a0, a1, a2, a3, a4, exec, ++g_fold_tick};
    if (hipGraphLaunch(exec, st) != hipSuccess) { (void)hipGetLastError(); g_fold_graphs_broken = true; return issue(st); }
    return LSA_OK;
}
}  // namespace
void fr_vec_release() {
    for (auto &g : g_fold_graphs) { if (g.exec) (void)hipGraphExecDestroy(g.exec); g = FoldGraph(); }
    g_fold_graphs_broken = false;
}

// CPPoly::prove witness coefficients: d_v (2^d, untouched), d_r (d), d_w (2^d; the first
// 2^d - 1 entries are written, the last is zeroed like the reference's value-initialised
// vector), d_tmp: scratch of 2^(d-1) + 2^(d-2) elements (ping-pong).  Asynchronous on st.
static int fr_cppoly_fold_issue(const Fr *d_v, size_t d, const Fr *d_r, Fr *d_w, Fr *d_tmp, hipStream_t st);
int fr_cppoly_fold_device(const Fr *d_v, size_t d, const Fr *d_r, Fr *d_w, Fr *d_tmp, hipStream_t st) {
    return fold_run(1, d, d_v, d_r, d_w, d_tmp, nullptr, st, [=](hipStream_t s) { return fr_cppoly_fold_issue(d_v, d, d_r, d_w, d_tmp, s); });
}
static int fr_cppoly_fold_issue(const Fr *d_v, size_t d, const Fr *d_r, Fr *d_w, Fr *d_tmp, hipStream_t st) {
    const size_t N = (size_t)1 << d;
    HIPCHK(hipMemsetAsync(d_w + (N - 1), 0, sizeof(Fr), st));
    const Fr *src = d_v;
    Fr *bufA = d_tmp, *bufB = d_tmp + (N >> 1);
    size_t start = 0;
    size_t i = 0;
    unsigned which = 0;                                  // the next launch writes bufA (0) or bufB (1); it never reads the one it writes
    while (i < d) {
        const size_t m = (size_t)1 << (d - i - 1);       // outputs of round i
        if (m <= FOLD_TAIL) {                            // this round and every later one: one workgroup
            hipLaunchKernelGGL(k_fold_pairs_tail, dim3(1), dim3(256), 0, st, src, m, d_r + i, d_w + start, bufA, bufB, which);
            break;
        }
        // up to three rounds per launch while their last round still has more than FOLD_TAIL outputs
        static const unsigned rmax = getenv("LSA_FOLD_ROUNDS") ? (unsigned)atoi(getenv("LSA_FOLD_ROUNDS")) : 2u;        // (measured at d = 24: 1 round 0.83, 2 rounds 0.74, 3 rounds 0.80 ms: eight 32-byte pieces per lane coalesce worse than four)
        unsigned R = 1;
        while (R < rmax && R < 3 && (m >> R) > FOLD_TAIL) R++;
        const size_t mo = m >> (R - 1);                  // outputs of the last of the R rounds
        Fr *dst = which ? bufB : bufA;                   // (every launch's output is at most half the previous one's: both halves of d_tmp are large enough)
        if (R == 3) hipLaunchKernelGGL((k_fold_pairs_multi<3>), dim3(stream_blocks(mo)), dim3(256), 0, st, src, mo, d_r + i, d_w + start, dst);
        else if (R == 2) hipLaunchKernelGGL((k_fold_pairs_multi<2>), dim3(stream_blocks(mo)), dim3(256), 0, st, src, mo, d_r + i, d_w + start, dst);
        else hipLaunchKernelGGL(k_fold_pairs, dim3(stream_blocks(mo)), dim3(256), 0, st, src, mo, d_r + i, d_w + start, dst);
        src = dst;
        which ^= 1u;
        for (unsigned j = 0; j < R; j++) start += m >> j;
        i += R;
    }
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

// evalMLE(v, r): d_v (2^d, untouched), d_r (d), d_tmp scratch of 2^(d-1) elements; the value
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
    const Fr *src = d_v;
    // d_tmp holds 2^(d-1) elements: a multi-round launch writes its (shorter) output behind what the next launch reads --
    // ping-pong between the two halves of d_tmp (the first launch reads d_v)
    Fr *pp[2] = {d_tmp, d_tmp + ((size_t)1 << (d - 1)) / 2};
    unsigned which = 0;
    size_t i = d;                                        // rounds left: the next one has half = 2^(i-1) and uses r[i-1]
    while (i > 0) {
        const size_t half = (size_t)1 << (i - 1);
        if (half <= FOLD_TAIL) {                         // this round and every later one: one workgroup, in place behind a copy
            hipLaunchKernelGGL(k_fold_halves_tail, dim3(1), dim3(256), 0, st, src, half, d_r, pp[which], d_out);
            break;
        }
        static const unsigned rmax = getenv("LSA_FOLD_ROUNDS_HALVES") ? (unsigned)atoi(getenv("LSA_FOLD_ROUNDS_HALVES")) : 2u; // (0.75 / 0.66 / 0.81 ms: eight read streams 2^21 elements apart are one too many)
        unsigned R = 1;
        while (R < rmax && R < 3 && (half >> R) > FOLD_TAIL) R++;
        const size_t q = half >> (R - 1);                // outputs after the R rounds
        Fr *dst = pp[which];
        if (R == 3) hipLaunchKernelGGL((k_fold_halves_multi<3>), dim3(stream_blocks(q)), dim3(256), 0, st, src, q, d_r + (i - 1), dst);
        else if (R == 2) hipLaunchKernelGGL((k_fold_halves_multi<2>), dim3(stream_blocks(q)), dim3(256), 0, st, src, q, d_r + (i - 1), dst);
        else hipLaunchKernelGGL((k_fold_halves_multi<1>), dim3(stream_blocks(q)), dim3(256), 0, st, src, q, d_r + (i - 1), dst);
        src = dst;
        which ^= 1u;
        i -= R;
    }
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
    const unsigned blocks = (unsigned)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
    hipLaunchKernelGGL(k_sumcheck_partial, dim3(blocks), dim3(256), 0, st, d_suff, tabs, (unsigned)m, half, d_partial);
    hipLaunchKernelGGL(k_sumcheck_finish, dim3(1), dim3(64), 0, st, d_partial, blocks, (unsigned)m, rho ? 1 : 0,
                       pre ? *pre : Fr::one(), rho ? *rho : Fr::zero(), d_out);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}
size_t fr_sumcheck_scratch_elems() { return (size_t)1024 * (SC_MAX_M + 1); }

int fr_scale_upper_device(const Fr *d_old, size_t half, const Fr &k, Fr *d_cur, hipStream_t st) {
    if (half == 0) return LSA_OK;
    hipLaunchKernelGGL(k_scale_upper, dim3(stream_blocks(half)), dim3(256), 0, st, d_old, half, k, d_cur);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}

}  // namespace lsa
