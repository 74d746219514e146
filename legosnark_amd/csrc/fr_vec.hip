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
// Both are HBM streams with two Montgomery products per 64 B read (32-bit CIOS, fp.h): about
// as many VALU cycles as HBM cycles at the chip's ratios, so large rounds are bound by
// whichever is slower and the small rounds by launch latency.  Algorithmic bytes per round
// over m output elements: pairs 64 B in + 64 B out, halves 64 B in + 32 B out.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fp.h"
#include "msm.h"

namespace lsa {

__global__ __launch_bounds__(256) void k_fold_pairs(const Fr *__restrict__ v, size_t m, const Fr *__restrict__ r_ptr,
                                                    Fr *__restrict__ w, Fr *__restrict__ vout) {
    const Fr r = *r_ptr;
    const Fr r1 = r - Fr::one();
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < m; p += (size_t)gridDim.x * blockDim.x) {
        const Fr a = v[2 * p], b = v[2 * p + 1];
        w[p] = b - a;
        vout[p] = b * r - a * r1;
    }
}

// cur may alias old: lane p reads old[p], old[p+half] and writes cur[p] only
__global__ __launch_bounds__(256) void k_fold_halves(const Fr *old, size_t half, const Fr *__restrict__ r_ptr, Fr *cur) {
    const Fr r = *r_ptr;
    const Fr r0 = Fr::one() - r;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < half; p += (size_t)gridDim.x * blockDim.x) {
        cur[p] = old[p] * r0 + old[p + half] * r;
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

static unsigned stream_blocks(size_t m) {
    size_t b = (m + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// CPPoly::prove witness coefficients: d_v (2^d, untouched), d_r (d), d_w (2^d; the first
// 2^d - 1 entries are written, the last is zeroed like the reference's value-initialised
// vector), d_tmp: scratch of 2^(d-1) + 2^(d-2) elements (ping-pong).  Asynchronous on st.
int fr_cppoly_fold_device(const Fr *d_v, size_t d, const Fr *d_r, Fr *d_w, Fr *d_tmp, hipStream_t st) {
    const size_t N = (size_t)1 << d;
    HIPCHK(hipMemsetAsync(d_w + (N - 1), 0, sizeof(Fr), st));
    const Fr *src = d_v;
    Fr *bufA = d_tmp, *bufB = d_tmp + (N >> 1);
    size_t start = 0;
    for (size_t i = 0; i < d; i++) {
        const size_t m = (size_t)1 << (d - i - 1);
        Fr *dst = (i & 1) ? bufB : bufA;
        hipLaunchKernelGGL(k_fold_pairs, dim3(stream_blocks(m)), dim3(256), 0, st, src, m, d_r + i, d_w + start, dst);
        src = dst;
        start += m;
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
int fr_eval_mle_device(const Fr *d_v, size_t d, const Fr *d_r, Fr *d_tmp, Fr *d_out, hipStream_t st) {
    if (d == 0) {
        HIPCHK(hipMemcpyAsync(d_out, d_v, sizeof(Fr), hipMemcpyDeviceToDevice, st));
        return LSA_OK;
    }
    const Fr *src = d_v;
    for (size_t i = d; i-- > 0;) {
        const size_t half = (size_t)1 << i;
        Fr *dst = i == 0 ? d_out : d_tmp;
        hipLaunchKernelGGL(k_fold_halves, dim3(stream_blocks(half)), dim3(256), 0, st, src, half, d_r + i, dst);
        src = dst;
    }
    HIPCHK(hipGetLastError());
    return LSA_OK;
}

}  // namespace lsa
