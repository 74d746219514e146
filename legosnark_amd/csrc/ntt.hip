// legosnark_amd/csrc/ntt.hip -- radix-2 number-theoretic transform over Fr on gfx950.
//
// Replaces libfqfft's basic_radix2_domain<Fr>::{FFT, iFFT, cosetFFT, icosetFFT}
// (_basic_radix2_FFT / _multiply_by_coset) as called by the Lipmaa Hadamard gadget,
// /root/reference/src/gadgets/lipmaa.cc:68-81,102-175:
//   FFT:   a[k] <- sum_i a[i] omega^(ik), natural order in and out; iFFT with omega^-1 and a
//   final factor 1/n; the coset variants scale a[i] by g^i before / by g^-i after.
// Fr values are canonical Montgomery residues, so any correct schedule gives libfqfft's bytes.
//
//   k_ntt_twiddles   tw[k] = omega^k, k < n/2 (each lane: one power by square-and-multiply, then
//                    a run of 64 successive products)
//   k_ntt_scale      a[i] *= c * h^i  (coset shifts, 1/n)
//   k_ntt_bitrev     in-place bit-reversal permutation (swap when i < rev(i))
//   k_ntt_local      the first min(log n, 10) butterfly stages on 1024 contiguous elements held
//                    in LDS (32 KiB): one HBM round trip for ten stages
//   k_ntt_stage      one butterfly stage per launch for the strides beyond the LDS tile
// HBM traffic at n = 2^20: 64 MB per global pass, 1 + (log n - 10) passes; about as many Fr
// products (n/2 log n) as a 2^20 MSM has field products per 1/16 of its work: HBM/latency bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fp.h"
#include "msm.h"

namespace lsa {

__device__ __forceinline__ Fr fr_pow(Fr base, uint64_t e) {
    Fr acc = Fr::one();
    bool started = false;
    for (int i = 63; i >= 0; --i) {
        if (started) acc = acc * acc;
        if ((e >> i) & 1) { acc = started ? acc * base : base; started = true; }
    }
    return acc;
}

static constexpr unsigned NTT_RUN = 64;     // successive powers per lane
__global__ __launch_bounds__(256) void k_ntt_twiddles(Fr omega, size_t count, Fr *__restrict__ tw) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t lo = t * NTT_RUN;
    if (lo >= count) return;
    Fr x = fr_pow(omega, lo);
    for (size_t k = lo; k < lo + NTT_RUN && k < count; k++) { tw[k] = x; x = x * omega; }
}

// a[i] *= c * h^i
__global__ __launch_bounds__(256) void k_ntt_scale(Fr *__restrict__ a, size_t n, Fr c, Fr h) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t lo = t * NTT_RUN;
    if (lo >= n) return;
    Fr x = c * fr_pow(h, lo);
    for (size_t k = lo; k < lo + NTT_RUN && k < n; k++) { a[k] = a[k] * x; x = x * h; }
}

__global__ __launch_bounds__(256) void k_ntt_bitrev(Fr *__restrict__ a, unsigned log_n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> log_n) return;
    const size_t j = (size_t)(__brevll((unsigned long long)i) >> (64 - log_n));
    if (i < j) { Fr x = a[i]; a[i] = a[j]; a[j] = x; }
}

// stages s = 0 .. ls-1 (butterfly spans 2 .. 2^ls <= 1024) on the block's 2^ls contiguous
// elements; twiddle of span len, position j: omega^(j * n/len) = tw[j << (log_n - s - 1)]
static constexpr unsigned NTT_LOCAL_LOG = 10;
__global__ __launch_bounds__(256) void k_ntt_local(Fr *__restrict__ a, unsigned log_n, unsigned ls, const Fr *__restrict__ tw) {
    __shared__ Fr tile[1u << NTT_LOCAL_LOG];
    const unsigned tsz = 1u << ls;
    Fr *base = a + (size_t)blockIdx.x * tsz;
    for (unsigned x = threadIdx.x; x < tsz; x += 256) tile[x] = base[x];
    __syncthreads();
    for (unsigned s = 0; s < ls; s++) {
        const unsigned hl = 1u << s;                      // half span
        for (unsigned b = threadIdx.x; b < tsz / 2; b += 256) {
            const unsigned j = b & (hl - 1), i0 = ((b >> s) << (s + 1)) | j;
            const Fr u = tile[i0], v = tile[i0 + hl] * tw[(size_t)j << (log_n - s - 1)];
            tile[i0] = u + v;
            tile[i0 + hl] = u - v;
        }
        __syncthreads();
    }
    for (unsigned x = threadIdx.x; x < tsz; x += 256) base[x] = tile[x];
}

// one stage s >= NTT_LOCAL_LOG in global memory
__global__ __launch_bounds__(256) void k_ntt_stage(Fr *__restrict__ a, unsigned log_n, unsigned s, const Fr *__restrict__ tw) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >> (log_n - 1)) return;
    const size_t hl = (size_t)1 << s;
    const size_t j = b & (hl - 1), i0 = ((b >> s) << (s + 1)) | j;
    const Fr u = a[i0], v = a[i0 + hl] * tw[j << (log_n - s - 1)];
    a[i0] = u + v;
    a[i0 + hl] = u - v;
}

#define HIPCHK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return LSA_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

static __host__ __device__ Fr host_pow(Fr base, uint64_t e) {
    Fr acc = Fr::one();
    for (int i = 63; i >= 0; --i) {
        acc = acc * acc;
        if ((e >> i) & 1) acc = acc * base;
    }
    return acc;
}

// In place on d_a (2^log_n elements).  omega: primitive 2^log_n-th root of unity.  coset: g or
// nullptr.  d_tw: scratch of 2^(log_n-1) elements.  Asynchronous on st.
int fr_ntt_device(Fr *d_a, unsigned log_n, const Fr &omega, bool inverse, const Fr *coset, Fr *d_tw, hipStream_t st) {
    const size_t n = (size_t)1 << log_n;
    if (log_n == 0) return LSA_OK;                        // the 1-point transform is the identity (1/n = 1, g^0 = 1)
    const Fr w = inverse ? omega.inverse() : omega;
    const unsigned run_blocks = (unsigned)((n / NTT_RUN + 255) / 256 + 1);
    if (!inverse && coset) hipLaunchKernelGGL(k_ntt_scale, dim3(run_blocks), dim3(256), 0, st, d_a, n, Fr::one(), *coset);
    hipLaunchKernelGGL(k_ntt_twiddles, dim3(run_blocks), dim3(256), 0, st, w, n / 2, d_tw);
    hipLaunchKernelGGL(k_ntt_bitrev, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_a, log_n);
    const unsigned ls = log_n < NTT_LOCAL_LOG ? log_n : NTT_LOCAL_LOG;
    hipLaunchKernelGGL(k_ntt_local, dim3((unsigned)(n >> ls)), dim3(256), 0, st, d_a, log_n, ls, d_tw);
    for (unsigned s = ls; s < log_n; s++)
        hipLaunchKernelGGL(k_ntt_stage, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st, d_a, log_n, s, d_tw);
    if (inverse) {
        // 1/n = 2^-log_n; with a coset also g^-i
        const Fr two = Fr::one() + Fr::one();
        const Fr ninv = host_pow(two, log_n).inverse();
        hipLaunchKernelGGL(k_ntt_scale, dim3(run_blocks), dim3(256), 0, st, d_a, n, ninv, coset ? coset->inverse() : Fr::one());
    }
    HIPCHK(hipGetLastError());
    return LSA_OK;
}

}  // namespace lsa
