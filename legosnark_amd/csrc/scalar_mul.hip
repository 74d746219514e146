// legosnark_amd/csrc/scalar_mul.hip -- variable-base batch scalar multiplication on G1 and
// the column-sparse "matrix in the exponent" product built on it.
//
// Replaces, for CPlink key generation,
//   mtxmultiexp(out, k, M)      /root/reference/src/gadgets/subspace.cc:18-25
//   -> simplesparsemexp(col, k) /root/reference/src/utils/sparsemexp.cc:15-24
//   -> sparsemexpG              /root/reference/src/utils/sparsemexp.h:62-90
// i.e. out[j] = sum over the non-zeros e of column j of k[row(e)] * M[e]: O(nnz) independent
// 254-bit scalar multiplications (the reference runs each as its own tiny multi_exp / acc*one
// on the CPU) followed by a per-column sum.
//
//   k_smul_g1   one lane per (point, scalar) item, persistent grid-stride loop.  GLV split
//               k = k1 + k2*lambda (glv.h), both halves recoded branch-free into 32 signed
//               4-bit digits in [-8, 7]; a per-lane table {1..8}*P in XYZZ coordinates lives in
//               a global-memory slot (1152 B per lane, contiguous so that every lookup is nine
//               16-byte loads from two cache lines; the slots of the resident lanes stay in
//               L2 / Infinity Cache); 31 x 4 doublings + at most 64 complete XYZZ additions
//               on 29-bit limbs, phi applied to a table entry as one product by beta.
//   k_col_sum   one lane per column: complete XYZZ additions over the column's segment.
// Results are group elements in libff's Jacobian layout; they agree with the reference after
// affine normalisation (the Jacobian representative depends on the addition order).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "curves.h"
#include "smul.h"
#include "msm.h"

namespace lsa {

static constexpr unsigned SMUL_BLOCKS = 512;   // persistent grid: 2 blocks (8 waves) per CU = the kernel's occupancy

__global__ __launch_bounds__(256) void k_smul_g1(const AffPacked *__restrict__ pts, const Fr *__restrict__ scalars,
                                                 const uint32_t *__restrict__ sidx, size_t n, XYZZ29 *__restrict__ tbl,
                                                 Jac<Fq> *__restrict__ out) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    XYZZ29 *T = tbl + gid * SMUL_TBL;
    for (size_t i = gid; i < n; i += stride) {
        const Aff29 P = unpack_affine(pts[i]);
        uint32_t s[8];
        scalars[sidx ? sidx[i] : i].to_canonical(s);
        out[i] = xyzz29_to_jac(smul_glv(P, s, T));
    }
}

// out[j] = sum of items[col_ptr[j] .. col_ptr[j+1])
__global__ __launch_bounds__(256) void k_col_sum_g1(const Jac<Fq> *__restrict__ items, const uint64_t *__restrict__ col_ptr, size_t ncols,
                                                    Jac<Fq> *__restrict__ out) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncols) return;
    XYZZ29 acc = XYZZ29::inf();
#pragma unroll 1
    for (uint64_t e = col_ptr[j]; e < col_ptr[j + 1]; e++) acc = xyzz29_add(acc, CurveG1::from_jac(items[e]));
    out[j] = xyzz29_to_jac(acc);
}

#define HIPCHK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return LSA_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

// d_out[i] = d_scalars[d_sidx ? d_sidx[i] : i] * d_pts[i]; d_pts: libff Jacobian, device.
// Synchronises the stream before returning (temporary buffers are freed).
int g1_scalar_mul_device(const Jac<Fq> *d_pts, const Fr *d_scalars, const uint32_t *d_sidx, size_t n, Jac<Fq> *d_out, hipStream_t st) {
    if (n == 0) return LSA_OK;
    const unsigned blocks = (unsigned)((n + 255) / 256 < SMUL_BLOCKS ? (n + 255) / 256 : SMUL_BLOCKS);
    AffPacked *d_aff = nullptr;
    XYZZ29 *d_tbl = nullptr;
    if (hipMalloc(&d_aff, n * sizeof(AffPacked)) != hipSuccess ||
        hipMalloc(&d_tbl, (size_t)blocks * 256 * SMUL_TBL * sizeof(XYZZ29)) != hipSuccess) {
        if (d_aff) (void)hipFree(d_aff);
        set_error("scalar_mul: workspace allocation failed");
        return LSA_ERR_NOMEM;
    }
    int rc = prepare_bases<Fq>(d_pts, d_aff, n, st);
    if (!rc) hipLaunchKernelGGL(k_smul_g1, dim3(blocks), dim3(256), 0, st, d_aff, d_scalars, d_sidx, n, d_tbl, d_out);
    hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(d_aff);
    (void)hipFree(d_tbl);
    if (rc) return rc;
    if (e != hipSuccess) { set_error("scalar_mul: %s", hipGetErrorString(e)); return LSA_ERR_HIP; }
    return LSA_OK;
}

int g1_column_sums_device(const Jac<Fq> *d_items, const uint64_t *d_col_ptr, size_t ncols, Jac<Fq> *d_out, hipStream_t st) {
    if (ncols == 0) return LSA_OK;
    hipLaunchKernelGGL(k_col_sum_g1, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, st, d_items, d_col_ptr, ncols, d_out);
    HIPCHK(hipGetLastError());
    return LSA_OK;
}

}  // namespace lsa
