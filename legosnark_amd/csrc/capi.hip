// legosnark_amd/csrc/capi.hip -- implementation of the C-ABI in include/legosnark_amd.h.
// No CPU fallback: every compute entry point requires lsa_init() to have found a gfx950
// device and fails with LSA_ERR_NO_DEVICE otherwise.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "msm.h"
#include "tower.h"

namespace lsa {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

struct State {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    void *d_result = nullptr;      // 192-byte device slot for MSM results
    void *h_result = nullptr;      // pinned host mirror
};
static State g;

#define HIPCHK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return LSA_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

static int require_ready() {
    if (!g.ready) {
        set_error("legosnark_amd: no initialised gfx950 device (call lsa_init first; there is no CPU fallback)");
        return LSA_ERR_NO_DEVICE;
    }
    return LSA_OK;
}

}  // namespace lsa

using namespace lsa;

static void release_stage_buffers();

struct lsa_bases {
    void *d_aff = nullptr;   // prepared bases (msm_base_bytes(group) each); with a table: window-major copies
    size_t n = 0;
    int group = 1;           // 1 = G1, 2 = G2
    size_t table_stride = 0; // n when the pre-shifted windows 2^(16k)*P are resident, else 0
};

extern "C" {

int lsa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *lsa_last_error(void) { return g_err; }

int lsa_init(int device) {
    if (g.ready && g.device == device) return LSA_OK;
    if (g.ready) lsa_shutdown();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("lsa_init: no HIP device visible (there is no CPU fallback)");
        return LSA_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("lsa_init: device %d out of range (%d visible)", device, n);
        return LSA_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("lsa_init: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return LSA_ERR_NO_DEVICE;
    }
    HIPCHK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    HIPCHK(hipMalloc(&g.d_result, 512));
    HIPCHK(hipHostMalloc(&g.h_result, 512, hipHostMallocDefault));
    g.device = device;
    g.ready = true;
    return LSA_OK;
}

void lsa_shutdown(void) {
    if (!g.ready) return;
    (void)hipStreamSynchronize(g.stream);
    msm_release_workspace();
    release_stage_buffers();
    (void)hipFree(g.d_result);
    (void)hipHostFree(g.h_result);
    (void)hipStreamDestroy(g.stream);
    g = State();
}

void *lsa_stream(void) { return g.ready ? (void *)g.stream : nullptr; }

int lsa_stream_join(void) {
    int rc = require_ready();
    if (rc) return rc;
    return msm_join(g.stream);
}

int lsa_synchronize(void) {
    int rc = require_ready();
    if (rc) return rc;
    rc = msm_join(g.stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    return LSA_OK;
}

unsigned lsa_msm_window_bits(size_t n) { return msm_window_bits(2 * n); }   // G1 (GLV: 2n virtual scalars)

int lsa_profile_enable(int on) { msm_profile_enable(on != 0); return LSA_OK; }
int lsa_profile_last_msm(float ms[LSA_MSM_STAGES]) { return msm_profile_last(ms); }

// ---------------------------------------------------------------- bases
}  // extern "C"
template <class F>
static int bases_create(const void *bases_jac, size_t n, int src_on_device, int group, lsa_bases **out) {
    int rc = require_ready();
    if (rc) return rc;
    if (!out || (n && !bases_jac)) { set_error("bases_create: null argument"); return LSA_ERR_INVALID; }
    lsa_bases *b = new lsa_bases();
    b->n = n;
    b->group = group;
    if (n) {
        const Jac<F> *d_in = nullptr;
        void *tmp = nullptr;
        // Large resident CRS vectors also keep the pre-shifted windows (msm.hip, "merged
        // windows"): nwin x the memory, no Horner fold per MSM.  LSA_PRECOMPUTE=0 opts out; a
        // table that does not fit falls back to the plain layout.
        const char *pe = getenv("LSA_PRECOMPUTE");
        bool table = n >= msm_merge_min() && !(pe && pe[0] == '0');
        const size_t tw = msm_table_windows(group);
        if (table && (uint64_t)n * tw >= (1u << 30)) table = false;
        if (table && hipMalloc(&b->d_aff, tw * n * msm_base_bytes(group)) != hipSuccess) { (void)hipGetLastError(); b->d_aff = nullptr; table = false; }
        if (!table && hipMalloc(&b->d_aff, n * msm_base_bytes(group)) != hipSuccess) {
            delete b;
            set_error("bases_create: hipMalloc of %zu bytes failed", n * msm_base_bytes(group));
            return LSA_ERR_NOMEM;
        }
        if (src_on_device) {
            d_in = (const Jac<F> *)bases_jac;
        } else {
            if (hipMalloc(&tmp, n * sizeof(Jac<F>)) != hipSuccess) {
                (void)hipFree(b->d_aff); delete b;
                set_error("bases_create: hipMalloc of %zu bytes failed", n * sizeof(Jac<F>));
                return LSA_ERR_NOMEM;
            }
            hipError_t e = hipMemcpyAsync(tmp, bases_jac, n * sizeof(Jac<F>), hipMemcpyHostToDevice, g.stream);
            if (e != hipSuccess) { (void)hipFree(tmp); (void)hipFree(b->d_aff); delete b; set_error("bases_create: H2D failed: %s", hipGetErrorString(e)); return LSA_ERR_HIP; }
            d_in = (const Jac<F> *)tmp;
        }
        rc = prepare_bases<F>(d_in, b->d_aff, n, g.stream);
        if (!rc && table) {
            rc = precompute_windows<F>(b->d_aff, n, g.stream);
            if (!rc) b->table_stride = n;
        }
        hipError_t e = hipStreamSynchronize(g.stream);
        if (tmp) (void)hipFree(tmp);
        if (rc || e != hipSuccess) {
            if (!rc) { set_error("bases_create: kernel failed: %s", hipGetErrorString(e)); rc = LSA_ERR_HIP; }
            (void)hipFree(b->d_aff); delete b;
            return rc;
        }
    }
    *out = b;
    return LSA_OK;
}

extern "C" {
int lsa_g1_bases_create(const void *bases_jac, size_t n, int src_on_device, lsa_bases **out) {
    return bases_create<Fq>(bases_jac, n, src_on_device, 1, out);
}
int lsa_g2_bases_create(const void *bases_jac, size_t n, int src_on_device, lsa_bases **out) {
    return bases_create<Fq2>(bases_jac, n, src_on_device, 2, out);
}
void lsa_bases_destroy(lsa_bases *b) {
    if (!b) return;
    if (b->d_aff) (void)hipFree(b->d_aff);
    delete b;
}
size_t lsa_bases_size(const lsa_bases *b) { return b ? b->n : 0; }
int lsa_bases_has_table(const lsa_bases *b) { return b && b->table_stride ? 1 : 0; }
void lsa_msm_set_table_threshold(size_t n) { msm_set_merge_min(n); }
const void *lsa_bases_device_ptr(const lsa_bases *b) { return b ? b->d_aff : nullptr; }

// ---------------------------------------------------------------- MSM
int lsa_msm_run_async(const lsa_bases *bases, size_t first, const void *d_scalars, size_t n, void *d_out_jac) {
    int rc = require_ready();
    if (rc) return rc;
    if (!bases || !d_out_jac || (n && !d_scalars)) { set_error("msm_run: null argument"); return LSA_ERR_INVALID; }
    if (first > bases->n || n > bases->n - first) { set_error("msm_run: range [%zu,%zu) exceeds %zu bases", first, first + n, bases->n); return LSA_ERR_INVALID; }
    if (bases->group == 1)
        return msm_device<Fq>(bases->d_aff, first, (const Fr *)d_scalars, n, (Jac<Fq> *)d_out_jac, g.stream, bases->table_stride);
    return msm_device<Fq2>(bases->d_aff, first, (const Fr *)d_scalars, n, (Jac<Fq2> *)d_out_jac, g.stream, bases->table_stride);
}

int lsa_msm_run(const lsa_bases *bases, size_t first, const void *d_scalars, size_t n, void *out_jac) {
    int rc = lsa_msm_run_async(bases, first, d_scalars, n, g.d_result);
    if (rc) return rc;
    rc = msm_join(g.stream);
    if (rc) return rc;
    size_t bytes = bases->group == 1 ? sizeof(Jac<Fq>) : sizeof(Jac<Fq2>);
    HIPCHK(hipMemcpyAsync(g.h_result, g.d_result, bytes, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    memcpy(out_jac, g.h_result, bytes);
    return LSA_OK;
}

}  // extern "C"
// grow-only device staging buffers of the host-buffer entry points (the libff shim calls
// lsa_g1_msm thousands of times with tiny inputs: no hipMalloc/hipFree per call)
namespace {
struct StageBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) { (void)hipStreamSynchronize(g.stream); (void)hipFree(p); }
        p = nullptr; cap = 0;
        size_t want = bytes < 4096 ? 4096 : bytes + bytes / 4;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return -1; }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
StageBuf g_stage_jac, g_stage_bases, g_stage_scalars;
StageBuf g_pair_p, g_pair_q, g_pair_f, g_pair_s, g_pair_o;     // pairing host path: points, Miller values, product scratch, result
}  // namespace
static void release_stage_buffers() {
    g_stage_jac.release(); g_stage_bases.release(); g_stage_scalars.release();
    g_pair_p.release(); g_pair_q.release(); g_pair_f.release(); g_pair_s.release(); g_pair_o.release();
}

template <class F>
static int msm_host(const void *bases_jac, const void *scalars, size_t n, void *out_jac, int group) {
    int rc = require_ready();
    if (rc) return rc;
    if (!out_jac || (n && (!bases_jac || !scalars))) { set_error("msm: null argument"); return LSA_ERR_INVALID; }
    if (g_stage_jac.ensure(n * sizeof(Jac<F>)) || g_stage_bases.ensure(n * msm_base_bytes(group)) || g_stage_scalars.ensure(n * sizeof(Fr))) {
        set_error("msm: staging allocation failed");
        return LSA_ERR_NOMEM;
    }
    if (n) {
        HIPCHK(hipMemcpyAsync(g_stage_jac.p, bases_jac, n * sizeof(Jac<F>), hipMemcpyHostToDevice, g.stream));
        HIPCHK(hipMemcpyAsync(g_stage_scalars.p, scalars, n * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
        rc = prepare_bases<F>((const Jac<F> *)g_stage_jac.p, g_stage_bases.p, n, g.stream);
        if (rc) return rc;
    }
    rc = msm_device<F>(g_stage_bases.p, 0, (const Fr *)g_stage_scalars.p, n, (Jac<F> *)g.d_result, g.stream);
    if (rc) return rc;
    rc = msm_join(g.stream);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(g.h_result, g.d_result, sizeof(Jac<F>), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    memcpy(out_jac, g.h_result, sizeof(Jac<F>));
    return LSA_OK;
}

extern "C" {
int lsa_g1_msm(const void *bases_jac, const void *scalars, size_t n, size_t chunks, void *out_jac) {
    (void)chunks;
    return msm_host<Fq>(bases_jac, scalars, n, out_jac, 1);
}
int lsa_g2_msm(const void *bases_jac, const void *scalars, size_t n, size_t chunks, void *out_jac) {
    (void)chunks;
    return msm_host<Fq2>(bases_jac, scalars, n, out_jac, 2);
}

// ---------------------------------------------------------------- normalisation
}  // extern "C"
template <class F>
static int normalize_host(const void *in_jac, size_t n, void *out_jac) {
    int rc = require_ready();
    if (rc) return rc;
    if (n == 0) return LSA_OK;
    if (!in_jac || !out_jac) { set_error("normalize: null argument"); return LSA_ERR_INVALID; }
    void *d_in = nullptr, *d_aff = nullptr;
    if (hipMalloc(&d_in, n * sizeof(Jac<F>)) != hipSuccess || hipMalloc(&d_aff, n * sizeof(Aff<F>)) != hipSuccess) {
        if (d_in) (void)hipFree(d_in);
        set_error("normalize: hipMalloc failed");
        return LSA_ERR_NOMEM;
    }
    hipError_t e = hipMemcpyAsync(d_in, in_jac, n * sizeof(Jac<F>), hipMemcpyHostToDevice, g.stream);
    if (e == hipSuccess) rc = normalize_to_affine<F>((const Jac<F> *)d_in, (Aff<F> *)d_aff, n, g.stream);
    std::vector<Aff<F>> host(n);
    if (e == hipSuccess && !rc) e = hipMemcpyAsync(host.data(), d_aff, n * sizeof(Aff<F>), hipMemcpyDeviceToHost, g.stream);
    if (e == hipSuccess && !rc) e = hipStreamSynchronize(g.stream);
    (void)hipFree(d_in);
    (void)hipFree(d_aff);
    if (rc) return rc;
    if (e != hipSuccess) { set_error("normalize: %s", hipGetErrorString(e)); return LSA_ERR_HIP; }
    Jac<F> *o = (Jac<F> *)out_jac;
    for (size_t i = 0; i < n; i++) {
        if (host[i].is_inf()) o[i] = Jac<F>::inf();
        else o[i] = Jac<F>{host[i].x, host[i].y, F::one()};
    }
    return LSA_OK;
}
extern "C" {
int lsa_g1_normalize(const void *in_jac, size_t n, void *out_jac) { return normalize_host<Fq>(in_jac, n, out_jac); }
int lsa_g2_normalize(const void *in_jac, size_t n, void *out_jac) { return normalize_host<Fq2>(in_jac, n, out_jac); }

}  // extern "C"

// ---------------------------------------------------------------- batch_exp / sum
template <class F>
static int batch_exp_any(const void *base_jac, const void *scalars, size_t n, void *out_jac, int on_device) {
    int rc = require_ready();
    if (rc) return rc;
    if (n == 0) return LSA_OK;
    if (!base_jac || !scalars || !out_jac) { set_error("batch_exp: null argument"); return LSA_ERR_INVALID; }
    Jac<F> base;
    memcpy(&base, base_jac, sizeof base);
    if (on_device) return batch_exp_device<F>(base, (const Fr *)scalars, n, (Jac<F> *)out_jac, g.stream);
    void *d_sc = nullptr, *d_out = nullptr;
    if (hipMalloc(&d_sc, n * sizeof(Fr)) != hipSuccess || hipMalloc(&d_out, n * sizeof(Jac<F>)) != hipSuccess) {
        if (d_sc) (void)hipFree(d_sc);
        set_error("batch_exp: hipMalloc failed");
        return LSA_ERR_NOMEM;
    }
    hipError_t e = hipMemcpyAsync(d_sc, scalars, n * sizeof(Fr), hipMemcpyHostToDevice, g.stream);
    if (e == hipSuccess) rc = batch_exp_device<F>(base, (const Fr *)d_sc, n, (Jac<F> *)d_out, g.stream);
    if (e == hipSuccess && !rc) e = hipMemcpy(out_jac, d_out, n * sizeof(Jac<F>), hipMemcpyDeviceToHost);
    (void)hipFree(d_sc);
    (void)hipFree(d_out);
    if (rc) return rc;
    if (e != hipSuccess) { set_error("batch_exp: %s", hipGetErrorString(e)); return LSA_ERR_HIP; }
    return LSA_OK;
}

extern "C" {
int lsa_g1_batch_exp(const void *base_jac, const void *scalars, size_t n, void *out_jac, int on_device) {
    return batch_exp_any<Fq>(base_jac, scalars, n, out_jac, on_device);
}
int lsa_g2_batch_exp(const void *base_jac, const void *scalars, size_t n, void *out_jac, int on_device) {
    return batch_exp_any<Fq2>(base_jac, scalars, n, out_jac, on_device);
}
// the same on a caller-supplied stream, without touching lsa_stream(): lets the collective and
// the fold of step i run beside the MSM front of step i+1
}  // extern "C"
template <class F>
static int sum_on_any(const void *d_pts, size_t n, void *d_out, void *stream) {
    int rc = require_ready();
    if (rc) return rc;
    if (!d_out || (n && !d_pts)) { set_error("sum: null argument"); return LSA_ERR_INVALID; }
    return sum_points_device<F>((const Jac<F> *)d_pts, n, (Jac<F> *)d_out, (hipStream_t)stream);
}
extern "C" {
int lsa_g1_sum_on(const void *d_pts, size_t n, void *d_out, void *stream) { return sum_on_any<Fq>(d_pts, n, d_out, stream); }
int lsa_g2_sum_on(const void *d_pts, size_t n, void *d_out, void *stream) { return sum_on_any<Fq2>(d_pts, n, d_out, stream); }
int lsa_stream_join_to(void *stream) {
    int rc = require_ready();
    if (rc) return rc;
    return msm_join_to((hipStream_t)stream);
}
int lsa_g1_sum_async(const void *d_pts, size_t n, void *d_out) {
    int rc = require_ready();
    if (rc) return rc;
    if (!d_out || (n && !d_pts)) { set_error("sum: null argument"); return LSA_ERR_INVALID; }
    rc = msm_join(g.stream);
    if (rc) return rc;
    return sum_points_device<Fq>((const Jac<Fq> *)d_pts, n, (Jac<Fq> *)d_out, g.stream);
}
int lsa_g2_sum_async(const void *d_pts, size_t n, void *d_out) {
    int rc = require_ready();
    if (rc) return rc;
    if (!d_out || (n && !d_pts)) { set_error("sum: null argument"); return LSA_ERR_INVALID; }
    rc = msm_join(g.stream);
    if (rc) return rc;
    return sum_points_device<Fq2>((const Jac<Fq2> *)d_pts, n, (Jac<Fq2> *)d_out, g.stream);
}
}  // extern "C"

namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess ? 0 : -1; }
};
}  // namespace

// ---------------------------------------------------------------- variable-base scalar mul / sparse matrix
extern "C" {
int lsa_g1_scalar_mul_batch(const void *pts_jac, const void *scalars, size_t n, void *out_jac, int on_device) {
    int rc = require_ready();
    if (rc) return rc;
    if (n == 0) return LSA_OK;
    if (!pts_jac || !scalars || !out_jac) { set_error("scalar_mul_batch: null argument"); return LSA_ERR_INVALID; }
    rc = msm_join(g.stream);
    if (rc) return rc;
    if (on_device) return g1_scalar_mul_device((const Jac<Fq> *)pts_jac, (const Fr *)scalars, nullptr, n, (Jac<Fq> *)out_jac, g.stream);
    DevBuf d_p, d_s, d_o;
    if (d_p.alloc(n * sizeof(Jac<Fq>)) || d_s.alloc(n * sizeof(Fr)) || d_o.alloc(n * sizeof(Jac<Fq>))) {
        set_error("scalar_mul_batch: hipMalloc failed");
        return LSA_ERR_NOMEM;
    }
    HIPCHK(hipMemcpyAsync(d_p.p, pts_jac, n * sizeof(Jac<Fq>), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(d_s.p, scalars, n * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
    rc = g1_scalar_mul_device((const Jac<Fq> *)d_p.p, (const Fr *)d_s.p, nullptr, n, (Jac<Fq> *)d_o.p, g.stream);
    if (rc) return rc;
    HIPCHK(hipMemcpy(out_jac, d_o.p, n * sizeof(Jac<Fq>), hipMemcpyDeviceToHost));
    return LSA_OK;
}

int lsa_g1_sparse_matrix_msm(const void *vals_jac, const uint32_t *rows, const uint64_t *col_ptr, size_t ncols,
                             const void *exps, size_t nrows, void *out_jac) {
    int rc = require_ready();
    if (rc) return rc;
    if (ncols == 0) return LSA_OK;
    if (!col_ptr || !out_jac) { set_error("sparse_matrix_msm: null argument"); return LSA_ERR_INVALID; }
    const uint64_t nnz = col_ptr[ncols];
    if (col_ptr[0] != 0) { set_error("sparse_matrix_msm: col_ptr[0] must be 0"); return LSA_ERR_INVALID; }
    for (size_t j = 0; j < ncols; j++)
        if (col_ptr[j] > col_ptr[j + 1]) { set_error("sparse_matrix_msm: col_ptr not monotone at column %zu", j); return LSA_ERR_INVALID; }
    if (nnz && (!vals_jac || !rows || !exps)) { set_error("sparse_matrix_msm: null argument"); return LSA_ERR_INVALID; }
    for (uint64_t e = 0; e < nnz; e++)
        if (rows[e] >= nrows) { set_error("sparse_matrix_msm: row index %u out of range (%zu rows)", rows[e], nrows); return LSA_ERR_INVALID; }
    rc = msm_join(g.stream);
    if (rc) return rc;
    DevBuf d_v, d_r, d_c, d_e, d_items, d_o;
    if (d_v.alloc(nnz * sizeof(Jac<Fq>)) || d_r.alloc(nnz * sizeof(uint32_t)) || d_c.alloc((ncols + 1) * sizeof(uint64_t)) ||
        d_e.alloc(nrows * sizeof(Fr)) || d_items.alloc(nnz * sizeof(Jac<Fq>)) || d_o.alloc(ncols * sizeof(Jac<Fq>))) {
        set_error("sparse_matrix_msm: hipMalloc failed");
        return LSA_ERR_NOMEM;
    }
    if (nnz) {
        HIPCHK(hipMemcpyAsync(d_v.p, vals_jac, nnz * sizeof(Jac<Fq>), hipMemcpyHostToDevice, g.stream));
        HIPCHK(hipMemcpyAsync(d_r.p, rows, nnz * sizeof(uint32_t), hipMemcpyHostToDevice, g.stream));
        HIPCHK(hipMemcpyAsync(d_e.p, exps, nrows * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
    }
    HIPCHK(hipMemcpyAsync(d_c.p, col_ptr, (ncols + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, g.stream));
    rc = g1_scalar_mul_device((const Jac<Fq> *)d_v.p, (const Fr *)d_e.p, (const uint32_t *)d_r.p, nnz, (Jac<Fq> *)d_items.p, g.stream);
    if (rc) return rc;
    rc = g1_column_sums_device((const Jac<Fq> *)d_items.p, (const uint64_t *)d_c.p, ncols, (Jac<Fq> *)d_o.p, g.stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipMemcpy(out_jac, d_o.p, ncols * sizeof(Jac<Fq>), hipMemcpyDeviceToHost));
    return LSA_OK;
}
}  // extern "C"

// ---------------------------------------------------------------- Fr vectors
extern "C" {
int lsa_fr_cppoly_witness(const void *v, size_t d, const void *r, void *w, int on_device) {
    int rc = require_ready();
    if (rc) return rc;
    if (d > 40) { set_error("cppoly_witness: d = %zu too large", d); return LSA_ERR_INVALID; }
    if (!v || !w || (d && !r)) { set_error("cppoly_witness: null argument"); return LSA_ERR_INVALID; }
    const size_t N = (size_t)1 << d;
    DevBuf d_tmp, d_v, d_r, d_w;
    if (d_tmp.alloc((N / 2 + N / 4 + 1) * sizeof(Fr))) { set_error("cppoly_witness: hipMalloc failed"); return LSA_ERR_NOMEM; }
    if (on_device) {
        rc = fr_cppoly_fold_device((const Fr *)v, d, (const Fr *)r, (Fr *)w, (Fr *)d_tmp.p, g.stream);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(g.stream));   // scratch is freed on return
        return LSA_OK;
    }
    if (d_v.alloc(N * sizeof(Fr)) || d_r.alloc((d + 1) * sizeof(Fr)) || d_w.alloc(N * sizeof(Fr))) { set_error("cppoly_witness: hipMalloc failed"); return LSA_ERR_NOMEM; }
    HIPCHK(hipMemcpyAsync(d_v.p, v, N * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
    if (d) HIPCHK(hipMemcpyAsync(d_r.p, r, d * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
    rc = fr_cppoly_fold_device((const Fr *)d_v.p, d, (const Fr *)d_r.p, (Fr *)d_w.p, (Fr *)d_tmp.p, g.stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipMemcpy(w, d_w.p, N * sizeof(Fr), hipMemcpyDeviceToHost));
    return LSA_OK;
}

int lsa_fr_eval_mle(const void *v, size_t d, const void *r, void *out, int on_device) {
    int rc = require_ready();
    if (rc) return rc;
    if (d > 40) { set_error("eval_mle: d = %zu too large", d); return LSA_ERR_INVALID; }
    if (!v || !out || (d && !r)) { set_error("eval_mle: null argument"); return LSA_ERR_INVALID; }
    const size_t N = (size_t)1 << d;
    DevBuf d_tmp, d_v, d_r, d_o;
    if (d_tmp.alloc((N / 2 + 1) * sizeof(Fr))) { set_error("eval_mle: hipMalloc failed"); return LSA_ERR_NOMEM; }
    if (on_device) {
        rc = fr_eval_mle_device((const Fr *)v, d, (const Fr *)r, (Fr *)d_tmp.p, (Fr *)out, g.stream);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(g.stream));
        return LSA_OK;
    }
    if (d_v.alloc(N * sizeof(Fr)) || d_r.alloc((d + 1) * sizeof(Fr)) || d_o.alloc(sizeof(Fr))) { set_error("eval_mle: hipMalloc failed"); return LSA_ERR_NOMEM; }
    HIPCHK(hipMemcpyAsync(d_v.p, v, N * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
    if (d) HIPCHK(hipMemcpyAsync(d_r.p, r, d * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
    rc = fr_eval_mle_device((const Fr *)d_v.p, d, (const Fr *)d_r.p, (Fr *)d_tmp.p, (Fr *)d_o.p, g.stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipMemcpy(out, d_o.p, sizeof(Fr), hipMemcpyDeviceToHost));
    return LSA_OK;
}

int lsa_fr_sumcheck_round(const void *suff, const void *const *tables, size_t m, size_t half, const void *pre, const void *rho_j,
                          void *out_coeffs, int on_device) {
    int rc = require_ready();
    if (rc) return rc;
    if (!tables || !out_coeffs || m == 0 || m > 4 || half == 0) { set_error("sumcheck_round: invalid argument (1 <= m <= 4, half > 0)"); return LSA_ERR_INVALID; }
    if (rho_j && !pre) { set_error("sumcheck_round: rho_j without pre"); return LSA_ERR_INVALID; }
    for (size_t t = 0; t < m; t++) if (!tables[t]) { set_error("sumcheck_round: null table"); return LSA_ERR_INVALID; }
    const size_t ncoef = m + (rho_j ? 2 : 1);
    DevBuf d_partial, d_out, d_suff, d_tab[4];
    if (d_partial.alloc(fr_sumcheck_scratch_elems() * sizeof(Fr)) || d_out.alloc(8 * sizeof(Fr))) { set_error("sumcheck_round: hipMalloc failed"); return LSA_ERR_NOMEM; }
    const Fr *tabs[4] = {nullptr, nullptr, nullptr, nullptr};
    const Fr *sf = (const Fr *)suff;
    if (on_device) {
        for (size_t t = 0; t < m; t++) tabs[t] = (const Fr *)tables[t];
    } else {
        for (size_t t = 0; t < m; t++) {
            if (d_tab[t].alloc(2 * half * sizeof(Fr))) { set_error("sumcheck_round: hipMalloc failed"); return LSA_ERR_NOMEM; }
            HIPCHK(hipMemcpyAsync(d_tab[t].p, tables[t], 2 * half * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
            tabs[t] = (const Fr *)d_tab[t].p;
        }
        if (suff) {
            if (d_suff.alloc(half * sizeof(Fr))) { set_error("sumcheck_round: hipMalloc failed"); return LSA_ERR_NOMEM; }
            HIPCHK(hipMemcpyAsync(d_suff.p, suff, half * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
            sf = (const Fr *)d_suff.p;
        }
    }
    rc = fr_sumcheck_round_device(sf, tabs, m, half, (const Fr *)pre, (const Fr *)rho_j, (Fr *)d_partial.p, (Fr *)d_out.p, g.stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipMemcpy(out_coeffs, d_out.p, ncoef * sizeof(Fr), hipMemcpyDeviceToHost));
    return LSA_OK;
}

int lsa_fr_scale_upper(const void *old, size_t half, const void *k, void *cur, int on_device) {
    int rc = require_ready();
    if (rc) return rc;
    if (half == 0) return LSA_OK;
    if (!old || !cur || !k) { set_error("fr_scale_upper: null argument"); return LSA_ERR_INVALID; }
    Fr kk;
    memcpy(&kk, k, sizeof kk);
    if (on_device) return fr_scale_upper_device((const Fr *)old, half, kk, (Fr *)cur, g.stream);
    DevBuf d_v;
    if (d_v.alloc(2 * half * sizeof(Fr))) { set_error("fr_scale_upper: hipMalloc failed"); return LSA_ERR_NOMEM; }
    HIPCHK(hipMemcpyAsync(d_v.p, old, 2 * half * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
    rc = fr_scale_upper_device((const Fr *)d_v.p, half, kk, (Fr *)d_v.p, g.stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipMemcpy(cur, d_v.p, half * sizeof(Fr), hipMemcpyDeviceToHost));
    return LSA_OK;
}

int lsa_fr_ntt(void *a, size_t log_n, const void *omega, int inverse, const void *coset_g, int on_device) {
    int rc = require_ready();
    if (rc) return rc;
    if (log_n > 28) { set_error("fr_ntt: log_n = %zu exceeds the 2-adicity of Fr (28)", log_n); return LSA_ERR_INVALID; }
    if (!a || !omega) { set_error("fr_ntt: null argument"); return LSA_ERR_INVALID; }
    if (log_n == 0) return LSA_OK;
    const size_t n = (size_t)1 << log_n;
    Fr w, gco;
    memcpy(&w, omega, sizeof w);
    if (coset_g) memcpy(&gco, coset_g, sizeof gco);
    DevBuf d_tw, d_a;
    if (d_tw.alloc((n / 2 + 1) * sizeof(Fr))) { set_error("fr_ntt: hipMalloc failed"); return LSA_ERR_NOMEM; }
    Fr *da = (Fr *)a;
    if (!on_device) {
        if (d_a.alloc(n * sizeof(Fr))) { set_error("fr_ntt: hipMalloc failed"); return LSA_ERR_NOMEM; }
        HIPCHK(hipMemcpyAsync(d_a.p, a, n * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
        da = (Fr *)d_a.p;
    }
    rc = fr_ntt_device(da, (unsigned)log_n, w, inverse != 0, coset_g ? &gco : nullptr, (Fr *)d_tw.p, g.stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));      // the twiddle table is freed on return
    if (!on_device) HIPCHK(hipMemcpy(a, d_a.p, n * sizeof(Fr), hipMemcpyDeviceToHost));
    return LSA_OK;
}

int lsa_fr_fold(const void *old, size_t half, const void *r, void *cur, int on_device) {
    int rc = require_ready();
    if (rc) return rc;
    if (half == 0) return LSA_OK;
    if (!old || !cur || !r) { set_error("fr_fold: null argument"); return LSA_ERR_INVALID; }
    if (on_device) return fr_fold_halves_device((const Fr *)old, half, (const Fr *)r, (Fr *)cur, g.stream);
    DevBuf d_v, d_r;
    if (d_v.alloc(2 * half * sizeof(Fr)) || d_r.alloc(sizeof(Fr))) { set_error("fr_fold: hipMalloc failed"); return LSA_ERR_NOMEM; }
    HIPCHK(hipMemcpyAsync(d_v.p, old, 2 * half * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(d_r.p, r, sizeof(Fr), hipMemcpyHostToDevice, g.stream));
    rc = fr_fold_halves_device((const Fr *)d_v.p, half, (const Fr *)d_r.p, (Fr *)d_v.p, g.stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipMemcpy(cur, d_v.p, half * sizeof(Fr), hipMemcpyDeviceToHost));
    return LSA_OK;
}
}  // extern "C"

// ---------------------------------------------------------------- pairing
namespace {

// uploads n (P,Q) pairs and runs the Miller loops into the grow-only staging buffer g_pair_f (a
// verifier calls this thousands of times: no hipMalloc / hipFree, which also keeps the device
// from idling down between the upload and the kernel)
int miller_upload_run(const void *g1, const void *g2, size_t n) {
    if (g_pair_p.ensure(n * sizeof(Jac<Fq>)) || g_pair_q.ensure(n * sizeof(Jac<Fq2>)) || g_pair_f.ensure(n * fq12_bytes())) {
        set_error("pairing: hipMalloc failed");
        return LSA_ERR_NOMEM;
    }
    HIPCHK(hipMemcpyAsync(g_pair_p.p, g1, n * sizeof(Jac<Fq>), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(g_pair_q.p, g2, n * sizeof(Jac<Fq2>), hipMemcpyHostToDevice, g.stream));
    return miller_device(g_pair_p.p, g_pair_q.p, n, g_pair_f.p, g.stream);
}

int miller_product_host(const void *g1, const void *g2, size_t n, void *out, bool final_exp) {
    int rc = require_ready();
    if (rc) return rc;
    if (!out || (n && (!g1 || !g2))) { set_error("pairing: null argument"); return LSA_ERR_INVALID; }
    void *res = nullptr;
    if (n == 0) {
        // empty product = 1; final_exponentiation(1) = 1
        Fq12 one = Fq12::one();
        memcpy(out, &one, sizeof one);
        return LSA_OK;
    }
    rc = miller_upload_run(g1, g2, n);
    if (rc) return rc;
    if (g_pair_s.ensure(((n + 7) / 8) * fq12_bytes()) || g_pair_o.ensure(fq12_bytes())) { set_error("pairing: hipMalloc failed"); return LSA_ERR_NOMEM; }
    rc = fq12_product_device(g_pair_f.p, g_pair_s.p, n, &res, g.stream);
    if (rc) return rc;
    if (final_exp) {
        rc = final_exp_device(res, 1, g_pair_o.p, g.stream);
        if (rc) return rc;
        res = g_pair_o.p;
    }
    HIPCHK(hipMemcpyAsync(g.h_result, res, fq12_bytes(), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    memcpy(out, g.h_result, fq12_bytes());
    return LSA_OK;
}
}  // namespace

extern "C" {
int lsa_miller_loop(const void *g1, const void *g2, size_t n, void *out, int on_device) {
    int rc = require_ready();
    if (rc) return rc;
    if (n == 0) return LSA_OK;
    if (!g1 || !g2 || !out) { set_error("miller_loop: null argument"); return LSA_ERR_INVALID; }
    if (on_device) return miller_device(g1, g2, n, out, g.stream);
    rc = miller_upload_run(g1, g2, n);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out, g_pair_f.p, n * fq12_bytes(), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return LSA_OK;
}
int lsa_miller_loop_product(const void *g1, const void *g2, size_t n, void *out) {
    return miller_product_host(g1, g2, n, out, false);
}
int lsa_pairing_product(const void *g1, const void *g2, size_t n, void *out) {
    return miller_product_host(g1, g2, n, out, true);
}
int lsa_fq12_product(const void *in, size_t n, void *out) {
    int rc = require_ready();
    if (rc) return rc;
    if (!out || (n && !in)) { set_error("fq12_product: null argument"); return LSA_ERR_INVALID; }
    if (n == 0) {
        Fq12 one = Fq12::one();
        memcpy(out, &one, sizeof one);
        return LSA_OK;
    }
    DevBuf d_f, d_s;
    void *res = nullptr;
    if (d_f.alloc(n * fq12_bytes()) || d_s.alloc(((n + 7) / 8) * fq12_bytes())) { set_error("fq12_product: hipMalloc failed"); return LSA_ERR_NOMEM; }
    HIPCHK(hipMemcpyAsync(d_f.p, in, n * fq12_bytes(), hipMemcpyHostToDevice, g.stream));
    rc = fq12_product_device(d_f.p, d_s.p, n, &res, g.stream);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(g.h_result, res, fq12_bytes(), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    memcpy(out, g.h_result, fq12_bytes());
    return LSA_OK;
}
int lsa_final_exponentiation(const void *in, size_t n, void *out, int on_device) {
    int rc = require_ready();
    if (rc) return rc;
    if (n == 0) return LSA_OK;
    if (!in || !out) { set_error("final_exponentiation: null argument"); return LSA_ERR_INVALID; }
    if (on_device) return final_exp_device(in, n, out, g.stream);
    if (g_pair_f.ensure(n * fq12_bytes()) || g_pair_s.ensure(n * fq12_bytes())) { set_error("final_exponentiation: hipMalloc failed"); return LSA_ERR_NOMEM; }
    HIPCHK(hipMemcpyAsync(g_pair_f.p, in, n * fq12_bytes(), hipMemcpyHostToDevice, g.stream));
    rc = final_exp_device(g_pair_f.p, n, g_pair_s.p, g.stream);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out, g_pair_s.p, n * fq12_bytes(), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return LSA_OK;
}
}  // extern "C"
