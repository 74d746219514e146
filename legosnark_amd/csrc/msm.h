// legosnark_amd/csrc/msm.h -- internal interface between the C-ABI (capi.hip) and the
// MSM / normalisation kernels (msm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "../../include/legosnark_amd.h"
#include "ec.h"

namespace lsa {

// printf-style error text retrievable through lsa_last_error()
void set_error(const char *fmt, ...);

unsigned msm_window_bits(size_t n);

// Jacobian (libff layout, device) -> affine (device), batch inversion per lane.
template <class F>
int normalize_to_affine(const Jac<F> *d_in, Aff<F> *d_out, size_t n, hipStream_t st);

// Jacobian (libff layout, device) -> the MSM pipeline's device-resident base format
// (G1: 64-B packed 29-bit-limb Montgomery affine; G2: 128-B affine), msm_base_bytes(group)
// per point.
template <class F>
int prepare_bases(const Jac<F> *d_in, void *d_out, size_t n, hipStream_t st);
size_t msm_base_bytes(int group);

// out = sum scalars[i] * bases[first + i]; everything device-resident; asynchronous on `st`.
// table_stride != 0: d_bases is a window-major table of pre-shifted bases
// (entry k*table_stride + i = 2^(s_k) * P_i, k < msm_table_windows(group, table_stride); the
// window starts s_k are a function of table_stride alone) built by precompute_windows(); used
// when n >= msm_merge_min() ("wide windows", msm.hip).
// blocking: the caller synchronises right after this call, so the tail runs on `st` itself (two cross-stream
// hand-overs of ~13 us less on the critical path of a lone call) and d_out may be pinned host memory
template <class F>
int msm_device(const void *d_bases, size_t first, const Fr *d_scalars, size_t n, Jac<F> *d_out, hipStream_t st, size_t table_stride = 0,
               bool blocking = false);
// nseg independent MSMs over prefixes of the same table-carrying bases in one pass (msm.hip):
// result j = sum_i d_scalars[seg_off[j] + i] * bases[first + i], i < seg_off[j+1] - seg_off[j].
template <class F>
int msm_segments_device(const void *d_bases, size_t first, const Fr *d_scalars, const uint64_t *seg_off, size_t nseg, Jac<F> *d_out,
                        hipStream_t st, size_t table_stride);
// G1 and G2 MSM over the same n scalars with ONE shared sort (CommScheme::commit); both base tables
// carry the copies and have table_stride points
int msm_commit_pair_device(const void *d_g1_bases, const void *d_g2_bases, const Fr *d_scalars, size_t n, Jac<Fq> *d_out1,
                           Jac<Fq2> *d_out2, hipStream_t st, size_t table_stride);
unsigned msm_table_windows(int group, size_t n);
unsigned msm_field_mults_per_pair(size_t n, size_t table_n);
size_t msm_merge_min();
bool msm_uses_table(size_t n);    // an MSM of n pairs on a table-carrying handle takes the wide-window pipeline
void msm_set_merge_min(size_t n);
bool msm_merge_min_is_explicit();  // lsa_msm_set_table_threshold(n != 0) is in force
// fills windows 1.. of a table whose window 0 holds the n prepared bases
template <class F>
int precompute_windows(void *d_table, size_t n, hipStream_t st);
// copy k (1 <= k < msm_table_windows) from copy k - 1: no allocation, no synchronisation; d_tmp: n Jacobian points of scratch
template <class F>
int precompute_window_step(void *d_table, size_t n, void *d_tmp, unsigned k, hipStream_t st);

// out[i] = scalars[i] * base (fixed base); d_scalars/d_out device-resident.
template <class F>
int batch_exp_device(const Jac<F> &base, const Fr *d_scalars, size_t n, Jac<F> *d_out, hipStream_t st);
unsigned batch_exp_window_bits(size_t n);
void batch_exp_release();      // the grow-only table workspace (lsa_shutdown)

// scalar_mul.hip: d_out[i] = d_scalars[d_sidx ? d_sidx[i] : i] * d_pts[i] (variable base, G1),
// and per-column sums of a CSC-ordered item array.  Device pointers.
int g1_scalar_mul_device(const Jac<Fq> *d_pts, const Fr *d_scalars, const uint32_t *d_sidx, size_t n, Jac<Fq> *d_out, hipStream_t st);
int g1_column_sums_device(const Jac<Fq> *d_items, const uint64_t *d_col_ptr, size_t ncols, Jac<Fq> *d_out, hipStream_t st);

// fr_vec.hip: Fr streaming kernels (device pointers, asynchronous on st).
int fr_cppoly_fold_device(const Fr *d_v, size_t d, const Fr *d_r, Fr *d_w, Fr *d_tmp, hipStream_t st);
int fr_fold_halves_device(const Fr *d_old, size_t half, const Fr *d_r, Fr *d_cur, hipStream_t st);
int fr_eval_mle_device(const Fr *d_v, size_t d, const Fr *d_r, Fr *d_tmp, Fr *d_out, hipStream_t st);
int fr_sumcheck_round_device(const Fr *d_suff, const Fr *const *d_tables, size_t m, size_t half, const Fr *pre, const Fr *rho,
                             Fr *d_partial, Fr *d_out, hipStream_t st);
size_t fr_sumcheck_scratch_elems();
void fr_vec_release();            // the cached hipGraphs of the recursions (lsa_shutdown)
int fr_scale_upper_device(const Fr *d_old, size_t half, const Fr &k, Fr *d_cur, hipStream_t st);
int fr_eq_table_device(const Fr *d_r, size_t d, int variant, Fr *d_tmp, Fr *d_out, hipStream_t st);   // d_tmp: fr_eq_table_scratch_elems(d)
size_t fr_eq_table_scratch_elems(size_t d);

// ntt.hip: in-place radix-2 NTT of 2^log_n Fr values (device) in at most three passes; d_tmp: 2^log_n scratch elements
// (unused up to 2^10 points); the twiddle tables are cached per domain (ntt_release frees them)
int fr_ntt_device(Fr *d_a, unsigned log_n, const Fr &omega, bool inverse, const Fr *coset, Fr *d_tmp, hipStream_t st);
void ntt_release();
// libfqfft step_radix2_domain (m = 2^big_log + 2^small_log) in place on d_a; d_scratch: 2^big_log elements; omega: primitive
// 2^(big_log + 1)-th root of unity
int fr_ntt_step_device(Fr *d_a, unsigned big_log, unsigned small_log, const Fr &omega, bool inverse, const Fr *coset, Fr *d_scratch, hipStream_t st);

// d_out = sum of n Jacobian points in d_in (device-resident).
template <class F>
int sum_points_device(const Jac<F> *d_in, size_t n, Jac<F> *d_out, hipStream_t st);

// pairing (pairing.hip): all pointers device-resident, asynchronous on `st`.
int miller_device(const void *d_g1_jac, const void *d_g2_jac, size_t n, void *d_out_fq12, hipStream_t st);
int final_exp_device(const void *d_in_fq12, size_t n, void *d_out_fq12, hipStream_t st);
int fq12_product_device(void *d_buf, void *d_scratch, size_t n, void **result, hipStream_t st);
int fq12_segment_products_device(const void *d_in, const uint64_t *d_off, size_t nseg, void *d_out, hipStream_t st);
size_t fq12_bytes();

// msm_compact.hip: a whole MSM of n <= msm_compact_max() pairs over a table-carrying handle in four launches (G1, G2)
template <class F>
int msm_compact_device(const void *d_table, size_t first, const Fr *d_scalars, size_t n, Jac<F> *d_out, hipStream_t st, size_t table_stride,
                       bool blocking = false);
size_t msm_compact_max();
size_t msm_compact_max_g2();
// all pre-shifted copies of n points in one kernel: d_table holds msm_table_windows() * stride packed points, copy 0
// (points [0, n)) filled; d_scratch: table_build_scratch_bytes(n)
int table_build_g1_device(void *d_table, size_t n, size_t stride, void *d_scratch, hipStream_t st);
int table_build_g2_device(void *d_table, size_t n, size_t stride, void *d_scratch, hipStream_t st);
size_t table_build_scratch_bytes(size_t n, int group = 1);

// The tail slots of the MSM pipelines (msm.hip): a call's front runs on the caller's stream, its tail on the slot's
// internal stream with the slot's workspace; results are ordered again at msm_join().
struct MsmSlot {
    hipStream_t tail;     // the slot's stream (the caller's stream itself under LSA_NO_OVERLAP=1)
    void *ws;             // >= the bytes asked for; owned by this call until its tail has run
    void *aux;            // 4 KiB that only msm_compact_device uses: zero at creation, left zero by every call
    int index;
};
// picks the next slot, grows its workspace, makes `st` wait for the slot's previous tail
int msm_slot_begin(hipStream_t st, size_t ws_bytes, const void *d_out, MsmSlot *slot, bool inline_tail = false);
// the front (on st) is issued: the slot's stream continues from here (and behind any earlier tail that writes d_out)
int msm_slot_handover(MsmSlot *slot, hipStream_t st);
// the tail is issued
int msm_slot_end(MsmSlot *slot, hipStream_t st);

// Orders the results of earlier msm_device calls (whose tails run on an internal stream) on `st`.
int msm_join(hipStream_t st);
// Makes `other` wait for them instead and leaves `st` alone.
int msm_join_to(hipStream_t other);

int msm_warmup(hipStream_t st);       // lsa_init: streams, events, function attributes, code object, first workspaces
void msm_release_workspace();
void msm_profile_enable(bool on);
int msm_profile_last(float ms[LSA_MSM_STAGES]);

}  // namespace lsa
