/* include/legosnark_amd.h -- C-ABI of the MI355X (gfx950) implementation of LegoSNARK's
 * elliptic-curve hot path: alt_bn128 G1/G2 multi-scalar multiplication, fixed-base
 * batch exponentiation, and batched Miller-loop / final-exponentiation.
 *
 * The reference has no FFI layer: its gadgets call libff's C++ templates directly
 * (SURVEY.md section 8b).  This header is the boundary a libff-compatible shim binds
 * (legosnark_amd/shim/libff/...; INTEGRATION.md shows the forwarding wrappers).  Every
 * entry point names the reference interface it replaces (file:line under
 * /root/reference).
 *
 * Conventions
 *  - Plain pointers and sizes only; all buffers caller-owned.
 *  - Byte layout = libff's in-memory layout: Fr/Fq = 32 B, 4 x u64 little-endian limbs in
 *    Montgomery form (R = 2^256); G1 = {X,Y,Z} Jacobian 96 B (Z == 0 <=> infinity);
 *    Fq2 = {c0,c1} 64 B; G2 = 192 B; GT = Fq12 = {c0:{c0,c1,c2}, c1:{..}} 384 B.
 *  - Return 0 on success, negative lsa_status on failure; lsa_last_error() gives text.
 *    No exceptions cross the boundary.  There is NO CPU fallback: without a usable
 *    gfx950 device every compute entry point fails with LSA_ERR_NO_DEVICE.
 *  - One process drives one GPU (lsa_init(device)); calls are issued from one host
 *    thread at a time, matching the reference (single caller thread).
 *  - Results are the same group / field elements libff computes.  Jacobian outputs are
 *    valid representatives (compare with libff operator== or after
 *    to_affine_coordinates()); field outputs (GT) are canonical, hence bit-identical.
 */
#ifndef LEGOSNARK_AMD_H
#define LEGOSNARK_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden (it is meant to be linked beside libsnark / libff builds and other
 * C++ code): the entry points declared in this header are its ONLY dynamic symbols (tests/test_capi_symbols.py). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef enum {
    LSA_OK = 0,
    LSA_ERR_NO_DEVICE = -1,   /* no HIP device / wrong arch / lsa_init not called */
    LSA_ERR_INVALID = -2,     /* bad argument */
    LSA_ERR_HIP = -3,         /* HIP runtime error, see lsa_last_error() */
    LSA_ERR_NOMEM = -4
} lsa_status;

/* ---- lifecycle -------------------------------------------------------------------- */
/* Replaces default_ec_pp::init_public_params() as the one-time setup call
 * (src/examples/cplink.cc:81).  Selects `device`, creates the library's HIP stream, and pays here what the first calls
 * of a prover would otherwise pay inside themselves: the code object, the internal streams and their copy-engine queues,
 * the pinned slots of the host-copy path, and workspaces / staging buffers sized for 2^20 pairs of G1 or G2 (about 1 GB
 * of device memory; LSA_WARM=0: everything lazily, LSA_WARM_MB=m: another size) -- a hipFree + hipMalloc pair inside a
 * call costs 0.5 to 10+ ms depending on the box.  A failed warm-up is reported and leaves the library shut down. */
int lsa_init(int device);
void lsa_shutdown(void);
int lsa_device_count(void);
const char *lsa_last_error(void);
/* hipStream_t on which every kernel of this library is launched (for HIP-event timing). */
void *lsa_stream(void);
/* Block until all work queued by this library has finished. */
int lsa_synchronize(void);
/* The tail of an lsa_msm_run_async call (bucket reduction + fold) runs on an internal stream so
 * that it overlaps the next call's front phase.  lsa_stream_join() makes lsa_stream() wait for
 * every tail issued so far: call it before ordering foreign work (another stream's wait, a
 * collective) after lsa_stream().  All synchronous entry points and lsa_*_sum_async join
 * implicitly. */
int lsa_stream_join(void);
/* The same for a stream of the caller's (hipStream_t): it waits for the tails issued so far,
 * lsa_stream() does not -- the next MSM's front keeps overlapping them.  With LSA_NO_OVERLAP=1
 * there are no internal streams and the caller must order `stream` after lsa_stream() itself. */
int lsa_stream_join_to(void *stream);

/* ---- variable-base MSM from host buffers ------------------------------------------- */
/* out = sum_i scalars[i] * bases[i], i < n.
 * Replaces libff::multi_exp_with_mixed_addition<G1<pp>,Fr<pp>,multi_exp_method_BDLO12>
 * (src/utils/globl.h:74-77, multiExpMA) and libff::multi_exp<..BDLO12>
 * (src/utils/sparsemexp.h:58,89).  `chunks` is libff's CPU work split
 * (src/utils/globl.h:67-71; omp_get_max_threads() in the reference's -DMULTICORE=ON build); it does not change the
 * result and is accepted for signature compatibility only.  bases: n x 96 B (G1) / 192 B (G2); scalars: n x 32 B;
 * out: one Jacobian point.
 * Host buffers the library has not been handed before travel through its own pinned slots (a range is copied directly
 * by the runtime only from its third sighting on): the first call on a 2^20-point G1 vector is ~7 ms whatever pages
 * the caller's allocator uses.  Callers should not free large buffers right before a call: unmapping memory the HIP
 * runtime has touched stalls the next submission by 15-25 ms (the libff-compatible shim keeps vectors on the heap). */
int lsa_g1_msm(const void *bases_jac, const void *scalars_mont, size_t n, size_t chunks, void *out_jac);
int lsa_g2_msm(const void *bases_jac, const void *scalars_mont, size_t n, size_t chunks, void *out_jac);

/* CRS cache behind lsa_g1_msm / lsa_g2_msm.  The reference passes the SAME std::vector<G> on
 * every call of a prover (crs->P, src/gadgets/subspace.cc:82; g1s/g2s, src/prototools/commit.h:
 * 154-155; prefixes of g1s, src/gadgets/poly.h:77-86), so vectors of >= 1024 points stay
 * resident on the GPU in prepared form, keyed by host pointer and verified by content before
 * every use.  mode 2 (default; env LSA_CRS_CACHE=full): every byte of the n points passed is
 * fingerprinted (64-point units, host thread pool, overlapped with the scalar upload) -- a
 * vector modified in place is a miss, never a stale result.  mode 1 (LSA_CRS_CACHE=sampled):
 * ~3*log2(n) sampled points are compared; for callers that do not modify CRS vectors in place.
 * mode 0 (LSA_CRS_CACHE=0): every call uploads and normalises its bases.  Vectors below 1024 points are not kept,
 * but are recognised as prefixes of a kept one (whole 64-point units, or fewer than 64 points) and then served from it.  Entries are evicted
 * least-recently-used beyond max_bytes of device memory (0 keeps the current budget; default
 * min(1/4 of the device, 64 GiB), env LSA_CRS_CACHE_MB).  The first re-use of an entry of at
 * least the table threshold that has been hit often enough also STARTS building its pre-shifted window copies
 * (lsa_crs_cache_table_after). */
int lsa_crs_cache_configure(int mode, size_t max_bytes);
/* The pre-shifted copies of an entry are built in the BACKGROUND (a low-priority stream, 26 copies ~ 27 ms of GPU time
 * at 2^20 G1 points, 64 ms for G2) once the entry has been hit `hits` times (default 23, the break-even of 27 ms
 * against the ~1.2 ms a 2^20-pair MSM saves with them; 0: never; env LSA_CRS_TABLE_AFTER): the call that starts the
 * build and the calls during it run at plain-pipeline speed, the entry switches to the copies when they are complete
 * -- results are the same point either way.  They hold 26 x 64 B per G1 point (128 B per G2 point).  A prover that
 * knows it will re-use a key many times sets 1. */
int lsa_crs_cache_table_after(unsigned hits);
/* Blocks until every background build has finished and its entry has switched (tests, benchmarks). */
int lsa_crs_cache_wait_tables(void);
void lsa_crs_cache_clear(void);
int lsa_crs_cache_stats(uint64_t *hits, uint64_t *misses, uint64_t *resident_bytes, uint64_t *entries);
/* Host-side wall-clock split of the most recent lsa_g1_msm / lsa_g2_msm call, as the caller of
 * multiExpMA experiences it (SURVEY.md 8e "Host staging": H2D reported separately). */
typedef struct {
    size_t n;
    int cache_hit;               /* 1: bases were resident */
    int table;                   /* 1: the call used the entry's pre-shifted window copies; 2: the copies of the entry's first
                                  * points only (a G1 entry without full copies serving a request of <= 2^16 pairs: built in one
                                  * kernel on the first such request, LSA_CRS_PREFIX_TABLE) */
    double h2d_scalars_ms;       /* pageable host -> device copy of n x 32 B */
    double fingerprint_wait_ms;  /* what was left of the fingerprint pass after that copy */
    double bases_prepare_ms;     /* miss: upload + normalise (+ table on first re-use); hit: lookup */
    double msm_ms;               /* kernels + 96/192-byte result copy */
    double total_ms;
    int table_building;          /* 1: the entry's pre-shifted copies are being built in the background */
} lsa_host_stats;
int lsa_msm_host_stats(lsa_host_stats *out);

/* ---- device-resident bases (CRS) ---------------------------------------------------- */
/* The CRS vectors passed to multiExpMA are fixed per key (crs->P at
 * src/gadgets/subspace.cc:82, g1s/g2s at src/prototools/commit.h:154-155), so they are
 * uploaded once, batch-normalised to affine in HBM (64 B / 128 B per point) and reused.
 * `bases_jac` may be a host pointer or (src_on_device != 0) a device pointer.
 * Handles of >= 2^19 points (LSA_PRECOMPUTE_MIN, lsa_msm_set_table_threshold) also keep 26 (24) pre-shifted copies of
 * every point (lsa_bases_has_table); so do G1 handles of up to 2^16 points -- their copies come from ONE kernel, ~1.2 ms --
 * unless a threshold was set explicitly.  G1 MSMs of up to 2^17 pairs over such a handle take a four-launch pipeline
 * (csrc/msm_compact.hip: 0.12-0.25 ms blocking for n <= 2^14). */
typedef struct lsa_bases lsa_bases;
int lsa_g1_bases_create(const void *bases_jac, size_t n, int src_on_device, lsa_bases **out);
int lsa_g2_bases_create(const void *bases_jac, size_t n, int src_on_device, lsa_bases **out);
void lsa_bases_destroy(lsa_bases *b);
size_t lsa_bases_size(const lsa_bases *b);
/* Resident vectors of at least 2^19 points (env LSA_PRECOMPUTE_MIN) also keep 26 pre-shifted
 * copies 2^(pos_j)*P_i at bit positions about 10 apart (24 copies about 11 apart from 6*2^20
 * points on; G1 64 B, G2 128 B per point and copy; LSA_PRECOMPUTE=0 disables).  Every MSM on such a handle then runs over ONE bucket space shared by
 * all windows: 13 (12) bucket additions per pair instead of 16 and no GLV beta-multiplications for
 * n >= 2^16, 26 narrow digits and only 512 buckets to reduce below; one bucket reduction, no
 * Horner fold.  lsa_msm_set_table_threshold(t), t != 0: vectors of at least t points get the
 * copies and only MSMs of at least t pairs use them (tests); 0 restores the defaults.
 * lsa_bases_has_table() tells whether a handle carries the copies. */
void lsa_msm_set_table_threshold(size_t threshold);
int lsa_bases_has_table(const lsa_bases *b);
unsigned lsa_bases_table_windows(const lsa_bases *b);   /* 0 without the copies */
/* Field multiplications per point-scalar pair in the bucket-accumulation kernel for an MSM of n
 * pairs on this handle (NULL: host-buffer path) -- the operation count behind bench.py's
 * integer-throughput roofline: 10 per bucket addition (XYZZ mixed add, 8M + 2S), plus one
 * beta-multiplication per GLV half on the plain path. */
unsigned lsa_msm_field_mults_per_pair(const lsa_bases *b, size_t n);
/* Device pointer to the normalised affine array (n x 64 B / 128 B), for inspection. */
const void *lsa_bases_device_ptr(const lsa_bases *b);

/* MSM over bases[first .. first+n) with scalars already resident in HBM
 * (d_scalars_mont: device pointer, n x 32 B, Montgomery Fr).  out_jac is a HOST buffer
 * (96 B / 192 B); the call returns after the result has landed there (the whole call runs on lsa_stream() and its last
 * kernel writes into pinned host memory: no cross-stream hand-over, no copy command behind the kernels). */
int lsa_msm_run(const lsa_bases *bases, size_t first, const void *d_scalars_mont, size_t n, void *out_jac);
/* Same, asynchronous: result is written to the DEVICE buffer d_out_jac; it is ordered on
 * lsa_stream() after lsa_stream_join() (or any synchronous call). */
int lsa_msm_run_async(const lsa_bases *bases, size_t first, const void *d_scalars_mont, size_t n, void *d_out_jac);

/* Several independent MSMs over PREFIXES of the same resident bases in one pass -- the ladder
 * of CPPoly::prove (src/gadgets/poly.h:77-86: witness[i] = multiExpMA(g1s, w[start_i ..
 * start_i + 2^(d-1-i))), whose scalar slices are consecutive in one array):
 *   d_out_jac[j] = sum_i d_scalars[seg_offsets[j] + i] * bases[first + i],
 *                  i < seg_offsets[j+1] - seg_offsets[j],      j < nseg <= 64.
 * seg_offsets: HOST array of nseg + 1 non-decreasing element offsets into d_scalars_mont (device).
 * d_out_jac: DEVICE, nseg points (infinity for an empty segment).  The bases must carry the
 * pre-shifted copies (lsa_bases_has_table).  Meant for segments of up to about 2^16 pairs (each
 * is cut into 26 ten-bit digits over 512 buckets of its own); longer ones are better issued
 * through lsa_msm_run_async.  Asynchronous like lsa_msm_run_async. */
int lsa_msm_run_segments_async(const lsa_bases *bases, size_t first, const void *d_scalars_mont, const uint64_t *seg_offsets, size_t nseg,
                               void *d_out_jac);

/* CommScheme::commit (src/prototools/commit.h:149-158; CPPoly::commitPoly, src/gadgets/poly.h:30-32):
 *   c = multiExpMA<LG1>(g1s, v),   kc = multiExpMA<LG2>(g2s, v)
 * -- a G1 and a G2 MSM over the SAME scalar vector.  The scalar sort (digits, ranks, scatter) does
 * not depend on the bases, so it runs once for both when the two handles carry pre-shifted copies
 * over the same number of points (otherwise this is two lsa_msm_run_async calls).  bases[0 .. n)
 * of both handles; d_scalars_mont, d_out_g1 (96 B), d_out_g2 (192 B): DEVICE.  Asynchronous like
 * lsa_msm_run_async. */
int lsa_commit_run_async(const lsa_bases *g1_bases, const lsa_bases *g2_bases, const void *d_scalars_mont, size_t n, void *d_out_g1,
                         void *d_out_g2);

/* Window width (bits) the MSM uses for n pairs -- exposed for DESIGN.md / tests. */
unsigned lsa_msm_window_bits(size_t n);

/* ---- per-stage timing (HIP events on lsa_stream()) ----------------------------------- */
#define LSA_MSM_STAGES 8
/* 0 digits+histogram, 1 scan, 2 scatter (+ fine sort), 3 bucket accumulate (dominant; the
 * accumulate kernels alone), 4 bucket reduce, 5 window fold, 6 ordering of the buckets by
 * population (between scatter and accumulate), 7 total */
int lsa_profile_enable(int on);   /* also resets the recorded-call counter */
/* Average milliseconds per stage over the MSM calls (at most 64) recorded since
 * lsa_profile_enable(1); synchronises on their events; returns the number of calls. */
int lsa_profile_last_msm(float ms[LSA_MSM_STAGES]);

/* ---- fixed-base batch exponentiation ------------------------------------------------ */
/* out[i] = scalars[i] * base, i < n.  Replaces libff get_window_table + batch_exp as used
 * by cputil::simpleBatchExp (src/utils/util.h:119-134) and Interpolator::mkG1Exp/mkG2Exp
 * (src/prototools/interp.h:36-59).  base: one HOST Jacobian point; scalars: n x 32 B
 * Montgomery Fr; out: n Jacobian points (valid representatives, not normalised).
 * `on_device` != 0: scalars and out are device pointers, else host pointers.
 * libff builds the window table once per base and calls batch_exp many times over it; this interface takes the base
 * with every call, so the device tables of the last four bases are kept (keyed by the base's bytes and the window
 * width, LSA_BEXP_TABLES): the first call on a base pays its 254 dependent doublings (~1.0 ms G1 / 2.6 ms G2),
 * later ones ~0.08 / 0.15 ms for a handful of scalars.  The result is complete when the call returns. */
int lsa_g1_batch_exp(const void *base_jac, const void *scalars_mont, size_t n, void *out_jac, int on_device);
int lsa_g2_batch_exp(const void *base_jac, const void *scalars_mont, size_t n, void *out_jac, int on_device);

/* ---- sum of points --------------------------------------------------------------------- */
/* out = sum_i pts[i]; pts: n Jacobian points on the DEVICE, d_out_jac: DEVICE (async on
 * lsa_stream()).  Folds the per-GPU MSM partials after the RCCL all-gather that replaces
 * libff's `final = final + partial[i]` loop over chunks (multi_exp). */
int lsa_g1_sum_async(const void *d_pts_jac, size_t n, void *d_out_jac);
int lsa_g2_sum_async(const void *d_pts_jac, size_t n, void *d_out_jac);
/* The same on a stream of the caller's (hipStream_t) instead of lsa_stream(). */
int lsa_g1_sum_on(const void *d_pts_jac, size_t n, void *d_out_jac, void *stream);
int lsa_g2_sum_on(const void *d_pts_jac, size_t n, void *d_out_jac, void *stream);

/* ---- variable-base batch scalar multiplication / sparse matrix in the exponent ---------- */
/* out[i] = scalars[i] * pts[i], i < n: the independent 254-bit scalar multiplications of
 * CPlink key generation -- `acc*one` and the one- and two-term multi_exp calls of sparsemexpG
 * (src/utils/sparsemexp.h:62-90) -- and of init_as_random-style input generation
 * (src/examples/cplink.cc:51-58) when the bases differ.  pts: libff G1 (96 B), scalars: Fr
 * (32 B), out: libff G1.  on_device != 0: device pointers, else host pointers. */
int lsa_g1_scalar_mul_batch(const void *pts_jac, const void *scalars_mont, size_t n, void *out_jac, int on_device);
/* mtxmultiexp(out, exps, M) for a column-major sparse matrix of G1 elements
 * (src/gadgets/subspace.cc:18-25 -> simplesparsemexp, src/utils/sparsemexp.cc:15-24):
 *   out[j] = sum_{e = col_ptr[j]}^{col_ptr[j+1]-1} exps[rows[e]] * vals[e],   j < ncols.
 * CSC layout: vals (nnz x 96 B libff G1) and rows (nnz x u32, the CoeffPos::pos of
 * src/utils/matrix.h:35-42) ordered by column, col_ptr (ncols+1 x u64, col_ptr[0] = 0,
 * col_ptr[ncols] = nnz).  exps: nrows x Fr.  out: ncols x libff G1 (infinity for an empty
 * column).  Host pointers.  Row indices >= nrows are rejected (LSA_ERR_INVALID). */
int lsa_g1_sparse_matrix_msm(const void *vals_jac, const uint32_t *rows, const uint64_t *col_ptr, size_t ncols,
                             const void *exps_mont, size_t nrows, void *out_jac);

/* ---- Fr vectors around the MSMs --------------------------------------------------------- */
/* Witness coefficients of CPPoly::prove (src/gadgets/poly.h:55-67): with v of length 2^d and the
 * evaluation point r of length d, round i = 0..d-1 over m = 2^(d-1-i) pairs writes
 *   w[start + p] = v[2p+1] - v[2p],   v'[p] = -v[2p]*(r[i]-1) + v[2p+1]*r[i],   start += m.
 * w has 2^d entries (the last one stays zero, as in the reference's value-initialised vector);
 * w[start .. start+m) is the scalar vector of the i-th MSM of the proof (poly.h:77-86) and can
 * be passed to lsa_msm_run_async without leaving the device.  v is not modified.
 * on_device != 0: v, r, w are device pointers; else host pointers. */
int lsa_fr_cppoly_witness(const void *v_mont, size_t d, const void *r_mont, void *w_mont, int on_device);
/* out = evalMLE(v, r) (MultiVPolyT::evalMLE, src/prototools/polytools.h:207-234): the
 * multilinear extension of v (2^d entries, index bit i pairs with r[i]) at r.  One Fr.
 * on_device != 0: v, r, out are device pointers; else host pointers. */
int lsa_fr_eval_mle(const void *v_mont, size_t d, const void *r_mont, void *out_mont, int on_device);
/* DPMle::pushRandomness (src/prototools/mle.h:199-210): cur[p] = old[p]*(1-r) + old[p+half]*r,
 * p < half; old has 2*half entries, r is ONE Fr; cur may alias old.  on_device != 0: all three
 * are device pointers and the call is asynchronous on lsa_stream(); else host pointers. */
int lsa_fr_fold(const void *old_mont, size_t half, const void *r_mont, void *cur_mont, int on_device);

/* One round polynomial of the sumcheck prover (CPSumcheck::make_new_h_poly,
 * src/gadgets/sumcheck.h:85-106):
 *   h_j(X) = sum_{p < half} betaPoly(j,p)(X) * prod_{t < m} mlePoly_t(j,p)(X),
 * mlePoly_t(j,p) = tables[t][p]*(1-X) + tables[t][p+half]*X (DPMle::getMLEPoly, mle.h:217-226; the
 * tables are the DPMle curVTable's, 2*half entries each), betaPoly(j,p) =
 * eqbit_poly(rho_j) * pre * suff[p] (DPBeta::getBetaPoly, mle.h:74-82: pre = getBetaPre(j-1),
 * suff = beta_suff_rho_cur, half entries, NULL = all ones as for j+1 > d-1).  rho_j == NULL drops
 * the beta factor altogether (DPBetaDummy, mle.h:148-166).  1 <= m <= 4.
 * out_coeffs (HOST): m+2 coefficients of h_j, lowest degree first (m+1 when rho_j == NULL).
 * pre, rho_j: HOST pointers to one Fr each.  on_device != 0: suff and tables[t] are device
 * pointers (the array `tables` itself is a host array of m pointers); else host pointers. */
int lsa_fr_sumcheck_round(const void *suff_mont, const void *const *tables, size_t m, size_t half, const void *pre_mont,
                          const void *rho_j_mont, void *out_coeffs_mont, int on_device);
/* DPBeta::pushRandomness, suffix table (src/prototools/mle.h:46-53): cur[p] = old[half + p] * k,
 * p < half (k = rhoInvs[j+1], ONE Fr, HOST pointer); cur may alias old.  on_device != 0: old,
 * cur are device pointers, asynchronous on lsa_stream(); else host pointers. */
int lsa_fr_scale_upper(const void *old_mont, size_t half, const void *k_mont, void *cur_mont, int on_device);
/* DPBeta::compute_eq_tbl (src/prototools/mle.h:93-105), the table DPBeta::precomputeAll (mle.h:121-137) starts from;
 * 2^d entries, 1 <= d <= 30, eqbit(1, x) = x, eqbit(0, x) = 1 - x (mle.cc:12-15).
 *   variant 0 -- what the reference's loop computes: its doubling step reads dst[p >> 1], so every factor is selected by
 *                the TOP bit of p:  out[p] = prod_j (1 - r[j]) for p < 2^(d-1),  prod_j r[j] above.  Use this one to
 *                reproduce the reference's proofs.
 *   variant 1 -- the table the loop's comment describes:  out[p] = prod_{j < d} eqbit(bit j of p, r[j]).
 * The prover's first suffix table is then lsa_fr_scale_upper(out, 2^(d-1), rhoInvs[0], suff) (mle.h:132-137).
 * on_device != 0: r (d entries) and out are device pointers and the call is asynchronous on lsa_stream(); else host
 * pointers. */
int lsa_fr_eq_table(const void *r_mont, size_t d, int variant, void *out_mont, int on_device);

/* In-place radix-2 NTT of 2^log_n Fr values, natural order in and out: replaces libfqfft
 * basic_radix2_domain<Fr>::FFT / iFFT / cosetFFT / icosetFFT as used by the Lipmaa gadget
 * (src/gadgets/lipmaa.cc:68-81,102-175).  omega: a primitive 2^log_n-th root of unity (HOST, one
 * Fr; libff::get_root_of_unity).  inverse == 0: a[k] <- sum_i a[i] omega^(ik), after a[i] *= g^i
 * when coset_g != NULL (cosetFFT).  inverse != 0: the same with omega^-1, then a[k] *= 1/n and,
 * with coset_g, *= g^-k (iFFT / icosetFFT).  coset_g: HOST, one Fr, or NULL.
 * on_device != 0: a is a device pointer; else a host pointer.  log_n <= 28.
 * At most three passes over the data (csrc/ntt.hip); the twiddle tables of the last eight domains (log_n, omega,
 * direction; each up to one vector's size while that is at most LSA_NTT_T1_MB = 512 MB) and of the last eight coset
 * generators stay on the device, least recently used first out once they hold more than LSA_NTT_CACHE_MB (default
 * 2048) together; a second vector of the same size is kept as scratch. */
int lsa_fr_ntt(void *a_mont, size_t log_n, const void *omega_mont, int inverse, const void *coset_g_mont, int on_device);

/* The same four transforms on libfqfft's step_radix2_domain<Fr> of m = 2^big_log + 2^small_log points, small_log <
 * big_log <= 27: the domain libfqfft's get_evaluation_domain selects for every size that is not a power of two
 * (src/prototools/interp.h:62, src/gadgets/lipmaa.cc:102 with such an n) [upstream, recalled:
 * libfqfft/evaluation_domain/domains/step_radix2_domain.tcc].  Evaluation points: omega^(2k) for k < 2^big_log, then
 * omega sigma^j for j < 2^small_log, sigma = omega^(2^(big_log + 1 - small_log)).  a: m Fr values, in place, coefficients
 * <-> values in that order.  omega: a primitive 2^(big_log + 1)-th root of unity (HOST, one Fr: the domain's `omega`,
 * libff::get_root_of_unity(2^(big_log + 1))).  inverse / coset_g / on_device as for lsa_fr_ntt. */
int lsa_fr_ntt_step(void *a_mont, size_t big_log, size_t small_log, const void *omega_mont, int inverse, const void *coset_g_mont, int on_device);

/* ---- pairing ---------------------------------------------------------------------------- */
/* out[i] = miller_loop(precompute_G1(P_i), precompute_G2(Q_i)), i < n: replaces libff
 * alt_bn128_pp::precompute_G1 / precompute_G2 / miller_loop (src/utils/globl.h:96-102,
 * src/gadgets/subspace.cc:142-160, src/gadgets/poly.h:105-119).  P_i: libff G1 (96 B),
 * Q_i: libff G2 (192 B), out: n x Fq12 (384 B).  on_device != 0: all three are device
 * pointers, else host pointers. */
int lsa_miller_loop(const void *g1_jac, const void *g2_jac, size_t n, void *out_fq12, int on_device);
/* out = prod_i miller_loop(P_i, Q_i) (one Fq12, HOST): n = 2 is libff double_miller_loop
 * (src/gadgets/subspace.cc:147-163, src/gadgets/lipmaa.cc:187-207). Host pointers. */
int lsa_miller_loop_product(const void *g1_jac, const void *g2_jac, size_t n, void *out_fq12);
/* out = prod_i in[i] (one Fq12, HOST; 1 for n = 0): the GT/Fqk operator* chain of the verifiers
 * (src/gadgets/subspace.cc:163-166, src/gadgets/poly.h:108-110) and the fold of the per-GPU
 * partial products when a pairing batch is split across ranks.  Host pointers. */
int lsa_fq12_product(const void *in_fq12, size_t n, void *out_fq12);
/* out[i] = final_exponentiation(in[i]): replaces alt_bn128_pp::final_exponentiation
 * (src/utils/globl.h:103, src/gadgets/subspace.cc:166, src/gadgets/poly.h:110,122). */
int lsa_final_exponentiation(const void *in_fq12, size_t n, void *out_fq12, int on_device);
/* out = final_exponentiation(prod_i miller_loop(P_i, Q_i)) (one GT element, HOST).  n = 1 is
 * reduced_pairing (src/gadgets/subspace.cc:88-102,123-124); the batched form is the CPhad /
 * CPsc verifier shape (BASELINE.json configs[4]).  Host pointers.
 * Wherever a final exponentiation follows inside the library (this call, lsa_pairing_terms / _segments / _sharded with
 * final_exp) the Miller loops of pairs without a resident line table may walk the SIGNED digits of 6u + 2 (65 doubling and
 * 21 addition steps instead of 64 and 36): their Miller values differ from libff's by vertical lines, which the final
 * exponent kills, so every GT value returned is libff's bit for bit -- for every input: workgroups that hold a point at
 * infinity in G2 or a pair off its curve (where libff's loop is still defined, as the value of its formulas, and the identity
 * is not) keep libff's binary loop.  Raw Miller values (lsa_miller_loop*, final_exp == 0) always come from the binary loop.
 * Env LSA_MILLER_NAF=0 switches the signed-digit loop off. */
int lsa_pairing_product(const void *g1_jac, const void *g2_jac, size_t n, void *out_gt);

/* ---- G2 precomputation (libff G2_precomp) ------------------------------------------------- */
/* libff's alt_bn128_ate_G2_precomp as bytes: QX, QY (the affine point), then the coefficient triple
 * {ell_0, ell_VW, ell_VV} of each of the 102 steps of the ate loop (64 doublings, 36 additions, 2
 * Frobenius steps), 308 Fq2 of 64 B in libff's layout. */
#define LSA_ATE_NUM_COEFFS 102
#define LSA_G2_PRECOMP_BYTES ((2 + 3 * LSA_ATE_NUM_COEFFS) * 64)
size_t lsa_g2_precomp_bytes(void);   /* == LSA_G2_PRECOMP_BYTES */
/* out_precomp[i] = precompute_G2(Q_i), i < n: replaces libff alt_bn128_pp::precompute_G2 where the
 * reference keeps the result in a key and re-uses it in every verification (src/gadgets/subspace.cc:48,
 * 66-70: C_precomp, a_precomp; src/gadgets/lipmaa.h:95-96: gammazg2_precomp; src/gadgets/poly.h:97-98).
 * Q_i: libff G2 (192 B), out: n x LSA_G2_PRECOMP_BYTES.  Host pointers. */
int lsa_g2_precompute(const void *g2_jac, size_t n, void *out_precomp);
/* out[i] = miller_loop(precompute_G1(P_i), *q_precomp[i]), i < n, over precomputed G2 values
 * (src/gadgets/subspace.cc:152-166 verifyLin3or4, src/gadgets/lipmaa.cc:194-200): only the Fq12 chain
 * f <- f^2 * line(P) runs, 166 dependent rounds instead of 344.  q_precomp: n host pointers to
 * LSA_G2_PRECOMP_BYTES each (equal pointers / equal contents share one device table).  Host pointers. */
int lsa_miller_loop_precomp(const void *g1_jac, const void *const *q_precomp, size_t n, void *out_fq12);
/* The general form every pairing entry point is a shape of:
 *   out[j] = [final_exponentiation] ( prod_{i = seg_offsets[j]}^{seg_offsets[j+1]-1} m_i ),  j < nseg,
 *   m_i = miller_loop(P_i, Q_i), or its conjugate (libff unitary_inverse) when flags[i] & 1.
 * Q_i is *q_precomp[i] when q_precomp and q_precomp[i] are non-null, else the point g2_jac[i].
 * flags may be null (no conjugates).  A verifier's whole check
 *   final_exponentiation(lhs * rhs.unitary_inverse()) == GT::one()
 * (src/utils/globl.h:94-105, src/gadgets/subspace.cc:166, src/gadgets/lipmaa.cc:203, src/gadgets/poly.h:110,122)
 * is ONE product here: the pairs of a product share accumulators on the device, conjugated terms run as
 * Miller loops on -P (the same Fq12 element).  Host pointers. */
int lsa_pairing_terms(const void *g1_jac, const void *g2_jac, const void *const *q_precomp, const uint8_t *flags,
                      const uint64_t *seg_offsets, size_t nseg, void *out_fq12, int final_exp);
/* The device keeps one line table (22 KB) per distinct Q it has seen -- keyed by a 128-bit fingerprint of
 * the bytes passed (the 192-byte point, or the precomp blob) -- so that a CRS element costs its G2
 * arithmetic once.  max_tables: capacity (LRU; default 4096, env LSA_G2_TABLES); 0 turns the cache off
 * (every call recomputes its tables).  Calls with more than 1024 terms or device-resident points bypass
 * it.  stats: hits, misses, resident tables, evictions. */
int lsa_g2_table_cache(size_t max_tables);
/* Promises the tables of the points the cache does not hold yet and returns without waiting (they are built in one
 * launch when the next Miller call arrives, or earlier once LSA_G2_PREFETCH_BATCH = 60 of them are waiting): what
 * libff's precompute_G2 costs the caller when the result is only ever handed back to miller_loop
 * (src/gadgets/poly.h:116-118: precompute_G2(pts[i] * g2) inside a host loop).  Host pointer. */
int lsa_g2_tables_prefetch(const void *g2_jac, size_t n);
/* Pairs of one product that share an accumulator f <- f^2 * prod_i line_i (1..4; 0 = chosen from the
 * batch size: 1 up to 8192 terms, so that every pair has its own lanes while the chip is not full).
 * The value of a product does not depend on it. */
int lsa_pairing_set_chunk(unsigned pairs_per_accumulator);
int lsa_g2_table_cache_stats(uint64_t out[4]);

/* ---- multi-GPU: one process per GPU, one exchange step per MSM (RCCL over xGMI) --------------- */
/* libff's multi_exp splits [0, n) into `chunks` contiguous ranges, runs multi_exp_inner on each
 * and sums the partials; multiExpMA forwards `chunks` (src/utils/globl.h:67-77).  Here a chunk is
 * a GPU and a rank is a process: every rank runs the single-GPU pipeline on its slice and the
 * 96-byte (G1) / 192-byte (G2) Jacobian partials are combined with ONE ncclAllGather + a sum in
 * rank order (RCCL has no elliptic-curve reduce op).  Every rank ends up with the same point.
 *
 * lsa_comm_unique_id: rank 0 fills 128 bytes (an ncclUniqueId) that the caller distributes to
 * the other ranks over its own transport (MPI, a torch.distributed broadcast, a file ...).
 * lsa_comm_init: collective over all ranks; binds the communicator to the device of lsa_init.
 * lsa_comm_init_file: single-node bootstrap without a transport: rank 0 writes the id to `path`,
 * the others poll for it (timeout_s <= 0: 60 s).  The file must not exist beforehand. */
#define LSA_COMM_ID_BYTES 128
int lsa_comm_unique_id(void *out_id128);
int lsa_comm_init(int rank, int world, const void *id128);
int lsa_comm_init_file(int rank, int world, const char *path, int timeout_s);
void lsa_comm_destroy(void);
int lsa_comm_rank(void);    /* 0 without a communicator */
int lsa_comm_world(void);   /* 1 without a communicator */
/* libff's chunk split: one = n / world, the last rank takes the remainder (n < world: rank 0
 * takes everything).  The range of (base, scalar) pairs rank `rank` owns. */
void lsa_shard_range(size_t n, int world, int rank, size_t *lo, size_t *hi);
/* This rank's part of a sharded MSM over its resident bases[first .. first+n) (device scalars),
 * then the exchange step on an internal side stream; d_out_jac (DEVICE) receives the sum over
 * all ranks.  Asynchronous: lsa_stream() may start the next call's front meanwhile;
 * lsa_comm_join() orders every exchange issued so far on lsa_stream(). */
int lsa_msm_run_sharded_async(const lsa_bases *bases, size_t first, const void *d_scalars_mont, size_t n, void *d_out_jac);
int lsa_comm_join(void);
/* The same, blocking, result in a HOST buffer. */
int lsa_msm_run_sharded(const lsa_bases *bases, size_t first, const void *d_scalars_mont, size_t n, void *out_jac);
/* multiExpMA for an SPMD prover: every rank passes ITS slice of the vectors (host buffers, the
 * range lsa_shard_range gives it; the CRS cache applies) and receives the sum over all ranks.
 * Without a communicator (world 1) identical to lsa_g1_msm / lsa_g2_msm. */
int lsa_g1_msm_sharded(const void *bases_jac, const void *scalars_mont, size_t n_local, void *out_jac);
int lsa_g2_msm_sharded(const void *bases_jac, const void *scalars_mont, size_t n_local, void *out_jac);
/* final_exponentiation(prod over all ranks' pairs): per-rank Miller product, all-gather of the
 * 384-byte Fq12 partials, product in rank order, one final exponentiation on every rank. */
int lsa_pairing_product_sharded(const void *g1_jac, const void *g2_jac, size_t n_local, void *out_gt);
/* Building block: all-gather of one device-resident partial per rank (kind 1: G1 point, 2: G2
 * point, 12: Fq12) into d_gathered (world x the same), ordered on lsa_stream(). */
int lsa_comm_all_gather(const void *d_partial, void *d_gathered, int kind);

/* Many independent pairing products in one pass -- a verifier's shape (CPPoly::verify,
 * src/gadgets/poly.h:105-122: d products of two or three Miller values, each followed by a final
 * exponentiation; CPhad verify at d = 20: 181 Miller loops in 62 products):
 *   out[j] = [final_exponentiation] ( prod_{i = seg_offsets[j]}^{seg_offsets[j+1]-1} miller_loop(P_i, Q_i) ),  j < nseg
 * (Fq12 one for an empty segment).  seg_offsets: nseg + 1 non-decreasing offsets, the first 0.
 * One upload, one Miller launch over all pairs, one product workgroup per segment, one batched
 * final exponentiation, one download.  final_exp == 0 leaves the Miller products
 * (double_miller_loop values).  Host pointers. */
int lsa_pairing_product_segments(const void *g1_jac, const void *g2_jac, const uint64_t *seg_offsets, size_t nseg, void *out_gt, int final_exp);

/* ---- point helpers ------------------------------------------------------------------- */
/* Jacobian -> libff "special" form (affine with Z = 1, or (0,1,0)), n points, host
 * buffers; replaces G::to_affine_coordinates()/batch_to_special on result vectors. */
int lsa_g1_normalize(const void *in_jac, size_t n, void *out_jac);
int lsa_g2_normalize(const void *in_jac, size_t n, void *out_jac);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
