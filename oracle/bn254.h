/* oracle/bn254.h -- CPU restatement of the libff algorithms that LegoSNARK's hot
 * path bottoms out in (alt_bn128 G1/G2 multi_exp, batch_exp, ate pairing).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under legosnark_amd/ or include/ may
 * include, link or call this; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, as the checker / reported baseline.
 *
 * PARITY PINNING -- "parity unpinned" in the strict sense: the reference holds no golden
 * vectors or known-answer tests for this path and cannot be built or run here, so nothing below
 * is checked against bytes produced by libff itself.  What pins it instead is listed under
 * (i)-(iii).  libff / libsnark are un-vendored submodules of the reference
 * (/root/reference/.gitmodules:1-6, depends/libsnark is empty, pinned commit
 * unknown), so there is no reference binary and the reference holds no golden
 * vectors.  This restatement follows libff's published
 * algorithms [upstream, recalled] and the in-tree mirror of the filter stage
 * (/root/reference/src/utils/sparsemexp.h:12-59).  It is pinned by
 *   (i)  tests/golden/ *.json produced by the independent affine big-int model
 *        oracle/pymodel/bn254_model.py (tests/golden/make_golden.py),
 *   (ii) public alt_bn128 known answers (2*G1, r*G1 = O, r*G2 = O),
 *   (iii) algebraic identities (bilinearity, known-discrete-log MSM).
 *
 * Memory layout = libff's: Fq/Fr = 4 x u64 little-endian limbs in Montgomery
 * form (R = 2^256); G1 = {X,Y,Z} Jacobian 96 B, infinity <=> Z == 0;
 * Fq2 = {c0,c1}; G2 = 192 B; Fq12 = {c0:Fq6{c0,c1,c2}, c1:Fq6} 384 B.
 */
#ifndef ORACLE_BN254_H
#define ORACLE_BN254_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } ofp_t;           /* element of Fq or Fr */
typedef struct { ofp_t c0, c1; } ofq2_t;
typedef struct { ofq2_t c0, c1, c2; } ofq6_t;
typedef struct { ofq6_t c0, c1; } ofq12_t;
typedef struct { ofp_t X, Y, Z; } og1_t;
typedef struct { ofq2_t X, Y, Z; } og2_t;

/* G2 line coefficients, libff alt_bn128_ate_ell_coeffs {ell_0, ell_VW, ell_VV}. */
typedef struct { ofq2_t ell_0, ell_VW, ell_VV; } oell_t;
#define ORACLE_ATE_NUM_COEFFS 102 /* 64 doublings + 36 additions (popcount(6u+2)-1) + 2 Frobenius steps */
typedef struct { ofp_t PX, PY; } og1_precomp_t;
typedef struct { ofq2_t QX, QY; oell_t coeffs[ORACLE_ATE_NUM_COEFFS]; } og2_precomp_t;

/* ---- fields (which = 0: Fq, 1: Fr) ---- */
void ofp_mul(ofp_t *r, const ofp_t *a, const ofp_t *b, int which);
void ofp_add(ofp_t *r, const ofp_t *a, const ofp_t *b, int which);
void ofp_sub(ofp_t *r, const ofp_t *a, const ofp_t *b, int which);
void ofp_inv(ofp_t *r, const ofp_t *a, int which);
void ofp_from_canonical(ofp_t *r, const uint64_t x[4], int which); /* -> Montgomery */
void ofp_to_canonical(uint64_t x[4], const ofp_t *a, int which);   /* libff as_bigint */

void ofq12_mul(ofq12_t *r, const ofq12_t *a, const ofq12_t *b);
void ofq12_one(ofq12_t *r);
void ofq12_unitary_inverse(ofq12_t *r, const ofq12_t *a);
void ofq12_inverse(ofq12_t *r, const ofq12_t *a);
void ofq12_frobenius(ofq12_t *r, const ofq12_t *a, unsigned power);
void ofq12_pow_u64(ofq12_t *r, const ofq12_t *a, uint64_t e);
void ofq12_pow_p(ofq12_t *r, const ofq12_t *a); /* generic a^p by square-and-multiply (test of Frobenius) */

/* ---- groups ---- */
void og1_zero(og1_t *r);
void og1_one(og1_t *r);
int og1_is_zero(const og1_t *a);
void og1_add(og1_t *r, const og1_t *a, const og1_t *b);       /* libff operator+ */
void og1_mixed_add(og1_t *r, const og1_t *a, const og1_t *b); /* b must have Z==1 or be zero */
void og1_dbl(og1_t *r, const og1_t *a);
void og1_neg(og1_t *r, const og1_t *a);
int og1_eq(const og1_t *a, const og1_t *b);                   /* projective equality */
void og1_to_affine(og1_t *r, const og1_t *a);                 /* libff to_affine_coordinates */
void og1_mul(og1_t *r, const og1_t *a, const ofp_t *k_mont);  /* libff scalar * point */
int og1_is_well_formed(const og1_t *a);

void og2_zero(og2_t *r);
void og2_one(og2_t *r);
int og2_is_zero(const og2_t *a);
void og2_add(og2_t *r, const og2_t *a, const og2_t *b);
void og2_mixed_add(og2_t *r, const og2_t *a, const og2_t *b);
void og2_dbl(og2_t *r, const og2_t *a);
void og2_neg(og2_t *r, const og2_t *a);
int og2_eq(const og2_t *a, const og2_t *b);
void og2_to_affine(og2_t *r, const og2_t *a);
void og2_mul(og2_t *r, const og2_t *a, const ofp_t *k_mont);
int og2_is_well_formed(const og2_t *a);

/* Canonical affine bytes: x,y as non-Montgomery LE limbs (+flag). Returns 1 if infinity. */
int og1_canonical_affine(uint64_t out_xy[8], const og1_t *a);
int og2_canonical_affine(uint64_t out_xy[16], const og2_t *a);

/* ---- multi_exp (libff multiexp.tcc restated) ---- */
size_t oracle_libff_log2(size_t n);
size_t oracle_bdlo12_window(size_t n);
/* multi_exp_inner<BDLO12> */
void oracle_g1_multi_exp_inner(og1_t *r, const og1_t *bases, const ofp_t *scalars, size_t n);
void oracle_g2_multi_exp_inner(og2_t *r, const og2_t *bases, const ofp_t *scalars, size_t n);
/* multi_exp<BDLO12>(.., chunks); threads!=0 runs the chunk loop on that many pthreads
 * (libff MULTICORE's "#pragma omp parallel for" over chunks). */
void oracle_g1_multi_exp(og1_t *r, const og1_t *bases, const ofp_t *scalars, size_t n, size_t chunks, int threads);
void oracle_g2_multi_exp(og2_t *r, const og2_t *bases, const ofp_t *scalars, size_t n, size_t chunks, int threads);
/* multi_exp_with_mixed_addition<BDLO12>: zero/one filter, then multi_exp
 * (reference call site: /root/reference/src/utils/globl.h:74-77). */
void oracle_g1_multi_exp_with_mixed_addition(og1_t *r, const og1_t *bases, const ofp_t *scalars, size_t n, size_t chunks, int threads);
void oracle_g2_multi_exp_with_mixed_addition(og2_t *r, const og2_t *bases, const ofp_t *scalars, size_t n, size_t chunks, int threads);

/* ---- fixed-base batch_exp (libff get_exp_window_size/get_window_table/batch_exp;
 * reference call site /root/reference/src/utils/util.h:119-134) ---- */
size_t oracle_g1_exp_window_size(size_t num_scalars);
size_t oracle_g2_exp_window_size(size_t num_scalars);
void oracle_g1_batch_exp(og1_t *out, const og1_t *base, const ofp_t *scalars, size_t n, size_t window);
void oracle_g2_batch_exp(og2_t *out, const og2_t *base, const ofp_t *scalars, size_t n, size_t window);

/* ---- sparse matrix in the exponent (CPlink keygen): sparsemexpG for one column
 * (/root/reference/src/utils/sparsemexp.h:62-90) and mtxmultiexp over a CSC matrix
 * (/root/reference/src/gadgets/subspace.cc:18-25) ---- */
void oracle_g1_sparsemexp_column(og1_t *out, const og1_t *vals, const uint32_t *pos, size_t nnz, const ofp_t *exps);
void oracle_g1_mtxmultiexp(og1_t *out, const og1_t *vals, const uint32_t *rows, const uint64_t *col_ptr, size_t ncols, const ofp_t *exps);

/* ---- Fr-vector loops around the MSMs ---- */
/* CPPoly::prove witness coefficients (/root/reference/src/gadgets/poly.h:51-67) */
void oracle_fr_cppoly_witness(ofp_t *w, const ofp_t *v, const ofp_t *r, size_t d);
/* MultiVPolyT::evalMLE (/root/reference/src/prototools/polytools.h:207-234) */
void oracle_fr_eval_mle(ofp_t *out, const ofp_t *v, const ofp_t *r, size_t d);
/* DPMle::pushRandomness (/root/reference/src/prototools/mle.h:199-210) */
void oracle_fr_push_randomness(ofp_t *cur, const ofp_t *old, const ofp_t *r, size_t half);

/* CPSumcheck::make_new_h_poly (/root/reference/src/gadgets/sumcheck.h:85-106); rho_j == NULL: DPBetaDummy */
void oracle_fr_sumcheck_round(ofp_t *out, const ofp_t *suff, const ofp_t *const *tables, size_t m, size_t half,
                              const ofp_t *pre, const ofp_t *rho_j);
/* DPBeta::pushRandomness suffix update (/root/reference/src/prototools/mle.h:46-53) */
void oracle_fr_scale_upper(ofp_t *cur, const ofp_t *old, const ofp_t *k, size_t half);

/* libfqfft basic_radix2_domain FFT / iFFT / cosetFFT / icosetFFT over Fr [upstream, recalled];
 * reference call sites /root/reference/src/gadgets/lipmaa.cc:68-81,102-175 */
void oracle_fr_radix2_fft(ofp_t *a, size_t log_n, const ofp_t *omega);
void oracle_fr_domain_transform(ofp_t *a, size_t log_n, const ofp_t *omega, int inverse, const ofp_t *coset_g);
/* libfqfft step_radix2_domain (2^big_log + 2^small_log points); omega: primitive 2^(big_log + 1)-th root of unity */
void oracle_fr_step_domain_transform(ofp_t *a, size_t big_log, size_t small_log, const ofp_t *omega, int inverse, const ofp_t *coset_g);

/* ---- test-input helper: out = sum_i a[i]*b[i] in Fr ---- */
void oracle_fr_dot(ofp_t *out, const ofp_t *a, const ofp_t *b, size_t n);
/* ---- test-input helper: out[i] = (a + i*b) * generator, un-normalised Jacobian ---- */
void oracle_g1_arith_bases(og1_t *out, const ofp_t *a_mont, const ofp_t *b_mont, size_t n);
void oracle_g2_arith_bases(og2_t *out, const ofp_t *a_mont, const ofp_t *b_mont, size_t n);

/* ---- pairing (libff alt_bn128_pairing.cpp restated; reference call sites
 * /root/reference/src/utils/globl.h:94-105, src/gadgets/subspace.cc:88-170) ---- */
void oracle_precompute_g1(og1_precomp_t *r, const og1_t *p);
void oracle_precompute_g2(og2_precomp_t *r, const og2_t *q);
void oracle_miller_loop(ofq12_t *r, const og1_precomp_t *p, const og2_precomp_t *q);
void oracle_double_miller_loop(ofq12_t *r, const og1_precomp_t *p1, const og2_precomp_t *q1,
                               const og1_precomp_t *p2, const og2_precomp_t *q2);
void oracle_final_exponentiation(ofq12_t *r, const ofq12_t *f);
void oracle_reduced_pairing(ofq12_t *r, const og1_t *p, const og2_t *q);
/* batch helpers used by tests / cpu_baseline */
void oracle_miller_loop_batch(ofq12_t *out, const og1_t *p, const og2_t *q, size_t n);
void oracle_pairing_product(ofq12_t *out, const og1_t *p, const og2_t *q, size_t n); /* final_exp(prod miller) */

#ifdef __cplusplus
}
#endif
#endif
