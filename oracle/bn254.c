/* oracle/bn254.c -- see bn254.h.  TEST INFRASTRUCTURE ONLY (checker + reported
 * CPU baseline); "libff-algorithm restatement", never "libff".
 * Plain C11, no GMP: 4 x u64 Montgomery limbs with unsigned __int128.
 */
#include "bn254.h"
#include "bn254_consts.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ Fp (Fq / Fr) */
static inline const uint64_t *modulus(int which) { return which ? BN254_R : BN254_P; }
static inline uint64_t mod_inv(int which) { return which ? BN254_R_INV : BN254_P_INV; }

static inline int limbs_geq(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; --i) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
    return 1;
}
static inline uint64_t limbs_sub(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)t; borrow = (uint64_t)(t >> 64) & 1;
    }
    return borrow;
}
static inline uint64_t limbs_add(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t carry = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a[i] + b[i] + carry;
        r[i] = (uint64_t)t; carry = (uint64_t)(t >> 64);
    }
    return carry;
}

/* CIOS Montgomery product: r = a*b*R^-1 mod m, inputs/outputs in [0,m). */
static inline __attribute__((always_inline)) void mont_mul(uint64_t r[4], const uint64_t a[4], const uint64_t b[4], const int which) {
    const uint64_t *m = modulus(which);
    const uint64_t inv = mod_inv(which);
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a[j] * b[i] + t[j];
            t[j] = (uint64_t)c; c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t q = t[0] * inv;
        c = (u128)q * m[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)q * m[j] + t[j];
            t[j - 1] = (uint64_t)c; c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || limbs_geq(t, m)) limbs_sub(r, t, m);
    else memcpy(r, t, 32);
}

void ofp_mul(ofp_t *r, const ofp_t *a, const ofp_t *b, int which) { mont_mul(r->l, a->l, b->l, which); }
void ofp_add(ofp_t *r, const ofp_t *a, const ofp_t *b, int which) {
    uint64_t t[4];
    uint64_t carry = limbs_add(t, a->l, b->l);
    if (carry || limbs_geq(t, modulus(which))) limbs_sub(r->l, t, modulus(which));
    else memcpy(r->l, t, 32);
}
void ofp_sub(ofp_t *r, const ofp_t *a, const ofp_t *b, int which) {
    uint64_t t[4];
    if (limbs_sub(t, a->l, b->l)) limbs_add(r->l, t, modulus(which));
    else memcpy(r->l, t, 32);
}
void ofp_from_canonical(ofp_t *r, const uint64_t x[4], int which) {
    mont_mul(r->l, x, which ? BN254_FR_R2 : BN254_FQ_R2, which);
}
void ofp_to_canonical(uint64_t x[4], const ofp_t *a, int which) {
    static const uint64_t one[4] = {1, 0, 0, 0};
    mont_mul(x, a->l, one, which);
}
static void fp_pow(ofp_t *r, const ofp_t *a, const uint64_t e[4], int which) {
    ofp_t acc;
    memcpy(acc.l, which ? BN254_FR_ONE : BN254_FQ_ONE, 32);
    for (int i = 255; i >= 0; --i) {
        ofp_mul(&acc, &acc, &acc, which);
        if ((e[i >> 6] >> (i & 63)) & 1) ofp_mul(&acc, &acc, a, which);
    }
    *r = acc;
}
void ofp_inv(ofp_t *r, const ofp_t *a, int which) { /* Fermat: a^(m-2) */
    uint64_t e[4]; static const uint64_t two[4] = {2, 0, 0, 0};
    limbs_sub(e, modulus(which), two);
    fp_pow(r, a, e, which);
}

/* Fq helpers with the uniform F(x) signature used by jacobian_tmpl.h */
static inline void fq_zero(ofp_t *r) { memset(r, 0, sizeof *r); }
static inline void fq_one(ofp_t *r) { memcpy(r->l, BN254_FQ_ONE, 32); }
static inline void fr_zero(ofp_t *r) { memset(r, 0, sizeof *r); }
static inline void fr_one(ofp_t *r) { memcpy(r->l, BN254_FR_ONE, 32); }
static inline int fq_is_zero(const ofp_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fq_eq(const ofp_t *a, const ofp_t *b) { return memcmp(a, b, 32) == 0; }
static inline void fq_add(ofp_t *r, const ofp_t *a, const ofp_t *b) { ofp_add(r, a, b, 0); }
static inline void fq_sub(ofp_t *r, const ofp_t *a, const ofp_t *b) { ofp_sub(r, a, b, 0); }
static inline void fq_mul(ofp_t *r, const ofp_t *a, const ofp_t *b) { mont_mul(r->l, a->l, b->l, 0); }
static inline void fq_sqr(ofp_t *r, const ofp_t *a) { mont_mul(r->l, a->l, a->l, 0); }
static inline void fq_neg(ofp_t *r, const ofp_t *a) { ofp_t z; fq_zero(&z); ofp_sub(r, &z, a, 0); }
static inline void fq_inv(ofp_t *r, const ofp_t *a) { ofp_inv(r, a, 0); }

/* ------------------------------------------------------------------ Fq2 = Fq[u]/(u^2+1) */
static inline void fq2_zero(ofq2_t *r) { memset(r, 0, sizeof *r); }
static inline void fq2_one(ofq2_t *r) { fq_one(&r->c0); fq_zero(&r->c1); }
static inline int fq2_is_zero(const ofq2_t *a) { return fq_is_zero(&a->c0) && fq_is_zero(&a->c1); }
static inline int fq2_eq(const ofq2_t *a, const ofq2_t *b) { return memcmp(a, b, sizeof *a) == 0; }
static inline void fq2_add(ofq2_t *r, const ofq2_t *a, const ofq2_t *b) { fq_add(&r->c0, &a->c0, &b->c0); fq_add(&r->c1, &a->c1, &b->c1); }
static inline void fq2_sub(ofq2_t *r, const ofq2_t *a, const ofq2_t *b) { fq_sub(&r->c0, &a->c0, &b->c0); fq_sub(&r->c1, &a->c1, &b->c1); }
static inline void fq2_neg(ofq2_t *r, const ofq2_t *a) { fq_neg(&r->c0, &a->c0); fq_neg(&r->c1, &a->c1); }
static void fq2_mul(ofq2_t *r, const ofq2_t *a, const ofq2_t *b) {
    ofp_t aa, bb, s, t, c0;
    fq_mul(&aa, &a->c0, &b->c0);
    fq_mul(&bb, &a->c1, &b->c1);
    fq_add(&s, &a->c0, &a->c1);
    fq_add(&t, &b->c0, &b->c1);
    fq_sub(&c0, &aa, &bb);                 /* non-residue -1 */
    fq_mul(&s, &s, &t);
    fq_sub(&s, &s, &aa);
    fq_sub(&r->c1, &s, &bb);
    r->c0 = c0;
}
static void fq2_sqr(ofq2_t *r, const ofq2_t *a) {
    ofp_t s, d, m;
    fq_add(&s, &a->c0, &a->c1);
    fq_sub(&d, &a->c0, &a->c1);
    fq_mul(&m, &a->c0, &a->c1);
    fq_mul(&r->c0, &s, &d);
    fq_add(&r->c1, &m, &m);
}
static void fq2_mul_fq(ofq2_t *r, const ofq2_t *a, const ofp_t *k) { fq_mul(&r->c0, &a->c0, k); fq_mul(&r->c1, &a->c1, k); }
static void fq2_inv(ofq2_t *r, const ofq2_t *a) {
    ofp_t t0, t1;
    fq_sqr(&t0, &a->c0); fq_sqr(&t1, &a->c1);
    fq_add(&t0, &t0, &t1);
    fq_inv(&t0, &t0);
    fq_mul(&r->c0, &a->c0, &t0);
    fq_mul(&t1, &a->c1, &t0);
    fq_neg(&r->c1, &t1);
}
static inline void fq2_conj(ofq2_t *r, const ofq2_t *a) { r->c0 = a->c0; fq_neg(&r->c1, &a->c1); }
static inline void fq2_load(ofq2_t *r, const uint64_t c[2][4]) { memcpy(r->c0.l, c[0], 32); memcpy(r->c1.l, c[1], 32); }
/* multiply by xi = 9 + u */
static void fq2_mul_xi(ofq2_t *r, const ofq2_t *a) {
    ofq2_t xi; fq2_load(&xi, BN254_XI);
    fq2_mul(r, a, &xi);
}

/* ------------------------------------------------------------------ Fq6 = Fq2[v]/(v^3 - xi) */
static inline void fq6_zero(ofq6_t *r) { memset(r, 0, sizeof *r); }
static inline void fq6_one(ofq6_t *r) { fq6_zero(r); fq2_one(&r->c0); }
static inline void fq6_add(ofq6_t *r, const ofq6_t *a, const ofq6_t *b) { fq2_add(&r->c0, &a->c0, &b->c0); fq2_add(&r->c1, &a->c1, &b->c1); fq2_add(&r->c2, &a->c2, &b->c2); }
static inline void fq6_sub(ofq6_t *r, const ofq6_t *a, const ofq6_t *b) { fq2_sub(&r->c0, &a->c0, &b->c0); fq2_sub(&r->c1, &a->c1, &b->c1); fq2_sub(&r->c2, &a->c2, &b->c2); }
static inline void fq6_neg(ofq6_t *r, const ofq6_t *a) { fq2_neg(&r->c0, &a->c0); fq2_neg(&r->c1, &a->c1); fq2_neg(&r->c2, &a->c2); }
static void fq6_mul(ofq6_t *r, const ofq6_t *a, const ofq6_t *b) {
    /* schoolbook with v^3 = xi */
    ofq2_t a0b0, a0b1, a0b2, a1b0, a1b1, a1b2, a2b0, a2b1, a2b2, t, c0, c1, c2;
    fq2_mul(&a0b0, &a->c0, &b->c0); fq2_mul(&a0b1, &a->c0, &b->c1); fq2_mul(&a0b2, &a->c0, &b->c2);
    fq2_mul(&a1b0, &a->c1, &b->c0); fq2_mul(&a1b1, &a->c1, &b->c1); fq2_mul(&a1b2, &a->c1, &b->c2);
    fq2_mul(&a2b0, &a->c2, &b->c0); fq2_mul(&a2b1, &a->c2, &b->c1); fq2_mul(&a2b2, &a->c2, &b->c2);
    fq2_add(&t, &a1b2, &a2b1); fq2_mul_xi(&t, &t); fq2_add(&c0, &a0b0, &t);
    fq2_mul_xi(&t, &a2b2); fq2_add(&c1, &a0b1, &a1b0); fq2_add(&c1, &c1, &t);
    fq2_add(&c2, &a0b2, &a1b1); fq2_add(&c2, &c2, &a2b0);
    r->c0 = c0; r->c1 = c1; r->c2 = c2;
}
static void fq6_mul_by_v(ofq6_t *r, const ofq6_t *a) { /* (c0,c1,c2)*v = (xi c2, c0, c1) */
    ofq2_t t; fq2_mul_xi(&t, &a->c2);
    ofq2_t c0 = a->c0, c1 = a->c1;
    r->c0 = t; r->c1 = c0; r->c2 = c1;
}
static void fq6_mul_fq2(ofq6_t *r, const ofq6_t *a, const ofq2_t *k) { fq2_mul(&r->c0, &a->c0, k); fq2_mul(&r->c1, &a->c1, k); fq2_mul(&r->c2, &a->c2, k); }
static void fq6_inv(ofq6_t *r, const ofq6_t *a) {
    /* standard: c0 = a0^2 - xi a1 a2; c1 = xi a2^2 - a0 a1; c2 = a1^2 - a0 a2; t = (a0 c0 + xi(a2 c1 + a1 c2))^-1 */
    ofq2_t t0, t1, t2, t3, t4, t5, c0, c1, c2, t6;
    fq2_sqr(&t0, &a->c0); fq2_sqr(&t1, &a->c1); fq2_sqr(&t2, &a->c2);
    fq2_mul(&t3, &a->c0, &a->c1); fq2_mul(&t4, &a->c0, &a->c2); fq2_mul(&t5, &a->c1, &a->c2);
    fq2_mul_xi(&c0, &t5); fq2_sub(&c0, &t0, &c0);
    fq2_mul_xi(&c1, &t2); fq2_sub(&c1, &c1, &t3);
    fq2_sub(&c2, &t1, &t4);
    fq2_mul(&t6, &a->c2, &c1); fq2_mul(&t0, &a->c1, &c2); fq2_add(&t6, &t6, &t0); fq2_mul_xi(&t6, &t6);
    fq2_mul(&t0, &a->c0, &c0); fq2_add(&t6, &t6, &t0);
    fq2_inv(&t6, &t6);
    fq2_mul(&r->c0, &c0, &t6); fq2_mul(&r->c1, &c1, &t6); fq2_mul(&r->c2, &c2, &t6);
}
static void fq2_frobenius(ofq2_t *r, const ofq2_t *a, unsigned power) {
    if (power & 1) fq2_conj(r, a); else *r = *a;
}
static void fq6_frobenius(ofq6_t *r, const ofq6_t *a, unsigned power) {
    ofq2_t k1, k2, t;
    fq2_load(&k1, BN254_FROB6_C1[power % 6]);
    fq2_load(&k2, BN254_FROB6_C2[power % 6]);
    fq2_frobenius(&r->c0, &a->c0, power);
    fq2_frobenius(&t, &a->c1, power); fq2_mul(&r->c1, &t, &k1);
    fq2_frobenius(&t, &a->c2, power); fq2_mul(&r->c2, &t, &k2);
}

/* ------------------------------------------------------------------ Fq12 = Fq6[w]/(w^2 - v) */
void ofq12_one(ofq12_t *r) { fq6_one(&r->c0); fq6_zero(&r->c1); }
void ofq12_mul(ofq12_t *r, const ofq12_t *a, const ofq12_t *b) {
    ofq6_t aa, bb, s, t, c0;
    fq6_mul(&aa, &a->c0, &b->c0);
    fq6_mul(&bb, &a->c1, &b->c1);
    fq6_add(&s, &a->c0, &a->c1);
    fq6_add(&t, &b->c0, &b->c1);
    fq6_mul(&s, &s, &t);
    fq6_mul_by_v(&c0, &bb); fq6_add(&c0, &c0, &aa);
    fq6_sub(&s, &s, &aa);
    fq6_sub(&r->c1, &s, &bb);
    r->c0 = c0;
}
static void fq12_sqr(ofq12_t *r, const ofq12_t *a) { ofq12_mul(r, a, a); }
void ofq12_unitary_inverse(ofq12_t *r, const ofq12_t *a) { r->c0 = a->c0; fq6_neg(&r->c1, &a->c1); }
void ofq12_inverse(ofq12_t *r, const ofq12_t *a) {
    /* (c0 + c1 w)^-1 = (c0 - c1 w) / (c0^2 - v c1^2) */
    ofq6_t t0, t1;
    fq6_mul(&t0, &a->c0, &a->c0);
    fq6_mul(&t1, &a->c1, &a->c1);
    fq6_mul_by_v(&t1, &t1);
    fq6_sub(&t0, &t0, &t1);
    fq6_inv(&t0, &t0);
    fq6_mul(&r->c0, &a->c0, &t0);
    fq6_mul(&t1, &a->c1, &t0);
    fq6_neg(&r->c1, &t1);
}
void ofq12_frobenius(ofq12_t *r, const ofq12_t *a, unsigned power) {
    ofq2_t k; fq2_load(&k, BN254_FROB12_C1[power % 12]);
    ofq6_t t;
    fq6_frobenius(&r->c0, &a->c0, power);
    fq6_frobenius(&t, &a->c1, power);
    fq6_mul_fq2(&r->c1, &t, &k);
}
void ofq12_pow_u64(ofq12_t *r, const ofq12_t *a, uint64_t e) {
    ofq12_t acc; ofq12_one(&acc);
    for (int i = 63; i >= 0; --i) {
        fq12_sqr(&acc, &acc);
        if ((e >> i) & 1) ofq12_mul(&acc, &acc, a);
    }
    *r = acc;
}
void ofq12_pow_p(ofq12_t *r, const ofq12_t *a) {
    ofq12_t acc; ofq12_one(&acc);
    for (int i = 255; i >= 0; --i) {
        fq12_sqr(&acc, &acc);
        if ((BN254_P[i >> 6] >> (i & 63)) & 1) ofq12_mul(&acc, &acc, a);
    }
    *r = acc;
}
/* libff Fp12::mul_by_024: this * (ell_0 + ell_VV v^2 + ell_VW v w), i.e. the sparse
 * element Fp12(Fp6(ell_0, 0, ell_VV), Fp6(0, ell_VW, 0)) [upstream, recalled]. */
static void fq12_mul_by_024(ofq12_t *r, const ofq12_t *a, const ofq2_t *ell_0, const ofq2_t *ell_VW, const ofq2_t *ell_VV) {
    ofq12_t s; memset(&s, 0, sizeof s);
    s.c0.c0 = *ell_0; s.c0.c2 = *ell_VV; s.c1.c1 = *ell_VW;
    ofq12_mul(r, a, &s);
}

/* ------------------------------------------------------------------ groups */
#include "bn254_groups.inc"

/* ------------------------------------------------------------------ pairing */
static ofq2_t g_twist_b, g_xi, g_q_x, g_q_y;
static ofp_t g_two_inv;
static int g_pair_init = 0;
static void pair_init(void) {
    if (g_pair_init) return;
    fq2_load(&g_twist_b, BN254_TWIST_B); fq2_load(&g_xi, BN254_XI);
    fq2_load(&g_q_x, BN254_TWIST_MUL_BY_Q_X); fq2_load(&g_q_y, BN254_TWIST_MUL_BY_Q_Y);
    memcpy(g_two_inv.l, BN254_FQ_TWO_INV, 32);
    g_pair_init = 1;
}

/* libff doubling_step_for_flipped_miller_loop [upstream, recalled] */
static void doubling_step(og2_t *cur, oell_t *c) {
    ofq2_t X = cur->X, Y = cur->Y, Z = cur->Z;
    ofq2_t A, B, C, D, E, Fv, Gv, H, I, J, E2, t;
    fq2_mul(&A, &X, &Y); fq2_mul_fq(&A, &A, &g_two_inv);   /* A = X1 Y1 / 2 */
    fq2_sqr(&B, &Y);                                        /* B = Y1^2 */
    fq2_sqr(&C, &Z);                                        /* C = Z1^2 */
    fq2_add(&D, &C, &C); fq2_add(&D, &D, &C);               /* D = 3C */
    fq2_mul(&E, &g_twist_b, &D);                            /* E = b' D */
    fq2_add(&Fv, &E, &E); fq2_add(&Fv, &Fv, &E);            /* F = 3E */
    fq2_add(&Gv, &B, &Fv); fq2_mul_fq(&Gv, &Gv, &g_two_inv);/* G = (B+F)/2 */
    fq2_add(&H, &Y, &Z); fq2_sqr(&H, &H); fq2_add(&t, &B, &C); fq2_sub(&H, &H, &t); /* H = (Y+Z)^2-(B+C) */
    fq2_sub(&I, &E, &B);                                    /* I = E - B */
    fq2_sqr(&J, &X);                                        /* J = X1^2 */
    fq2_sqr(&E2, &E);
    fq2_sub(&t, &B, &Fv); fq2_mul(&cur->X, &A, &t);         /* X3 = A (B-F) */
    fq2_sqr(&cur->Y, &Gv); fq2_add(&t, &E2, &E2); fq2_add(&t, &t, &E2); fq2_sub(&cur->Y, &cur->Y, &t); /* Y3 = G^2 - 3E^2 */
    fq2_mul(&cur->Z, &B, &H);                               /* Z3 = B H */
    fq2_mul(&c->ell_0, &g_xi, &I);                          /* ell_0 = xi I */
    fq2_neg(&c->ell_VW, &H);                                /* ell_VW = -H */
    fq2_add(&c->ell_VV, &J, &J); fq2_add(&c->ell_VV, &c->ell_VV, &J); /* ell_VV = 3J */
}
/* libff mixed_addition_step_for_flipped_miller_loop [upstream, recalled] */
static void mixed_addition_step(const og2_t *base, og2_t *cur, oell_t *c) {
    ofq2_t X1 = cur->X, Y1 = cur->Y, Z1 = cur->Z;
    const ofq2_t *x2 = &base->X, *y2 = &base->Y;
    ofq2_t D, E, Fv, Gv, H, I, J, t, u;
    fq2_mul(&t, x2, &Z1); fq2_sub(&D, &X1, &t);             /* D = X1 - x2 Z1 */
    fq2_mul(&t, y2, &Z1); fq2_sub(&E, &Y1, &t);             /* E = Y1 - y2 Z1 */
    fq2_sqr(&Fv, &D); fq2_sqr(&Gv, &E);
    fq2_mul(&H, &D, &Fv);
    fq2_mul(&I, &X1, &Fv);
    fq2_mul(&t, &Z1, &Gv); fq2_add(&J, &H, &t); fq2_add(&t, &I, &I); fq2_sub(&J, &J, &t); /* J = H + Z1 G - 2I */
    fq2_mul(&cur->X, &D, &J);
    fq2_sub(&t, &I, &J); fq2_mul(&t, &E, &t); fq2_mul(&u, &H, &Y1); fq2_sub(&cur->Y, &t, &u);
    fq2_mul(&cur->Z, &Z1, &H);
    fq2_mul(&t, &E, x2); fq2_mul(&u, &D, y2); fq2_sub(&t, &t, &u); fq2_mul(&c->ell_0, &g_xi, &t);
    fq2_neg(&c->ell_VV, &E);
    c->ell_VW = D;
}
static void g2_mul_by_q(og2_t *r, const og2_t *a) {
    ofq2_t t;
    fq2_frobenius(&t, &a->X, 1); fq2_mul(&r->X, &g_q_x, &t);
    fq2_frobenius(&t, &a->Y, 1); fq2_mul(&r->Y, &g_q_y, &t);
    fq2_frobenius(&r->Z, &a->Z, 1);
}
static inline int ate_bit(int i) {
    if (i >= 64) return (int)((BN254_ATE_LOOP_COUNT_HI >> (i - 64)) & 1);
    return (int)((BN254_ATE_LOOP_COUNT_LO >> i) & 1);
}

void oracle_precompute_g1(og1_precomp_t *r, const og1_t *p) {
    og1_t a; og1_to_affine(&a, p);
    r->PX = a.X; r->PY = a.Y;
}
void oracle_precompute_g2(og2_precomp_t *r, const og2_t *q) {
    pair_init();
    og2_t Q; og2_to_affine(&Q, q);
    r->QX = Q.X; r->QY = Q.Y;
    og2_t Rr; Rr.X = Q.X; Rr.Y = Q.Y; fq2_one(&Rr.Z);
    int found_one = 0; size_t idx = 0;
    for (int i = 255; i >= 0; --i) {  /* loop_count.max_bits() downto 0 */
        int bit = i < 128 ? ate_bit(i) : 0;
        if (!found_one) { found_one |= bit; continue; }
        doubling_step(&Rr, &r->coeffs[idx++]);
        if (bit) mixed_addition_step(&Q, &Rr, &r->coeffs[idx++]);
    }
    og2_t Q1, Q2;
    g2_mul_by_q(&Q1, &Q);
    g2_mul_by_q(&Q2, &Q1);
    /* alt_bn128_ate_is_loop_count_neg == false */
    fq2_neg(&Q2.Y, &Q2.Y);
    mixed_addition_step(&Q1, &Rr, &r->coeffs[idx++]);
    mixed_addition_step(&Q2, &Rr, &r->coeffs[idx++]);
    if (idx != ORACLE_ATE_NUM_COEFFS) abort();
}
static void miller_apply(ofq12_t *f, const og1_precomp_t *p, const oell_t *c) {
    ofq2_t vw, vv;
    fq2_mul_fq(&vw, &c->ell_VW, &p->PY);
    fq2_mul_fq(&vv, &c->ell_VV, &p->PX);
    fq12_mul_by_024(f, f, &c->ell_0, &vw, &vv);
}
void oracle_miller_loop(ofq12_t *r, const og1_precomp_t *p, const og2_precomp_t *q) {
    ofq12_t f; ofq12_one(&f);
    int found_one = 0; size_t idx = 0;
    for (int i = 255; i >= 0; --i) {
        int bit = i < 128 ? ate_bit(i) : 0;
        if (!found_one) { found_one |= bit; continue; }
        fq12_sqr(&f, &f);
        miller_apply(&f, p, &q->coeffs[idx++]);
        if (bit) miller_apply(&f, p, &q->coeffs[idx++]);
    }
    miller_apply(&f, p, &q->coeffs[idx++]);
    miller_apply(&f, p, &q->coeffs[idx++]);
    *r = f;
}
void oracle_double_miller_loop(ofq12_t *r, const og1_precomp_t *p1, const og2_precomp_t *q1,
                               const og1_precomp_t *p2, const og2_precomp_t *q2) {
    ofq12_t f; ofq12_one(&f);
    int found_one = 0; size_t idx = 0;
    for (int i = 255; i >= 0; --i) {
        int bit = i < 128 ? ate_bit(i) : 0;
        if (!found_one) { found_one |= bit; continue; }
        fq12_sqr(&f, &f);
        miller_apply(&f, p1, &q1->coeffs[idx]); miller_apply(&f, p2, &q2->coeffs[idx]); idx++;
        if (bit) { miller_apply(&f, p1, &q1->coeffs[idx]); miller_apply(&f, p2, &q2->coeffs[idx]); idx++; }
    }
    miller_apply(&f, p1, &q1->coeffs[idx]); miller_apply(&f, p2, &q2->coeffs[idx]); idx++;
    miller_apply(&f, p1, &q1->coeffs[idx]); miller_apply(&f, p2, &q2->coeffs[idx]); idx++;
    *r = f;
}
/* libff final exponentiation [upstream, recalled]: first chunk (q^6-1)(q^2+1),
 * last chunk by the Fuentes-Castaneda et al. chain with three exp_by_neg_z. */
static void exp_by_neg_z(ofq12_t *r, const ofq12_t *a) {
    ofq12_t t; ofq12_pow_u64(&t, a, BN254_FINAL_EXP_Z);   /* cyclotomic_exp(z) == plain pow on the same element */
    ofq12_unitary_inverse(r, &t);                          /* z is positive -> invert */
}
void oracle_final_exponentiation(ofq12_t *r, const ofq12_t *elt) {
    ofq12_t A, B, C, D, E, Fv, Gv, H, I, J, K, L, M, N, O, Pp, Q, Rr, S, T, Uu, first;
    /* first chunk */
    ofq12_unitary_inverse(&A, elt);
    ofq12_inverse(&B, elt);
    ofq12_mul(&C, &A, &B);
    ofq12_frobenius(&D, &C, 2);
    ofq12_mul(&first, &D, &C);
    /* last chunk */
    exp_by_neg_z(&A, &first);
    fq12_sqr(&B, &A);
    fq12_sqr(&C, &B);
    ofq12_mul(&D, &C, &B);
    exp_by_neg_z(&E, &D);
    fq12_sqr(&Fv, &E);
    exp_by_neg_z(&Gv, &Fv);
    ofq12_unitary_inverse(&H, &D);
    ofq12_unitary_inverse(&I, &Gv);
    ofq12_mul(&J, &I, &E);
    ofq12_mul(&K, &J, &H);
    ofq12_mul(&L, &K, &B);
    ofq12_mul(&M, &K, &E);
    ofq12_mul(&N, &M, &first);
    ofq12_frobenius(&O, &L, 1);
    ofq12_mul(&Pp, &O, &N);
    ofq12_frobenius(&Q, &K, 2);
    ofq12_mul(&Rr, &Q, &Pp);
    ofq12_unitary_inverse(&S, &first);
    ofq12_mul(&T, &S, &L);
    ofq12_frobenius(&Uu, &T, 3);
    ofq12_mul(r, &Uu, &Rr);
}
void oracle_reduced_pairing(ofq12_t *r, const og1_t *p, const og2_t *q) {
    og1_precomp_t pp; og2_precomp_t qp; ofq12_t f;
    oracle_precompute_g1(&pp, p);
    oracle_precompute_g2(&qp, q);
    oracle_miller_loop(&f, &pp, &qp);
    oracle_final_exponentiation(r, &f);
}
void oracle_miller_loop_batch(ofq12_t *out, const og1_t *p, const og2_t *q, size_t n) {
    og2_precomp_t *qp = (og2_precomp_t *)malloc(sizeof *qp);
    for (size_t i = 0; i < n; i++) {
        og1_precomp_t pp;
        oracle_precompute_g1(&pp, &p[i]);
        oracle_precompute_g2(qp, &q[i]);
        oracle_miller_loop(&out[i], &pp, qp);
    }
    free(qp);
}
void oracle_pairing_product(ofq12_t *out, const og1_t *p, const og2_t *q, size_t n) {
    ofq12_t acc, f; ofq12_one(&acc);
    og2_precomp_t *qp = (og2_precomp_t *)malloc(sizeof *qp);
    for (size_t i = 0; i < n; i++) {
        og1_precomp_t pp;
        oracle_precompute_g1(&pp, &p[i]);
        oracle_precompute_g2(qp, &q[i]);
        oracle_miller_loop(&f, &pp, qp);
        ofq12_mul(&acc, &acc, &f);
    }
    free(qp);
    oracle_final_exponentiation(out, &acc);
}

/* ---- Fr-vector loops around the MSMs (restated line by line from the reference) ---- */
static inline void fr_add_(ofp_t *r, const ofp_t *a, const ofp_t *b) { ofp_add(r, a, b, 1); }
static inline void fr_sub_(ofp_t *r, const ofp_t *a, const ofp_t *b) { ofp_sub(r, a, b, 1); }
static inline void fr_mul_(ofp_t *r, const ofp_t *a, const ofp_t *b) { ofp_mul(r, a, b, 1); }
static inline void fr_neg_(ofp_t *r, const ofp_t *a) { ofp_t z; fr_zero(&z); ofp_sub(r, &z, a, 1); }

/* CPPoly::prove, witness coefficients: /root/reference/src/gadgets/poly.h:51-67.
 * v: 2^d, r: d, w: 2^d (value-initialised to zero like `Scalars w_coeffs(1 << d)`). */
void oracle_fr_cppoly_witness(ofp_t *w, const ofp_t *v, const ofp_t *r, size_t d) {
    size_t N = (size_t)1 << d;
    ofp_t *tmp = (ofp_t *)malloc(sizeof(ofp_t) * N);
    ofp_t one; fr_one(&one);
    memcpy(tmp, v, sizeof(ofp_t) * N);
    memset(w, 0, sizeof(ofp_t) * N);
    size_t start = 0;
    for (size_t i = 0; i < d; i++) {
        size_t bound = (size_t)1 << (d - i - 1);
        ofp_t rm1; fr_sub_(&rm1, &r[i], &one);
        for (size_t p = 0; p < bound; p++) {
            size_t p0 = p << 1, p1 = (p << 1) + 1;
            ofp_t n0, t0, t1;
            fr_neg_(&n0, &tmp[p0]);
            fr_add_(&w[start + p], &n0, &tmp[p1]);              /* -tmp_v[p0] + tmp_v[p1] */
            fr_mul_(&t0, &n0, &rm1);                           /* -tmp_v[p0]*(r[i]-1) */
            fr_mul_(&t1, &tmp[p1], &r[i]);
            fr_add_(&tmp[p], &t0, &t1);
        }
        start += bound;
    }
    free(tmp);
}

/* MultiVPolyT::evalMLE: /root/reference/src/prototools/polytools.h:207-234
 * (table of eq-monomials, then the dot product with v). */
void oracle_fr_eval_mle(ofp_t *out, const ofp_t *v, const ofp_t *r, size_t d) {
    size_t N = (size_t)1 << d;
    ofp_t *products = (ofp_t *)malloc(sizeof(ofp_t) * N);
    ofp_t f1; fr_one(&f1);
    memset(products, 0, sizeof(ofp_t) * N);
    products[0] = f1;
    size_t idx = 1;
    for (size_t i = 0; i < d; i++) {
        size_t bound = (size_t)1 << i;
        ofp_t omr; fr_sub_(&omr, &f1, &r[i]);
        for (size_t p = 0; p < bound; p++) {
            fr_mul_(&products[p + idx], &products[p], &r[i]);
            fr_mul_(&products[p], &products[p], &omr);
        }
        idx += (size_t)1 << i;
    }
    ofp_t acc; fr_zero(&acc);
    for (size_t p = 0; p < N; p++) {
        ofp_t t; fr_mul_(&t, &v[p], &products[p]);
        fr_add_(&acc, &acc, &t);
    }
    *out = acc;
    free(products);
}

/* DPMle::pushRandomness: /root/reference/src/prototools/mle.h:199-210 with
 * eqbit(bool, r) of src/prototools/mle.cc:12-15.  old: 2*half, cur: half. */
void oracle_fr_push_randomness(ofp_t *cur, const ofp_t *old, const ofp_t *r, size_t half) {
    ofp_t one, omr; fr_one(&one);
    fr_sub_(&omr, &one, r);
    for (size_t p = 0; p < half; p++) {
        ofp_t t0, t1;
        fr_mul_(&t0, &old[p], &omr);
        fr_mul_(&t1, &old[p + half], r);
        fr_add_(&cur[p], &t0, &t1);
    }
}

/* CPSumcheck::make_new_h_poly: /root/reference/src/gadgets/sumcheck.h:85-106, with
 * DPBeta::getBetaPoly (src/prototools/mle.h:74-82; eqbit_poly(r) = {1-r, 2r-1},
 * src/prototools/mle.cc:23-29; PolyT::mul(scalar) and PolyT::mul(PolyT), polytools.h:54-71)
 * and DPMle::getMLEPoly (mle.h:217-226: eqbit_poly(0)*V[p] + eqbit_poly(1)*V[p+half],
 * eqbit_poly(bool) = 1-X or X, mle.cc:17-20).  Restated literally: per p the beta polynomial,
 * then one polynomial product per table, then the sum.  rho_j == NULL: DPBetaDummy (beta
 * polynomial = 1).  out: m+2 coefficients (m+1 without beta). */
void oracle_fr_sumcheck_round(ofp_t *out, const ofp_t *suff, const ofp_t *const *tables, size_t m, size_t half,
                              const ofp_t *pre, const ofp_t *rho_j) {
    ofp_t one; fr_one(&one);
    size_t nout = m + (rho_j ? 2 : 1);
    for (size_t i = 0; i < nout; i++) fr_zero(&out[i]);
    for (size_t p = 0; p < half; p++) {
        ofp_t poly[8], tmp[8];
        size_t deg;
        if (rho_j) {
            /* eqbit_poly(rho_j).mul(pre * suff) */
            ofp_t s, two_r, e0, e1;
            if (suff) fr_mul_(&s, pre, &suff[p]); else fr_mul_(&s, pre, &one);
            fr_sub_(&e0, &one, rho_j);
            fr_add_(&two_r, rho_j, rho_j);
            fr_sub_(&e1, &two_r, &one);
            fr_mul_(&poly[0], &e0, &s);
            fr_mul_(&poly[1], &e1, &s);
            deg = 1;
        } else {
            poly[0] = one;
            deg = 0;
        }
        for (size_t t = 0; t < m; t++) {
            /* mle_poly = (1-X)*v0 + X*v1 = {v0, v1 - v0}: eqbit_poly(0).mul(v0) = {v0, -v0}, eqbit_poly(1).mul(v1) = {0, v1}, add */
            ofp_t c0 = tables[t][p], nv0, c1;
            fr_neg_(&nv0, &tables[t][p]);
            fr_add_(&c1, &nv0, &tables[t][p + half]);
            for (size_t i = 0; i <= deg + 1; i++) fr_zero(&tmp[i]);
            for (size_t i = 0; i <= deg; i++) {
                ofp_t x;
                fr_mul_(&x, &poly[i], &c0); fr_add_(&tmp[i], &tmp[i], &x);
                fr_mul_(&x, &poly[i], &c1); fr_add_(&tmp[i + 1], &tmp[i + 1], &x);
            }
            deg++;
            for (size_t i = 0; i <= deg; i++) poly[i] = tmp[i];
        }
        for (size_t i = 0; i <= deg; i++) fr_add_(&out[i], &out[i], &poly[i]);
    }
}

/* DPBeta::pushRandomness, suffix table: /root/reference/src/prototools/mle.h:46-53 */
void oracle_fr_scale_upper(ofp_t *cur, const ofp_t *old, const ofp_t *k, size_t half) {
    for (size_t p = 0; p < half; p++) fr_mul_(&cur[p], &old[half + p], k);
}
/* DPBeta::compute_eq_tbl: /root/reference/src/prototools/mle.h:93-105 with eqbit(bool, r) of src/prototools/mle.cc:12-15,
 * restated literally (the table doubles d - 1 times, the new bit on top).  dst, tmp: 2^d entries; r: d >= 1 entries.
 * At the end dst[p] = eq(p, r). */
void oracle_fr_eq_table(ofp_t *dst, const ofp_t *r, size_t d) {
    size_t N = (size_t)1 << d;
    ofp_t *a = (ofp_t *)malloc(sizeof(ofp_t) * N), *b = (ofp_t *)malloc(sizeof(ofp_t) * N), one;
    fr_one(&one);
    memset(a, 0, sizeof(ofp_t) * N);
    memset(b, 0, sizeof(ofp_t) * N);
    fr_sub_(&a[0], &one, &r[0]);                                /* dst[0] = eqbit(false, r[0]) */
    a[1] = r[0];                                                /* dst[1] = eqbit(true, r[0]) */
    for (size_t j = 1; j < d; j++) {
        ofp_t omr; fr_sub_(&omr, &one, &r[j]);
        for (size_t p = 0; p < ((size_t)1 << (j + 1)); p++) {
            int msb = p >= ((size_t)1 << j);
            fr_mul_(&b[p], msb ? &r[j] : &omr, &a[p >> 1]);      /* tmp[p] = eqbit(msb, r[j]) * dst[p >> 1] */
        }
        ofp_t *t = a; a = b; b = t;                             /* swap(tmp, dst) */
    }
    memcpy(dst, a, sizeof(ofp_t) * N);
    free(a); free(b);
}
/* test-input helper: out = sum_i a[i] * b[i] in Fr (the O(n) check value of the
 * known-discrete-log identity at n = 2^20 .. 2^24) */
void oracle_fr_dot(ofp_t *out, const ofp_t *a, const ofp_t *b, size_t n) {
    ofp_t acc, t;
    memset(&acc, 0, sizeof acc);
    for (size_t i = 0; i < n; i++) {
        fr_mul_(&t, &a[i], &b[i]);
        ofp_add(&acc, &acc, &t, 1);
    }
    *out = acc;
}

/* libfqfft _basic_radix2_FFT [upstream, recalled] as used through basic_radix2_domain by
 * /root/reference/src/gadgets/lipmaa.cc:102-175: in-place bit-reversal, then log n
 * butterfly levels with w_m = omega^(n/m).  a: n = 2^log_n elements of Fr. */
void oracle_fr_radix2_fft(ofp_t *a, size_t log_n, const ofp_t *omega) {
    const size_t n = (size_t)1 << log_n;
    for (size_t k = 0; k < n; k++) {
        size_t rk = 0;
        for (size_t b = 0; b < log_n; b++) if (k & ((size_t)1 << b)) rk |= (size_t)1 << (log_n - 1 - b);
        if (k < rk) { ofp_t t = a[k]; a[k] = a[rk]; a[rk] = t; }
    }
    size_t m = 1;
    for (size_t s = 1; s <= log_n; s++) {
        /* w_m = omega^(n / 2m) */
        ofp_t w_m = *omega;
        for (size_t e = 2 * m; e < n; e <<= 1) fr_mul_(&w_m, &w_m, &w_m);
        for (size_t k = 0; k < n; k += 2 * m) {
            ofp_t w; fr_one(&w);
            for (size_t j = 0; j < m; j++) {
                ofp_t t;
                fr_mul_(&t, &w, &a[k + j + m]);
                fr_sub_(&a[k + j + m], &a[k + j], &t);
                fr_add_(&a[k + j], &a[k + j], &t);
                fr_mul_(&w, &w, &w_m);
            }
        }
        m *= 2;
    }
}
/* basic_radix2_domain::iFFT (FFT with omega^-1, then * 1/n), cosetFFT (_multiply_by_coset then
 * FFT), icosetFFT (iFFT then _multiply_by_coset with g^-1) [upstream, recalled] */
void oracle_fr_domain_transform(ofp_t *a, size_t log_n, const ofp_t *omega, int inverse, const ofp_t *coset_g) {
    const size_t n = (size_t)1 << log_n;
    ofp_t one; fr_one(&one);
    if (!inverse) {
        if (coset_g) { ofp_t u = *coset_g; for (size_t i = 1; i < n; i++) { fr_mul_(&a[i], &a[i], &u); fr_mul_(&u, &u, coset_g); } }
        oracle_fr_radix2_fft(a, log_n, omega);
        return;
    }
    ofp_t wi, ninv, nn;
    ofp_inv(&wi, omega, 1);
    oracle_fr_radix2_fft(a, log_n, &wi);
    fr_zero(&nn);
    for (size_t i = 0; i < n; i++) fr_add_(&nn, &nn, &one);      /* FieldT(n) */
    ofp_inv(&ninv, &nn, 1);
    for (size_t i = 0; i < n; i++) fr_mul_(&a[i], &a[i], &ninv);
    if (coset_g) {
        ofp_t gi, u;
        ofp_inv(&gi, coset_g, 1);
        u = gi;
        for (size_t i = 1; i < n; i++) { fr_mul_(&a[i], &a[i], &u); fr_mul_(&u, &u, &gi); }
    }
}

/* libfqfft step_radix2_domain<FieldT>::{FFT, iFFT, cosetFFT, icosetFFT} (m = big_m + small_m, big_m = 2^big_log, small_m =
 * 2^small_log < big_m; omega a primitive 2 big_m-th root of unity, big_omega = omega^2, small_omega =
 * get_root_of_unity(small_m) = omega^(2 big_m / small_m)) [upstream, recalled:
 * libfqfft/evaluation_domain/domains/step_radix2_domain.tcc -- the c / d / e vectors of FFT, the U0 / U1 / tmp vectors
 * of iFFT, in upstream's order of operations].  Checked against the definition (values of the polynomial at
 * omega^(2k), then at omega small_omega^j) in tests/test_oracle_golden.py. */
static void fr_from_size(ofp_t *out, size_t n) {
    ofp_t one, acc, bit;
    fr_one(&one);
    fr_zero(&acc);
    bit = one;
    for (size_t v = n; v; v >>= 1) {
        if (v & 1) fr_add_(&acc, &acc, &bit);
        fr_add_(&bit, &bit, &bit);
    }
    *out = acc;
}
static void fr_scale_by_powers(ofp_t *a, size_t n, const ofp_t *g) {   /* _multiply_by_coset */
    ofp_t u = *g;
    for (size_t i = 1; i < n; i++) { fr_mul_(&a[i], &a[i], &u); fr_mul_(&u, &u, g); }
}
void oracle_fr_step_domain_transform(ofp_t *a, size_t big_log, size_t small_log, const ofp_t *omega, int inverse, const ofp_t *coset_g) {
    const size_t big_m = (size_t)1 << big_log, small_m = (size_t)1 << small_log, m = big_m + small_m, compr = big_m / small_m;
    ofp_t big_omega, small_omega, one;
    fr_one(&one);
    fr_mul_(&big_omega, omega, omega);
    small_omega = *omega;
    for (size_t e = small_m; e < 2 * big_m; e <<= 1) fr_mul_(&small_omega, &small_omega, &small_omega);
    ofp_t *c = (ofp_t *)calloc(big_m, sizeof(ofp_t)), *d = (ofp_t *)calloc(big_m, sizeof(ofp_t)), *e = (ofp_t *)calloc(small_m, sizeof(ofp_t));
    if (!inverse) {
        if (coset_g) fr_scale_by_powers(a, m, coset_g);
        ofp_t omega_i = one;
        for (size_t i = 0; i < big_m; i++) {
            if (i < small_m) { fr_add_(&c[i], &a[i], &a[i + big_m]); fr_sub_(&d[i], &a[i], &a[i + big_m]); }
            else { c[i] = a[i]; d[i] = a[i]; }
            fr_mul_(&d[i], &omega_i, &d[i]);
            fr_mul_(&omega_i, &omega_i, omega);
        }
        for (size_t i = 0; i < small_m; i++) {
            fr_zero(&e[i]);
            for (size_t j = 0; j < compr; j++) fr_add_(&e[i], &e[i], &d[i + j * small_m]);
        }
        oracle_fr_radix2_fft(c, big_log, &big_omega);
        oracle_fr_radix2_fft(e, small_log, &small_omega);
        for (size_t i = 0; i < big_m; i++) a[i] = c[i];
        for (size_t i = 0; i < small_m; i++) a[i + big_m] = e[i];
    } else {
        ofp_t *U0 = c, *U1 = e, *tmp = d, inv, sz, omega_i, over_two, two;
        for (size_t i = 0; i < big_m; i++) U0[i] = a[i];
        for (size_t i = 0; i < small_m; i++) U1[i] = a[big_m + i];
        ofp_inv(&inv, &big_omega, 1);
        oracle_fr_radix2_fft(U0, big_log, &inv);
        ofp_inv(&inv, &small_omega, 1);
        oracle_fr_radix2_fft(U1, small_log, &inv);
        fr_from_size(&sz, big_m);
        ofp_inv(&inv, &sz, 1);
        for (size_t i = 0; i < big_m; i++) fr_mul_(&U0[i], &U0[i], &inv);
        fr_from_size(&sz, small_m);
        ofp_inv(&inv, &sz, 1);
        for (size_t i = 0; i < small_m; i++) fr_mul_(&U1[i], &U1[i], &inv);
        omega_i = one;
        for (size_t i = 0; i < big_m; i++) { fr_mul_(&tmp[i], &U0[i], &omega_i); fr_mul_(&omega_i, &omega_i, omega); }
        for (size_t i = small_m; i < big_m; i++) a[i] = U0[i];                       /* A_suffix */
        for (size_t i = 0; i < small_m; i++)
            for (size_t j = 1; j < compr; j++) fr_sub_(&U1[i], &U1[i], &tmp[i + j * small_m]);
        ofp_inv(&inv, omega, 1);
        omega_i = one;
        for (size_t i = 0; i < small_m; i++) { fr_mul_(&U1[i], &U1[i], &omega_i); fr_mul_(&omega_i, &omega_i, &inv); }
        fr_add_(&two, &one, &one);
        ofp_inv(&over_two, &two, 1);
        for (size_t i = 0; i < small_m; i++) {                                       /* A_prefix */
            ofp_t s, t;
            fr_add_(&s, &U0[i], &U1[i]);
            fr_sub_(&t, &U0[i], &U1[i]);
            fr_mul_(&a[i], &s, &over_two);
            fr_mul_(&a[big_m + i], &t, &over_two);
        }
        if (coset_g) {
            ofp_inv(&inv, coset_g, 1);
            fr_scale_by_powers(a, m, &inv);
        }
    }
    free(c); free(d); free(e);
}
