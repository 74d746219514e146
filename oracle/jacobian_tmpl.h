/* oracle/jacobian_tmpl.h -- Jacobian short-Weierstrass (a = 0) group law, included
 * twice by bn254.c with
 *   FT      field element type          (ofp_t for G1, ofq2_t for G2)
 *   GT_     point type                  (og1_t / og2_t)
 *   F(x)    field function prefix       (fq_##x / fq2_##x)
 *   G(x)    group function prefix       (og1_##x / og2_##x)
 *   COEFF_B pointer to curve b in FT
 * TEST INFRASTRUCTURE (see bn254.h).  Formulas restate libff's
 * alt_bn128_G1/G2::{operator+, mixed_add, dbl, to_affine_coordinates, operator==}
 * [upstream, recalled]: add-2007-bl, madd-2007-bl, dbl-2009-l with explicit
 * handling of O and of the P+P case (SURVEY.md section 7 "Hard parts").
 */

void G(zero)(GT_ *r) { F(zero)(&r->X); F(one)(&r->Y); F(zero)(&r->Z); }
int G(is_zero)(const GT_ *a) { return F(is_zero)(&a->Z); }

void G(dbl)(GT_ *r, const GT_ *a) {
    if (G(is_zero)(a)) { *r = *a; return; }
    FT A, B, C, D, E, Fv, X3, Y3, Z3, t, eightC;
    F(sqr)(&A, &a->X);            /* A = X1^2 */
    F(sqr)(&B, &a->Y);            /* B = Y1^2 */
    F(sqr)(&C, &B);               /* C = B^2 */
    F(add)(&t, &a->X, &B);
    F(sqr)(&D, &t);
    F(sub)(&D, &D, &A);
    F(sub)(&D, &D, &C);
    F(add)(&D, &D, &D);           /* D = 2((X1+B)^2 - A - C) */
    F(add)(&E, &A, &A);
    F(add)(&E, &E, &A);           /* E = 3A */
    F(sqr)(&Fv, &E);              /* F = E^2 */
    F(add)(&t, &D, &D);
    F(sub)(&X3, &Fv, &t);         /* X3 = F - 2D */
    F(add)(&eightC, &C, &C);
    F(add)(&eightC, &eightC, &eightC);
    F(add)(&eightC, &eightC, &eightC);
    F(sub)(&t, &D, &X3);
    F(mul)(&Y3, &E, &t);
    F(sub)(&Y3, &Y3, &eightC);    /* Y3 = E(D - X3) - 8C */
    F(mul)(&Z3, &a->Y, &a->Z);
    F(add)(&Z3, &Z3, &Z3);        /* Z3 = 2 Y1 Z1 */
    r->X = X3; r->Y = Y3; r->Z = Z3;
}

void G(add)(GT_ *r, const GT_ *a, const GT_ *b) {
    if (G(is_zero)(a)) { *r = *b; return; }
    if (G(is_zero)(b)) { *r = *a; return; }
    FT Z1Z1, Z2Z2, U1, U2, Z1c, Z2c, S1, S2;
    F(sqr)(&Z1Z1, &a->Z);
    F(sqr)(&Z2Z2, &b->Z);
    F(mul)(&U1, &a->X, &Z2Z2);
    F(mul)(&U2, &b->X, &Z1Z1);
    F(mul)(&Z1c, &a->Z, &Z1Z1);
    F(mul)(&Z2c, &b->Z, &Z2Z2);
    F(mul)(&S1, &a->Y, &Z2c);
    F(mul)(&S2, &b->Y, &Z1c);
    if (F(eq)(&U1, &U2) && F(eq)(&S1, &S2)) { G(dbl)(r, a); return; }
    FT H, S2mS1, I, J, rr, V, X3, Y3, Z3, t;
    F(sub)(&H, &U2, &U1);
    F(sub)(&S2mS1, &S2, &S1);
    F(add)(&t, &H, &H);
    F(sqr)(&I, &t);               /* I = (2H)^2 */
    F(mul)(&J, &H, &I);
    F(add)(&rr, &S2mS1, &S2mS1);  /* r = 2(S2-S1) */
    F(mul)(&V, &U1, &I);
    F(sqr)(&X3, &rr);
    F(sub)(&X3, &X3, &J);
    F(sub)(&X3, &X3, &V);
    F(sub)(&X3, &X3, &V);         /* X3 = r^2 - J - 2V */
    F(mul)(&t, &S1, &J);
    F(add)(&t, &t, &t);
    F(sub)(&Y3, &V, &X3);
    F(mul)(&Y3, &rr, &Y3);
    F(sub)(&Y3, &Y3, &t);         /* Y3 = r(V-X3) - 2 S1 J */
    F(add)(&Z3, &a->Z, &b->Z);
    F(sqr)(&Z3, &Z3);
    F(sub)(&Z3, &Z3, &Z1Z1);
    F(sub)(&Z3, &Z3, &Z2Z2);
    F(mul)(&Z3, &Z3, &H);         /* Z3 = ((Z1+Z2)^2 - Z1Z1 - Z2Z2) H */
    r->X = X3; r->Y = Y3; r->Z = Z3;
}

void G(mixed_add)(GT_ *r, const GT_ *a, const GT_ *b) {
    if (G(is_zero)(a)) { *r = *b; return; }
    if (G(is_zero)(b)) { *r = *a; return; }
    FT Z1Z1, U2, Z1c, S2;
    F(sqr)(&Z1Z1, &a->Z);
    F(mul)(&U2, &b->X, &Z1Z1);
    F(mul)(&Z1c, &a->Z, &Z1Z1);
    F(mul)(&S2, &b->Y, &Z1c);
    if (F(eq)(&a->X, &U2) && F(eq)(&a->Y, &S2)) { G(dbl)(r, a); return; }
    FT H, HH, I, J, rr, V, X3, Y3, Z3, t;
    F(sub)(&H, &U2, &a->X);
    F(sqr)(&HH, &H);
    F(add)(&I, &HH, &HH);
    F(add)(&I, &I, &I);           /* I = 4 HH */
    F(mul)(&J, &H, &I);
    F(sub)(&rr, &S2, &a->Y);
    F(add)(&rr, &rr, &rr);
    F(mul)(&V, &a->X, &I);
    F(sqr)(&X3, &rr);
    F(sub)(&X3, &X3, &J);
    F(sub)(&X3, &X3, &V);
    F(sub)(&X3, &X3, &V);
    F(mul)(&t, &a->Y, &J);
    F(add)(&t, &t, &t);
    F(sub)(&Y3, &V, &X3);
    F(mul)(&Y3, &rr, &Y3);
    F(sub)(&Y3, &Y3, &t);
    F(add)(&Z3, &a->Z, &H);
    F(sqr)(&Z3, &Z3);
    F(sub)(&Z3, &Z3, &Z1Z1);
    F(sub)(&Z3, &Z3, &HH);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}

void G(neg)(GT_ *r, const GT_ *a) { r->X = a->X; F(neg)(&r->Y, &a->Y); r->Z = a->Z; }

int G(eq)(const GT_ *a, const GT_ *b) {
    if (G(is_zero)(a)) return G(is_zero)(b);
    if (G(is_zero)(b)) return 0;
    FT Z1Z1, Z2Z2, l, rr, Z1c, Z2c;
    F(sqr)(&Z1Z1, &a->Z);
    F(sqr)(&Z2Z2, &b->Z);
    F(mul)(&l, &a->X, &Z2Z2);
    F(mul)(&rr, &b->X, &Z1Z1);
    if (!F(eq)(&l, &rr)) return 0;
    F(mul)(&Z1c, &a->Z, &Z1Z1);
    F(mul)(&Z2c, &b->Z, &Z2Z2);
    F(mul)(&l, &a->Y, &Z2c);
    F(mul)(&rr, &b->Y, &Z1c);
    return F(eq)(&l, &rr);
}

void G(to_affine)(GT_ *r, const GT_ *a) {
    if (G(is_zero)(a)) { G(zero)(r); return; }
    FT zi, zi2, zi3;
    F(inv)(&zi, &a->Z);
    F(sqr)(&zi2, &zi);
    F(mul)(&zi3, &zi2, &zi);
    F(mul)(&r->X, &a->X, &zi2);
    F(mul)(&r->Y, &a->Y, &zi3);
    F(one)(&r->Z);
}

int G(is_well_formed)(const GT_ *a) {
    if (G(is_zero)(a)) return 1;
    /* Y^2 = X^3 + b Z^6 */
    FT X2, Y2, Z2, X3, Z3, Z6, t;
    F(sqr)(&X2, &a->X); F(sqr)(&Y2, &a->Y); F(sqr)(&Z2, &a->Z);
    F(mul)(&X3, &a->X, &X2);
    F(mul)(&Z3, &a->Z, &Z2);
    F(sqr)(&Z6, &Z3);
    F(mul)(&t, COEFF_B, &Z6);
    F(add)(&t, &t, &X3);
    return F(eq)(&Y2, &t);
}

/* libff "scalar * point" = power<GroupT>(base, scalar.as_bigint()): MSB-first
 * double-and-add on the canonical (non-Montgomery) scalar. */
void G(mul)(GT_ *r, const GT_ *a, const ofp_t *k_mont) {
    uint64_t k[4];
    ofp_to_canonical(k, k_mont, 1);
    GT_ res; G(zero)(&res);
    int found = 0;
    for (int i = 255; i >= 0; --i) {
        if (found) G(dbl)(&res, &res);
        if ((k[i >> 6] >> (i & 63)) & 1) { found = 1; G(add)(&res, &res, a); }
    }
    *r = res;
}

/* ---- libff multi_exp_inner<multi_exp_method_BDLO12> [upstream, recalled]:
 * c = L - (L/3 - 2), L = ceil(log2 n); scalars -> canonical bigints; num_bits =
 * max bit length; for k = groups-1..0: c doublings of result; 2^c buckets;
 * bucket[id] += base; running-sum sweep from 2^c-1 down to 1 into result. */
void G(multi_exp_inner_)(GT_ *out, const GT_ *bases, const ofp_t *scalars, size_t length) {
    size_t c = oracle_bdlo12_window(length);
    uint64_t (*bn)[4] = (uint64_t (*)[4])malloc(sizeof(uint64_t[4]) * (length ? length : 1));
    size_t num_bits = 0;
    for (size_t i = 0; i < length; i++) {
        ofp_to_canonical(bn[i], &scalars[i], 1);
        size_t nb = 0;
        for (int l = 3; l >= 0; --l) if (bn[i][l]) { nb = 64 * l + (64 - __builtin_clzll(bn[i][l])); break; }
        if (nb > num_bits) num_bits = nb;
    }
    size_t num_groups = (num_bits + c - 1) / c;
    GT_ result; G(zero)(&result);
    int result_nonzero = 0;
    size_t nbuckets = (size_t)1 << c;
    GT_ *buckets = (GT_ *)malloc(sizeof(GT_) * nbuckets);
    unsigned char *bucket_nonzero = (unsigned char *)malloc(nbuckets);
    for (size_t k = num_groups - 1; k <= num_groups; k--) {
        if (result_nonzero) for (size_t i = 0; i < c; i++) G(dbl)(&result, &result);
        memset(bucket_nonzero, 0, nbuckets);
        for (size_t i = 0; i < length; i++) {
            size_t id = 0;
            for (size_t j = 0; j < c; j++) {
                size_t bit = k * c + j;
                if (bit < 256 && ((bn[i][bit >> 6] >> (bit & 63)) & 1)) id |= (size_t)1 << j;
            }
            if (id == 0) continue;
            if (bucket_nonzero[id]) G(add)(&buckets[id], &buckets[id], &bases[i]);
            else { buckets[id] = bases[i]; bucket_nonzero[id] = 1; }
        }
        GT_ running_sum; G(zero)(&running_sum);
        int running_sum_nonzero = 0;
        for (size_t i = nbuckets - 1; i > 0; i--) {
            if (bucket_nonzero[i]) {
                if (running_sum_nonzero) G(add)(&running_sum, &running_sum, &buckets[i]);
                else { running_sum = buckets[i]; running_sum_nonzero = 1; }
            }
            if (running_sum_nonzero) {
                if (result_nonzero) G(add)(&result, &result, &running_sum);
                else { result = running_sum; result_nonzero = 1; }
            }
        }
    }
    free(buckets); free(bucket_nonzero); free(bn);
    *out = result;
}

typedef struct { GT_ *out; const GT_ *bases; const ofp_t *scalars; size_t n; } G(chunk_job);
typedef struct { G(chunk_job) *jobs; size_t njobs; size_t *next; pthread_mutex_t *mu; } G(pool);

static void *G(worker)(void *arg) {
    G(pool) *p = (G(pool) *)arg;
    for (;;) {
        pthread_mutex_lock(p->mu);
        size_t i = (*p->next)++;
        pthread_mutex_unlock(p->mu);
        if (i >= p->njobs) break;
        G(multi_exp_inner_)(p->jobs[i].out, p->jobs[i].bases, p->jobs[i].scalars, p->jobs[i].n);
    }
    return NULL;
}

/* libff multi_exp<T,FieldT,Method>(.., chunks) [upstream, recalled]: total < chunks
 * or chunks == 1 -> inner; else `one = total/chunks` contiguous ranges (the last
 * takes the remainder), partials summed left to right. */
void G(multi_exp_)(GT_ *out, const GT_ *bases, const ofp_t *scalars, size_t total, size_t chunks, int threads) {
    if (total < chunks || chunks == 1) { G(multi_exp_inner_)(out, bases, scalars, total); return; }
    size_t one = total / chunks;
    GT_ *partial = (GT_ *)malloc(sizeof(GT_) * chunks);
    G(chunk_job) *jobs = (G(chunk_job) *)malloc(sizeof(G(chunk_job)) * chunks);
    for (size_t i = 0; i < chunks; i++) {
        jobs[i].out = &partial[i];
        jobs[i].bases = bases + i * one;
        jobs[i].scalars = scalars + i * one;
        jobs[i].n = (i == chunks - 1) ? total - i * one : one;
    }
    if (threads <= 1) {
        for (size_t i = 0; i < chunks; i++) G(multi_exp_inner_)(jobs[i].out, jobs[i].bases, jobs[i].scalars, jobs[i].n);
    } else {
        pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
        size_t next = 0;
        G(pool) pool = { jobs, chunks, &next, &mu };
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
        for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, G(worker), &pool);
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
        free(th);
    }
    GT_ final; G(zero)(&final);
    for (size_t i = 0; i < chunks; i++) G(add)(&final, &final, &partial[i]);
    free(partial); free(jobs);
    *out = final;
}

/* libff multi_exp_with_mixed_addition [upstream, recalled]; the same filter is
 * mirrored in-tree at /root/reference/src/utils/sparsemexp.h:12-59:
 * scalar == 0 -> skip, == 1 -> acc += base, else collect; acc + multi_exp(rest). */
void G(multi_exp_with_mixed_addition_)(GT_ *out, const GT_ *bases, const ofp_t *scalars, size_t n, size_t chunks, int threads) {
    ofp_t zero, one;
    fr_zero(&zero); fr_one(&one);
    GT_ *g = (GT_ *)malloc(sizeof(GT_) * (n ? n : 1));
    ofp_t *p = (ofp_t *)malloc(sizeof(ofp_t) * (n ? n : 1));
    size_t m = 0;
    GT_ acc; G(zero)(&acc);
    for (size_t i = 0; i < n; i++) {
        if (fq_eq(&scalars[i], &zero)) continue;
        if (fq_eq(&scalars[i], &one)) { G(add)(&acc, &acc, &bases[i]); continue; }
        p[m] = scalars[i]; g[m] = bases[i]; m++;
    }
    GT_ rest;
    G(multi_exp_)(&rest, g, p, m, chunks, threads);
    G(add)(out, &acc, &rest);
    free(g); free(p);
}

/* libff get_window_table / windowed_exp / batch_exp [upstream, recalled];
 * reference call sites /root/reference/src/utils/util.h:119-134,
 * src/prototools/interp.h:36-59. scalar_size = Fr::size_in_bits() = 254. */
void G(batch_exp_)(GT_ *out, const GT_ *base, const ofp_t *scalars, size_t n, size_t window) {
    const size_t scalar_size = 254;
    const size_t in_window = (size_t)1 << window;
    const size_t outerc = (scalar_size + window - 1) / window;
    const size_t last_in_window = (size_t)1 << (scalar_size - (outerc - 1) * window);
    GT_ *tbl = (GT_ *)malloc(sizeof(GT_) * outerc * in_window);
    GT_ gouter = *base;
    for (size_t outer = 0; outer < outerc; ++outer) {
        GT_ ginner; G(zero)(&ginner);
        size_t cur = (outer == outerc - 1) ? last_in_window : in_window;
        for (size_t inner = 0; inner < in_window; ++inner) G(zero)(&tbl[outer * in_window + inner]);
        for (size_t inner = 0; inner < cur; ++inner) {
            tbl[outer * in_window + inner] = ginner;
            G(add)(&ginner, &ginner, &gouter);
        }
        for (size_t i = 0; i < window; ++i) G(add)(&gouter, &gouter, &gouter);
    }
    for (size_t s = 0; s < n; s++) {
        uint64_t k[4];
        ofp_to_canonical(k, &scalars[s], 1);
        GT_ res = tbl[0];
        for (size_t outer = 0; outer < outerc; ++outer) {
            size_t inner = 0;
            for (size_t i = 0; i < window; ++i) {
                size_t bit = outer * window + i;
                if (bit < 256 && ((k[bit >> 6] >> (bit & 63)) & 1)) inner |= (size_t)1 << i;
            }
            G(add)(&res, &res, &tbl[outer * in_window + inner]);
        }
        out[s] = res;
    }
    free(tbl);
}

/* Test-input helper (not a libff function): out[i] = (a + i*b) * base, produced by
 * repeated addition so that large MSM inputs have known discrete logs
 * (SURVEY.md section 8c(iii)).  Outputs are un-normalised Jacobian points. */
void G(arith_bases_)(GT_ *out, const GT_ *base, const ofp_t *a_mont, const ofp_t *b_mont, size_t n) {
    GT_ cur, step;
    G(mul)(&cur, base, a_mont);
    G(mul)(&step, base, b_mont);
    for (size_t i = 0; i < n; i++) { out[i] = cur; G(add)(&cur, &cur, &step); }
}
