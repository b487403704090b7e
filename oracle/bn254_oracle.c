/*
 * oracle/bn254_oracle.c -- see bn254_oracle.h.  TEST INFRASTRUCTURE ONLY:
 * the product path must never link this file.
 *
 * Plain C, 4 x 64-bit limbs with unsigned __int128 (the arkworks Fq layout:
 * Fp<MontBackend<_,4>,4>, R = 2^256 -- utils/mont_reduction.rs:9-40 of the
 * reference uses exactly these INV / MODULUS words).
 */
#include "bn254_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fq;
typedef struct { fq x, y, z; } jac;   /* identity <=> z == 0 */
typedef struct { fq x, y; int inf; } aff;

/* p, SH/constants.metal:30-47 (BN254_BASEFIELD_MODULUS) */
static const fq FQ_P = {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
/* R mod p, SH/constants.metal:175-192 (BN254_ZERO_XR) */
static const fq FQ_R1 = {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}};
/* R^2 mod p */
static const fq FQ_R2 = {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}};
/* -p^-1 mod 2^64  (low 16 bits = 25481 = N0, SH/constants.metal:9) */
#define FQ_INV 0x87d20782e4866389ULL
/* group order r (scalar field modulus), 254 bits */
static const uint64_t FR_R[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
#define SCALAR_BITS 254u

/* ------------------------------------------------------------------ Fq --- */
static inline int fq_is_zero(const fq *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fq_eq(const fq *a, const fq *b) {
    return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline int fq_gte_p(const fq *a) {
    for (int i = 3; i >= 0; i--) {
        if (a->l[i] > FQ_P.l[i]) return 1;
        if (a->l[i] < FQ_P.l[i]) return 0;
    }
    return 1;
}
static inline void fq_sub_p(fq *a) {
    u128 bw = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - FQ_P.l[i] - bw;
        a->l[i] = (uint64_t)d;
        bw = (d >> 64) & 1;
    }
}
/* ff_add, SH/field/ff.metal:9-20 */
static inline void fq_add(fq *o, const fq *a, const fq *b) {
    u128 c = 0;
    fq t;
    for (int i = 0; i < 4; i++) {
        c += (u128)a->l[i] + b->l[i];
        t.l[i] = (uint64_t)c;
        c >>= 64;
    }
    /* p < 2^254 so a+b < 2^255: no carry out of 256 bits */
    if (fq_gte_p(&t)) fq_sub_p(&t);
    *o = t;
}
/* ff_sub, SH/field/ff.metal:22-35 */
static inline void fq_sub(fq *o, const fq *a, const fq *b) {
    u128 bw = 0;
    fq t;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - bw;
        t.l[i] = (uint64_t)d;
        bw = (d >> 64) & 1;
    }
    if (bw) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)t.l[i] + FQ_P.l[i];
            t.l[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    *o = t;
}
static inline void fq_neg(fq *o, const fq *a) {
    if (fq_is_zero(a)) { *o = *a; return; }
    fq z = {{0, 0, 0, 0}};
    fq_sub(o, &z, a);
}
static inline void fq_dbl(fq *o, const fq *a) { fq_add(o, a, a); }

/* Montgomery product a*b*R^-1 mod p: CIOS (mont_mul_cios, SH/mont_backend/mont.metal:105-181,
 * restated on 64-bit limbs) */
static void fq_mul(fq *o, const fq *a, const fq *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)t[j] + (u128)a->l[j] * b->l[i];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * FQ_INV;
        c = (u128)t[0] + (u128)m * FQ_P.l[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)t[j] + (u128)m * FQ_P.l[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fq r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fq_gte_p(&r)) fq_sub_p(&r);
    *o = r;
}
static inline void fq_sqr(fq *o, const fq *a) { fq_mul(o, a, a); }
static inline void fq_to_mont(fq *o, const fq *a) { fq_mul(o, a, &FQ_R2); }
/* raw_reduction, utils/mont_reduction.rs:15-40 */
static inline void fq_from_mont(fq *o, const fq *a) {
    fq one = {{1, 0, 0, 0}};
    fq_mul(o, a, &one);
}
/* a^(p-2) */
static void fq_inv(fq *o, const fq *a) {
    uint64_t e[4] = {FQ_P.l[0] - 2, FQ_P.l[1], FQ_P.l[2], FQ_P.l[3]};
    fq acc = FQ_R1, base = *a;
    for (int i = 0; i < 256; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) fq_mul(&acc, &acc, &base);
        fq_sqr(&base, &base);
    }
    *o = acc;
}
static inline void fq_load(fq *o, const uint32_t w[8]) {
    for (int i = 0; i < 4; i++) o->l[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}
static inline void fq_store(uint32_t w[8], const fq *a) {
    for (int i = 0; i < 4; i++) {
        w[2 * i] = (uint32_t)a->l[i];
        w[2 * i + 1] = (uint32_t)(a->l[i] >> 32);
    }
}

/* ------------------------------------------------------------------ G1 --- */
static inline void jac_set_inf(jac *o) { o->x = FQ_R1; o->y = FQ_R1; memset(&o->z, 0, sizeof(fq)); } /* (1,1,0): SH/constants.metal:67-120 */
static inline int jac_is_inf(const jac *a) { return fq_is_zero(&a->z); }

/* dbl-2009-l, SH/curve/jacobian.metal:11-44 */
static void jac_dbl(jac *o, const jac *p) {
    if (jac_is_inf(p)) { jac_set_inf(o); return; }
    fq a, b, c, d, e, f, t, x3, y3, z3;
    fq_sqr(&a, &p->x);
    fq_sqr(&b, &p->y);
    fq_sqr(&c, &b);
    fq_add(&t, &p->x, &b);
    fq_sqr(&t, &t);
    fq_sub(&t, &t, &a);
    fq_sub(&t, &t, &c);
    fq_dbl(&d, &t);
    fq_dbl(&e, &a);
    fq_add(&e, &e, &a);
    fq_sqr(&f, &e);
    fq_dbl(&t, &d);
    fq_sub(&x3, &f, &t);
    fq_dbl(&c, &c); fq_dbl(&c, &c); fq_dbl(&c, &c);
    fq_sub(&t, &d, &x3);
    fq_mul(&y3, &e, &t);
    fq_sub(&y3, &y3, &c);
    fq_mul(&z3, &p->y, &p->z);
    fq_dbl(&z3, &z3);
    o->x = x3; o->y = y3; o->z = z3;
}

/* add-2007-bl, SH/curve/jacobian.metal:46-100.  The reference decides "same point"
 * by limb equality (curve/utils.metal:14-26); arkworks -- the behaviour to match --
 * implements the complete group law, so equality is decided projectively (H == 0). */
static void jac_add(jac *o, const jac *a, const jac *b) {
    if (jac_is_inf(a)) { *o = *b; return; }
    if (jac_is_inf(b)) { *o = *a; return; }
    fq z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t, x3, y3, z3;
    fq_sqr(&z1z1, &a->z);
    fq_sqr(&z2z2, &b->z);
    fq_mul(&u1, &a->x, &z2z2);
    fq_mul(&u2, &b->x, &z1z1);
    fq_mul(&s1, &a->y, &b->z); fq_mul(&s1, &s1, &z2z2);
    fq_mul(&s2, &b->y, &a->z); fq_mul(&s2, &s2, &z1z1);
    fq_sub(&h, &u2, &u1);
    fq_sub(&r, &s2, &s1);
    if (fq_is_zero(&h)) {
        if (fq_is_zero(&r)) { jac_dbl(o, a); return; }
        jac_set_inf(o); return;
    }
    fq_dbl(&r, &r);
    fq_dbl(&i, &h); fq_sqr(&i, &i);
    fq_mul(&j, &h, &i);
    fq_mul(&v, &u1, &i);
    fq_sqr(&x3, &r);
    fq_sub(&x3, &x3, &j);
    fq_dbl(&t, &v);
    fq_sub(&x3, &x3, &t);
    fq_sub(&t, &v, &x3);
    fq_mul(&y3, &r, &t);
    fq_mul(&t, &s1, &j); fq_dbl(&t, &t);
    fq_sub(&y3, &y3, &t);
    fq_mul(&z3, &a->z, &b->z);
    fq_mul(&z3, &z3, &h);
    fq_dbl(&z3, &z3);
    o->x = x3; o->y = y3; o->z = z3;
}

/* madd-2007-bl, SH/curve/jacobian.metal:102-166 (b affine, Montgomery coordinates) */
static void jac_madd(jac *o, const jac *a, const aff *b) {
    if (b->inf) { *o = *a; return; }
    if (jac_is_inf(a)) { o->x = b->x; o->y = b->y; o->z = FQ_R1; return; }
    fq z1z1, u2, s2, h, hh, i, j, r, v, t, x3, y3, z3;
    fq_sqr(&z1z1, &a->z);
    fq_mul(&u2, &b->x, &z1z1);
    fq_mul(&s2, &b->y, &a->z); fq_mul(&s2, &s2, &z1z1);
    fq_sub(&h, &u2, &a->x);
    fq_sub(&r, &s2, &a->y);
    if (fq_is_zero(&h)) {
        if (fq_is_zero(&r)) { jac_dbl(o, a); return; }
        jac_set_inf(o); return;
    }
    fq_sqr(&hh, &h);
    fq_dbl(&i, &hh); fq_dbl(&i, &i);
    fq_mul(&j, &h, &i);
    fq_dbl(&r, &r);
    fq_mul(&v, &a->x, &i);
    fq_sqr(&x3, &r);
    fq_sub(&x3, &x3, &j);
    fq_dbl(&t, &v);
    fq_sub(&x3, &x3, &t);
    fq_sub(&t, &v, &x3);
    fq_mul(&y3, &r, &t);
    fq_mul(&t, &a->y, &j); fq_dbl(&t, &t);
    fq_sub(&y3, &y3, &t);
    fq_add(&z3, &a->z, &h); fq_sqr(&z3, &z3);
    fq_sub(&z3, &z3, &z1z1);
    fq_sub(&z3, &z3, &hh);
    o->x = x3; o->y = y3; o->z = z3;
}
static inline void aff_neg(aff *o, const aff *a) { o->x = a->x; fq_neg(&o->y, &a->y); o->inf = a->inf; }
static inline void jac_neg(jac *o, const jac *a) { o->x = a->x; fq_neg(&o->y, &a->y); o->z = a->z; }

static int jac_to_affine(const jac *a, fq *x_std, fq *y_std) {
    if (jac_is_inf(a)) { memset(x_std, 0, sizeof(fq)); memset(y_std, 0, sizeof(fq)); return 1; }
    fq zi, zi2, zi3, x, y;
    fq_inv(&zi, &a->z);
    fq_sqr(&zi2, &zi);
    fq_mul(&zi3, &zi2, &zi);
    fq_mul(&x, &a->x, &zi2);
    fq_mul(&y, &a->y, &zi3);
    fq_from_mont(x_std, &x);
    fq_from_mont(y_std, &y);
    return 0;
}

static void jac_load(jac *o, const uint32_t w[24]) { fq_load(&o->x, w); fq_load(&o->y, w + 8); fq_load(&o->z, w + 16); }
static void jac_store(uint32_t w[24], const jac *a) { fq_store(w, &a->x); fq_store(w + 8, &a->y); fq_store(w + 16, &a->z); }

static void aff_load(aff *o, const uint32_t *w, uint32_t form, int inf) {
    fq_load(&o->x, w);
    fq_load(&o->y, w + 8);
    o->inf = inf;
    if (form == ORACLE_FORM_STD) { fq_to_mont(&o->x, &o->x); fq_to_mont(&o->y, &o->y); }
}

/* scalar bit access, scalar = 8 LE u32 words */
static inline uint32_t sc_bit(const uint32_t *s, unsigned i) { return (s[i >> 5] >> (i & 31)) & 1u; }
static inline uint64_t sc_bits(const uint32_t *s, unsigned off, unsigned cnt) { /* cnt <= 32 */
    uint64_t v = 0;
    for (unsigned k = 0; k < cnt; k++) {
        unsigned i = off + k;
        if (i < 256) v |= (uint64_t)sc_bit(s, i) << k;
    }
    return v;
}

static void jac_scalar_mul(jac *o, const aff *b, const uint32_t k[8]) {
    jac acc;
    jac_set_inf(&acc);
    for (int i = 255; i >= 0; i--) {
        jac_dbl(&acc, &acc);
        if (sc_bit(k, (unsigned)i)) jac_madd(&acc, &acc, b);
    }
    *o = acc;
}

/* ------------------------------------------------------------- C API: Fq --- */
void oracle_fq_constants(uint32_t p[8], uint32_t r1[8], uint32_t r2[8], uint64_t *inv64) {
    fq_store(p, &FQ_P); fq_store(r1, &FQ_R1); fq_store(r2, &FQ_R2); *inv64 = FQ_INV;
}
void oracle_fq_to_mont(const uint32_t a[8], uint32_t out[8]) { fq x; fq_load(&x, a); fq_to_mont(&x, &x); fq_store(out, &x); }
void oracle_fq_from_mont(const uint32_t a[8], uint32_t out[8]) { fq x; fq_load(&x, a); fq_from_mont(&x, &x); fq_store(out, &x); }
void oracle_fq_mont_mul(const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) {
    fq x, y; fq_load(&x, a); fq_load(&y, b); fq_mul(&x, &x, &y); fq_store(out, &x);
}
void oracle_fq_add(const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) {
    fq x, y; fq_load(&x, a); fq_load(&y, b); fq_add(&x, &x, &y); fq_store(out, &x);
}
void oracle_fq_sub(const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) {
    fq x, y; fq_load(&x, a); fq_load(&y, b); fq_sub(&x, &x, &y); fq_store(out, &x);
}
void oracle_fq_inv_mont(const uint32_t a[8], uint32_t out[8]) { fq x; fq_load(&x, a); fq_inv(&x, &x); fq_store(out, &x); }

/* ------------------------------------------------------------- C API: G1 --- */
void oracle_g1_dbl(const uint32_t a[24], uint32_t out[24]) { jac p; jac_load(&p, a); jac_dbl(&p, &p); jac_store(out, &p); }
/* 2^k * a: k doublings (the window-table levels T_j = 2^(c*j) P of the engine's resident path, row f4; checker only) */
void oracle_g1_dbl_n(const uint32_t a[24], uint32_t k, uint32_t out[24]) {
    jac p; jac_load(&p, a);
    for (uint32_t i = 0; i < k; i++) jac_dbl(&p, &p);
    jac_store(out, &p);
}
void oracle_g1_add(const uint32_t a[24], const uint32_t b[24], uint32_t out[24]) {
    jac p, q, r; jac_load(&p, a); jac_load(&q, b); jac_add(&r, &p, &q); jac_store(out, &r);
}
void oracle_g1_madd(const uint32_t a[24], const uint32_t b[16], uint32_t out[24]) {
    jac p, r; aff q; jac_load(&p, a); aff_load(&q, b, ORACLE_FORM_MONT, 0); jac_madd(&r, &p, &q); jac_store(out, &r);
}
int oracle_g1_to_affine_std(const uint32_t a[24], uint32_t out_xy[16]) {
    jac p; fq x, y; jac_load(&p, a);
    int inf = jac_to_affine(&p, &x, &y);
    fq_store(out_xy, &x); fq_store(out_xy + 8, &y);
    return inf;
}
void oracle_g1_scalar_mul(const uint32_t base_xy_std[16], const uint32_t k[8], uint32_t out[24]) {
    aff b; jac r; aff_load(&b, base_xy_std, ORACLE_FORM_STD, 0); jac_scalar_mul(&r, &b, k); jac_store(out, &r);
}

static void finish(const jac *r, uint32_t out_xy_std[16], uint8_t *out_inf, uint32_t out_jac[24]) {
    fq x, y;
    int inf = jac_to_affine(r, &x, &y);
    if (out_xy_std) { fq_store(out_xy_std, &x); fq_store(out_xy_std + 8, &y); }
    if (out_inf) *out_inf = (uint8_t)inf;
    if (out_jac) jac_store(out_jac, r);
}

static aff *load_bases(const uint32_t *bases, uint32_t form, const uint8_t *inf, size_t n) {
    aff *b = (aff *)malloc(sizeof(aff) * (n ? n : 1));
    if (!b) return NULL;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) aff_load(&b[i], bases + 16 * (size_t)i, form, inf ? inf[i] != 0 : 0);
    return b;
}

int oracle_threads_available(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------- naive --- */
int oracle_msm_naive(const uint32_t *bases, uint32_t form, const uint8_t *inf, const uint32_t *scalars,
                     size_t n, uint32_t out_xy_std[16], uint8_t *out_inf, uint32_t out_jac[24]) {
    if (!bases || !scalars || n == 0) return -1;
    aff *b = load_bases(bases, form, inf, n);
    if (!b) return -2;
    jac acc;
    jac_set_inf(&acc);
    for (size_t i = 0; i < n; i++) {
        jac t;
        if (b[i].inf) continue;
        jac_scalar_mul(&t, &b[i], scalars + 8 * i);
        jac_add(&acc, &acc, &t);
    }
    free(b);
    finish(&acc, out_xy_std, out_inf, out_jac);
    return 0;
}

/* ------------------------------------------------ arkworks-0.4 Pippenger --- */
/* ark_std::log2: ceil(log2(x)) */
static unsigned ark_log2(size_t x) {
    if (x <= 1) return 0;
    unsigned l = 0;
    size_t v = x - 1;
    while (v) { l++; v >>= 1; }
    return l;
}
/* ark-ec 0.4.1 scalar_mul/variable_base/mod.rs: make_digits (signed radix-2^w) */
static void ark_make_digits(const uint32_t *s, unsigned w, unsigned num_bits, int64_t *digits, unsigned digits_count) {
    uint64_t radix = 1ULL << w, mask = radix - 1, carry = 0;
    (void)num_bits;
    for (unsigned i = 0; i < digits_count; i++) {
        uint64_t coef = carry + (sc_bits(s, i * w, w) & mask);
        carry = (coef + radix / 2) >> w;
        digits[i] = (int64_t)coef - (int64_t)(carry << w);
    }
    digits[digits_count - 1] += (int64_t)(carry << w);
}

int oracle_msm_pippenger(const uint32_t *bases, uint32_t form, const uint8_t *inf, const uint32_t *scalars,
                         size_t n, int threads, uint32_t out_xy_std[16], uint8_t *out_inf, uint32_t out_jac[24]) {
    if (!bases || !scalars || n == 0) return -1;
    aff *b = load_bases(bases, form, inf, n);
    if (!b) return -2;
    /* c = 3 for n < 32 else ln_without_floats(n) + 2, ln_without_floats(a) = log2(a)*69/100 */
    unsigned c = n < 32 ? 3 : (ark_log2(n) * 69 / 100) + 2;
    unsigned W = (SCALAR_BITS + c - 1) / c;
    int64_t *dig = (int64_t *)malloc(sizeof(int64_t) * n * W);
    jac *wsum = (jac *)malloc(sizeof(jac) * W);
    if (!dig || !wsum) { free(b); free(dig); free(wsum); return -2; }
#ifdef _OPENMP
    int nt = threads > 0 ? threads : omp_get_max_threads();
#else
    int nt = 1; (void)threads;
#endif
#pragma omp parallel for schedule(static) num_threads(nt)
    for (long i = 0; i < (long)n; i++) ark_make_digits(scalars + 8 * (size_t)i, c, SCALAR_BITS, dig + (size_t)i * W, W);
    int oom = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
    for (int w = 0; w < (int)W; w++) {
        size_t nb = (size_t)1 << c; /* arkworks allocates 1<<c buckets and walks all of them */
        jac *bk = (jac *)malloc(sizeof(jac) * nb);
        if (!bk) { oom = 1; continue; }
        for (size_t k = 0; k < nb; k++) jac_set_inf(&bk[k]);
        for (size_t i = 0; i < n; i++) {
            int64_t d = dig[i * W + (size_t)w];
            if (d > 0) jac_madd(&bk[d - 1], &bk[d - 1], &b[i]);
            else if (d < 0) { aff nb_; aff_neg(&nb_, &b[i]); jac_madd(&bk[-d - 1], &bk[-d - 1], &nb_); }
        }
        jac run, res;
        jac_set_inf(&run); jac_set_inf(&res);
        for (size_t k = nb; k-- > 0;) { jac_add(&run, &run, &bk[k]); jac_add(&res, &res, &run); }
        wsum[w] = res;
        free(bk);
    }
    jac total;
    jac_set_inf(&total);
    for (int w = (int)W - 1; w >= 1; w--) {
        jac_add(&total, &total, &wsum[w]);
        for (unsigned k = 0; k < c; k++) jac_dbl(&total, &total);
    }
    jac_add(&total, &total, &wsum[0]);
    free(b); free(dig); free(wsum);
    if (oom) return -2;
    finish(&total, out_xy_std, out_inf, out_jac);
    return 0;
}

/* ------------------------------------ reference stage mirrors (cuZK path) --- */
uint32_t oracle_ref_window_bits(size_t n) { /* metal_msm.rs:661-673 */
    if (n < 16384) return 8;
    if (n < 524288) return 13;
    if (n <= 16777216) return 15;
    return 16;
}
uint32_t oracle_num_windows(uint32_t w) { return (SCALAR_BITS + w - 1) / w; } /* ceil(254/w), metal_msm.rs:84-85 */

/* extract_word_from_bytes_le, SH/cuzk/extract_word_from_bytes_le.metal:7-31, on the 16
 * big-endian-ordered halfwords built at convert_point...metal:83-91 */
static uint32_t ref_extract_word(const uint32_t hw[16], uint32_t word_idx, uint32_t window) {
    uint32_t start_idx = 15 - ((word_idx * window + window) / 16);
    uint32_t end_idx = 15 - ((word_idx * window) / 16);
    uint32_t start_off = (word_idx * window + window) % 16;
    uint32_t end_off = (word_idx * window) % 16;
    uint32_t mask = 0, word;
    if (start_off > 0) mask = (2u << (start_off - 1)) - 1;
    if (start_idx == end_idx) word = (hw[start_idx] & mask) >> end_off;
    else { word = (hw[start_idx] & mask) << (16 - end_off); word += hw[end_idx] >> end_off; }
    return word;
}
/* K1 scalar half: convert_point_coords_and_decompose_scalars.metal:80-121 */
void oracle_decompose_signed(const uint32_t *scalars, size_t n, uint32_t window, uint32_t *chunks) {
    uint32_t W = oracle_num_windows(window);
    uint32_t l = 1u << window, s = l / 2;
    for (size_t id = 0; id < n; id++) {
        uint32_t hw[16];
        for (uint32_t i = 0; i < 8; i++) {
            uint32_t v = scalars[id * 8 + i];
            hw[15 - 2 * i] = v & 0xFFFFu;
            hw[15 - 2 * i - 1] = v >> 16;
        }
        uint32_t carry = 0;
        for (uint32_t i = 0; i < W; i++) {
            uint32_t chunk;
            if (i < W - 1) chunk = ref_extract_word(hw, i, window);
            else chunk = hw[0] >> ((((W * window - 256u) + 16u) - window) & 31u); /* uint32 wrap-around, Appendix B */
            int32_t slice = (int32_t)(chunk + carry);
            if (slice >= (int32_t)s) { slice = ((int32_t)l - slice) * (-1); carry = 1; }
            else carry = 0;
            chunks[(size_t)i * n + id] = (uint32_t)slice + s;
        }
    }
}
/* K2: serial CSR->CSC counting sort, transpose.metal:27-64 / tests/cuzk/transpose.rs:95-118 */
void oracle_transpose(const uint32_t *chunks, size_t n, uint32_t W, uint32_t C, uint32_t *col_ptr, uint32_t *val_idxs) {
    uint32_t *curr = (uint32_t *)calloc(C, sizeof(uint32_t));
    for (uint32_t w = 0; w < W; w++) {
        uint32_t *cp = col_ptr + (size_t)w * (C + 1);
        const uint32_t *col = chunks + (size_t)w * n;
        uint32_t *vi = val_idxs + (size_t)w * n;
        memset(cp, 0, sizeof(uint32_t) * (C + 1));
        memset(curr, 0, sizeof(uint32_t) * C);
        for (size_t j = 0; j < n; j++) cp[col[j] + 1]++;
        for (uint32_t i = 1; i < C + 1; i++) cp[i] += cp[i - 1];
        for (size_t j = 0; j < n; j++) { uint32_t loc = cp[col[j]] + curr[col[j]]++; vi[loc] = (uint32_t)j; }
    }
    free(curr);
}

int oracle_msm_cuzk(const uint32_t *bases, uint32_t form, const uint8_t *inf, const uint32_t *scalars,
                    size_t n, uint32_t window, uint32_t out_xy_std[16], uint8_t *out_inf, uint32_t out_jac[24]) {
    if (!bases || !scalars || n == 0) return -1;
    if (window == 0) window = oracle_ref_window_bits(n);
    if (window < 2 || window > 20) return -1;
    uint32_t W = oracle_num_windows(window), C = 1u << window, H = C / 2;
    aff *b = load_bases(bases, form, inf, n);
    uint32_t *chunks = (uint32_t *)malloc(sizeof(uint32_t) * n * W);
    uint32_t *col_ptr = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)W * (C + 1));
    uint32_t *val = (uint32_t *)malloc(sizeof(uint32_t) * n * W);
    jac *wsum = (jac *)malloc(sizeof(jac) * W);
    if (!b || !chunks || !col_ptr || !val || !wsum) { free(b); free(chunks); free(col_ptr); free(val); free(wsum); return -2; }
    oracle_decompose_signed(scalars, n, window, chunks);
    oracle_transpose(chunks, n, W, C, col_ptr, val);
#pragma omp parallel for schedule(dynamic, 1)
    for (int w = 0; w < (int)W; w++) {
        /* SMVP (smvp.metal:46-105): slot t>0 = sum(digit=+t) - sum(digit=-t); slot 0 = -sum(digit=-H), magnitude H */
        jac *bk = (jac *)malloc(sizeof(jac) * H);
        const uint32_t *cp = col_ptr + (size_t)w * (C + 1);
        const uint32_t *vi = val + (size_t)w * n;
        for (uint32_t t = 0; t < H; t++) {
            jac pos, negs;
            jac_set_inf(&pos); jac_set_inf(&negs);
            if (t > 0) for (uint32_t k = cp[H + t]; k < cp[H + t + 1]; k++) jac_madd(&pos, &pos, &b[vi[k]]);
            uint32_t row = t > 0 ? H - t : 0;
            for (uint32_t k = cp[row]; k < cp[row + 1]; k++) jac_madd(&negs, &negs, &b[vi[k]]);
            jac_neg(&negs, &negs);
            jac_add(&bk[t], &pos, &negs);
        }
        /* PBPR (pbpr.metal:33-148) computes sum_k k*B_k in wg partials; serial running sum gives the
         * same group element: magnitudes H (slot 0), H-1, ..., 1 */
        jac run, res;
        jac_set_inf(&run); jac_set_inf(&res);
        jac_add(&run, &run, &bk[0]); jac_add(&res, &res, &run);
        for (uint32_t t = H - 1; t >= 1; t--) { jac_add(&run, &run, &bk[t]); jac_add(&res, &res, &run); }
        wsum[w] = res;
        free(bk);
    }
    /* Horner high->low with multiplier 2^w, metal_msm.rs:249-258 */
    jac total;
    jac_set_inf(&total);
    for (int w = (int)W - 1; w >= 0; w--) {
        for (uint32_t k = 0; k < window; k++) jac_dbl(&total, &total);
        jac_add(&total, &total, &wsum[w]);
    }
    free(b); free(chunks); free(col_ptr); free(val); free(wsum);
    finish(&total, out_xy_std, out_inf, out_jac);
    return 0;
}

/* ---- stage mirrors for the stage-level GPU tests (counterparts of tests/cuzk/smvp.rs:119-303 and pbpr.rs:26-247) ---------
 * digits: W x n signed digits (int32, window-major).  Bucket b of window w collects the points whose digit has MAGNITUDE b + 1,
 * negated when the digit is negative: the SMVP sign folding of smvp.metal:46-105 with the engine's index convention (the
 * reference keeps magnitude t in slot t and magnitude H in slot 0).  out: W*nb Jacobian Montgomery points. */
int oracle_bucket_sums(const uint32_t *bases, uint32_t form, const uint8_t *inf, const int32_t *digits, size_t n, uint32_t W,
                       uint32_t nb, uint32_t *out_jac) {
    if (!bases || !digits || !out_jac || n == 0) return -1;
    aff *b = load_bases(bases, form, inf, n);
    if (!b) return -2;
#pragma omp parallel for schedule(dynamic, 1)
    for (int w = 0; w < (int)W; w++) {
        jac *bk = (jac *)malloc(sizeof(jac) * nb);
        for (uint32_t t = 0; t < nb; t++) jac_set_inf(&bk[t]);
        for (size_t i = 0; i < n; i++) {
            int32_t d = digits[(size_t)w * n + i];
            if (d == 0 || b[i].inf) continue;
            uint32_t mag = (uint32_t)(d < 0 ? -d : d);
            if (mag > nb) continue; /* caller error: left out so that the comparison fails */
            aff q = b[i];
            if (d < 0) aff_neg(&q, &q);
            jac_madd(&bk[mag - 1], &bk[mag - 1], &q);
        }
        for (uint32_t t = 0; t < nb; t++) jac_store(out_jac + ((size_t)w * nb + t) * 24, &bk[t]);
        free(bk);
    }
    free(b);
    return 0;
}
/* The weights of the bucket reduction pushed to bit sums (what the engine's reduction produces instead of the running sums
 * of pbpr.metal:33-148): Q[w][u] = sum of the buckets whose index has bit u set (u < kb), Q[w][kb] = sum of all buckets, so
 * that sum_b (b+1)*B[w][b] = Q[w][kb] + sum_u 2^u Q[w][u].  buckets: W*nb Jacobian points; out: W*(kb+1). */
int oracle_bit_sums(const uint32_t *buckets_jac, uint32_t W, uint32_t nb, uint32_t kb, uint32_t *out_jac) {
    if (!buckets_jac || !out_jac) return -1;
    for (uint32_t w = 0; w < W; w++)
        for (uint32_t u = 0; u <= kb; u++) {
            jac acc;
            jac_set_inf(&acc);
            for (uint32_t t = 0; t < nb; t++) {
                if (u < kb && !((t >> u) & 1u)) continue;
                jac q;
                jac_load(&q, buckets_jac + ((size_t)w * nb + t) * 24);
                jac_add(&acc, &acc, &q);
            }
            jac_store(out_jac + ((size_t)w * (kb + 1) + u) * 24, &acc);
        }
    return 0;
}

/* ------------------------------------------------------ synthetic inputs --- */
static inline uint64_t splitmix64(uint64_t *st) {
    uint64_t z = (*st += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static int sc_lt_r(const uint64_t v[4]) {
    for (int i = 3; i >= 0; i--) {
        if (v[i] < FR_R[i]) return 1;
        if (v[i] > FR_R[i]) return 0;
    }
    return 0;
}
/* element i is a pure function of (seed, i): stream i = SplitMix64 seeded with seed + i*0xD1342543DE82EF95,
 * rejection-sampled on the low 254 bits */
void oracle_gen_scalars(uint64_t seed, size_t n, int nonzero, uint32_t *out) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) {
        uint64_t st = seed + (uint64_t)i * 0xD1342543DE82EF95ULL;
        uint64_t v[4];
        for (;;) {
            for (int k = 0; k < 4; k++) v[k] = splitmix64(&st);
            v[3] &= 0x3FFFFFFFFFFFFFFFULL;
            if (!sc_lt_r(v)) continue;
            if (nonzero && (v[0] | v[1] | v[2] | v[3]) == 0) continue;
            break;
        }
        for (int k = 0; k < 4; k++) { out[8 * i + 2 * k] = (uint32_t)v[k]; out[8 * i + 2 * k + 1] = (uint32_t)(v[k] >> 32); }
    }
}

/* P_i = k_i*G by an 8-bit fixed-base table (32 x 255 affine multiples of G: <=32 mixed adds per
 * point), normalised to affine with one Fermat inversion per point */
void oracle_gen_bases_from_logs(const uint32_t *k, size_t n, uint32_t form, uint32_t *out_xy) {
    enum { WB = 8, NW = 32, TS = 255 };
    aff *tab = (aff *)malloc(sizeof(aff) * NW * TS);
    aff g;
    g.x = FQ_R1; fq_dbl(&g.y, &FQ_R1); g.inf = 0; /* (1,2) */
    jac base;
    base.x = g.x; base.y = g.y; base.z = FQ_R1;
    for (int w = 0; w < NW; w++) {
        jac acc = base;
        for (int d = 1; d <= TS; d++) {
            fq x, y;
            jac_to_affine(&acc, &x, &y);
            aff *e = &tab[w * TS + d - 1];
            fq_to_mont(&e->x, &x); fq_to_mont(&e->y, &y); e->inf = 0;
            jac_add(&acc, &acc, &base);
        }
        for (int d = 0; d < WB; d++) jac_dbl(&base, &base);
    }
#pragma omp parallel for schedule(dynamic, 64)
    for (long i = 0; i < (long)n; i++) {
        jac acc;
        jac_set_inf(&acc);
        const uint32_t *s = k + 8 * (size_t)i;
        for (int w = 0; w < NW; w++) {
            uint32_t d = (s[w >> 2] >> (8 * (w & 3))) & 0xFFu;
            if (d) jac_madd(&acc, &acc, &tab[w * TS + d - 1]);
        }
        fq x, y;
        int inf = jac_to_affine(&acc, &x, &y);
        (void)inf;
        if (form == ORACLE_FORM_MONT) { fq_to_mont(&x, &x); fq_to_mont(&y, &y); }
        fq_store(out_xy + 16 * (size_t)i, &x);
        fq_store(out_xy + 16 * (size_t)i + 8, &y);
    }
    free(tab);
}

/* ------------------------------------------------ arkworks compressed images --- */
/* sqrt in Fq for p = 3 (mod 4): a^((p+1)/4), checked by squaring.  Montgomery in/out.  Returns 0 if a is a non-residue. */
static int fq_sqrt(fq *o, const fq *a) {
    /* (p+1)/4 from the limbs of p: (p+1) has no carry out of limb 0 (p ends in ...47) */
    uint64_t q[4] = {FQ_P.l[0] + 1, FQ_P.l[1], FQ_P.l[2], FQ_P.l[3]};
    uint64_t e[4];
    for (int i = 0; i < 4; i++) e[i] = (q[i] >> 2) | (i < 3 ? q[i + 1] << 62 : 0);
    fq acc = FQ_R1, base = *a, chk;
    for (int i = 0; i < 256; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) fq_mul(&acc, &acc, &base);
        fq_sqr(&base, &base);
    }
    fq_sqr(&chk, &acc);
    *o = acc;
    return fq_eq(&chk, a);
}
/* integer comparison a > b of canonical standard-form values */
static int fq_gt(const fq *a, const fq *b) {
    for (int i = 3; i >= 0; i--)
        if (a->l[i] != b->l[i]) return a->l[i] > b->l[i];
    return 0;
}
size_t oracle_g1_decompress(const uint8_t *compressed, size_t n, uint32_t form, uint32_t *out_xy, uint8_t *out_inf) {
    size_t first_bad = 0;
    for (size_t i = 0; i < n; i++) {
        uint32_t w[8];
        memcpy(w, compressed + 32 * i, 32);
        const int neg = (int)(w[7] >> 31), inf = (int)((w[7] >> 30) & 1);
        w[7] &= 0x3FFFFFFFu;
        fq x, y, ny, t, three, ys, nys;
        fq_load(&x, w);
        memset(out_xy + 16 * i, 0, 64);
        out_inf[i] = 0;
        if ((neg && inf) || fq_gte_p(&x)) { if (!first_bad) first_bad = i + 1; continue; }
        if (inf) { out_inf[i] = 1; continue; }
        fq_to_mont(&x, &x);
        fq_sqr(&t, &x); fq_mul(&t, &t, &x);
        fq_add(&three, &FQ_R1, &FQ_R1); fq_add(&three, &three, &FQ_R1);
        fq_add(&t, &t, &three);
        if (!fq_sqrt(&y, &t)) { if (!first_bad) first_bad = i + 1; continue; }
        fq_neg(&ny, &y);
        fq_from_mont(&ys, &y); fq_from_mont(&nys, &ny);
        /* get_ys_from_x_unchecked orders (smaller, larger); YIsNegative selects the larger */
        const int y_is_larger = fq_gt(&ys, &nys);
        const fq *pick = (y_is_larger == neg) ? &y : &ny;
        fq ox = x, oy = *pick;
        if (form == ORACLE_FORM_STD) { fq_from_mont(&ox, &ox); fq_from_mont(&oy, &oy); }
        fq_store(out_xy + 16 * i, &ox);
        fq_store(out_xy + 16 * i + 8, &oy);
    }
    return first_bad;
}
void oracle_g1_compress(const uint32_t *bases_xy, uint32_t form, const uint8_t *inf, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (inf && inf[i]) {
            w[7] = 1u << 30;
        } else {
            fq x, y, ny;
            fq_load(&x, bases_xy + 16 * i); fq_load(&y, bases_xy + 16 * i + 8);
            if (form == ORACLE_FORM_MONT) { fq_from_mont(&x, &x); fq_from_mont(&y, &y); }
            /* standard-form negation: p - y (y != 0 on this curve) */
            fq zero = {{0, 0, 0, 0}};
            fq_sub(&ny, &zero, &y);
            fq_store(w, &x);
            if (fq_gt(&y, &ny)) w[7] |= 1u << 31;
        }
        memcpy(out + 32 * i, w, 32);
    }
}
