/* CPU restatement of the BN254 hot path — TEST INFRASTRUCTURE ONLY (see bn254_oracle.h).
 *
 * PARITY UNPINNED: follows the published algorithm of gnark-crypto v0.19.0
 * ecc/bn254 (absent from /root/reference, pinned by reference go.mod:5), as exercised by the
 * reference call sites
 *   bn254.Pair                cpabe/bsw07/bsw07_cpabe.go:75,184; access/tree/access_tree_node.go:106-119
 *   bn254.PairingCheck        signature/bls01_signature/bls_signature.go:81
 *   G1/G2 ScalarMultiplication signature/bls01_signature/bls_signature.go:45,63
 *   GT.Exp / Mul / Div        access/tree/access_tree_node.go:114,156-157
 * Structure restated:
 *   fp      4x64-bit Montgomery (CIOS), values always fully reduced to [0,p)
 *   E2      Fp[i]/(i^2+1);  E6 = E2[v]/(v^3-(9+i));  E12 = E6[w]/(w^2-v)
 *   Miller  homogeneous-projective doubling / mixed addition lines (Costello-Lange-Naehrig
 *           ePrint 2013/722 §4.3), line = r0*yP + r1*xP*w + r2*w^3 ("034" sparse), NAF(6u+2),
 *           two Frobenius lines pi(Q), -pi^2(Q)
 *   FE      easy part (p^6-1)(p^2+1); hard part = Fuentes-Castaneda chain (exponent multiplied by
 *           the cofactor s = 2u(6u^2+3u+1)), Granger-Scott cyclotomic squarings in x^u
 *   G1/G2   Jacobian double-and-add, affine output (canonical => algorithm-independent bits)
 */
#include "bn254_oracle.h"
#include "bn254_constants.h"
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fp;
typedef struct { fp a0, a1; } fp2;
typedef struct { fp2 b0, b1, b2; } fp6;
typedef struct { fp6 c0, c1; } fp12;
typedef struct { fp x, y; } g1a;
typedef struct { fp2 x, y; } g2a;
typedef struct { fp x, y, z; } g1j;
typedef struct { fp2 x, y, z; } g2j;

static const fp FP_P = {BN254_P_LIMBS};
static const fp FP_ONE = {BN254_FP_ONE};
static const fp FP_TWO_INV = {BN254_FP_TWO_INV};
static const fp FP_PM2 = {BN254_P_MINUS_2};
static const fp2 B_TWIST = {{BN254_B_G2_A0}, {BN254_B_G2_A1}};
static const fp2 GAMMA1[5] = BN254_GAMMA1;
static const fp2 GAMMA2[5] = BN254_GAMMA2;
static const fp2 GAMMA3[5] = BN254_GAMMA3;
static const int8_t ATE_NAF[BN254_ATE_NAF_LEN] = BN254_ATE_NAF;

#ifdef GPBC_COUNT_MULS
static _Thread_local uint64_t g_mul_count;
#define COUNT_MUL() (g_mul_count++)
#else
#define COUNT_MUL() ((void)0)
#endif

/* ------------------------------------------------------------------------------------------ Fp */
static inline int fp_is_zero(const fp *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fp_eq(const fp *a, const fp *b) {
    return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline int fp_geq_p(const uint64_t t[4]) {
    for (int i = 3; i >= 0; i--) {
        if (t[i] > FP_P.l[i]) return 1;
        if (t[i] < FP_P.l[i]) return 0;
    }
    return 1;
}
static inline void fp_sub_p(uint64_t t[4]) {
    u128 b = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)t[i] - FP_P.l[i] - b;
        t[i] = (uint64_t)d;
        b = (d >> 64) & 1;
    }
}
static inline void fp_add(fp *z, const fp *x, const fp *y) {
    u128 c = 0;
    uint64_t t[4];
    for (int i = 0; i < 4; i++) {
        c += (u128)x->l[i] + y->l[i];
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    if (c || fp_geq_p(t)) fp_sub_p(t);
    memcpy(z->l, t, 32);
}
static inline void fp_sub(fp *z, const fp *x, const fp *y) {
    u128 b = 0;
    uint64_t t[4];
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)x->l[i] - y->l[i] - b;
        t[i] = (uint64_t)d;
        b = (d >> 64) & 1;
    }
    if (b) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)t[i] + FP_P.l[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    memcpy(z->l, t, 32);
}
static inline void fp_neg(fp *z, const fp *x) {
    if (fp_is_zero(x)) { *z = *x; return; }
    fp zero = {{0, 0, 0, 0}};
    fp_sub(z, &zero, x);
}
static inline void fp_dbl(fp *z, const fp *x) { fp_add(z, x, x); }
static void fp_mul(fp *z, const fp *x, const fp *y) {
    COUNT_MUL();
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)x->l[j] * y->l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * BN254_P_INV_NEG;
        c = (u128)m * FP_P.l[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * FP_P.l[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || fp_geq_p(t)) fp_sub_p(t);
    memcpy(z->l, t, 32);
}
static inline void fp_sqr(fp *z, const fp *x) { fp_mul(z, x, x); }
static void fp_inv(fp *z, const fp *x) { /* Fermat: x^(p-2); 0 -> 0 (gnark convention) */
    fp r = FP_ONE, b = *x;
    for (int i = 0; i < 254; i++) {
        if ((FP_PM2.l[i >> 6] >> (i & 63)) & 1) fp_mul(&r, &r, &b);
        fp_sqr(&b, &b);
    }
    *z = r;
}
static inline void fp_halve(fp *z, const fp *x) { fp_mul(z, x, &FP_TWO_INV); }

/* ------------------------------------------------------------------------------------------ Fp2 */
static inline int fp2_is_zero(const fp2 *a) { return fp_is_zero(&a->a0) && fp_is_zero(&a->a1); }
static inline int fp2_eq(const fp2 *a, const fp2 *b) { return fp_eq(&a->a0, &b->a0) && fp_eq(&a->a1, &b->a1); }
static inline void fp2_add(fp2 *z, const fp2 *x, const fp2 *y) { fp_add(&z->a0, &x->a0, &y->a0); fp_add(&z->a1, &x->a1, &y->a1); }
static inline void fp2_sub(fp2 *z, const fp2 *x, const fp2 *y) { fp_sub(&z->a0, &x->a0, &y->a0); fp_sub(&z->a1, &x->a1, &y->a1); }
static inline void fp2_dbl(fp2 *z, const fp2 *x) { fp2_add(z, x, x); }
static inline void fp2_neg(fp2 *z, const fp2 *x) { fp_neg(&z->a0, &x->a0); fp_neg(&z->a1, &x->a1); }
static inline void fp2_conj(fp2 *z, const fp2 *x) { z->a0 = x->a0; fp_neg(&z->a1, &x->a1); }
static inline void fp2_halve(fp2 *z, const fp2 *x) { fp_halve(&z->a0, &x->a0); fp_halve(&z->a1, &x->a1); }
static void fp2_mul(fp2 *z, const fp2 *x, const fp2 *y) {
    fp t0, t1, s0, s1, m;
    fp_mul(&t0, &x->a0, &y->a0);
    fp_mul(&t1, &x->a1, &y->a1);
    fp_add(&s0, &x->a0, &x->a1);
    fp_add(&s1, &y->a0, &y->a1);
    fp_mul(&m, &s0, &s1);
    fp_sub(&m, &m, &t0);
    fp_sub(&z->a1, &m, &t1);
    fp_sub(&z->a0, &t0, &t1);
}
static void fp2_sqr(fp2 *z, const fp2 *x) {
    fp s, d, m;
    fp_add(&s, &x->a0, &x->a1);
    fp_sub(&d, &x->a0, &x->a1);
    fp_mul(&m, &x->a0, &x->a1);
    fp_mul(&z->a0, &s, &d);
    fp_dbl(&z->a1, &m);
}
static inline void fp2_mul_fp(fp2 *z, const fp2 *x, const fp *k) { fp_mul(&z->a0, &x->a0, k); fp_mul(&z->a1, &x->a1, k); }
static void fp2_mul_xi(fp2 *z, const fp2 *x) { /* (a0 + a1 i)(9 + i) */
    fp t0, t1, n0, n1;
    fp_dbl(&t0, &x->a0); fp_dbl(&t0, &t0); fp_dbl(&t0, &t0); fp_add(&t0, &t0, &x->a0); /* 9 a0 */
    fp_dbl(&t1, &x->a1); fp_dbl(&t1, &t1); fp_dbl(&t1, &t1); fp_add(&t1, &t1, &x->a1); /* 9 a1 */
    fp_sub(&n0, &t0, &x->a1);
    fp_add(&n1, &t1, &x->a0);
    z->a0 = n0; z->a1 = n1;
}
static void fp2_inv(fp2 *z, const fp2 *x) {
    fp n, t;
    fp_sqr(&n, &x->a0);
    fp_sqr(&t, &x->a1);
    fp_add(&n, &n, &t);
    fp_inv(&n, &n);
    fp_mul(&z->a0, &x->a0, &n);
    fp_mul(&t, &x->a1, &n);
    fp_neg(&z->a1, &t);
}

/* ------------------------------------------------------------------------------------------ Fp6 */
static inline void fp6_add(fp6 *z, const fp6 *x, const fp6 *y) { fp2_add(&z->b0, &x->b0, &y->b0); fp2_add(&z->b1, &x->b1, &y->b1); fp2_add(&z->b2, &x->b2, &y->b2); }
static inline void fp6_sub(fp6 *z, const fp6 *x, const fp6 *y) { fp2_sub(&z->b0, &x->b0, &y->b0); fp2_sub(&z->b1, &x->b1, &y->b1); fp2_sub(&z->b2, &x->b2, &y->b2); }
static inline void fp6_neg(fp6 *z, const fp6 *x) { fp2_neg(&z->b0, &x->b0); fp2_neg(&z->b1, &x->b1); fp2_neg(&z->b2, &x->b2); }
static inline void fp6_dbl(fp6 *z, const fp6 *x) { fp6_add(z, x, x); }
static void fp6_mul_v(fp6 *z, const fp6 *x) {
    fp2 t;
    fp2_mul_xi(&t, &x->b2);
    z->b2 = x->b1; z->b1 = x->b0; z->b0 = t;
}
static void fp6_mul(fp6 *z, const fp6 *x, const fp6 *y) {
    fp2 t0, t1, t2, s, u, c0, c1, c2;
    fp2_mul(&t0, &x->b0, &y->b0);
    fp2_mul(&t1, &x->b1, &y->b1);
    fp2_mul(&t2, &x->b2, &y->b2);
    fp2_add(&s, &x->b1, &x->b2); fp2_add(&u, &y->b1, &y->b2);
    fp2_mul(&c0, &s, &u); fp2_sub(&c0, &c0, &t1); fp2_sub(&c0, &c0, &t2); fp2_mul_xi(&c0, &c0); fp2_add(&c0, &c0, &t0);
    fp2_add(&s, &x->b0, &x->b1); fp2_add(&u, &y->b0, &y->b1);
    fp2_mul(&c1, &s, &u); fp2_sub(&c1, &c1, &t0); fp2_sub(&c1, &c1, &t1); fp2_mul_xi(&s, &t2); fp2_add(&c1, &c1, &s);
    fp2_add(&s, &x->b0, &x->b2); fp2_add(&u, &y->b0, &y->b2);
    fp2_mul(&c2, &s, &u); fp2_sub(&c2, &c2, &t0); fp2_sub(&c2, &c2, &t2); fp2_add(&c2, &c2, &t1);
    z->b0 = c0; z->b1 = c1; z->b2 = c2;
}
static void fp6_mul_fp2(fp6 *z, const fp6 *x, const fp2 *k) { fp2_mul(&z->b0, &x->b0, k); fp2_mul(&z->b1, &x->b1, k); fp2_mul(&z->b2, &x->b2, k); }
static void fp6_mul_01(fp6 *z, const fp6 *x, const fp2 *c0, const fp2 *c1) { /* x * (c0 + c1 v) */
    fp2 t, r0, r1, r2;
    fp2_mul(&r0, &x->b0, c0); fp2_mul(&t, &x->b2, c1); fp2_mul_xi(&t, &t); fp2_add(&r0, &r0, &t);
    fp2_mul(&r1, &x->b0, c1); fp2_mul(&t, &x->b1, c0); fp2_add(&r1, &r1, &t);
    fp2_mul(&r2, &x->b1, c1); fp2_mul(&t, &x->b2, c0); fp2_add(&r2, &r2, &t);
    z->b0 = r0; z->b1 = r1; z->b2 = r2;
}
static void fp6_inv(fp6 *z, const fp6 *x) {
    fp2 t0, t1, t2, a, d;
    fp2_sqr(&t0, &x->b0); fp2_mul(&a, &x->b1, &x->b2); fp2_mul_xi(&a, &a); fp2_sub(&t0, &t0, &a);
    fp2_sqr(&t1, &x->b2); fp2_mul_xi(&t1, &t1); fp2_mul(&a, &x->b0, &x->b1); fp2_sub(&t1, &t1, &a);
    fp2_sqr(&t2, &x->b1); fp2_mul(&a, &x->b0, &x->b2); fp2_sub(&t2, &t2, &a);
    fp2_mul(&d, &x->b0, &t0);
    fp2_mul(&a, &x->b2, &t1); fp2_mul_xi(&a, &a); fp2_add(&d, &d, &a);
    fp2_mul(&a, &x->b1, &t2); fp2_mul_xi(&a, &a); fp2_add(&d, &d, &a);
    fp2_inv(&d, &d);
    fp2_mul(&z->b0, &t0, &d); fp2_mul(&z->b1, &t1, &d); fp2_mul(&z->b2, &t2, &d);
}

/* ------------------------------------------------------------------------------------------ Fp12 */
static void fp12_set_one(fp12 *z) { memset(z, 0, sizeof *z); z->c0.b0.a0 = FP_ONE; }
static int fp12_is_one(const fp12 *z) { fp12 o; fp12_set_one(&o); return memcmp(z, &o, sizeof o) == 0; }
static void fp12_mul(fp12 *z, const fp12 *x, const fp12 *y) {
    fp6 t0, t1, s, u, c1;
    fp6_mul(&t0, &x->c0, &y->c0);
    fp6_mul(&t1, &x->c1, &y->c1);
    fp6_add(&s, &x->c0, &x->c1); fp6_add(&u, &y->c0, &y->c1);
    fp6_mul(&c1, &s, &u); fp6_sub(&c1, &c1, &t0); fp6_sub(&c1, &c1, &t1);
    fp6_mul_v(&t1, &t1);
    fp6_add(&z->c0, &t0, &t1);
    z->c1 = c1;
}
static void fp12_sqr(fp12 *z, const fp12 *x) {
    fp6 s, t, m, mv;
    fp6_add(&s, &x->c0, &x->c1);
    fp6_mul_v(&t, &x->c1); fp6_add(&t, &t, &x->c0);
    fp6_mul(&m, &x->c0, &x->c1);
    fp6_mul(&s, &s, &t);
    fp6_mul_v(&mv, &m);
    fp6_sub(&s, &s, &m); fp6_sub(&z->c0, &s, &mv);
    fp6_dbl(&z->c1, &m);
}
static void fp12_conj(fp12 *z, const fp12 *x) { z->c0 = x->c0; fp6_neg(&z->c1, &x->c1); }
static void fp12_inv(fp12 *z, const fp12 *x) {
    fp6 t0, t1;
    fp6_mul(&t0, &x->c0, &x->c0);
    fp6_mul(&t1, &x->c1, &x->c1); fp6_mul_v(&t1, &t1);
    fp6_sub(&t0, &t0, &t1);
    fp6_inv(&t0, &t0);
    fp6_mul(&z->c0, &x->c0, &t0);
    fp6_mul(&t1, &x->c1, &t0); fp6_neg(&z->c1, &t1);
}
/* x^(p^j) on the w-basis: coefficient of w^k -> (conj if j odd)(c_k) * gamma_j[k] */
static void fp12_frob(fp12 *z, const fp12 *x, int j) {
    const fp2 *g = j == 1 ? GAMMA1 : (j == 2 ? GAMMA2 : GAMMA3);
    const fp2 *src[6] = {&x->c0.b0, &x->c1.b0, &x->c0.b1, &x->c1.b1, &x->c0.b2, &x->c1.b2};
    fp2 *dst[6] = {&z->c0.b0, &z->c1.b0, &z->c0.b1, &z->c1.b1, &z->c0.b2, &z->c1.b2};
    for (int k = 0; k < 6; k++) {
        fp2 c = *src[k];
        if (j & 1) fp2_conj(&c, &c);
        if (k) fp2_mul(&c, &c, &g[k - 1]);
        *dst[k] = c;
    }
}
/* Granger-Scott squaring for elements of the cyclotomic subgroup */
static void fp12_cyclo_sqr(fp12 *z, const fp12 *x) {
    fp2 t0, t1, t2, t3, t4, t5, t6, t7, t8, s;
    fp2_sqr(&t0, &x->c1.b1); fp2_sqr(&t1, &x->c0.b0);
    fp2_add(&t6, &x->c1.b1, &x->c0.b0); fp2_sqr(&t6, &t6); fp2_sub(&t6, &t6, &t0); fp2_sub(&t6, &t6, &t1);
    fp2_sqr(&t2, &x->c0.b2); fp2_sqr(&t3, &x->c1.b0);
    fp2_add(&t7, &x->c0.b2, &x->c1.b0); fp2_sqr(&t7, &t7); fp2_sub(&t7, &t7, &t2); fp2_sub(&t7, &t7, &t3);
    fp2_sqr(&t4, &x->c1.b2); fp2_sqr(&t5, &x->c0.b1);
    fp2_add(&t8, &x->c1.b2, &x->c0.b1); fp2_sqr(&t8, &t8); fp2_sub(&t8, &t8, &t4); fp2_sub(&t8, &t8, &t5); fp2_mul_xi(&t8, &t8);
    fp2_mul_xi(&t0, &t0); fp2_add(&t0, &t0, &t1);
    fp2_mul_xi(&t2, &t2); fp2_add(&t2, &t2, &t3);
    fp2_mul_xi(&t4, &t4); fp2_add(&t4, &t4, &t5);
    fp12 r;
    fp2_sub(&s, &t0, &x->c0.b0); fp2_dbl(&s, &s); fp2_add(&r.c0.b0, &s, &t0);
    fp2_sub(&s, &t2, &x->c0.b1); fp2_dbl(&s, &s); fp2_add(&r.c0.b1, &s, &t2);
    fp2_sub(&s, &t4, &x->c0.b2); fp2_dbl(&s, &s); fp2_add(&r.c0.b2, &s, &t4);
    fp2_add(&s, &t8, &x->c1.b0); fp2_dbl(&s, &s); fp2_add(&r.c1.b0, &s, &t8);
    fp2_add(&s, &t6, &x->c1.b1); fp2_dbl(&s, &s); fp2_add(&r.c1.b1, &s, &t6);
    fp2_add(&s, &t7, &x->c1.b2); fp2_dbl(&s, &s); fp2_add(&r.c1.b2, &s, &t7);
    *z = r;
}
/* z = x * (c0 + c3 w + c4 v w) */
static void fp12_mul_034(fp12 *z, const fp12 *x, const fp2 *c0, const fp2 *c3, const fp2 *c4) {
    fp6 a, b, t;
    fp6_mul_fp2(&a, &x->c0, c0);          /* x0 * l0 */
    fp6_mul_01(&b, &x->c1, c3, c4);       /* x1 * l1 */
    fp6_mul_v(&b, &b);
    fp6_add(&a, &a, &b);
    fp6_mul_01(&b, &x->c0, c3, c4);       /* x0 * l1 */
    fp6_mul_fp2(&t, &x->c1, c0);          /* x1 * l0 */
    fp6_add(&z->c1, &b, &t);
    z->c0 = a;
}
static void fp12_expt(fp12 *z, const fp12 *x) { /* x^u, x in the cyclotomic subgroup */
    fp12 r = *x;
    for (int i = BN254_U_BITS - 2; i >= 0; i--) {
        fp12_cyclo_sqr(&r, &r);
        if ((BN254_U >> i) & 1) fp12_mul(&r, &r, x);
    }
    *z = r;
}

/* ------------------------------------------------------------------------------------------ pairing */
typedef struct { fp2 r0, r1, r2; } line_t;

static void g2_double_step(g2j *t, line_t *l) {
    fp2 A, B, C, D, E, F, G, H, I, J, EE, K, t1;
    fp2_mul(&A, &t->x, &t->y); fp2_halve(&A, &A);
    fp2_sqr(&B, &t->y);
    fp2_sqr(&C, &t->z);
    fp2_dbl(&D, &C); fp2_add(&D, &D, &C);
    fp2_mul(&E, &D, &B_TWIST);
    fp2_dbl(&F, &E); fp2_add(&F, &F, &E);
    fp2_add(&G, &B, &F); fp2_halve(&G, &G);
    fp2_add(&H, &t->y, &t->z); fp2_sqr(&H, &H); fp2_add(&t1, &B, &C); fp2_sub(&H, &H, &t1);
    fp2_sub(&I, &E, &B);
    fp2_sqr(&J, &t->x);
    fp2_sqr(&EE, &E);
    fp2_dbl(&K, &EE); fp2_add(&K, &K, &EE);
    fp2_sub(&t1, &B, &F); fp2_mul(&t->x, &A, &t1);
    fp2_sqr(&t1, &G); fp2_sub(&t->y, &t1, &K);
    fp2_mul(&t->z, &B, &H);
    fp2_neg(&l->r0, &H);
    fp2_dbl(&l->r1, &J); fp2_add(&l->r1, &l->r1, &J);
    l->r2 = I;
}
static void g2_add_mixed_step(g2j *t, line_t *l, const g2a *q) {
    fp2 Y2Z1, X2Z1, O, L, C, D, E, F, G, H, t0, t1, t2, J;
    fp2_mul(&Y2Z1, &q->y, &t->z); fp2_sub(&O, &t->y, &Y2Z1);
    fp2_mul(&X2Z1, &q->x, &t->z); fp2_sub(&L, &t->x, &X2Z1);
    fp2_sqr(&C, &O); fp2_sqr(&D, &L);
    fp2_mul(&E, &L, &D);
    fp2_mul(&F, &t->z, &C);
    fp2_mul(&G, &t->x, &D);
    fp2_dbl(&t0, &G);
    fp2_add(&H, &E, &F); fp2_sub(&H, &H, &t0);
    fp2_mul(&t1, &t->y, &E);
    fp2_mul(&t->x, &L, &H);
    fp2_sub(&t0, &G, &H); fp2_mul(&t0, &t0, &O); fp2_sub(&t->y, &t0, &t1);
    fp2_mul(&t->z, &E, &t->z);
    fp2_mul(&t2, &L, &q->y);
    fp2_mul(&J, &q->x, &O); fp2_sub(&J, &J, &t2);
    l->r0 = L;
    fp2_neg(&l->r1, &O);
    l->r2 = J;
}
static void line_apply(fp12 *f, const line_t *l, const g1a *p) {
    fp2 c0, c3;
    fp2_mul_fp(&c0, &l->r0, &p->y);
    fp2_mul_fp(&c3, &l->r1, &p->x);
    fp12_mul_034(f, f, &c0, &c3, &l->r2);
}
static int g1a_is_inf(const g1a *p) { return fp_is_zero(&p->x) && fp_is_zero(&p->y); }
static int g2a_is_inf(const g2a *q) { return fp2_is_zero(&q->x) && fp2_is_zero(&q->y); }

/* f *= Miller function of (p,q); pairs with infinity are skipped (gnark MillerLoop filter) */
static void miller_accumulate(fp12 *f_acc, const g1a *p, const g2a *q) {
    if (g1a_is_inf(p) || g2a_is_inf(q)) return;
    fp12 f;
    fp12_set_one(&f);
    g2j t = {q->x, q->y, {FP_ONE, {{0, 0, 0, 0}}}};
    g2a qn = *q;
    fp2_neg(&qn.y, &q->y);
    line_t l;
    for (int i = BN254_ATE_NAF_LEN - 2; i >= 0; i--) {
        fp12_sqr(&f, &f);
        g2_double_step(&t, &l);
        line_apply(&f, &l, p);
        if (ATE_NAF[i] == 1) { g2_add_mixed_step(&t, &l, q); line_apply(&f, &l, p); }
        else if (ATE_NAF[i] == -1) { g2_add_mixed_step(&t, &l, &qn); line_apply(&f, &l, p); }
    }
    g2a q1, q2;
    fp2_conj(&q1.x, &q->x); fp2_mul(&q1.x, &q1.x, &GAMMA1[1]);
    fp2_conj(&q1.y, &q->y); fp2_mul(&q1.y, &q1.y, &GAMMA1[2]);
    fp2_mul(&q2.x, &q->x, &GAMMA2[1]);
    fp2_mul(&q2.y, &q->y, &GAMMA2[2]); fp2_neg(&q2.y, &q2.y);
    g2_add_mixed_step(&t, &l, &q1); line_apply(&f, &l, p);
    g2_add_mixed_step(&t, &l, &q2); line_apply(&f, &l, p);
    fp12_mul(f_acc, f_acc, &f);
}

static void final_exp(fp12 *z, const fp12 *x) {
    fp12 r, t0, t1, t2, t3, t4;
    /* easy part */
    fp12_conj(&t0, x);
    fp12_inv(&r, x);
    fp12_mul(&t0, &t0, &r);
    fp12_frob(&r, &t0, 2);
    fp12_mul(&r, &r, &t0);
    if (fp12_is_one(&r)) { *z = r; return; }
    /* hard part, exponent s*(p^4-p^2+1)/r */
    fp12_expt(&t0, &r); fp12_conj(&t0, &t0);
    fp12_cyclo_sqr(&t0, &t0);
    fp12_cyclo_sqr(&t1, &t0);
    fp12_mul(&t1, &t0, &t1);
    fp12_expt(&t2, &t1); fp12_conj(&t2, &t2);
    fp12_conj(&t3, &t1);
    fp12_mul(&t1, &t2, &t3);
    fp12_cyclo_sqr(&t3, &t2);
    fp12_expt(&t4, &t3);
    fp12_mul(&t4, &t1, &t4);
    fp12_mul(&t3, &t0, &t4);
    fp12_mul(&t0, &t2, &t4);
    fp12_mul(&t0, &r, &t0);
    fp12_frob(&t2, &t3, 1);
    fp12_mul(&t0, &t2, &t0);
    fp12_frob(&t2, &t4, 2);
    fp12_mul(&t0, &t2, &t0);
    fp12_conj(&t2, &r);
    fp12_mul(&t2, &t2, &t3);
    fp12_frob(&t2, &t2, 3);
    fp12_mul(&t0, &t2, &t0);
    *z = t0;
}

/* ------------------------------------------------------------------------------------------ G1 / G2 Jacobian */
#define DEFINE_CURVE(PFX, F, JT, AT, F_add, F_sub, F_dbl, F_mul, F_sqr, F_inv, F_is_zero, F_ONE_INIT)           \
    static void PFX##_dbl(JT *r, const JT *p) {                                                                 \
        if (F_is_zero(&p->z)) { *r = *p; return; }                                                              \
        F A, B, C, D, E, FF, t, x3, y3, z3;                                                                     \
        F_sqr(&A, &p->x); F_sqr(&B, &p->y); F_sqr(&C, &B);                                                      \
        F_add(&t, &p->x, &B); F_sqr(&t, &t); F_sub(&t, &t, &A); F_sub(&t, &t, &C); F_dbl(&D, &t);               \
        F_dbl(&E, &A); F_add(&E, &E, &A);                                                                       \
        F_sqr(&FF, &E);                                                                                         \
        F_dbl(&t, &D); F_sub(&x3, &FF, &t);                                                                     \
        F_sub(&t, &D, &x3); F_mul(&y3, &E, &t);                                                                 \
        F_dbl(&t, &C); F_dbl(&t, &t); F_dbl(&t, &t); F_sub(&y3, &y3, &t);                                       \
        F_mul(&z3, &p->y, &p->z); F_dbl(&z3, &z3);                                                              \
        r->x = x3; r->y = y3; r->z = z3;                                                                        \
    }                                                                                                           \
    static void PFX##_add_mixed(JT *r, const JT *p, const AT *q) {                                              \
        if (F_is_zero(&q->x) && F_is_zero(&q->y)) { *r = *p; return; }                                          \
        if (F_is_zero(&p->z)) { r->x = q->x; r->y = q->y; F one = F_ONE_INIT; r->z = one; return; }             \
        F Z1Z1, U2, S2, H, HH, I, J, rr, V, t, x3, y3, z3;                                                      \
        F_sqr(&Z1Z1, &p->z); F_mul(&U2, &q->x, &Z1Z1);                                                          \
        F_mul(&S2, &q->y, &p->z); F_mul(&S2, &S2, &Z1Z1);                                                       \
        F_sub(&H, &U2, &p->x); F_sub(&rr, &S2, &p->y);                                                          \
        if (F_is_zero(&H)) {                                                                                    \
            if (F_is_zero(&rr)) { PFX##_dbl(r, p); return; }                                                    \
            memset(r, 0, sizeof *r); return;                                                                    \
        }                                                                                                       \
        F_dbl(&rr, &rr);                                                                                        \
        F_sqr(&HH, &H); F_dbl(&I, &HH); F_dbl(&I, &I); F_mul(&J, &H, &I);                                       \
        F_mul(&V, &p->x, &I);                                                                                   \
        F_sqr(&x3, &rr); F_sub(&x3, &x3, &J); F_dbl(&t, &V); F_sub(&x3, &x3, &t);                               \
        F_sub(&t, &V, &x3); F_mul(&y3, &rr, &t); F_mul(&t, &p->y, &J); F_dbl(&t, &t); F_sub(&y3, &y3, &t);      \
        F_add(&z3, &p->z, &H); F_sqr(&z3, &z3); F_sub(&z3, &z3, &Z1Z1); F_sub(&z3, &z3, &HH);                   \
        r->x = x3; r->y = y3; r->z = z3;                                                                        \
    }                                                                                                           \
    static void PFX##_to_affine(AT *r, const JT *p) {                                                           \
        if (F_is_zero(&p->z)) { memset(r, 0, sizeof *r); return; }                                              \
        F zi, zi2;                                                                                              \
        F_inv(&zi, &p->z); F_sqr(&zi2, &zi);                                                                    \
        F_mul(&r->x, &p->x, &zi2); F_mul(&zi2, &zi2, &zi); F_mul(&r->y, &p->y, &zi2);                           \
    }                                                                                                           \
    static void PFX##_scalar_mul(AT *r, const AT *base, const uint8_t k[32]) {                                  \
        JT acc; memset(&acc, 0, sizeof acc);                                                                    \
        for (int i = 255; i >= 0; i--) {                                                                        \
            PFX##_dbl(&acc, &acc);                                                                              \
            if ((k[i >> 3] >> (i & 7)) & 1) PFX##_add_mixed(&acc, &acc, base);                                  \
        }                                                                                                       \
        PFX##_to_affine(r, &acc);                                                                               \
    }                                                                                                           \
    static void PFX##_sum(AT *r, const AT *pts, size_t n) {                                                     \
        JT acc; memset(&acc, 0, sizeof acc);                                                                    \
        for (size_t i = 0; i < n; i++) PFX##_add_mixed(&acc, &acc, &pts[i]);                                    \
        PFX##_to_affine(r, &acc);                                                                               \
    }

#define FP_ONE_INIT {BN254_FP_ONE}
#define FP2_ONE_INIT {{BN254_FP_ONE}, {{0, 0, 0, 0}}}
DEFINE_CURVE(g1, fp, g1j, g1a, fp_add, fp_sub, fp_dbl, fp_mul, fp_sqr, fp_inv, fp_is_zero, FP_ONE_INIT)
DEFINE_CURVE(g2, fp2, g2j, g2a, fp2_add, fp2_sub, fp2_dbl, fp2_mul, fp2_sqr, fp2_inv, fp2_is_zero, FP2_ONE_INIT)

static void gt_exp(fp12 *z, const fp12 *x, const uint8_t k[32]) {
    fp12 r;
    fp12_set_one(&r);
    int started = 0;
    for (int i = 255; i >= 0; i--) {
        if (started) fp12_sqr(&r, &r);
        if ((k[i >> 3] >> (i & 7)) & 1) {
            if (started) fp12_mul(&r, &r, x); else { r = *x; started = 1; }
        }
    }
    *z = r;
}

/* ------------------------------------------------------------------------------------------ C entry points */
static int clamp_threads(int threads) {
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    return threads;
#else
    (void)threads;
    return 1;
#endif
}

void gpbc_oracle_miller_loop(const uint8_t *P, const uint8_t *Q, size_t n, uint8_t *f_out, int threads) {
    threads = clamp_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
    for (long i = 0; i < (long)n; i++) {
        g1a p; g2a q; fp12 f;
        memcpy(&p, P + 64 * i, 64); memcpy(&q, Q + 128 * i, 128);
        fp12_set_one(&f);
        miller_accumulate(&f, &p, &q);
        memcpy(f_out + 384 * i, &f, 384);
    }
}
void gpbc_oracle_final_exp(const uint8_t *f, size_t n, uint8_t *gt_out, int threads) {
    threads = clamp_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
    for (long i = 0; i < (long)n; i++) {
        fp12 x, z;
        memcpy(&x, f + 384 * i, 384);
        final_exp(&z, &x);
        memcpy(gt_out + 384 * i, &z, 384);
    }
}
void gpbc_oracle_multi_pair(const uint8_t *P, const uint8_t *Q, const uint64_t *seg_off, size_t k,
                            uint8_t *gt_out, int threads) {
    threads = clamp_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (long s = 0; s < (long)k; s++) {
        fp12 f, z;
        fp12_set_one(&f);
        for (uint64_t i = seg_off[s]; i < seg_off[s + 1]; i++) {
            g1a p; g2a q;
            memcpy(&p, P + 64 * i, 64); memcpy(&q, Q + 128 * i, 128);
            miller_accumulate(&f, &p, &q);
        }
        final_exp(&z, &f);
        memcpy(gt_out + 384 * s, &z, 384);
    }
}
void gpbc_oracle_pair_batch(const uint8_t *P, const uint8_t *Q, size_t n, uint8_t *gt_out, int threads) {
    threads = clamp_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
    for (long i = 0; i < (long)n; i++) {
        g1a p; g2a q; fp12 f, z;
        memcpy(&p, P + 64 * i, 64); memcpy(&q, Q + 128 * i, 128);
        fp12_set_one(&f);
        miller_accumulate(&f, &p, &q);
        final_exp(&z, &f);
        memcpy(gt_out + 384 * i, &z, 384);
    }
}
void gpbc_oracle_g1_scalar_mul(const uint8_t *base, size_t nbase, const uint8_t *scalars, size_t n,
                               uint8_t *out, int threads) {
    threads = clamp_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 8)
    for (long i = 0; i < (long)n; i++) {
        g1a b, r;
        memcpy(&b, base + 64 * (nbase == 1 ? 0 : i), 64);
        g1_scalar_mul(&r, &b, scalars + 32 * i);
        memcpy(out + 64 * i, &r, 64);
    }
}
void gpbc_oracle_g2_scalar_mul(const uint8_t *base, size_t nbase, const uint8_t *scalars, size_t n,
                               uint8_t *out, int threads) {
    threads = clamp_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 8)
    for (long i = 0; i < (long)n; i++) {
        g2a b, r;
        memcpy(&b, base + 128 * (nbase == 1 ? 0 : i), 128);
        g2_scalar_mul(&r, &b, scalars + 32 * i);
        memcpy(out + 128 * i, &r, 128);
    }
}
void gpbc_oracle_g1_sum(const uint8_t *pts, size_t n, uint8_t *out) {
    g1a r; g1_sum(&r, (const g1a *)pts, n); memcpy(out, &r, 64);
}
void gpbc_oracle_g2_sum(const uint8_t *pts, size_t n, uint8_t *out) {
    g2a r; g2_sum(&r, (const g2a *)pts, n); memcpy(out, &r, 128);
}
void gpbc_oracle_gt_exp(const uint8_t *x, const uint8_t *k, size_t n, uint8_t *out, int threads) {
    threads = clamp_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
    for (long i = 0; i < (long)n; i++) {
        fp12 a, z;
        memcpy(&a, x + 384 * i, 384);
        gt_exp(&z, &a, k + 32 * i);
        memcpy(out + 384 * i, &z, 384);
    }
}
void gpbc_oracle_gt_mul(const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        fp12 x, y, z;
        memcpy(&x, a + 384 * i, 384); memcpy(&y, b + 384 * i, 384);
        fp12_mul(&z, &x, &y);
        memcpy(out + 384 * i, &z, 384);
    }
}
void gpbc_oracle_gt_div(const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        fp12 x, y, z;
        memcpy(&x, a + 384 * i, 384); memcpy(&y, b + 384 * i, 384);
        fp12_inv(&y, &y);
        fp12_mul(&z, &x, &y);
        memcpy(out + 384 * i, &z, 384);
    }
}
void gpbc_oracle_gt_inverse(const uint8_t *a, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        fp12 x, z;
        memcpy(&x, a + 384 * i, 384);
        fp12_inv(&z, &x);
        memcpy(out + 384 * i, &z, 384);
    }
}
void gpbc_oracle_fp_mul(const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        fp x, y, z;
        memcpy(&x, a + 32 * i, 32); memcpy(&y, b + 32 * i, 32);
        fp_mul(&z, &x, &y);
        memcpy(out + 32 * i, &z, 32);
    }
}
void gpbc_oracle_fp_inv(const uint8_t *a, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        fp x, z;
        memcpy(&x, a + 32 * i, 32);
        fp_inv(&z, &x);
        memcpy(out + 32 * i, &z, 32);
    }
}
void gpbc_oracle_fp12_mul(const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out) { gpbc_oracle_gt_mul(a, b, n, out); }
void gpbc_oracle_fp12_cyclotomic_square(const uint8_t *a, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        fp12 x, z;
        memcpy(&x, a + 384 * i, 384);
        fp12_cyclo_sqr(&z, &x);
        memcpy(out + 384 * i, &z, 384);
    }
}
uint64_t gpbc_oracle_fp_mul_count(int reset) {
#ifdef GPBC_COUNT_MULS
    uint64_t v = g_mul_count;
    if (reset) g_mul_count = 0;
    return v;
#else
    (void)reset;
    return 0;
#endif
}
