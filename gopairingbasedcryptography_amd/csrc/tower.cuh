// Extension tower over Fp for BN254 (gnark E2/E6/E12 layout):
//   Fp2 = Fp[i]/(i^2+1)  {a0,a1};  Fp6 = Fp2[v]/(v^3-(9+i))  {b0,b1,b2};  Fp12 = Fp6[w]/(w^2-v)  {c0,c1}
// In-memory order C0.B0.A0 ... C1.B2.A1 = 12 x 32 B, identical to gnark's GT struct (SURVEY.md §8 header).
#ifndef GPBC_TOWER_CUH
#define GPBC_TOWER_CUH
#include "fp.cuh"

namespace gpbc {

struct Fp2 { Fp a0, a1; };
struct Fp6 { Fp2 b0, b1, b2; };
struct Fp12 { Fp6 c0, c1; };

// ------------------------------------------------------------------------------------------- Fp2
__device__ __forceinline__ Fp2 fp2_zero() { return Fp2{fp_zero(), fp_zero()}; }
__device__ __forceinline__ Fp2 fp2_one() { return Fp2{fp_one(), fp_zero()}; }
__device__ __forceinline__ bool fp2_is_zero(const Fp2 &a) { return fp_is_zero(a.a0) && fp_is_zero(a.a1); }
__device__ __forceinline__ bool fp2_eq(const Fp2 &a, const Fp2 &b) { return fp_eq(a.a0, b.a0) && fp_eq(a.a1, b.a1); }
__device__ __forceinline__ Fp2 fp2_add(const Fp2 &x, const Fp2 &y) { return Fp2{fp_add(x.a0, y.a0), fp_add(x.a1, y.a1)}; }
__device__ __forceinline__ Fp2 fp2_sub(const Fp2 &x, const Fp2 &y) { return Fp2{fp_sub(x.a0, y.a0), fp_sub(x.a1, y.a1)}; }
__device__ __forceinline__ Fp2 fp2_dbl(const Fp2 &x) { return Fp2{fp_dbl(x.a0), fp_dbl(x.a1)}; }
__device__ __forceinline__ Fp2 fp2_neg(const Fp2 &x) { return Fp2{fp_neg(x.a0), fp_neg(x.a1)}; }
__device__ __forceinline__ Fp2 fp2_conj(const Fp2 &x) { return Fp2{x.a0, fp_neg(x.a1)}; }
__device__ __forceinline__ Fp2 fp2_halve(const Fp2 &x) { return Fp2{fp_halve(x.a0), fp_halve(x.a1)}; }

__device__ __noinline__ Fp2 fp2_mul(const Fp2 &x, const Fp2 &y) {
    Fp t0 = fp_mul(x.a0, y.a0);
    Fp t1 = fp_mul(x.a1, y.a1);
    Fp m = fp_mul(fp_add(x.a0, x.a1), fp_add(y.a0, y.a1));
    return Fp2{fp_sub(t0, t1), fp_sub(fp_sub(m, t0), t1)};
}
__device__ __noinline__ Fp2 fp2_sqr(const Fp2 &x) {
    Fp m = fp_mul(x.a0, x.a1);
    Fp r0 = fp_mul(fp_add(x.a0, x.a1), fp_sub(x.a0, x.a1));
    return Fp2{r0, fp_dbl(m)};
}
__device__ __forceinline__ Fp2 fp2_mul_fp(const Fp2 &x, const Fp &k) { return Fp2{fp_mul(x.a0, k), fp_mul(x.a1, k)}; }
// (a0 + a1 i)(9 + i) = (9 a0 - a1) + (9 a1 + a0) i
__device__ __noinline__ Fp2 fp2_mul_xi(const Fp2 &x) {
    Fp t0 = fp_dbl(fp_dbl(fp_dbl(x.a0)));
    Fp t1 = fp_dbl(fp_dbl(fp_dbl(x.a1)));
    t0 = fp_add(t0, x.a0);
    t1 = fp_add(t1, x.a1);
    return Fp2{fp_sub(t0, x.a1), fp_add(t1, x.a0)};
}
__device__ __noinline__ Fp2 fp2_inv(const Fp2 &x) {
    Fp n = fp_inv(fp_add(fp_sqr(x.a0), fp_sqr(x.a1)));
    return Fp2{fp_mul(x.a0, n), fp_neg(fp_mul(x.a1, n))};
}
__device__ __forceinline__ Fp2 fp2_load(const uint8_t *p) { return Fp2{fp_load(p), fp_load(p + 32)}; }
__device__ __forceinline__ void fp2_store(uint8_t *p, const Fp2 &x) { fp_store(p, x.a0); fp_store(p + 32, x.a1); }

// constant tables (Montgomery form) ----------------------------------------------------------
__device__ __forceinline__ Fp2 fp2_from_limbs(const u64 (&t)[2][4]) {
    return Fp2{fp_from_u64(t[0][0], t[0][1], t[0][2], t[0][3]), fp_from_u64(t[1][0], t[1][1], t[1][2], t[1][3])};
}
__device__ const u64 GAMMA_TBL[3][5][2][4] = {BN254_GAMMA1, BN254_GAMMA2, BN254_GAMMA3};
__device__ __forceinline__ Fp2 gamma(int j, int k) {  // xi^(k (p^j - 1)/6), j=1..3, k=1..5
    return fp2_from_limbs(GAMMA_TBL[j - 1][k - 1]);
}
__device__ __forceinline__ Fp2 b_twist() {
    constexpr u64 A0[4] = BN254_B_G2_A0;
    constexpr u64 A1[4] = BN254_B_G2_A1;
    return Fp2{fp_from_u64(A0[0], A0[1], A0[2], A0[3]), fp_from_u64(A1[0], A1[1], A1[2], A1[3])};
}

// ------------------------------------------------------------------------------------------- Fp6
__device__ __forceinline__ void fp6_add(Fp6 &z, const Fp6 &x, const Fp6 &y) { z.b0 = fp2_add(x.b0, y.b0); z.b1 = fp2_add(x.b1, y.b1); z.b2 = fp2_add(x.b2, y.b2); }
__device__ __forceinline__ void fp6_sub(Fp6 &z, const Fp6 &x, const Fp6 &y) { z.b0 = fp2_sub(x.b0, y.b0); z.b1 = fp2_sub(x.b1, y.b1); z.b2 = fp2_sub(x.b2, y.b2); }
__device__ __forceinline__ void fp6_neg(Fp6 &z, const Fp6 &x) { z.b0 = fp2_neg(x.b0); z.b1 = fp2_neg(x.b1); z.b2 = fp2_neg(x.b2); }
__device__ __forceinline__ void fp6_mul_v(Fp6 &z, const Fp6 &x) {
    Fp2 t = fp2_mul_xi(x.b2);
    z.b2 = x.b1; z.b1 = x.b0; z.b0 = t;
}
__device__ __noinline__ void fp6_mul(Fp6 &z, const Fp6 &x, const Fp6 &y) {
    Fp2 t0 = fp2_mul(x.b0, y.b0);
    Fp2 t1 = fp2_mul(x.b1, y.b1);
    Fp2 t2 = fp2_mul(x.b2, y.b2);
    Fp2 c0 = fp2_mul(fp2_add(x.b1, x.b2), fp2_add(y.b1, y.b2));
    c0 = fp2_add(fp2_mul_xi(fp2_sub(fp2_sub(c0, t1), t2)), t0);
    Fp2 c1 = fp2_mul(fp2_add(x.b0, x.b1), fp2_add(y.b0, y.b1));
    c1 = fp2_add(fp2_sub(fp2_sub(c1, t0), t1), fp2_mul_xi(t2));
    Fp2 c2 = fp2_mul(fp2_add(x.b0, x.b2), fp2_add(y.b0, y.b2));
    c2 = fp2_add(fp2_sub(fp2_sub(c2, t0), t2), t1);
    z.b0 = c0; z.b1 = c1; z.b2 = c2;
}
__device__ __noinline__ void fp6_sqr(Fp6 &z, const Fp6 &x) {  // CH-SQR2
    Fp2 s0 = fp2_sqr(x.b0);
    Fp2 ab = fp2_mul(x.b0, x.b1);
    Fp2 s1 = fp2_dbl(ab);
    Fp2 s2 = fp2_sqr(fp2_add(fp2_sub(x.b0, x.b1), x.b2));
    Fp2 bc = fp2_mul(x.b1, x.b2);
    Fp2 s3 = fp2_dbl(bc);
    Fp2 s4 = fp2_sqr(x.b2);
    z.b0 = fp2_add(s0, fp2_mul_xi(s3));
    z.b1 = fp2_add(s1, fp2_mul_xi(s4));
    z.b2 = fp2_sub(fp2_add(fp2_add(s1, s2), s3), fp2_add(s0, s4));
}
__device__ __forceinline__ void fp6_mul_fp2(Fp6 &z, const Fp6 &x, const Fp2 &k) { z.b0 = fp2_mul(x.b0, k); z.b1 = fp2_mul(x.b1, k); z.b2 = fp2_mul(x.b2, k); }
// x * (c0 + c1 v)
__device__ __noinline__ void fp6_mul_01(Fp6 &z, const Fp6 &x, const Fp2 &c0, const Fp2 &c1) {
    Fp2 a = fp2_mul(x.b0, c0);
    Fp2 b = fp2_mul(x.b1, c1);
    Fp2 t0 = fp2_add(fp2_mul_xi(fp2_sub(fp2_mul(fp2_add(x.b1, x.b2), c1), b)), a);      // xi*(x2 c1) + x0 c0
    Fp2 t1 = fp2_sub(fp2_sub(fp2_mul(fp2_add(x.b0, x.b1), fp2_add(c0, c1)), a), b);     // x0 c1 + x1 c0
    Fp2 t2 = fp2_add(fp2_sub(fp2_mul(fp2_add(x.b0, x.b2), c0), a), b);                  // x2 c0 + x1 c1
    z.b0 = t0; z.b1 = t1; z.b2 = t2;
}
__device__ __noinline__ void fp6_inv(Fp6 &z, const Fp6 &x) {
    Fp2 t0 = fp2_sub(fp2_sqr(x.b0), fp2_mul_xi(fp2_mul(x.b1, x.b2)));
    Fp2 t1 = fp2_sub(fp2_mul_xi(fp2_sqr(x.b2)), fp2_mul(x.b0, x.b1));
    Fp2 t2 = fp2_sub(fp2_sqr(x.b1), fp2_mul(x.b0, x.b2));
    Fp2 d = fp2_add(fp2_mul(x.b0, t0), fp2_mul_xi(fp2_add(fp2_mul(x.b2, t1), fp2_mul(x.b1, t2))));
    d = fp2_inv(d);
    z.b0 = fp2_mul(t0, d); z.b1 = fp2_mul(t1, d); z.b2 = fp2_mul(t2, d);
}

// ------------------------------------------------------------------------------------------- Fp12
__device__ __forceinline__ void fp12_set_one(Fp12 &z) {
    z.c0.b0 = fp2_one(); z.c0.b1 = fp2_zero(); z.c0.b2 = fp2_zero();
    z.c1.b0 = fp2_zero(); z.c1.b1 = fp2_zero(); z.c1.b2 = fp2_zero();
}
__device__ __forceinline__ bool fp12_is_one(const Fp12 &z) {
    return fp2_eq(z.c0.b0, fp2_one()) && fp2_is_zero(z.c0.b1) && fp2_is_zero(z.c0.b2) &&
           fp2_is_zero(z.c1.b0) && fp2_is_zero(z.c1.b1) && fp2_is_zero(z.c1.b2);
}
__device__ __noinline__ void fp12_mul(Fp12 &z, const Fp12 &x, const Fp12 &y) {
    Fp6 t0, t1, s, u, c1;
    fp6_mul(t0, x.c0, y.c0);
    fp6_mul(t1, x.c1, y.c1);
    fp6_add(s, x.c0, x.c1);
    fp6_add(u, y.c0, y.c1);
    fp6_mul(c1, s, u);
    fp6_sub(c1, c1, t0);
    fp6_sub(c1, c1, t1);
    fp6_mul_v(t1, t1);
    fp6_add(z.c0, t0, t1);
    z.c1 = c1;
}
__device__ __noinline__ void fp12_sqr(Fp12 &z, const Fp12 &x) {
    Fp6 s, t, m, mv;
    fp6_add(s, x.c0, x.c1);
    fp6_mul_v(t, x.c1);
    fp6_add(t, t, x.c0);
    fp6_mul(m, x.c0, x.c1);
    fp6_mul(s, s, t);
    fp6_mul_v(mv, m);
    fp6_sub(s, s, m);
    fp6_sub(z.c0, s, mv);
    fp6_add(z.c1, m, m);
}
__device__ __forceinline__ void fp12_conj(Fp12 &z, const Fp12 &x) { z.c0 = x.c0; fp6_neg(z.c1, x.c1); }
__device__ __noinline__ void fp12_inv(Fp12 &z, const Fp12 &x) {
    Fp6 t0, t1;
    fp6_sqr(t0, x.c0);
    fp6_sqr(t1, x.c1);
    fp6_mul_v(t1, t1);
    fp6_sub(t0, t0, t1);
    fp6_inv(t0, t0);
    fp6_mul(z.c0, x.c0, t0);
    fp6_mul(t1, x.c1, t0);
    fp6_neg(z.c1, t1);
}
// x^(p^j): coefficient of w^k -> (conj if j odd)(c_k) * gamma_j[k]; w-basis order C0.B0,C1.B0,C0.B1,C1.B1,C0.B2,C1.B2
__device__ __noinline__ void fp12_frob(Fp12 &z, const Fp12 &x, int j) {
    Fp2 c[6] = {x.c0.b0, x.c1.b0, x.c0.b1, x.c1.b1, x.c0.b2, x.c1.b2};
    const bool odd = j & 1;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if (odd) c[k] = fp2_conj(c[k]);
        if (k) c[k] = fp2_mul(c[k], gamma(j, k));
    }
    z.c0.b0 = c[0]; z.c1.b0 = c[1]; z.c0.b1 = c[2]; z.c1.b1 = c[3]; z.c0.b2 = c[4]; z.c1.b2 = c[5];
}
// Granger-Scott squaring in the cyclotomic subgroup (after the easy part of the final exponentiation)
__device__ __noinline__ void fp12_cyclo_sqr(Fp12 &z, const Fp12 &x) {
    Fp2 t0 = fp2_sqr(x.c1.b1), t1 = fp2_sqr(x.c0.b0);
    Fp2 t6 = fp2_sub(fp2_sub(fp2_sqr(fp2_add(x.c1.b1, x.c0.b0)), t0), t1);
    Fp2 t2 = fp2_sqr(x.c0.b2), t3 = fp2_sqr(x.c1.b0);
    Fp2 t7 = fp2_sub(fp2_sub(fp2_sqr(fp2_add(x.c0.b2, x.c1.b0)), t2), t3);
    Fp2 t4 = fp2_sqr(x.c1.b2), t5 = fp2_sqr(x.c0.b1);
    Fp2 t8 = fp2_mul_xi(fp2_sub(fp2_sub(fp2_sqr(fp2_add(x.c1.b2, x.c0.b1)), t4), t5));
    t0 = fp2_add(fp2_mul_xi(t0), t1);
    t2 = fp2_add(fp2_mul_xi(t2), t3);
    t4 = fp2_add(fp2_mul_xi(t4), t5);
    Fp2 r00 = fp2_add(fp2_dbl(fp2_sub(t0, x.c0.b0)), t0);
    Fp2 r01 = fp2_add(fp2_dbl(fp2_sub(t2, x.c0.b1)), t2);
    Fp2 r02 = fp2_add(fp2_dbl(fp2_sub(t4, x.c0.b2)), t4);
    Fp2 r10 = fp2_add(fp2_dbl(fp2_add(t8, x.c1.b0)), t8);
    Fp2 r11 = fp2_add(fp2_dbl(fp2_add(t6, x.c1.b1)), t6);
    Fp2 r12 = fp2_add(fp2_dbl(fp2_add(t7, x.c1.b2)), t7);
    z.c0.b0 = r00; z.c0.b1 = r01; z.c0.b2 = r02; z.c1.b0 = r10; z.c1.b1 = r11; z.c1.b2 = r12;
}
// z = x * (c0 + c3 w + c4 v w)   — the sparse line element of the Miller loop
__device__ __noinline__ void fp12_mul_034(Fp12 &z, const Fp12 &x, const Fp2 &c0, const Fp2 &c3, const Fp2 &c4) {
    Fp6 a, b, t;
    fp6_mul_fp2(a, x.c0, c0);
    fp6_mul_01(b, x.c1, c3, c4);
    fp6_mul_v(b, b);
    fp6_add(a, a, b);
    fp6_mul_01(b, x.c0, c3, c4);
    fp6_mul_fp2(t, x.c1, c0);
    fp6_add(z.c1, b, t);
    z.c0 = a;
}
// x^u, u = 0x44e992b44a6909f1, x in the cyclotomic subgroup
__device__ __noinline__ void fp12_expt(Fp12 &z, const Fp12 &x) {
    Fp12 r = x;
    for (int i = BN254_U_BITS - 2; i >= 0; i--) {
        fp12_cyclo_sqr(r, r);
        if ((BN254_U >> i) & 1) fp12_mul(r, r, x);
    }
    z = r;
}
__device__ __forceinline__ void fp12_load(Fp12 &z, const uint8_t *p) {
    z.c0.b0 = fp2_load(p); z.c0.b1 = fp2_load(p + 64); z.c0.b2 = fp2_load(p + 128);
    z.c1.b0 = fp2_load(p + 192); z.c1.b1 = fp2_load(p + 256); z.c1.b2 = fp2_load(p + 320);
}
__device__ __forceinline__ void fp12_store(uint8_t *p, const Fp12 &z) {
    fp2_store(p, z.c0.b0); fp2_store(p + 64, z.c0.b1); fp2_store(p + 128, z.c0.b2);
    fp2_store(p + 192, z.c1.b0); fp2_store(p + 256, z.c1.b1); fp2_store(p + 320, z.c1.b2);
}

}  // namespace gpbc
#endif
