// Optimal-ate pairing on BN254 for gfx950: Miller loop + final exponentiation, one pairing per lane.
// Replaces the inside of gnark-crypto's bn254.Pair / PairingCheck as called by the reference at
// cpabe/bsw07/bsw07_cpabe.go:75,184, access/tree/access_tree_node.go:106,110,119,
// bibe/afp25_bibe/afp25_bibe.go:227,395,399,403, signature/bls01_signature/bls_signature.go:81.
// The GT value is the unique residue e(P,Q)^s (s = 2u(6u^2+3u+1)), so any correct
// evaluation order gives gnark's bytes.
#ifndef GPBC_PAIRING_CUH
#define GPBC_PAIRING_CUH
#include "tower.cuh"

namespace gpbc {

struct G1Aff { Fp x, y; };
struct G2Aff { Fp2 x, y; };
struct G2Proj { Fp2 x, y, z; };     // homogeneous projective
struct Line { Fp2 r0, r1, r2; };    // l = r0*yP + r1*xP*w + r2*w^3

__device__ const int8_t ATE_NAF[BN254_ATE_NAF_LEN] = BN254_ATE_NAF;

__device__ __forceinline__ bool g1_is_inf(const G1Aff &p) { return fp_is_zero(p.x) && fp_is_zero(p.y); }
__device__ __forceinline__ bool g2_is_inf(const G2Aff &q) { return fp2_is_zero(q.x) && fp2_is_zero(q.y); }
__device__ __forceinline__ G1Aff g1_load(const uint8_t *p) { return G1Aff{fp_load(p), fp_load(p + 32)}; }
__device__ __forceinline__ G2Aff g2_load(const uint8_t *p) { return G2Aff{fp2_load(p), fp2_load(p + 64)}; }

// Tangent line at T and T <- 2T (Costello-Lange-Naehrig, ePrint 2013/722 §4.3, a=0 twist)
__device__ __noinline__ void g2_double_step(G2Proj &t, Line &l) {
    Fp2 A = fp2_halve(fp2_mul(t.x, t.y));
    Fp2 B = fp2_sqr(t.y);
    Fp2 C = fp2_sqr(t.z);
    Fp2 E = fp2_mul(fp2_add(fp2_dbl(C), C), b_twist());
    Fp2 F = fp2_add(fp2_dbl(E), E);
    Fp2 G = fp2_halve(fp2_add(B, F));
    Fp2 H = fp2_sub(fp2_sqr(fp2_add(t.y, t.z)), fp2_add(B, C));
    Fp2 J = fp2_sqr(t.x);
    Fp2 EE = fp2_sqr(E);
    t.x = fp2_mul(A, fp2_sub(B, F));
    t.y = fp2_sub(fp2_sqr(G), fp2_add(fp2_dbl(EE), EE));
    t.z = fp2_mul(B, H);
    l.r0 = fp2_neg(H);
    l.r1 = fp2_add(fp2_dbl(J), J);
    l.r2 = fp2_sub(E, B);
}
// Chord through T and affine Q, T <- T + Q
__device__ __noinline__ void g2_add_step(G2Proj &t, Line &l, const G2Aff &q) {
    Fp2 O = fp2_sub(t.y, fp2_mul(q.y, t.z));
    Fp2 L = fp2_sub(t.x, fp2_mul(q.x, t.z));
    Fp2 C = fp2_sqr(O);
    Fp2 D = fp2_sqr(L);
    Fp2 E = fp2_mul(L, D);
    Fp2 F = fp2_mul(t.z, C);
    Fp2 G = fp2_mul(t.x, D);
    Fp2 H = fp2_sub(fp2_add(E, F), fp2_dbl(G));
    Fp2 t1 = fp2_mul(t.y, E);
    t.x = fp2_mul(L, H);
    t.y = fp2_sub(fp2_mul(fp2_sub(G, H), O), t1);
    t.z = fp2_mul(E, t.z);
    l.r0 = L;
    l.r1 = fp2_neg(O);
    l.r2 = fp2_sub(fp2_mul(q.x, O), fp2_mul(L, q.y));
}
__device__ __forceinline__ void line_apply(Fp12 &f, const Line &l, const G1Aff &p) {
    fp12_mul_034(f, f, fp2_mul_fp(l.r0, p.y), fp2_mul_fp(l.r1, p.x), l.r2);
}

// f <- Miller function of (p,q); caller has checked neither is infinity
__device__ __noinline__ void miller_loop(Fp12 &f, const G1Aff &p, const G2Aff &q) {
    fp12_set_one(f);
    G2Proj t{q.x, q.y, fp2_one()};
    G2Aff qn{q.x, fp2_neg(q.y)};
    Line l;
    for (int i = BN254_ATE_NAF_LEN - 2; i >= 0; i--) {
        if (i != BN254_ATE_NAF_LEN - 2) fp12_sqr(f, f);
        g2_double_step(t, l);
        line_apply(f, l, p);
        int d = ATE_NAF[i];
        if (d != 0) {
            g2_add_step(t, l, d > 0 ? q : qn);
            line_apply(f, l, p);
        }
    }
    G2Aff q1{fp2_mul(fp2_conj(q.x), gamma(1, 2)), fp2_mul(fp2_conj(q.y), gamma(1, 3))};
    G2Aff q2{fp2_mul(q.x, gamma(2, 2)), fp2_neg(fp2_mul(q.y, gamma(2, 3)))};
    g2_add_step(t, l, q1);
    line_apply(f, l, p);
    g2_add_step(t, l, q2);
    line_apply(f, l, p);
}

// z = x^(s (p^12-1)/r): easy part, then Fuentes-Castaneda hard part (gnark's operation order)
__device__ __noinline__ void final_exp(Fp12 &z, const Fp12 &x) {
    Fp12 r, t0, t1, t2, t3, t4;
    fp12_conj(t0, x);
    fp12_inv(r, x);
    fp12_mul(t0, t0, r);
    fp12_frob(r, t0, 2);
    fp12_mul(r, r, t0);
    if (fp12_is_one(r)) { z = r; return; }
    fp12_expt(t0, r); fp12_conj(t0, t0);
    fp12_cyclo_sqr(t0, t0);
    fp12_cyclo_sqr(t1, t0);
    fp12_mul(t1, t0, t1);
    fp12_expt(t2, t1); fp12_conj(t2, t2);
    fp12_conj(t3, t1);
    fp12_mul(t1, t2, t3);
    fp12_cyclo_sqr(t3, t2);
    fp12_expt(t4, t3);
    fp12_mul(t4, t1, t4);
    fp12_mul(t3, t0, t4);
    fp12_mul(t0, t2, t4);
    fp12_mul(t0, r, t0);
    fp12_frob(t2, t3, 1);
    fp12_mul(t0, t2, t0);
    fp12_frob(t2, t4, 2);
    fp12_mul(t0, t2, t0);
    fp12_conj(t2, r);
    fp12_mul(t2, t2, t3);
    fp12_frob(t2, t2, 3);
    fp12_mul(t0, t2, t0);
    z = t0;
}

}  // namespace gpbc
#endif
