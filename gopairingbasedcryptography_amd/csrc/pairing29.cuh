// Optimal-ate pairing on BN254 over the 29-bit-limb tower: Miller loop + final exponentiation, one pairing per lane.
// Replaces the inside of gnark-crypto's bn254.Pair / PairingCheck as called by the reference at
// cpabe/bsw07/bsw07_cpabe.go:75,184, access/tree/access_tree_node.go:106,110,119,
// bibe/afp25_bibe/afp25_bibe.go:227,395,399,403, signature/bls01_signature/bls_signature.go:81.
// The GT value is the unique residue e(P,Q)^s (s = 2u(6u^2+3u+1)), so any correct
// evaluation order gives gnark's bytes once the result is made canonical.
#ifndef GPBC_PAIRING29_CUH
#define GPBC_PAIRING29_CUH
#include "bn254_constants.cuh"
#include "tower29.cuh"

namespace gpbc {

struct G1A { Fe x, y; };
struct G2A { F2 x, y; };
struct G2P { F2 x, y, z; };     // homogeneous projective
struct LineE { F2 r0, r1, r2; };  // l = r0*yP + r1*xP*w + r2*w^3

GPBC_INLINE int ate_naf_digit(int i) { constexpr int8_t D[BN254_ATE_NAF_LEN] = BN254_ATE_NAF; return D[i]; }

// Tangent line at T and T <- 2T (Costello-Lange-Naehrig, ePrint 2013/722 §4.3, a = 0 twist). T in/out N-class.
GPBC_INLINE void g2_double_step(G2P &t, LineE &l) {
    F2 A = f2_halve(f2_mul(t.x, t.y));
    F2 B = f2_sqr(t.y);
    F2 C = f2_sqr(t.z);
    F2 E = f2_mul(f2_norm(f2_add(f2_dbl(C), C)), b_twist29());
    F2 F = f2_add(f2_dbl(E), E);
    F2 G = f2_halve(f2_norm(f2_add(B, F)));
    F2 H = f2_sub(f2_sqr(f2_norm(f2_add(t.y, t.z))), f2_add(B, C));
    F2 J = f2_sqr(t.x);
    F2 EE = f2_sqr(E);
    t.x = f2_mul(f2_norm(A), f2_norm(f2_sub(B, F)));
    t.y = f2_norm(f2_sub(f2_sqr(f2_norm(G)), f2_add(f2_dbl(EE), EE)));
    t.z = f2_mul(B, f2_norm(H));
    l.r0 = f2_norm(f2_neg(H));
    l.r1 = f2_norm(f2_add(f2_dbl(J), J));
    l.r2 = f2_norm(f2_sub(E, B));
}
// Chord through T and affine Q, T <- T + Q
GPBC_INLINE void g2_add_step(G2P &t, LineE &l, const G2A &q) {
    F2 O = f2_norm(f2_sub(t.y, f2_mul(q.y, t.z)));
    F2 L = f2_norm(f2_sub(t.x, f2_mul(q.x, t.z)));
    F2 C = f2_sqr(O);
    F2 D = f2_sqr(L);
    F2 E = f2_mul(L, D);
    F2 F = f2_mul(t.z, C);
    F2 G = f2_mul(t.x, D);
    F2 H = f2_norm(f2_sub(f2_add(E, F), f2_dbl(G)));
    F2 t1 = f2_mul(t.y, E);
    t.x = f2_mul(L, H);
    t.y = f2_norm(f2_sub(f2_mul(f2_norm(f2_sub(G, H)), O), t1));
    t.z = f2_mul(E, t.z);
    l.r0 = L;
    l.r1 = f2_neg(O);
    l.r2 = f2_norm(f2_sub(f2_mul(q.x, O), f2_mul(L, q.y)));
}
GPBC_INLINE F12 line_apply(const F12 &f, const LineE &l, const G1A &p) {
    return f12_mul_034(f, f2_mul_fe(l.r0, p.y), f2_mul_fe(l.r1, p.x), l.r2);
}

// Miller function of (p,q); caller has checked neither is infinity
GPBC_INLINE F12 miller_loop29(const G1A &p, const G2A &q) {
    F12 f = f12_one();
    G2P t{q.x, q.y, f2_one()};
    G2A qn{q.x, f2_neg(q.y)};
    LineE l;
    for (int i = BN254_ATE_NAF_LEN - 2; i >= 0; i--) {
        if (i != BN254_ATE_NAF_LEN - 2) f = f12_sqr(f);
        g2_double_step(t, l);
        f = line_apply(f, l, p);
        int d = ate_naf_digit(i);
        if (d != 0) {
            g2_add_step(t, l, d > 0 ? q : qn);
            f = line_apply(f, l, p);
        }
    }
    G2A q1{f2_mul(f2_conj(q.x), gamma29(1, 2)), f2_mul(f2_conj(q.y), gamma29(1, 3))};
    G2A q2{f2_mul(q.x, gamma29(2, 2)), f2_neg(f2_mul(q.y, gamma29(2, 3)))};
    g2_add_step(t, l, q1);
    f = line_apply(f, l, p);
    g2_add_step(t, l, q2);
    f = line_apply(f, l, p);
    return f;
}

// x^u, u = 0x44e992b44a6909f1, x in the cyclotomic subgroup
GPBC_INLINE F12 f12_expt(const F12 &x) {
    F12 r = x;
    for (int i = BN254_U_BITS - 2; i >= 0; i--) {
        r = f12_cyclo_sqr(r);
        if ((BN254_U >> i) & 1) r = f12_mul(r, x);
    }
    return r;
}

// x^(s (p^12-1)/r): easy part, then the Fuentes-Castaneda hard part (gnark's operation order).
// gnark returns early when the easy part gives 1; the hard part maps 1 to 1, so the early exit is not needed here.
GPBC_INLINE F12 final_exp29(const F12 &x) {
    F12 t0 = f12_mul(f12_conj(x), f12_inv(x));
    F12 r = f12_mul(f12_frob(t0, 2), t0);
    t0 = f12_conj(f12_expt(r));
    t0 = f12_cyclo_sqr(t0);
    F12 t1 = f12_cyclo_sqr(t0);
    t1 = f12_mul(t0, t1);
    F12 t2 = f12_conj(f12_expt(t1));
    F12 t3 = f12_conj(t1);
    t1 = f12_mul(t2, t3);
    t3 = f12_cyclo_sqr(t2);
    F12 t4 = f12_expt(t3);
    t4 = f12_mul(t1, t4);
    t3 = f12_mul(t0, t4);
    t0 = f12_mul(t2, t4);
    t0 = f12_mul(r, t0);
    t2 = f12_frob(t3, 1);
    t0 = f12_mul(t2, t0);
    t2 = f12_frob(t4, 2);
    t0 = f12_mul(t2, t0);
    t2 = f12_mul(f12_conj(r), t3);
    t2 = f12_frob(t2, 3);
    return f12_mul(t2, t0);
}

}  // namespace gpbc
#endif
