// Hash to curve, the group part, one output point per lane: the tail of bn254.HashToG1 / HashToG2
// after hash_to_field (reference hash/hash_to.go:113-119,169-175,204-210,271-277) [EXT, parity unpinned] —
//   Q0 = map_to_curve(u0), Q1 = map_to_curve(u1), R = Q0 + Q1, and for G2 R = clear_cofactor(R).
// map_to_curve is the Shallue-van de Woestijne map of RFC 9380 §6.6.1 in its straight-line form (Appendix F.1, which is
// what gnark-crypto's MapToCurve1 / MapToCurve2 list step by step) with Z = 1 for both curves (the first value RFC 9380
// Appendix H.1 find_z_svdw accepts; tools/gen_constants.py derives it and c1..c4).  clear_cofactor is Fuentes-Castaneda et al. §6.1 as
// gnark's G2Jac.ClearCofactor does it:  [x]P + psi([3x]P) + psi^2([x]P) + psi^3(P),  x = the curve parameter u.
// The message hashing itself (expand_message_xmd with SHA-256, L = 48) is csrc/xmd29.hip.hpp: the k_g1_hash / k_g2_hash kernels
// run it in the same lane right before this map; the map_fields kernels take field elements a caller hashed itself.
#ifndef GPBC_H2C29_HIP_HPP
#define GPBC_H2C29_HIP_HPP
#include "wire29.hip.hpp"

namespace gpbc {

struct SvdwFe { Fe z, c1, c2, c3, c4, b; };
struct SvdwF2 { F2 z, c1, c2, c3, c4, b; };
GPBC_INLINE void svdw_load(SvdwFe &k) {
    constexpr int32_t Z[NL] = SVDW_G1_Z, C1[NL] = SVDW_G1_C1, C2[NL] = SVDW_G1_C2, C3[NL] = SVDW_G1_C3, C4[NL] = SVDW_G1_C4, B[NL] = F29_B_G1;
    k.z = fe_const(Z); k.c1 = fe_const(C1); k.c2 = fe_const(C2); k.c3 = fe_const(C3); k.c4 = fe_const(C4); k.b = fe_const(B);
}
GPBC_INLINE void svdw_load(SvdwF2 &k) {
    constexpr int32_t Z[2][NL] = SVDW_G2_Z, C1[2][NL] = SVDW_G2_C1, C2[2][NL] = SVDW_G2_C2, C3[2][NL] = SVDW_G2_C3, C4[2][NL] = SVDW_G2_C4, B[2][NL] = F29_B_G2;
    k.z = f2_const(Z); k.c1 = f2_const(C1); k.c2 = f2_const(C2); k.c3 = f2_const(C3); k.c4 = f2_const(C4); k.b = f2_const(B);
}
template <class F> struct SvdwOf;
template <> struct SvdwOf<Fe> { typedef SvdwFe type; };
template <> struct SvdwOf<F2> { typedef SvdwF2 type; };

// is_square: Legendre symbol (fe_legendre; a^((p-1)/2) != -1 where that does not decide; 0 counts as a square, as in RFC 9380);
// for Fp2 on the norm
GPBC_INLINE bool g_is_square(const Fe &a) { return fe_is_square(a); }
GPBC_INLINE bool g_is_square(const F2 &a) { return g_is_square(fe_norm(fe_add(fe_sqr(a.a0), fe_sqr(a.a1)))); }
GPBC_INLINE Fe g_sqrt(const Fe &a, bool &ok) { return fe_sqrt(a, ok); }
GPBC_INLINE F2 g_sqrt(const F2 &a, bool &ok) { return f2_sqrt(a, ok); }
// sgn0 of RFC 9380 §4.1: parity of the canonical value; for Fp2 sign_0 OR (zero_0 AND sign_1)
GPBC_INLINE int g_sgn0(const Fe &a) {
    uint32_t w[8];
    fe_to_plain_words(w, a);
    return (int)(w[0] & 1);
}
GPBC_INLINE int g_sgn0(const F2 &a) {
    uint32_t w0[8], w1[8];
    fe_to_plain_words(w0, a.a0);
    fe_to_plain_words(w1, a.a1);
    return (int)((w0[0] & 1) | ((wire_words_zero(w0) ? 1u : 0u) & (w1[0] & 1)));
}
template <class F, class K> GPBC_INLINE F svdw_g(const K &k, const F &x) { return g_norm(g_add(g_mul(g_sqr(x), x), k.b)); }

// RFC 9380 Appendix F.1, A = 0.  Every lane takes the same path: the three candidates are all formed and selected.
template <class F> GPBC_NOINLINE void map_to_curve_svdw(AffP<F> &out, const F &u) {
    typename SvdwOf<F>::type k;
    svdw_load(k);
    F one;
    g_set_one(one);
    F tv1 = g_mul(g_sqr(u), k.c1);                              // 1-2
    F tv2 = g_norm(g_add(one, tv1));                            // 3
    tv1 = g_norm(g_sub(one, tv1));                              // 4
    F tv3 = g_inv(g_mul(tv1, tv2));                             // 5-6   inv0(0) = 0
    F tv4 = g_mul(g_mul(g_mul(u, tv1), tv3), k.c3);             // 7-9
    F x1 = g_norm(g_sub(k.c2, tv4));                            // 10
    bool e1 = g_is_square(svdw_g(k, x1));                       // 11-15
    F x2 = g_norm(g_add(k.c2, tv4));                            // 16
    bool e2 = g_is_square(svdw_g(k, x2)) && !e1;                // 17-21
    F x3 = g_sqr(g_mul(g_sqr(tv2), tv3));                       // 22-24
    x3 = g_norm(g_add(g_mul(x3, k.c4), k.z));                   // 25-26
    F x = g_sel<F>(e1, x1, x3);                                 // 27
    x = g_sel<F>(e2, x2, x);                                    // 28
    bool ok;
    F y = g_sqrt(svdw_g(k, x), ok);                             // 29-33 (a square by construction)
    if (g_sgn0(u) != g_sgn0(y)) y = g_neg(y);                   // 34-35
    out.x = x; out.y = y; out.inf = false;
}

// psi^j on Jacobian coordinates of the twist: jac_psi_tw (wire29.hip.hpp)
GPBC_INLINE JacP<F2> jac_psi(const JacP<F2> &p, int j) { return jac_psi_tw(p, j); }
// [x]P + psi([3x]P) + psi^2([x]P) + psi^3(P) for an affine twist point P
GPBC_NOINLINE void g2_clear_cofactor29(JacP<F2> &out, const AffP<F2> &p) {
    constexpr uint64_t X = 4965661367192848881ull;              // BN254 parameter u
    JacP<F2> xq, t;
    jac_set_inf(xq);
    for (int i = 62; i >= 0; i--) {
        jac_dbl(xq, xq);
        if ((X >> i) & 1) jac_add_mixed(xq, xq, p);
    }
    jac_dbl(t, xq);
    jac_add(t, t, xq);                                          // [3x]P
    JacP<F2> pj{p.x, p.y, f2_one(), p.inf};
    JacP<F2> acc;
    jac_add(acc, xq, jac_psi(t, 1));
    jac_add(acc, acc, jac_psi(xq, 2));
    jac_add(out, acc, jac_psi(pj, 3));
}

// u0, u1 -> G1 point (affine): MapToCurve1(u0) + MapToCurve1(u1); cofactor 1
GPBC_INLINE void g1_map_fields(AffP<Fe> &out, const Fe &u0, const Fe &u1) {
    AffP<Fe> q0, q1;
    map_to_curve_svdw<Fe>(q0, u0);
    map_to_curve_svdw<Fe>(q1, u1);
    JacP<Fe> j{q0.x, q0.y, fe_one(), false}, s;
    jac_add_mixed(s, j, q1);
    jac_to_affine(out, s);
}
GPBC_INLINE void g2_map_fields(AffP<F2> &out, const F2 &u0, const F2 &u1) {
    AffP<F2> q0, q1, a;
    map_to_curve_svdw<F2>(q0, u0);
    map_to_curve_svdw<F2>(q1, u1);
    JacP<F2> j{q0.x, q0.y, f2_one(), false}, s, c;
    jac_add_mixed(s, j, q1);
    jac_to_affine(a, s);
    g2_clear_cofactor29(c, a);
    jac_to_affine(out, c);
}

}  // namespace gpbc
#endif
