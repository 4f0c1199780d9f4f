// Point arithmetic with ONE point per QUAD of lanes (device only): the products of a doubling or a mixed addition that do not depend on
// each other go to different lanes of the quad and come back by DPP broadcasts, so a dependent chain of point operations — the Horner
// tail of the bucket MSM, a scalar multiplication in a call too small to fill the chip with one point per lane — costs the DEPTH of
// its formulas in products instead of their number: a = 0 doubling 3 instead of 7, madd-2007-bl 5 instead of 11.  Every lane of the
// quad holds the whole point and runs the linear parts itself; the values are those of jac_dbl / jac_add_mixed (curve29.hip.hpp), with
// squares taken as general products.  All four lanes of a quad must be active together (they share every branch: one point).
#ifndef GPBC_CURVE29_QUAD_HIP_HPP
#define GPBC_CURVE29_QUAD_HIP_HPP
#include "curve29.hip.hpp"

namespace gpbc {

template <int SRC> __device__ __forceinline__ Fe quad_bcast(const Fe &a) {
    Fe r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = __builtin_amdgcn_mov_dpp(a.v[i], SRC * 0x55, 0xF, 0xF, true);   // quad_perm [SRC, SRC, SRC, SRC]
    return r;
}
template <int SRC> __device__ __forceinline__ F2 quad_bcast(const F2 &a) { return F2{quad_bcast<SRC>(a.a0), quad_bcast<SRC>(a.a1)}; }

// p <- 2 p.   levels: X^2, Y^2, Y 2Z | (3A)^2, B 8B, X 4B | E (S - x3)
template <class F> __device__ __forceinline__ void jac_dbl_quad(JacP<F> &p, int q) {
    if (p.inf) return;
    const F z2 = g_norm(g_dbl(p.z));
    const F p1 = g_mul(g_sel<F>(q == 0, p.x, p.y), g_sel<F>(q == 0, p.x, g_sel<F>(q == 1, p.y, z2)));   // lane 0: A = X^2, 1: B = Y^2, 2: z3 = Y 2Z
    const F A = quad_bcast<0>(p1), B = quad_bcast<1>(p1), z3 = quad_bcast<2>(p1);
    const F B4 = g_norm(g_dbl(g_dbl(B)));
    const F E = g_norm(g_add(g_dbl(A), A));
    const F p2 = g_mul(g_sel<F>(q == 0, E, g_sel<F>(q == 1, B, p.x)), g_sel<F>(q == 0, E, g_sel<F>(q == 1, g_norm(g_dbl(B4)), B4)));   // 0: FF = E^2, 1: C8 = B 8B, 2: S = X 4B
    const F FF = quad_bcast<0>(p2), C8 = quad_bcast<1>(p2), S = quad_bcast<2>(p2);
    const F x3 = g_norm(g_sub(FF, g_dbl(S)));
    p.y = g_sub(g_mul(E, g_sub(S, x3)), C8);
    p.x = x3; p.z = z3;
}

// p <- p + t (t affine), exceptional cases as in jac_add_mixed.
//   levels: Z^2, ty Z | tx Z1Z1, (ty Z) Z1Z1 | H^2, (Z + H)^2, r^2 | H I, X I | r (V - x3), Y J
template <class F> __device__ __forceinline__ void jac_add_mixed_quad(JacP<F> &p, const AffP<F> &t, int q) {
    if (t.inf) return;
    if (p.inf) { p.x = t.x; p.y = t.y; g_set_one(p.z); p.inf = false; return; }
    const F p1 = g_mul(g_sel<F>(q == 0, p.z, t.y), p.z);
    const F Z1Z1 = quad_bcast<0>(p1), YZ = quad_bcast<1>(p1);
    const F p2 = g_mul(g_sel<F>(q == 0, t.x, YZ), Z1Z1);
    const F U2 = quad_bcast<0>(p2), S2 = quad_bcast<1>(p2);
    const F H = g_sub(U2, p.x);
    F rr = g_norm(g_sub(S2, p.y));
    if (g_is_zero(H)) {                                       // the same on the four lanes
        if (g_is_zero(rr)) { jac_dbl_quad(p, q); return; }
        jac_set_inf(p);
        return;
    }
    rr = g_norm(g_dbl(rr));
    const F zh = g_norm(g_add(p.z, H));
    const F a3 = g_sel<F>(q == 0, H, g_sel<F>(q == 1, zh, rr));
    const F p3 = g_mul(a3, a3);
    const F HH = quad_bcast<0>(p3), ZH2 = quad_bcast<1>(p3), RR = quad_bcast<2>(p3);
    const F I = g_norm(g_dbl(g_dbl(HH)));
    const F p4 = g_mul(g_sel<F>(q == 0, H, p.x), I);
    const F J = quad_bcast<0>(p4), V = quad_bcast<1>(p4);
    const F x3 = g_norm(g_sub(g_sub(RR, J), g_dbl(V)));
    const F p5 = g_mul(g_sel<F>(q == 0, rr, p.y), g_sel<F>(q == 0, g_sub(V, x3), J));
    const F Ar = quad_bcast<0>(p5), Br = quad_bcast<1>(p5);
    p.y = g_norm(g_sub(Ar, g_dbl(Br)));
    p.z = g_norm(g_sub(g_sub(ZH2, Z1Z1), HH));
    p.x = x3;
}

// scalar_mul29_jac / scalar_mul29_gls with the loop's point operations on the quad.  The window table is built by every lane of the
// quad for itself (the same values into the same rows of `tab`: the quad shares one table block), the loop runs three wide.
template <class F> __device__ __forceinline__ void scalar_mul29_jac_quad(JacP<F> &acc, const AffP<F> &base, const uint32_t k[8], int32_t *tab, int q) {
    GlvSplit s;
    glv_split(s, k);
    jac_set_inf(acc);
    int top = 159;
    while (top >= 0 && !(((s.k1[top >> 5] | s.k2[top >> 5]) >> (top & 31)) & 1)) top--;
    if (base.inf || top < 0) return;
    AffP<F> p1{base.x, s.neg1 ? g_neg(base.y) : base.y, false};
    AffP<F> p2{glv_phi_x(base.x), s.neg2 ? g_neg(base.y) : base.y, false};
    F W;
    glv_table29<F>(tab, W, p1, p2, s.neg1 != s.neg2);
    for (int i = top >> 1; i >= 0; i--) {
        const int b = 2 * i;
        const int idx = 4 * (int)((s.k1[b >> 5] >> (b & 31)) & 3) + (int)((s.k2[b >> 5] >> (b & 31)) & 3);
        AffP<F> t;
        tab_load(tab, idx, t);
        jac_dbl_quad(acc, q);
        jac_dbl_quad(acc, q);
        if (idx) jac_add_mixed_quad(acc, t, q);
    }
    if (!acc.inf) acc.z = g_mul(acc.z, W);               // back from the curve scaled by W
}
__device__ __forceinline__ void scalar_mul29_gls_quad(JacP<F2> &acc, const AffP<F2> &base, const uint32_t k[8], int32_t *tab, int q) {
    GlsSplit s;
    gls_split(s, k);
    jac_set_inf(acc);
    int top = 95;
    while (top >= 0 && !(((s.k[0][top >> 5] | s.k[1][top >> 5] | s.k[2][top >> 5] | s.k[3][top >> 5]) >> (top & 31)) & 1)) top--;
    if (base.inf || top < 0) return;
    AffP<F2> P[4];
    P[0] = AffP<F2>{base.x, s.neg[0] ? f2_neg(base.y) : base.y, false};
    for (int i = 1; i < 4; i++) {
        const bool cj = i & 1;
        F2 y = f2_mul(cj ? f2_conj(base.y) : base.y, gamma29(i, 3));
        P[i] = AffP<F2>{f2_mul(cj ? f2_conj(base.x) : base.x, gamma29(i, 2)), s.neg[i] ? f2_neg(y) : y, false};
    }
    F2 W;
    gls_table29(tab, W, P);
    for (int i = top; i >= 0; i--) {
        const int w = i >> 5, b = i & 31;
        const int idx = (int)((s.k[0][w] >> b) & 1) | (int)(((s.k[1][w] >> b) & 1) << 1) | (int)(((s.k[2][w] >> b) & 1) << 2) | (int)(((s.k[3][w] >> b) & 1) << 3);
        AffP<F2> t;
        tab_load(tab, idx, t);
        jac_dbl_quad(acc, q);
        if (idx) jac_add_mixed_quad(acc, t, q);
    }
    if (!acc.inf) acc.z = f2_mul(acc.z, W);
}

}  // namespace gpbc
#endif
