// G2 point arithmetic with ONE point per OCTET of lanes (device only): the quad form (curve29_quad.hip.hpp) gives every product of a
// level its own lane; over Fp2 a product is two independent Fp halves (a0 b0 - a1 b1, a0 b1 + a1 b0: one 243-MAD leaf each instead of
// one 486-MAD leaf), and a lone wave issues one MAD per ~9 cycles whatever else it does, so the halves go to the two lanes of a PAIR
// and an octet holds four product slots.  A level then costs half the multiplier time of the quad form; the two halves meet again
// through one DPP pair swap and a product reaches the other lanes of the octet through ds_bpermute.  Same values as jac_dbl /
// jac_add_mixed / gls_table29 (the half products are the components of f2_mul's result, limb for limb).  Used for calls of up to 2 048
// elements, where the depth of the chain is all that counts: ScalarMultiplication in G2 (signature/bls01_signature/bls_signature.go:63:
// every BLS Sign; gpbc_curve.hip: k_g2_scalar_mul_oct), the [x]P chains of HashToG2's cofactor clearing (hash/hash_to.go:271-277) and of
// G2 Unmarshal's subgroup test (serialization/serialization_curve.go:23-27; gpbc_wire.hip: k_g2_hash_oct, k_g2_map_fields_oct,
// k_g2_decode_oct).  All eight lanes of an octet must be active together (one point: they share every branch).
#ifndef GPBC_CURVE29_OCT_HIP_HPP
#define GPBC_CURVE29_OCT_HIP_HPP
#include "curve29_quad.hip.hpp"

namespace gpbc {

// the product a b, whole, on both lanes of the pair (q & 1 = which half this lane computes)
__device__ __forceinline__ F2 oct_mul(const F2 &a, const F2 &b, int q) {
    const bool h = q & 1;
    const Fe r = fe_mul2_l(a.a0, fe_sel(h, b.a1, b.a0), fe_sel(h, a.a1, fe_neg(a.a1)), fe_sel(h, b.a0, b.a1));
    Fe o;
#pragma unroll
    for (int i = 0; i < NL; i++) o.v[i] = __builtin_amdgcn_mov_dpp(r.v[i], 0xB1, 0xF, 0xF, true);     // the partner's half (quad_perm [1,0,3,2])
    return F2{fe_sel(h, o, r), fe_sel(h, r, o)};
}
// the value product slot S holds (lanes 2 S, 2 S + 1 of the octet), on every lane of the octet
template <int S> __device__ __forceinline__ F2 oct_from(const F2 &v) {
    const int src = (int)(((threadIdx.x & 63u) & ~7u) + 2 * S) * 4;
    F2 r;
#pragma unroll
    for (int i = 0; i < NL; i++) { r.a0.v[i] = __builtin_amdgcn_ds_bpermute(src, v.a0.v[i]); r.a1.v[i] = __builtin_amdgcn_ds_bpermute(src, v.a1.v[i]); }
    return r;
}
// four independent products in one round: slot s multiplies a[s] b[s]; every lane gets back its own slot's product
__device__ __forceinline__ F2 oct_round(const F2 &a0, const F2 &b0, const F2 &a1, const F2 &b1, const F2 &a2, const F2 &b2, const F2 &a3, const F2 &b3, int q) {
    const int s = q >> 1;
    return oct_mul(f2_sel(s == 0, a0, f2_sel(s == 1, a1, f2_sel(s == 2, a2, a3))), f2_sel(s == 0, b0, f2_sel(s == 1, b1, f2_sel(s == 2, b2, b3))), q);
}

// p <- 2 p.   levels as jac_dbl_quad: X^2, Y^2, Y 2Z | (3A)^2, B 8B, X 4B | E (S - x3)
__device__ __forceinline__ void jac_dbl_oct(JacP<F2> &p, int q) {
    if (p.inf) return;
    const int s = q >> 1;
    const F2 z2 = f2_norm(f2_dbl(p.z));
    const F2 p1 = oct_mul(f2_sel(s == 0, p.x, p.y), f2_sel(s == 0, p.x, f2_sel(s == 1, p.y, z2)), q);
    const F2 A = oct_from<0>(p1), B = oct_from<1>(p1), z3 = oct_from<2>(p1);
    const F2 B4 = f2_norm(f2_dbl(f2_dbl(B)));
    const F2 E = f2_norm(f2_add(f2_dbl(A), A));
    const F2 p2 = oct_mul(f2_sel(s == 0, E, f2_sel(s == 1, B, p.x)), f2_sel(s == 0, E, f2_sel(s == 1, f2_norm(f2_dbl(B4)), B4)), q);
    const F2 FF = oct_from<0>(p2), C8 = oct_from<1>(p2), S = oct_from<2>(p2);
    const F2 x3 = f2_norm(f2_sub(FF, f2_dbl(S)));
    p.y = f2_sub(oct_mul(E, f2_sub(S, x3), q), C8);           // (every pair computes it: the product is whole on each lane)
    p.x = x3; p.z = z3;
}

// p <- p + t (t affine), exceptional cases as in jac_add_mixed; levels as jac_add_mixed_quad
__device__ __forceinline__ void jac_add_mixed_oct(JacP<F2> &p, const AffP<F2> &t, int q) {
    if (t.inf) return;
    if (p.inf) { p.x = t.x; p.y = t.y; p.z = f2_one(); p.inf = false; return; }
    const int s = q >> 1;
    const F2 p1 = oct_mul(f2_sel(s == 0, p.z, t.y), p.z, q);
    const F2 Z1Z1 = oct_from<0>(p1), YZ = oct_from<1>(p1);
    const F2 p2 = oct_mul(f2_sel(s == 0, t.x, YZ), Z1Z1, q);
    const F2 U2 = oct_from<0>(p2), S2 = oct_from<1>(p2);
    const F2 H = f2_sub(U2, p.x);
    F2 rr = f2_norm(f2_sub(S2, p.y));
    if (f2_is_zero(H)) {                                      // the same on the eight lanes
        if (f2_is_zero(rr)) { jac_dbl_oct(p, q); return; }
        jac_set_inf(p);
        return;
    }
    rr = f2_norm(f2_dbl(rr));
    const F2 zh = f2_norm(f2_add(p.z, H));
    const F2 a3 = f2_sel(s == 0, H, f2_sel(s == 1, zh, rr));
    const F2 p3 = oct_mul(a3, a3, q);
    const F2 HH = oct_from<0>(p3), ZH2 = oct_from<1>(p3), RR = oct_from<2>(p3);
    const F2 I = f2_norm(f2_dbl(f2_dbl(HH)));
    const F2 p4 = oct_mul(f2_sel(s == 0, H, p.x), I, q);
    const F2 J = oct_from<0>(p4), V = oct_from<1>(p4);
    const F2 x3 = f2_norm(f2_sub(f2_sub(RR, J), f2_dbl(V)));
    const F2 p5 = oct_mul(f2_sel(s == 0, rr, p.y), f2_sel(s == 0, f2_sub(V, x3), J), q);
    const F2 Ar = oct_from<0>(p5), Br = oct_from<1>(p5);
    p.y = f2_norm(f2_sub(Ar, f2_dbl(Br)));
    p.z = f2_norm(f2_sub(f2_sub(ZH2, Z1Z1), HH));
    p.x = x3;
}
// r <- a + t with a given (a copy): the table's additions
__device__ __forceinline__ JacP<F2> jac_sum_oct(JacP<F2> a, const AffP<F2> &t, int q) { jac_add_mixed_oct(a, t, q); return a; }

// gls_table29 on the octet: the eleven additions one after the other, each five rounds deep instead of eleven products long, and the
// scaling to one common Z with four products per round.  Every lane of the octet stores the same rows (one table block per point).
__device__ __noinline__ void gls_table29_oct(int32_t *tab, F2 &W, const AffP<F2> (&P)[4], int q) {
    const F2 one = f2_one();
    static constexpr int IDX[11] = {3, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15};
    JacP<F2> J[11];
    auto single = [&](int i) { return JacP<F2>{P[i].x, P[i].y, one, false}; };
    J[0] = jac_sum_oct(single(0), P[1], q);
    J[1] = jac_sum_oct(single(0), P[2], q);
    J[2] = jac_sum_oct(single(1), P[2], q);
    J[3] = jac_sum_oct(J[0], P[2], q);
    J[4] = jac_sum_oct(single(0), P[3], q);
    J[5] = jac_sum_oct(single(1), P[3], q);
    J[6] = jac_sum_oct(J[0], P[3], q);
    J[7] = jac_sum_oct(single(2), P[3], q);
    J[8] = jac_sum_oct(J[1], P[3], q);
    J[9] = jac_sum_oct(J[2], P[3], q);
    J[10] = jac_sum_oct(J[3], P[3], q);
    // prefix and suffix products of the eleven Z: two chains, one per product slot of a round (slots 2, 3 idle)
    F2 pre[11], suf[11];                                      // pre[i] = z_0 .. z_(i-1),  suf[i] = z_(i+1) .. z_10
    pre[0] = one; suf[10] = one;
    for (int i = 1; i < 11; i++) {
        const F2 m = oct_round(pre[i - 1], J[i - 1].z, suf[11 - i], J[11 - i].z, one, one, one, one, q);
        pre[i] = oct_from<0>(m); suf[10 - i] = oct_from<1>(m);
    }
    {
        const F2 m = oct_mul(pre[10], J[10].z, q);
        W = m;                                                // (whole on every lane: all pairs computed the same product)
    }
    const int s = q >> 1;
    for (int i = 0; i < 11; i++) {
        const F2 l = oct_mul(pre[i], suf[i], q);              // the product of the other ten Z
        const F2 l2 = oct_mul(l, l, q);
        const F2 m = oct_round(J[i].x, l2, l2, l, one, one, one, one, q);              // slot 0: x l^2, slot 1: l^3
        const F2 x = oct_from<0>(m), l3 = oct_from<1>(m);
        const F2 y = oct_mul(J[i].y, l3, q);
        tab_store(tab, IDX[i], AffP<F2>{x, y, false});
    }
    const F2 w2 = oct_mul(W, W, q), w3 = oct_mul(w2, W, q);
    {
        const F2 m = oct_round(P[0].x, w2, P[1].x, w2, P[2].x, w2, P[3].x, w2, q), n = oct_round(P[0].y, w3, P[1].y, w3, P[2].y, w3, P[3].y, w3, q);
        const F2 x0 = oct_from<0>(m), x1 = oct_from<1>(m), x2 = oct_from<2>(m), x3 = oct_from<3>(m);
        const F2 y0 = oct_from<0>(n), y1 = oct_from<1>(n), y2 = oct_from<2>(n), y3 = oct_from<3>(n);
        tab_store(tab, 1, AffP<F2>{x0, y0, false}); tab_store(tab, 2, AffP<F2>{x1, y1, false});
        tab_store(tab, 4, AffP<F2>{x2, y2, false}); tab_store(tab, 8, AffP<F2>{x3, y3, false});
    }
    (void)s;
}

// scalar_mul29_gls with the table and the loop on the octet
__device__ __forceinline__ void scalar_mul29_gls_oct(JacP<F2> &acc, const AffP<F2> &base, const uint32_t k[8], int32_t *tab, int q) {
    GlsSplit s;
    gls_split(s, k);
    jac_set_inf(acc);
    int top = 95;
    while (top >= 0 && !(((s.k[0][top >> 5] | s.k[1][top >> 5] | s.k[2][top >> 5] | s.k[3][top >> 5]) >> (top & 31)) & 1)) top--;
    if (base.inf || top < 0) return;
    AffP<F2> P[4];
    P[0] = AffP<F2>{base.x, s.neg[0] ? f2_neg(base.y) : base.y, false};
    {
        // psi^i(base), i = 1..3: six products in two rounds
        const F2 cx = f2_conj(base.x), cy = f2_conj(base.y);
        const F2 m = oct_round(cx, gamma29(1, 2), cy, gamma29(1, 3), base.x, gamma29(2, 2), base.y, gamma29(2, 3), q);
        const F2 n = oct_round(cx, gamma29(3, 2), cy, gamma29(3, 3), f2_one(), f2_one(), f2_one(), f2_one(), q);
        const F2 x1 = oct_from<0>(m), y1 = oct_from<1>(m), x2 = oct_from<2>(m), y2 = oct_from<3>(m), x3 = oct_from<0>(n), y3 = oct_from<1>(n);
        P[1] = AffP<F2>{x1, s.neg[1] ? f2_neg(y1) : y1, false};
        P[2] = AffP<F2>{x2, s.neg[2] ? f2_neg(y2) : y2, false};
        P[3] = AffP<F2>{x3, s.neg[3] ? f2_neg(y3) : y3, false};
    }
    F2 W;
    gls_table29_oct(tab, W, P, q);
    for (int i = top; i >= 0; i--) {
        const int w = i >> 5, b = i & 31;
        const int idx = (int)((s.k[0][w] >> b) & 1) | (int)(((s.k[1][w] >> b) & 1) << 1) | (int)(((s.k[2][w] >> b) & 1) << 2) | (int)(((s.k[3][w] >> b) & 1) << 3);
        AffP<F2> t;
        tab_load(tab, idx, t);
        jac_dbl_oct(acc, q);
        if (idx) jac_add_mixed_oct(acc, t, q);
    }
    if (!acc.inf) acc.z = oct_mul(acc.z, W, q);
}

}  // namespace gpbc
#endif
