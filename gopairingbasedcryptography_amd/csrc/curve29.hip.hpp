// G1 (y^2 = x^3 + 3 over Fp) and G2 (twist y^2 = x^3 + 3/(9+i) over Fp2) scalar multiplication over the
// 29-bit-limb field, one point per lane.  Replaces gnark-crypto's G1Affine/G2Affine.ScalarMultiplication(Base) as
// called at signature/bls01_signature/bls_signature.go:45,63, cpabe/bsw07/bsw07_cpabe.go:69-160,
// bibe/afp25_bibe/afp25_bibe_utils.go:48,51.  The affine result is canonical, so the algorithm is free
// (gnark: GLV + Jacobian); here: Jacobian coordinates, a = 0 doubling, mixed addition, endomorphism splits (GLV for G1,
// four-dimensional GLS for G2) with fixed joint windows over a common-Z table, safegcd inversion for the affine result.
#ifndef GPBC_CURVE29_HIP_HPP
#define GPBC_CURVE29_HIP_HPP
#include "tower29.hip.hpp"

namespace gpbc {

// field-generic wrappers; every function returns an N-class value unless noted
GPBC_INLINE Fe g_add(const Fe &a, const Fe &b) { return fe_add(a, b); }
GPBC_INLINE Fe g_sub(const Fe &a, const Fe &b) { return fe_sub(a, b); }
GPBC_INLINE Fe g_dbl(const Fe &a) { return fe_dbl(a); }
GPBC_INLINE Fe g_norm(const Fe &a) { return fe_norm(a); }
GPBC_INLINE Fe g_mul8n(const Fe &a) { return fe_mul8_norm(a); }
GPBC_INLINE Fe g_mul(const Fe &a, const Fe &b) { return fe_mul(a, b); }
GPBC_INLINE Fe g_sqr(const Fe &a) { return fe_sqr(a); }
GPBC_INLINE Fe g_inv(const Fe &a) { return fe_inv(a); }
GPBC_INLINE bool g_is_zero(const Fe &a) { return fe_is_zero(a); }
GPBC_INLINE void g_set_one(Fe &a) { a = fe_one(); }
GPBC_INLINE void g_set_zero(Fe &a) { a = fe_zero(); }
GPBC_INLINE F2 g_add(const F2 &a, const F2 &b) { return f2_add(a, b); }
GPBC_INLINE F2 g_sub(const F2 &a, const F2 &b) { return f2_sub(a, b); }
GPBC_INLINE F2 g_dbl(const F2 &a) { return f2_dbl(a); }
GPBC_INLINE F2 g_norm(const F2 &a) { return f2_norm(a); }
GPBC_INLINE F2 g_mul8n(const F2 &a) { return f2_mul8_norm(a); }
GPBC_INLINE F2 g_mul(const F2 &a, const F2 &b) { return f2_mul(a, b); }
GPBC_INLINE F2 g_sqr(const F2 &a) { return f2_sqr(a); }
GPBC_INLINE F2 g_inv(const F2 &a) { return f2_inv(a); }
GPBC_INLINE bool g_is_zero(const F2 &a) { return f2_is_zero(a); }
GPBC_INLINE void g_set_one(F2 &a) { a = f2_one(); }
GPBC_INLINE void g_set_zero(F2 &a) { a = f2_zero(); }

template <class F> struct AffP { F x, y; bool inf; };     // inf decided on the raw input bytes ((0,0) in gnark)
template <class F> struct JacP { F x, y, z; bool inf; };

template <class F> GPBC_INLINE void jac_set_inf(JacP<F> &p) { g_set_one(p.x); g_set_one(p.y); g_set_zero(p.z); p.inf = true; }

// a = 0 doubling.  With lazy reduction every product resets the worst-case value bound to ~p, while sums and small
// multiples of results inflate it; so constants are folded into product INPUTS (S = x * 4y^2, 8y^4 = y^2 * 8y^2)
// instead of scaling outputs, which keeps the bound from growing from doubling to doubling.  In/out N-class.
template <class F> GPBC_INLINE void jac_dbl(JacP<F> &r, const JacP<F> &p) {
    if (p.inf) { r = p; return; }
    F A = g_sqr(p.x), B = g_sqr(p.y);
    F B4 = g_norm(g_dbl(g_dbl(B)));
    F S = g_mul(p.x, B4);
    F C8 = g_mul(B, g_norm(g_dbl(B4)));
    F E = g_norm(g_add(g_dbl(A), A));
    F FF = g_sqr(E);
    F x3 = g_norm(g_sub(FF, g_dbl(S)));
    F y3 = g_sub(g_mul(E, g_sub(S, x3)), C8);             // differences of normalised values stay within +-2^29: no normalisation (interval harness)
    F z3 = g_mul(p.y, g_norm(g_dbl(p.z)));
    r.x = x3; r.y = y3; r.z = z3; r.inf = false;
}
// madd-2007-bl with the exceptional cases handled (any 256-bit scalar must give [s mod r]P)
template <class F> GPBC_INLINE void jac_add_mixed(JacP<F> &r, const JacP<F> &p, const AffP<F> &q) {
    if (q.inf) { r = p; return; }
    if (p.inf) { r.x = q.x; r.y = q.y; g_set_one(r.z); r.inf = false; return; }
    F Z1Z1 = g_sqr(p.z);
    F U2 = g_mul(q.x, Z1Z1);
    F S2 = g_mul(g_mul(q.y, p.z), Z1Z1);
    F H = g_sub(U2, p.x);
    F rr = g_norm(g_sub(S2, p.y));
    if (g_is_zero(H)) {
        if (g_is_zero(rr)) { jac_dbl(r, p); return; }
        jac_set_inf(r);
        return;
    }
    rr = g_norm(g_dbl(rr));
    F HH = g_sqr(H);
    F I = g_norm(g_dbl(g_dbl(HH)));
    F J = g_mul(H, I);
    F V = g_mul(p.x, I);
    F x3 = g_norm(g_sub(g_sub(g_sqr(rr), J), g_dbl(V)));
    F y3 = g_norm(g_sub(g_mul(rr, g_sub(V, x3)), g_dbl(g_mul(p.y, J))));
    F z3 = g_norm(g_sub(g_sub(g_sqr(g_norm(g_add(p.z, H))), Z1Z1), HH));
    r.x = x3; r.y = y3; r.z = z3; r.inf = false;
}
// add-2007-bl with the exceptional cases; in/out N-class
template <class F> GPBC_INLINE void jac_add(JacP<F> &r, const JacP<F> &p, const JacP<F> &q) {
    if (q.inf) { r = p; return; }
    if (p.inf) { r = q; return; }
    F Z1Z1 = g_sqr(p.z), Z2Z2 = g_sqr(q.z);
    F U1 = g_mul(p.x, Z2Z2), U2 = g_mul(q.x, Z1Z1);
    F S1 = g_mul(g_mul(p.y, q.z), Z2Z2), S2 = g_mul(g_mul(q.y, p.z), Z1Z1);
    F H = g_sub(U2, U1);
    F rr = g_sub(S2, S1);
    if (g_is_zero(H)) {
        if (g_is_zero(rr)) { jac_dbl(r, p); return; }
        jac_set_inf(r);
        return;
    }
    rr = g_norm(g_dbl(rr));
    F HH = g_sqr(H);
    F I = g_norm(g_dbl(g_dbl(HH)));
    F J = g_mul(H, I);
    F V = g_mul(U1, I);
    F x3 = g_norm(g_sub(g_sub(g_sqr(rr), J), g_dbl(V)));
    F y3 = g_norm(g_sub(g_mul(rr, g_sub(V, x3)), g_dbl(g_mul(S1, J))));
    F z3 = g_mul(g_norm(g_sub(g_sub(g_sqr(g_norm(g_add(p.z, q.z))), Z1Z1), Z2Z2)), H);
    r.x = x3; r.y = y3; r.z = z3; r.inf = false;
}
template <class F> GPBC_INLINE void jac_to_affine(AffP<F> &r, const JacP<F> &p) {
    if (p.inf) { g_set_zero(r.x); g_set_zero(r.y); r.inf = true; return; }
    F zi = g_inv(p.z);
    F zi2 = g_sqr(zi);
    r.x = g_mul(p.x, zi2);
    r.y = g_mul(p.y, g_mul(zi2, zi));
    r.inf = false;
}

// ------------------------------------------------------------------------------------------- GLV scalar decomposition
// BN254 has j = 0: phi(x, y) = (beta x, y) is an endomorphism acting as [lambda] on the order-r groups (G1 with
// beta1, the twist subgroup G2 with beta2 = beta1^2; tools/gen_constants.py derives and checks the constants).  A scalar
// splits as k = k1 + k2*lambda (mod r) with |k1|, |k2| < 2^130, halving the doublings of [k]P = [k1]P + [k2]phi(P).
// (gnark-crypto's ScalarMultiplication uses the same endomorphism; the affine result is the unique point [k mod r]P.)
//   c1 = (k * g1) >> 256,  c2 = (k * g2) >> 256     (g1 ~ 2^256 b2 / r, g2 ~ 2^256 |b1| / r; any rounding error only
//   k1 = k - c1*a1 - c2*a2,  k2 = c1*|b1| - c2*b2     widens k1, k2 by a few bits: the identity holds exactly)
struct GlvSplit { uint32_t k1[5], k2[5]; bool neg1, neg2; };

template <int NA, int NB, int NO> GPBC_INLINE void mp_mul(uint32_t (&out)[NO], const uint32_t (&a)[NA], const uint32_t (&b)[NB]) {
    // out = low NO limbs of a * b (column-wise; a column sum of up to 8 32x32 products is kept as two 64-bit halves)
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < NO; k++) {
        uint64_t lo = carry, hi = 0;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int j = k - i;
            if (j < 0 || j >= NB) continue;
            uint64_t p = (uint64_t)a[i] * (uint64_t)b[j];
            lo += p & 0xffffffffu;
            hi += p >> 32;
        }
        out[k] = (uint32_t)lo;
        carry = (lo >> 32) + hi;
    }
}
template <int N> GPBC_INLINE void mp_sub(uint32_t (&r)[N], const uint32_t (&a)[N], const uint32_t (&b)[N]) {
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t d = (uint64_t)a[i] - b[i] - borrow;
        r[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
}
GPBC_INLINE void glv_split(GlvSplit &o, const uint32_t kin[8]) {
    constexpr uint32_t R32[8] = GLV_R32, G1[3] = GLV_G1, G2[5] = GLV_G2, A1[2] = GLV_A1, A2[4] = GLV_A2, B1[4] = GLV_B1ABS, B2[2] = GLV_B2;
    uint32_t k[8];
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = kin[i];
    for (int rep = 0; rep < 6; rep++) {            // k < 2^256 < 6r: reduce to [0, r)
        uint32_t d[8];
        uint64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { uint64_t t = (uint64_t)k[i] - R32[i] - borrow; d[i] = (uint32_t)t; borrow = (t >> 32) & 1; }
#pragma unroll
        for (int i = 0; i < 8; i++) k[i] = borrow ? k[i] : d[i];
    }
    uint32_t p1[11], p2[13];
    mp_mul<8, 3, 11>(p1, k, G1);
    mp_mul<8, 5, 13>(p2, k, G2);
    uint32_t c1[3] = {p1[8], p1[9], p1[10]}, c2[5] = {p2[8], p2[9], p2[10], p2[11], p2[12]};
    uint32_t t1[5], t2[5], t3[5], t4[5], k5[5] = {k[0], k[1], k[2], k[3], k[4]}, r1[5], r2[5];
    mp_mul<3, 2, 5>(t1, c1, A1);
    mp_mul<5, 4, 5>(t2, c2, A2);
    mp_sub<5>(r1, k5, t1);
    mp_sub<5>(r1, r1, t2);                           // k1 mod 2^160 (two's complement)
    mp_mul<3, 4, 5>(t3, c1, B1);
    mp_mul<5, 2, 5>(t4, c2, B2);
    mp_sub<5>(r2, t3, t4);                           // k2 mod 2^160
    o.neg1 = (r1[4] >> 31) != 0;
    o.neg2 = (r2[4] >> 31) != 0;
    uint32_t zero[5] = {0, 0, 0, 0, 0}, n1[5], n2[5];
    mp_sub<5>(n1, zero, r1);
    mp_sub<5>(n2, zero, r2);
#pragma unroll
    for (int i = 0; i < 5; i++) { o.k1[i] = o.neg1 ? n1[i] : r1[i]; o.k2[i] = o.neg2 ? n2[i] : r2[i]; }
}

GPBC_INLINE Fe glv_phi_x(const Fe &x) { constexpr int32_t B[NL] = GLV_BETA_G1; return fe_mul(x, fe_const(B)); }
GPBC_INLINE F2 glv_phi_x(const F2 &x) { constexpr int32_t B[NL] = GLV_BETA_G2; return f2_mul_fe(x, fe_const(B)); }
GPBC_INLINE Fe g_neg(const Fe &a) { return fe_neg(a); }
GPBC_INLINE F2 g_neg(const F2 &a) { return f2_neg(a); }

template <class F> GPBC_INLINE F g_sel(bool c, const F &a, const F &b);
template <> GPBC_INLINE Fe g_sel<Fe>(bool c, const Fe &a, const Fe &b) {
    Fe r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = c ? a.v[i] : b.v[i];
#ifdef GPBC_BOUNDS
    for (int i = 0; i < NL; i++) { r.lo[i] = a.lo[i] < b.lo[i] ? a.lo[i] : b.lo[i]; r.hi[i] = a.hi[i] > b.hi[i] ? a.hi[i] : b.hi[i]; }
    r.vb = a.vb > b.vb ? a.vb : b.vb;
#endif
    return r;
}
template <> GPBC_INLINE F2 g_sel<F2>(bool c, const F2 &a, const F2 &b) { return F2{g_sel<Fe>(c, a.a0, b.a0), g_sel<Fe>(c, a.a1, b.a1)}; }

// [k]base for any 256-bit k (Jacobian result).  GLV split k = k1 + k2*lambda, then ONE joint loop over the ~130-bit halves
// in FIXED 2-bit windows: two doublings and one mixed addition of T[d1][d2] = d1*P1 + d2*P2 (P1 = +-P, P2 = +-phi(P)) per
// step.  Why fixed windows: in a 64-lane wave a per-lane `if (digit) add` costs the addition whenever ANY lane has a
// non-zero digit, i.e. practically always, so sparse recodings (NAF, JSF) buy nothing; what counts is the number of
// addition SLOTS, and aligned 2-bit windows halve them (65 instead of 130).
// The 15 table points are built in Jacobian coordinates (2 doublings, 8 mixed and 1 general addition; 2P2 and 3P2 are
// phi(2P1), phi(3P1) with the sign s1*s2) and then brought to ONE common Z = W = product of their 11 distinct Z's without
// any inversion:  (X, Y, Z) ~ (X l^2, Y l^3, W),  l = W / Z  from prefix / suffix products.  Points sharing a Z are affine
// points of the isomorphic curve y^2 = x^3 + b W^6; the a = 0 doubling and the mixed addition never touch b, so the
// whole loop runs there and the result (X, Y, Z') maps back as (X, Y, Z' W).
template <class F> GPBC_INLINE JacP<F> jac_phi(const JacP<F> &p, bool flip) {
    return JacP<F>{glv_phi_x(p.x), flip ? g_neg(p.y) : p.y, p.z, p.inf};
}
// The table lives in a caller-provided memory block, NOT in a private array: private (scratch) memory is dword-swizzled
// across the 64 lanes, so a per-lane dynamic index makes every one of the 18 / 36 dwords of an entry touch a different
// cache line (measured: 18.6 GB of HBM-side traffic per 262 144 G1 multiplications, 4.7 TB/s).  Here an entry is one
// contiguous, 128-byte-aligned row per lane — 20 dwords used of 32 (G1), 36 of 64 (G2) — read with 128-bit loads.
struct alignas(16) TabQuad { int32_t a, b, c, d; };
template <class F> struct TabLayout;
template <> struct TabLayout<Fe> { static constexpr int ENTRY_DWORDS = 32; };
template <> struct TabLayout<F2> { static constexpr int ENTRY_DWORDS = 64; };
template <class F> constexpr int glv_table_dwords() { return 16 * TabLayout<F>::ENTRY_DWORDS; }
GPBC_INLINE void tab_put(TabQuad *q, const Fe &x, const Fe &y) {
    q[0] = TabQuad{x.v[0], x.v[1], x.v[2], x.v[3]};
    q[1] = TabQuad{x.v[4], x.v[5], x.v[6], x.v[7]};
    q[2] = TabQuad{x.v[8], y.v[0], y.v[1], y.v[2]};
    q[3] = TabQuad{y.v[3], y.v[4], y.v[5], y.v[6]};
    q[4] = TabQuad{y.v[7], y.v[8], 0, 0};
}
GPBC_INLINE void tab_get(const TabQuad *q, Fe &x, Fe &y) {
    TabQuad t0 = q[0], t1 = q[1], t2 = q[2], t3 = q[3], t4 = q[4];
    x.v[0] = t0.a; x.v[1] = t0.b; x.v[2] = t0.c; x.v[3] = t0.d; x.v[4] = t1.a; x.v[5] = t1.b; x.v[6] = t1.c; x.v[7] = t1.d; x.v[8] = t2.a;
    y.v[0] = t2.b; y.v[1] = t2.c; y.v[2] = t2.d; y.v[3] = t3.a; y.v[4] = t3.b; y.v[5] = t3.c; y.v[6] = t3.d; y.v[7] = t4.a; y.v[8] = t4.b;
    GPBC_B(set_class_n(x, 1.5); set_class_n(y, 1.5);)      // entries are products: value < (1 + eps) p, limbs masked
}
GPBC_INLINE void tab_store(int32_t *mem, int idx, const AffP<Fe> &p) { tab_put(reinterpret_cast<TabQuad *>(mem + idx * 32), p.x, p.y); }
GPBC_INLINE void tab_store(int32_t *mem, int idx, const AffP<F2> &p) {
    TabQuad *q = reinterpret_cast<TabQuad *>(mem + idx * 64);
    tab_put(q, p.x.a0, p.x.a1);
    tab_put(q + 5, p.y.a0, p.y.a1);
}
GPBC_INLINE void tab_load(const int32_t *mem, int idx, AffP<Fe> &p) { tab_get(reinterpret_cast<const TabQuad *>(mem + idx * 32), p.x, p.y); p.inf = false; }
GPBC_INLINE void tab_load(const int32_t *mem, int idx, AffP<F2> &p) {
    const TabQuad *q = reinterpret_cast<const TabQuad *>(mem + idx * 64);
    tab_get(q, p.x.a0, p.x.a1);
    tab_get(q + 5, p.y.a0, p.y.a1);
    p.inf = false;
}
template <class F> GPBC_NOINLINE void glv_table29(int32_t *tab, F &W, const AffP<F> &p1, const AffP<F> &p2, bool flip) {
    F one;
    g_set_one(one);
    // J: 0 D1=2P1, 1 T1=3P1, 2 (1,1), 3 (2,2), 4 (3,3), 5 (1,2), 6 (2,1), 7 (1,3), 8 (3,1), 9 (2,3), 10 (3,2)
    JacP<F> J[11];
    JacP<F> p1j{p1.x, p1.y, one, false};
    jac_dbl(J[0], p1j);
    jac_add_mixed(J[1], J[0], p1);
    jac_add_mixed(J[2], p1j, p2);
    jac_dbl(J[3], J[2]);
    jac_add(J[4], J[3], J[2]);
    JacP<F> d2 = jac_phi(J[0], flip), t2 = jac_phi(J[1], flip);        // 2P2, 3P2 (same Z as 2P1, 3P1)
    jac_add_mixed(J[5], d2, p1);
    jac_add_mixed(J[6], J[0], p2);
    jac_add_mixed(J[7], t2, p1);
    jac_add_mixed(J[8], J[1], p2);
    jac_add_mixed(J[9], J[7], p1);
    jac_add_mixed(J[10], J[8], p2);
    // l[i] = product of all Z's except J[i].z
    F pre[11], l[11];
    F run = one;
    for (int i = 0; i < 11; i++) { pre[i] = run; run = g_mul(run, J[i].z); }
    W = run;
    run = one;
    for (int i = 10; i >= 0; i--) { l[i] = g_mul(pre[i], run); run = g_mul(run, J[i].z); }
    // index = 4 * d1 + d2
    static constexpr int IDX[11] = {8, 12, 5, 10, 15, 6, 9, 7, 13, 11, 14};
    F l2d = one, l3d = one, l2t = one, l3t = one;
    for (int i = 0; i < 11; i++) {
        F l2 = g_sqr(l[i]), l3 = g_mul(l2, l[i]);
        tab_store(tab, IDX[i], AffP<F>{g_mul(J[i].x, l2), g_mul(J[i].y, l3), false});
        if (i == 0) { l2d = l2; l3d = l3; }
        if (i == 1) { l2t = l2; l3t = l3; }
    }
    tab_store(tab, 2, AffP<F>{g_mul(d2.x, l2d), g_mul(d2.y, l3d), false});
    tab_store(tab, 3, AffP<F>{g_mul(t2.x, l2t), g_mul(t2.y, l3t), false});
    F w2 = g_sqr(W), w3 = g_mul(w2, W);
    tab_store(tab, 4, AffP<F>{g_mul(p1.x, w2), g_mul(p1.y, w3), false});
    tab_store(tab, 1, AffP<F>{g_mul(p2.x, w2), g_mul(p2.y, w3), false});
}
// tab: this lane's block of glv_table_dwords<F>() int32 (16-byte aligned)
template <class F> GPBC_INLINE void scalar_mul29_jac(JacP<F> &acc, const AffP<F> &base, const uint32_t k[8], int32_t *tab) {
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
        tab_load(tab, idx, t);                           // requested before the doublings: its HBM latency passes behind them
                                                         // (entry 0 is never used; its row exists)
        jac_dbl(acc, acc);
        jac_dbl(acc, acc);
        if (idx) jac_add_mixed(acc, acc, t);
    }
    if (!acc.inf) acc.z = g_mul(acc.z, W);               // back from the curve scaled by W
}
// ------------------------------------------------------------------------------------------- 4-dimensional GLS for G2
// On the order-r subgroup of the twist the endomorphism psi (untwist, Frobenius, twist) acts as [mu], mu = 6u^2, so
// k = k0 + k1 mu + k2 mu^2 + k3 mu^3 (mod r) with |k_i| < 2^66 (Galbraith-Scott lattice for BN curves, Babai rounding) and
// [k]Q = sum [k_i] psi^i(Q): ONE joint loop of ~66 doublings and ~66 mixed additions of T[b0 b1 b2 b3] = sum b_i P_i,
// P_i = +-psi^i(Q), instead of the 130 doublings + 65 additions of the two-dimensional GLV loop.  All arithmetic on the
// small results is done modulo 2^96 (the large terms c_j * B_ji cancel).
struct GlsSplit { uint32_t k[4][3]; bool neg[4]; };
GPBC_INLINE void gls_split(GlsSplit &o, const uint32_t kin[8]) {
    constexpr uint32_t R32[8] = GLV_R32, G[4][7] = GLS_G, BM[4][4][3] = GLS_BMAG;
    constexpr int GN[4] = GLS_GNEG, BN[4][4] = GLS_BNEG;
    uint32_t k[8];
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = kin[i];
    for (int rep = 0; rep < 6; rep++) {            // k < 2^256 < 6r: reduce to [0, r)
        uint32_t d[8];
        uint64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { uint64_t t = (uint64_t)k[i] - R32[i] - borrow; d[i] = (uint32_t)t; borrow = (t >> 32) & 1; }
#pragma unroll
        for (int i = 0; i < 8; i++) k[i] = borrow ? k[i] : d[i];
    }
    uint32_t c[4][3];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t p[11];
        mp_mul<8, 7, 11>(p, k, G[j]);              // c_j = (k * g_j) >> 256, needed modulo 2^96 only
        c[j][0] = p[8]; c[j][1] = p[9]; c[j][2] = p[10];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t acc[3] = {i == 0 ? k[0] : 0u, i == 0 ? k[1] : 0u, i == 0 ? k[2] : 0u};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t t[3], r[3];
            mp_mul<3, 3, 3>(t, c[j], BM[j][i]);
            const bool plus = (GN[j] != 0) != (BN[j][i] != 0);     // acc -= sign(c_j) sign(B_ji) |c_j| |B_ji|
            if (plus) {
                uint64_t cy = 0;
#pragma unroll
                for (int w = 0; w < 3; w++) { uint64_t v = (uint64_t)acc[w] + t[w] + cy; r[w] = (uint32_t)v; cy = v >> 32; }
            } else mp_sub<3>(r, acc, t);
            acc[0] = r[0]; acc[1] = r[1]; acc[2] = r[2];
        }
        o.neg[i] = (acc[2] >> 31) != 0;
        if (o.neg[i]) {
            uint32_t z[3] = {0, 0, 0}, r[3];
            mp_sub<3>(r, z, acc);
            acc[0] = r[0]; acc[1] = r[1]; acc[2] = r[2];
        }
        o.k[i][0] = acc[0]; o.k[i][1] = acc[1]; o.k[i][2] = acc[2];
    }
}
// table index = b0 + 2 b1 + 4 b2 + 8 b3; entries brought to one common Z like glv_table29 (rows of the same 128-byte layout)
GPBC_NOINLINE void gls_table29(int32_t *tab, F2 &W, const AffP<F2> (&P)[4]) {
    F2 one = f2_one();
    // J[e], e = 0..10: the 11 entries with two or more points, in increasing index order
    static constexpr int IDX[11] = {3, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15};
    JacP<F2> J[11];
    auto single = [&](int i) { return JacP<F2>{P[i].x, P[i].y, one, false}; };
    jac_add_mixed(J[0], single(0), P[1]);           // 3  = P0+P1
    jac_add_mixed(J[1], single(0), P[2]);           // 5  = P0+P2
    jac_add_mixed(J[2], single(1), P[2]);           // 6  = P1+P2
    jac_add_mixed(J[3], J[0], P[2]);                // 7  = P0+P1+P2
    jac_add_mixed(J[4], single(0), P[3]);           // 9  = P0+P3
    jac_add_mixed(J[5], single(1), P[3]);           // 10 = P1+P3
    jac_add_mixed(J[6], J[0], P[3]);                // 11 = P0+P1+P3
    jac_add_mixed(J[7], single(2), P[3]);           // 12 = P2+P3
    jac_add_mixed(J[8], J[1], P[3]);                // 13 = P0+P2+P3
    jac_add_mixed(J[9], J[2], P[3]);                // 14 = P1+P2+P3
    jac_add_mixed(J[10], J[3], P[3]);               // 15 = all four
    F2 pre[11], l[11];
    F2 run = one;
    for (int i = 0; i < 11; i++) { pre[i] = run; run = f2_mul(run, J[i].z); }
    W = run;
    run = one;
    for (int i = 10; i >= 0; i--) { l[i] = f2_mul(pre[i], run); run = f2_mul(run, J[i].z); }
    for (int i = 0; i < 11; i++) {
        F2 l2 = f2_sqr(l[i]), l3 = f2_mul(l2, l[i]);
        tab_store(tab, IDX[i], AffP<F2>{f2_mul(J[i].x, l2), f2_mul(J[i].y, l3), false});
    }
    F2 w2 = f2_sqr(W), w3 = f2_mul(w2, W);
    for (int i = 0; i < 4; i++) tab_store(tab, 1 << i, AffP<F2>{f2_mul(P[i].x, w2), f2_mul(P[i].y, w3), false});
}
GPBC_INLINE void scalar_mul29_gls(JacP<F2> &acc, const AffP<F2> &base, const uint32_t k[8], int32_t *tab) {
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
        tab_load(tab, idx, t);                           // before the doubling, as in the GLV loop
        jac_dbl(acc, acc);
        if (idx) jac_add_mixed(acc, acc, t);
    }
    if (!acc.inf) acc.z = f2_mul(acc.z, W);
}
// field-generic front: G1 takes the two-dimensional GLV loop, G2 the four-dimensional GLS loop
GPBC_INLINE void scalar_mul29_best(JacP<Fe> &acc, const AffP<Fe> &base, const uint32_t k[8], int32_t *tab) { scalar_mul29_jac<Fe>(acc, base, k, tab); }
GPBC_INLINE void scalar_mul29_best(JacP<F2> &acc, const AffP<F2> &base, const uint32_t k[8], int32_t *tab) { scalar_mul29_gls(acc, base, k, tab); }

template <class F> GPBC_INLINE void scalar_mul29(AffP<F> &out, const AffP<F> &base, const uint32_t k[8], int32_t *tab) {
    JacP<F> acc;
    scalar_mul29_jac(acc, base, k, tab);
    jac_to_affine(out, acc);
}

// K Jacobian points -> affine with ONE field inversion (Montgomery's trick): the inversion (~380 Fp products by Fermat)
// is a sixth of a G1 scalar multiplication, so a lane that owns K points amortises it K-fold.
template <class F, int K> GPBC_INLINE void jac_to_affine_batch(AffP<F> (&out)[K], const JacP<F> (&p)[K]) {
    F one, pre[K];
    g_set_one(one);
    F run = one;
    for (int j = 0; j < K; j++) {
        pre[j] = run;                                    // product of the z's before j
        run = g_mul(run, p[j].inf ? one : p[j].z);
    }
    F inv = g_inv(run);
    for (int j = K - 1; j >= 0; j--) {
        F zi = g_mul(inv, pre[j]);
        inv = g_mul(inv, p[j].inf ? one : p[j].z);
        if (p[j].inf) { g_set_zero(out[j].x); g_set_zero(out[j].y); out[j].inf = true; continue; }
        F zi2 = g_sqr(zi);
        out[j].x = g_mul(p[j].x, zi2);
        out[j].y = g_mul(p[j].y, g_mul(zi2, zi));
        out[j].inf = false;
    }
}

}  // namespace gpbc
#endif
