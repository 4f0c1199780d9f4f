// G1 (y^2 = x^3 + 3 over Fp) and G2 (twist y^2 = x^3 + 3/(9+i) over Fp2) scalar multiplication over the
// 29-bit-limb field, one point per lane.  Replaces gnark-crypto's G1Affine/G2Affine.ScalarMultiplication(Base) as
// called at signature/bls01_signature/bls_signature.go:45,63, cpabe/bsw07/bsw07_cpabe.go:69-160,
// bibe/afp25_bibe/afp25_bibe_utils.go:48,51.  The affine result is canonical, so the algorithm is free
// (gnark: GLV + Jacobian); here: Jacobian coordinates, a = 0 doubling, mixed addition, left-to-right binary.
#ifndef GPBC_CURVE29_CUH
#define GPBC_CURVE29_CUH
#include "tower29.cuh"

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
    F y3 = g_norm(g_sub(g_mul(E, g_norm(g_sub(S, x3))), C8));
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
    F H = g_norm(g_sub(U2, p.x));
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
    F y3 = g_norm(g_sub(g_mul(rr, g_norm(g_sub(V, x3))), g_dbl(g_mul(p.y, J))));
    F z3 = g_norm(g_sub(g_sub(g_sqr(g_norm(g_add(p.z, H))), Z1Z1), HH));
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

// [k]base, k = 256-bit little-endian plain integer (8 x u32). Left-to-right binary double-and-add.
template <class F> GPBC_INLINE void scalar_mul29(AffP<F> &out, const AffP<F> &base, const uint32_t k[8]) {
    JacP<F> acc;
    jac_set_inf(acc);
    int top = 255;
    while (top >= 0 && !((k[top >> 5] >> (top & 31)) & 1)) top--;
    for (int i = top; i >= 0; i--) {
        jac_dbl(acc, acc);
        if ((k[i >> 5] >> (i & 31)) & 1) jac_add_mixed(acc, acc, base);
    }
    jac_to_affine(out, acc);
}

}  // namespace gpbc
#endif
