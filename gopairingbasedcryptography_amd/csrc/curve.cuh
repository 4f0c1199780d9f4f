// G1 (y^2 = x^3 + 3 over Fp) and G2 (twist y^2 = x^3 + 3/(9+i) over Fp2) scalar multiplication for gfx950,
// one point per lane.  Replaces gnark-crypto's G1Affine/G2Affine.ScalarMultiplication(Base) as called at
// signature/bls01_signature/bls_signature.go:45,63, cpabe/bsw07/bsw07_cpabe.go:69-160,
// bibe/afp25_bibe/afp25_bibe_utils.go:48,51.  The affine result is canonical, so the algorithm is free
// (gnark: GLV + Jacobian); here: Jacobian coordinates, a = 0 doubling, mixed addition, left-to-right
// binary double-and-add over the 256-bit scalar.
#ifndef GPBC_CURVE_CUH
#define GPBC_CURVE_CUH
#include "tower.cuh"

namespace gpbc {

// field-generic wrappers -----------------------------------------------------------------------
__device__ __forceinline__ Fp f_add(const Fp &a, const Fp &b) { return fp_add(a, b); }
__device__ __forceinline__ Fp f_sub(const Fp &a, const Fp &b) { return fp_sub(a, b); }
__device__ __forceinline__ Fp f_dbl(const Fp &a) { return fp_dbl(a); }
__device__ __forceinline__ Fp f_neg(const Fp &a) { return fp_neg(a); }
__device__ __forceinline__ Fp f_mul(const Fp &a, const Fp &b) { return fp_mul(a, b); }
__device__ __forceinline__ Fp f_sqr(const Fp &a) { return fp_sqr(a); }
__device__ __forceinline__ Fp f_inv(const Fp &a) { return fp_inv(a); }
__device__ __forceinline__ bool f_is_zero(const Fp &a) { return fp_is_zero(a); }
__device__ __forceinline__ void f_set_one(Fp &a) { a = fp_one(); }
__device__ __forceinline__ void f_set_zero(Fp &a) { a = fp_zero(); }
__device__ __forceinline__ Fp2 f_add(const Fp2 &a, const Fp2 &b) { return fp2_add(a, b); }
__device__ __forceinline__ Fp2 f_sub(const Fp2 &a, const Fp2 &b) { return fp2_sub(a, b); }
__device__ __forceinline__ Fp2 f_dbl(const Fp2 &a) { return fp2_dbl(a); }
__device__ __forceinline__ Fp2 f_neg(const Fp2 &a) { return fp2_neg(a); }
__device__ __forceinline__ Fp2 f_mul(const Fp2 &a, const Fp2 &b) { return fp2_mul(a, b); }
__device__ __forceinline__ Fp2 f_sqr(const Fp2 &a) { return fp2_sqr(a); }
__device__ __forceinline__ Fp2 f_inv(const Fp2 &a) { return fp2_inv(a); }
__device__ __forceinline__ bool f_is_zero(const Fp2 &a) { return fp2_is_zero(a); }
__device__ __forceinline__ void f_set_one(Fp2 &a) { a = fp2_one(); }
__device__ __forceinline__ void f_set_zero(Fp2 &a) { a = fp2_zero(); }

template <class F> struct Aff { F x, y; };          // (0,0) = infinity (gnark convention)
template <class F> struct Jac { F x, y, z; };        // z = 0 = infinity

template <class F> __device__ __forceinline__ bool aff_is_inf(const Aff<F> &p) { return f_is_zero(p.x) && f_is_zero(p.y); }
template <class F> __device__ __forceinline__ void jac_set_inf(Jac<F> &p) { f_set_one(p.x); f_set_one(p.y); f_set_zero(p.z); }

// dbl-2009-l (a = 0)
template <class F> __device__ __noinline__ void jac_dbl(Jac<F> &r, const Jac<F> &p) {
    if (f_is_zero(p.z)) { r = p; return; }
    F A = f_sqr(p.x), B = f_sqr(p.y), C = f_sqr(B);
    F D = f_dbl(f_sub(f_sub(f_sqr(f_add(p.x, B)), A), C));
    F E = f_add(f_dbl(A), A);
    F FF = f_sqr(E);
    F x3 = f_sub(FF, f_dbl(D));
    F y3 = f_sub(f_mul(E, f_sub(D, x3)), f_dbl(f_dbl(f_dbl(C))));
    F z3 = f_dbl(f_mul(p.y, p.z));
    r.x = x3; r.y = y3; r.z = z3;
}
// madd-2007-bl with the exceptional cases handled (any 256-bit scalar must give [s mod r]P)
template <class F> __device__ __noinline__ void jac_add_mixed(Jac<F> &r, const Jac<F> &p, const Aff<F> &q) {
    if (aff_is_inf(q)) { r = p; return; }
    if (f_is_zero(p.z)) { r.x = q.x; r.y = q.y; f_set_one(r.z); return; }
    F Z1Z1 = f_sqr(p.z);
    F U2 = f_mul(q.x, Z1Z1);
    F S2 = f_mul(f_mul(q.y, p.z), Z1Z1);
    F H = f_sub(U2, p.x);
    F rr = f_sub(S2, p.y);
    if (f_is_zero(H)) {
        if (f_is_zero(rr)) { jac_dbl(r, p); return; }
        jac_set_inf(r);
        return;
    }
    rr = f_dbl(rr);
    F HH = f_sqr(H);
    F I = f_dbl(f_dbl(HH));
    F J = f_mul(H, I);
    F V = f_mul(p.x, I);
    F x3 = f_sub(f_sub(f_sqr(rr), J), f_dbl(V));
    F y3 = f_sub(f_mul(rr, f_sub(V, x3)), f_dbl(f_mul(p.y, J)));
    F z3 = f_sub(f_sub(f_sqr(f_add(p.z, H)), Z1Z1), HH);
    r.x = x3; r.y = y3; r.z = z3;
}
template <class F> __device__ __noinline__ void jac_to_affine(Aff<F> &r, const Jac<F> &p) {
    if (f_is_zero(p.z)) { f_set_zero(r.x); f_set_zero(r.y); return; }
    F zi = f_inv(p.z);
    F zi2 = f_sqr(zi);
    r.x = f_mul(p.x, zi2);
    r.y = f_mul(p.y, f_mul(zi2, zi));
}

// [k]base, k = 256-bit little-endian plain integer (8 x u32). Left-to-right binary double-and-add.
template <class F> __device__ __noinline__ void scalar_mul(Aff<F> &out, const Aff<F> &base, const u32 k[8]) {
    Jac<F> acc;
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
