// Extension tower over the 29-bit-limb field (fe29.hip.hpp); gnark E2/E6/E12 structure:
//   F2 = Fp[i]/(i^2+1) {a0,a1};  F6 = F2[v]/(v^3-(9+i)) {b0,b1,b2};  F12 = F6[w]/(w^2-v) {c0,c1}
// Convention: every tower function takes operands whose limbs are "N-class" (|limb| <= 2^29 + small) and returns
// N-class results; sums and differences in between run carry-free and are re-normalised (fe_norm) only where a
// product needs it.  F2 products are lazy: two wide column accumulations + two reductions (4 x 81 + 2 x 81 MADs)
// instead of Karatsuba's three full multiplications — MADs cost the same as adds on this machine.
#ifndef GPBC_TOWER29_HIP_HPP
#define GPBC_TOWER29_HIP_HPP
#include "fe29.hip.hpp"

namespace gpbc {

struct F2 { Fe a0, a1; };
struct F6 { F2 b0, b1, b2; };
struct F12 { F6 c0, c1; };

// ------------------------------------------------------------------------------------------- F2
GPBC_INLINE F2 f2_zero() { return F2{fe_zero(), fe_zero()}; }
GPBC_INLINE F2 f2_one() { return F2{fe_one(), fe_zero()}; }
GPBC_INLINE F2 f2_add(const F2 &x, const F2 &y) { return F2{fe_add(x.a0, y.a0), fe_add(x.a1, y.a1)}; }
GPBC_INLINE F2 f2_sub(const F2 &x, const F2 &y) { return F2{fe_sub(x.a0, y.a0), fe_sub(x.a1, y.a1)}; }
GPBC_INLINE F2 f2_dbl(const F2 &x) { return F2{fe_dbl(x.a0), fe_dbl(x.a1)}; }
GPBC_INLINE F2 f2_neg(const F2 &x) { return F2{fe_neg(x.a0), fe_neg(x.a1)}; }
GPBC_INLINE F2 f2_conj(const F2 &x) { return F2{x.a0, fe_neg(x.a1)}; }
GPBC_INLINE F2 f2_norm(const F2 &x) { return F2{fe_norm(x.a0), fe_norm(x.a1)}; }
GPBC_INLINE F2 f2_reduce(const F2 &x) { return F2{fe_reduce(x.a0), fe_reduce(x.a1)}; }
GPBC_INLINE F2 f2_halve(const F2 &x) { return F2{fe_halve(x.a0), fe_halve(x.a1)}; }
GPBC_INLINE bool f2_is_zero(const F2 &x) { return fe_is_zero(x.a0) && fe_is_zero(x.a1); }

// limb-wise selection (v_cndmask): values picked by a flag stay in registers, where `c ? a : b` on the structs makes the
// compiler keep both in private memory and load the chosen one
GPBC_INLINE Fe fe_sel(bool c, const Fe &a, const Fe &b) {
    Fe r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = c ? a.v[i] : b.v[i];
#ifdef GPBC_BOUNDS
    for (int i = 0; i < NL; i++) { r.lo[i] = a.lo[i] < b.lo[i] ? a.lo[i] : b.lo[i]; r.hi[i] = a.hi[i] > b.hi[i] ? a.hi[i] : b.hi[i]; }   // must hold for either lane
    r.vb = a.vb > b.vb ? a.vb : b.vb;
#endif
    return r;
}
GPBC_INLINE F2 f2_sel(bool c, const F2 &a, const F2 &b) { return F2{fe_sel(c, a.a0, b.a0), fe_sel(c, a.a1, b.a1)}; }

// Lazy schoolbook product and complex squaring.  NORM variants first re-normalise their operands (used for the sums
// of the Karatsuba layers above), so that the normalisation code lives inside the leaf instead of at every call site.
template <bool NORM> GPBC_INLINE F2 f2_mul_core(const F2 &xx, const F2 &yy) {
    F2 x = NORM ? f2_norm(xx) : xx, y = NORM ? f2_norm(yy) : yy;
    F2 r;
    fe_mul2_dual(r.a0, r.a1, x.a0, y.a0, fe_neg(x.a1), y.a1, x.a0, y.a1, x.a1, y.a0);
    return r;
}
template <bool NORM> GPBC_INLINE F2 f2_sqr_core(const F2 &xx) {
    F2 x = NORM ? f2_norm(xx) : xx;
    F2 r;
    fe_sqrdiff_mul_dual(r.a0, r.a1, x.a0, x.a1);          // (a0^2 - a1^2, 2 a0 a1) without the operand sums
    return r;
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GPBC_BOUNDS)
// F2-level leaves: one call per F2 product / squaring, all 36 / 18 limbs as scalar arguments (VGPRs; the last few of
// the 36 travel through the stack), the 18 result limbs returned in registers as one vector value.
typedef int32_t i32x18 __attribute__((ext_vector_type(18)));
#define GPBC_PACK_F2(a, b) F2{GPBC_PACK9(a), GPBC_PACK9(b)}
GPBC_INLINE i32x18 f2_to_vec(const F2 &r) {
    i32x18 v;
#pragma unroll
    for (int i = 0; i < NL; i++) { v[i] = r.a0.v[i]; v[NL + i] = r.a1.v[i]; }
    return v;
}
GPBC_INLINE F2 f2_from_vec(const i32x18 &v) {
    F2 r;
#pragma unroll
    for (int i = 0; i < NL; i++) { r.a0.v[i] = v[i]; r.a1.v[i] = v[NL + i]; }
    return r;
}
// An F2 product has 36 operand limbs and the calling convention 31 argument VGPRs (v31 carries the work-item id).  The last
// five used to travel through the stack, i.e. private memory — and the PMC counters show that on this machine every
// private-memory access of these kernels reaches HBM (profiles/r02_final5_pmc_traffic.json: k_final_exp wrote 17 KB and fetched
// 20 KB per lane, its dynamic count of scratch stores and loads times four bytes): ten HBM transactions per product, more than
// half of the kernel's traffic.  They go through LDS instead: a 16-byte and a 4-byte slot per lane of the (64-thread) workgroup,
// written by the caller right before the call and read first thing by the leaf, in order on the wave's LDS queue.  The slot is
// addressed by threadIdx.x (the work-item id travels into the leaf in v31).
// (The stack form measured 57.5 against 55.1 ms on k_miller_accumulate: profiles/r02_variant_lds_args.txt.)
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
constexpr int F2_ARG_LANES = 128;               // workgroups have one wave, except the latency form's Miller kernel (two: gpbc_pairing.hip)
__shared__ i32x4 g_f2_arg_slot[F2_ARG_LANES];
__shared__ int32_t g_f2_arg_slot4[F2_ARG_LANES];
GPBC_INLINE unsigned f2_arg_lane() { return threadIdx.x; }
template <bool NORM> __device__ __noinline__ i32x18 f2_mul_leaf(GPBC_ARGS9(a), GPBC_ARGS9(b), GPBC_ARGS9(c), int32_t d0, int32_t d1, int32_t d2, int32_t d3) {
    const unsigned lane = f2_arg_lane();
    const i32x4 t = g_f2_arg_slot[lane];
    const int32_t d4 = g_f2_arg_slot4[lane], d5 = t.x, d6 = t.y, d7 = t.z, d8 = t.w;
    return f2_to_vec(f2_mul_core<NORM>(GPBC_PACK_F2(a, b), GPBC_PACK_F2(c, d)));
}
template <bool NORM> GPBC_INLINE F2 f2_mul_call(const F2 &x, const F2 &y) {
    const unsigned lane = f2_arg_lane();
    g_f2_arg_slot[lane] = i32x4{y.a1.v[5], y.a1.v[6], y.a1.v[7], y.a1.v[8]};
    g_f2_arg_slot4[lane] = y.a1.v[4];
    return f2_from_vec(f2_mul_leaf<NORM>(GPBC_PASS9(x.a0), GPBC_PASS9(x.a1), GPBC_PASS9(y.a0), y.a1.v[0], y.a1.v[1], y.a1.v[2], y.a1.v[3]));
}
// a b + c d in one reduction, as a leaf of its own with the same LDS argument slots (36 operand limbs again): one HALF of an F2 product.
// The latency form (wide29.hip.hpp) gives the two halves of every F2 product to two lanes.
__device__ __noinline__ Fe fe_mul2_lds_leaf(GPBC_ARGS9(a), GPBC_ARGS9(b), GPBC_ARGS9(c), int32_t d0, int32_t d1, int32_t d2, int32_t d3) {
    const unsigned lane = f2_arg_lane();
    const i32x4 t = g_f2_arg_slot[lane];
    const int32_t d4 = g_f2_arg_slot4[lane], d5 = t.x, d6 = t.y, d7 = t.z, d8 = t.w;
    Fe a = GPBC_PACK9(a), b = GPBC_PACK9(b), c = GPBC_PACK9(c), d = GPBC_PACK9(d);
    return fe_mul_core<true>(a, b, c, d);
}
GPBC_INLINE Fe fe_mul2_l(const Fe &a, const Fe &b, const Fe &c, const Fe &d) {
    const unsigned lane = f2_arg_lane();
    g_f2_arg_slot[lane] = i32x4{d.v[5], d.v[6], d.v[7], d.v[8]};
    g_f2_arg_slot4[lane] = d.v[4];
    return fe_mul2_lds_leaf(GPBC_PASS9(a), GPBC_PASS9(b), GPBC_PASS9(c), d.v[0], d.v[1], d.v[2], d.v[3]);
}
template <bool NORM> __device__ __noinline__ i32x18 f2_sqr_leaf(GPBC_ARGS9(a), GPBC_ARGS9(b)) {
    return f2_to_vec(f2_sqr_core<NORM>(GPBC_PACK_F2(a, b)));
}
GPBC_INLINE F2 f2_mul(const F2 &x, const F2 &y) { return f2_mul_call<false>(x, y); }
GPBC_INLINE F2 f2_mul_nn(const F2 &x, const F2 &y) { return f2_mul_call<true>(x, y); }
GPBC_INLINE F2 f2_sqr(const F2 &x) { return f2_from_vec(f2_sqr_leaf<false>(GPBC_PASS9(x.a0), GPBC_PASS9(x.a1))); }
GPBC_INLINE F2 f2_sqr_n(const F2 &x) { return f2_from_vec(f2_sqr_leaf<true>(GPBC_PASS9(x.a0), GPBC_PASS9(x.a1))); }
#else
GPBC_INLINE Fe fe_mul2_l(const Fe &a, const Fe &b, const Fe &c, const Fe &d) { return fe_mul_core<true>(a, b, c, d); }
GPBC_INLINE F2 f2_mul(const F2 &x, const F2 &y) { return f2_mul_core<false>(x, y); }
GPBC_INLINE F2 f2_mul_nn(const F2 &x, const F2 &y) { return f2_mul_core<true>(x, y); }      // both operands un-normalised sums
GPBC_INLINE F2 f2_sqr(const F2 &x) { return f2_sqr_core<false>(x); }
GPBC_INLINE F2 f2_sqr_n(const F2 &x) { return f2_sqr_core<true>(x); }                        // operand an un-normalised sum
#endif
GPBC_INLINE F2 f2_mul_fe(const F2 &x, const Fe &k) { return F2{fe_mul(x.a0, k), fe_mul(x.a1, k)}; }
// (a0 + a1 i)(9 + i) = (9 a0 - a1) + (9 a1 + a0) i ; input N-class, output limbs < 3 * 2^29 (not normalised)
GPBC_INLINE F2 f2_mul_xi(const F2 &x) {
    return F2{fe_sub(fe_add(fe_mul8_norm(x.a0), x.a0), x.a1), fe_add(fe_add(fe_mul8_norm(x.a1), x.a1), x.a0)};
}
// N-class in, N-class out, and value-reduced (|value| < 0.51p): the factor |9+i| would otherwise compound
GPBC_INLINE F2 f2_mul_xi_n(const F2 &x) {
    F2 t = f2_norm(f2_mul_xi(x));
    return F2{fe_reduce(t.a0), fe_reduce(t.a1)};
}
GPBC_INLINE F2 f2_mul_xi_nn(const F2 &x) { return f2_norm(f2_mul_xi(x)); }      // N-class out, value NOT reduced (|xi| ~ 10 x)
template <bool RX> GPBC_INLINE F2 f2_mul_xi_t(const F2 &x) { return RX ? f2_mul_xi_n(x) : f2_mul_xi_nn(x); }
GPBC_INLINE F2 f2_mul8_norm(const F2 &x) { return F2{fe_mul8_norm(x.a0), fe_mul8_norm(x.a1)}; }
GPBC_INLINE F2 f2_inv(const F2 &x) {
    Fe n = fe_inv(fe_add(fe_sqr(x.a0), fe_sqr(x.a1)));       // fe_inv starts with a product by one: limbs up to 2^30 fit
    return F2{fe_mul(x.a0, n), fe_neg(fe_mul(x.a1, n))};
}
GPBC_INLINE F2 f2_load(const uint8_t *p) { return F2{fe_load(p), fe_load(p + 32)}; }
GPBC_INLINE void f2_store(uint8_t *p, const F2 &x) { fe_store(p, x.a0); fe_store(p + 32, x.a1); }

GPBC_INLINE F2 f2_const(const int32_t (&t)[2][NL]) { return F2{fe_const(t[0]), fe_const(t[1])}; }
GPBC_INLINE F2 gamma29(int j, int k) {   // xi^(k (p^j - 1)/6), j = 1..3, k = 1..5
    constexpr int32_t G[3][5][2][NL] = {F29_GAMMA1, F29_GAMMA2, F29_GAMMA3};
    F2 r;
#pragma unroll
    for (int i = 0; i < NL; i++) { r.a0.v[i] = G[j - 1][k - 1][0][i]; r.a1.v[i] = G[j - 1][k - 1][1][i]; }
    GPBC_B(set_class_n(r.a0, 1.0); set_class_n(r.a1, 1.0);)
    return r;
}
GPBC_INLINE F2 b_twist29() { constexpr int32_t B[2][NL] = F29_B_G2; return f2_const(B); }

// ------------------------------------------------------------------------------------------- F6
GPBC_INLINE F6 f6_add(const F6 &x, const F6 &y) { return F6{f2_add(x.b0, y.b0), f2_add(x.b1, y.b1), f2_add(x.b2, y.b2)}; }
GPBC_INLINE F6 f6_sub(const F6 &x, const F6 &y) { return F6{f2_sub(x.b0, y.b0), f2_sub(x.b1, y.b1), f2_sub(x.b2, y.b2)}; }
GPBC_INLINE F6 f6_neg(const F6 &x) { return F6{f2_neg(x.b0), f2_neg(x.b1), f2_neg(x.b2)}; }
GPBC_INLINE F6 f6_norm(const F6 &x) { return F6{f2_norm(x.b0), f2_norm(x.b1), f2_norm(x.b2)}; }
GPBC_INLINE F6 f6_reduce(const F6 &x) { return F6{f2_reduce(x.b0), f2_reduce(x.b1), f2_reduce(x.b2)}; }
GPBC_INLINE F2 f2_reduce_arith(const F2 &x) { return F2{fe_reduce_arith(x.a0), fe_reduce_arith(x.a1)}; }
GPBC_INLINE F6 f6_reduce_arith(const F6 &x) { return F6{f2_reduce_arith(x.b0), f2_reduce_arith(x.b1), f2_reduce_arith(x.b2)}; }
GPBC_INLINE F2 f2_reduce_arith_norm(const F2 &x) { return F2{fe_reduce_arith_norm(x.a0), fe_reduce_arith_norm(x.a1)}; }
GPBC_INLINE F6 f6_reduce_arith_norm(const F6 &x) { return F6{f2_reduce_arith_norm(x.b0), f2_reduce_arith_norm(x.b1), f2_reduce_arith_norm(x.b2)}; }
// x * v: (xi b2, b0, b1); N-class in and out
template <bool RX> GPBC_INLINE F6 f6_mul_v_t(const F6 &x) { return F6{f2_mul_xi_t<RX>(x.b2), x.b0, x.b1}; }
GPBC_INLINE F6 f6_mul_v(const F6 &x) { return f6_mul_v_t<true>(x); }

// RX = false leaves the two xi products un-reduced in value (the caller reduces its own outputs instead)
template <bool RX> GPBC_INLINE F6 f6_mul_t(const F6 &x, const F6 &y) {
    F2 t0 = f2_mul(x.b0, y.b0), t1 = f2_mul(x.b1, y.b1), t2 = f2_mul(x.b2, y.b2);
    F2 m12 = f2_mul_nn(f2_add(x.b1, x.b2), f2_add(y.b1, y.b2));
    F2 m01 = f2_mul_nn(f2_add(x.b0, x.b1), f2_add(y.b0, y.b1));
    F2 m02 = f2_mul_nn(f2_add(x.b0, x.b2), f2_add(y.b0, y.b2));
    F2 c0 = f2_add(f2_mul_xi_t<RX>(f2_norm(f2_sub(f2_sub(m12, t1), t2))), t0);
    F2 c1 = f2_add(f2_sub(f2_sub(m01, t0), t1), f2_mul_xi_t<RX>(t2));        // within (-2^30, 2^30 + 2^29): one normalisation, below
    F2 c2 = f2_add(f2_sub(f2_sub(m02, t0), t2), t1);
    return F6{f2_norm(c0), f2_norm(c1), f2_norm(c2)};
}
GPBC_INLINE F6 f6_mul(const F6 &x, const F6 &y) { return f6_mul_t<true>(x, y); }
// The same product for operands that come straight out of fe_norm ("positive-normalised": limbs 0..7 within [-2^4, 2^29 + 2^4],
// whatever the sign of the value — the top limb carries it): Karatsuba in SUBTRACTIVE form, (x_i - x_j)(y_i - y_j) = t_i + t_j -
// (x_i y_j + x_j y_i).  A difference of two such coefficients is an N-class operand as it stands, where the sums of the additive form
// must be normalised first (six F2 normalisations per product: the plain leaf instead of the normalising one, 3 x 96 instructions).
template <bool RX> GPBC_INLINE F6 f6_mul_pn_t(const F6 &x, const F6 &y) {
    F2 t0 = f2_mul(x.b0, y.b0), t1 = f2_mul(x.b1, y.b1), t2 = f2_mul(x.b2, y.b2);
    F2 m12 = f2_mul(f2_sub(x.b1, x.b2), f2_sub(y.b1, y.b2));
    F2 m01 = f2_mul(f2_sub(x.b0, x.b1), f2_sub(y.b0, y.b1));
    F2 m02 = f2_mul(f2_sub(x.b0, x.b2), f2_sub(y.b0, y.b2));
    F2 c0 = f2_add(f2_mul_xi_t<RX>(f2_norm(f2_sub(f2_add(t1, t2), m12))), t0);
    F2 c1 = f2_add(f2_sub(f2_add(t0, t1), m01), f2_mul_xi_t<RX>(t2));
    F2 c2 = f2_add(f2_sub(f2_add(t0, t2), m02), t1);
    return F6{f2_norm(c0), f2_norm(c1), f2_norm(c2)};
}
GPBC_INLINE F6 f6_sqr(const F6 &x) {   // CH-SQR2
    F2 s0 = f2_sqr(x.b0);
    F2 m01 = f2_mul(x.b0, x.b1);
    F2 s2 = f2_sqr_n(f2_add(f2_sub(x.b0, x.b1), x.b2));
    F2 m12 = f2_mul(x.b1, x.b2);
    F2 s4 = f2_sqr(x.b2);
    F2 c0 = f2_add(s0, f2_mul_xi_n(f2_norm(f2_dbl(m12))));
    F2 c1 = f2_add(f2_dbl(m01), f2_mul_xi_n(s4));
    F2 c2 = f2_add(f2_norm(f2_dbl(f2_add(m01, m12))), f2_sub(f2_sub(s2, s0), s4));
    return F6{c0, f2_norm(c1), f2_norm(c2)};                  // c0 = s0 + a reduced xi product: already within [-2^29, 2^30]
}
GPBC_INLINE F6 f6_mul_f2(const F6 &x, const F2 &k) { return F6{f2_mul(x.b0, k), f2_mul(x.b1, k), f2_mul(x.b2, k)}; }
// x * (c0 + c1 v); s01 = norm(c0 + c1) supplied by the caller (shared between the two uses in the sparse F12 product)
// NORM01 = false leaves the first output coefficient un-normalised as well (limbs within (-2^30, 2^30]): for callers that only
// add it to something and normalise the sum; the second one is never normalised here, the third one always (a multiplication
// by v sends it through the xi product, whose eightfold term needs the headroom).  Which normalisations are redundant was
// found by dropping them one at a time under the signed-interval bounds harness (tools/prune_norms.py).
template <bool RX, bool NORM01 = true> GPBC_INLINE F6 f6_mul_01_t(const F6 &x, const F2 &c0, const F2 &c1, const F2 &s01) {
    F2 a = f2_mul(x.b0, c0), b = f2_mul(x.b1, c1);
    F2 t0 = f2_add(f2_mul_xi_t<RX>(f2_sub(f2_mul(f2_norm(f2_add(x.b1, x.b2)), c1), b)), a);      // the xi product re-splits its eightfold term itself
    F2 t1 = f2_sub(f2_sub(f2_mul(f2_norm(f2_add(x.b0, x.b1)), s01), a), b);
    F2 t2 = f2_add(f2_sub(f2_mul(f2_norm(f2_add(x.b0, x.b2)), c0), a), b);
    return F6{NORM01 ? f2_norm(t0) : t0, t1, f2_norm(t2)};    // t1 = m - a - b stays within (-2^30, 2^29): every caller only adds it
}
GPBC_INLINE F6 f6_mul_01(const F6 &x, const F2 &c0, const F2 &c1, const F2 &s01) { return f6_mul_01_t<true>(x, c0, c1, s01); }
// The same sparse product for a positive-normalised x (see f6_mul_pn_t), subtractive form: the caller supplies d01 = norm(c0 - c1);
//   x2 c1 = b - (x1 - x2) c1,   x0 c1 + x1 c0 = a + b - (x0 - x1)(c0 - c1),   x2 c0 = a - (x0 - x2) c0        (a = x0 c0, b = x1 c1)
// — no operand of x's needs a normalisation.  Outputs as f6_mul_01_t<RX, false>: first and second coefficient un-normalised (limbs
// within (-2^30, 2^30]), third normalised.
template <bool RX> GPBC_INLINE F6 f6_mul_01_pn_t(const F6 &x, const F2 &c0, const F2 &c1, const F2 &d01) {
    F2 a = f2_mul(x.b0, c0), b = f2_mul(x.b1, c1);
    F2 t0 = f2_add(f2_mul_xi_t<RX>(f2_sub(b, f2_mul(f2_sub(x.b1, x.b2), c1))), a);
    F2 t1 = f2_sub(f2_add(a, b), f2_mul(f2_sub(x.b0, x.b1), d01));
    F2 t2 = f2_add(f2_sub(a, f2_mul(f2_sub(x.b0, x.b2), c0)), b);
    return F6{t0, t1, f2_norm(t2)};
}
GPBC_INLINE F6 f6_inv(const F6 &x) {
    F2 t0 = f2_norm(f2_sub(f2_sqr(x.b0), f2_mul_xi_n(f2_mul(x.b1, x.b2))));
    F2 t1 = f2_norm(f2_sub(f2_mul_xi_n(f2_sqr(x.b2)), f2_mul(x.b0, x.b1)));
    F2 t2 = f2_sub(f2_sqr(x.b1), f2_mul(x.b0, x.b2));
    F2 inner = f2_norm(f2_add(f2_mul(x.b2, t1), f2_mul(x.b1, t2)));
    F2 d = f2_norm(f2_add(f2_mul(x.b0, t0), f2_mul_xi_n(inner)));
    d = f2_inv(d);
    return F6{f2_mul(t0, d), f2_mul(t1, d), f2_mul(t2, d)};
}

// ------------------------------------------------------------------------------------------- F12
GPBC_INLINE F12 f12_one() { return F12{F6{f2_one(), f2_zero(), f2_zero()}, F6{f2_zero(), f2_zero(), f2_zero()}}; }
GPBC_INLINE F12 f12_conj(const F12 &x) { return F12{x.c0, f6_neg(x.c1)}; }
GPBC_INLINE F12 f12_mul(const F12 &x, const F12 &y) {
    F6 t0 = f6_mul(x.c0, y.c0), t1 = f6_mul(x.c1, y.c1);
    F6 m = f6_mul(f6_norm(f6_add(x.c0, x.c1)), f6_norm(f6_add(y.c0, y.c1)));
    F6 c1 = f6_sub(f6_sub(m, t0), t1);
    F6 c0 = f6_add(t0, f6_mul_v(t1));
    return F12{f6_reduce(f6_norm(c0)), f6_reduce(f6_norm(c1))};   // three-term sums of F6 products: reset the value bound
}
GPBC_INLINE F12 f12_sqr(const F12 &x) {
    F6 m = f6_mul(x.c0, x.c1);
    F6 s = f6_norm(f6_add(x.c0, x.c1));
    F6 t = f6_norm(f6_add(x.c0, f6_mul_v(x.c1)));
    F6 st = f6_mul(s, t);
    F6 mv = f6_mul_v(m);
    return F12{f6_norm(f6_sub(f6_sub(st, m), mv)), f6_norm(f6_add(m, m))};
}
GPBC_INLINE F12 f12_inv(const F12 &x) {
    F6 t1 = f6_mul_v(f6_sqr(x.c1));
    F6 d = f6_inv(f6_norm(f6_sub(f6_sqr(x.c0), t1)));
    return F12{f6_mul(x.c0, d), f6_neg(f6_mul(x.c1, d))};
}
// 1 / y = conj(y) / N(y) with the norm N(y) = c0^2 - v c1^2 in Fp6.  What the reference divides by is a pairing value (GT.Div of a
// ciphertext component by a pairing product, bibe/afp25_bibe/afp25_bibe.go:408-413, cpabe/bsw07/bsw07_cpabe.go:189-190), and those have
// norm one: when every element of the wavefront does (`all`: the agreement of the active lanes), the Fp6 inversion and the two Fp6
// products are skipped (40 k -> 8 k instructions); any other element in the wavefront and all of it takes the general path — the
// same field element either way.
GPBC_INLINE bool f6_is_one(const F6 &v) {
    const F2 d = f2_norm(f2_sub(v.b0, f2_one()));
    return f2_is_zero(d) && f2_is_zero(v.b1) && f2_is_zero(v.b2);
}
template <class All> GPBC_INLINE F12 f12_inv_gt(const F12 &y, All &&all) {
    const F6 nrm = f6_norm(f6_sub(f6_sqr(y.c0), f6_mul_v(f6_sqr(y.c1))));
    if (all(f6_is_one(nrm))) return f12_conj(y);
    const F6 d = f6_inv(nrm);
    return F12{f6_mul(y.c0, d), f6_neg(f6_mul(y.c1, d))};
}
// x^(p^j), j = 1..3: coefficient of w^k -> (conj if j odd)(c_k) * gamma_j[k]; w-basis order C0.B0,C1.B0,C0.B1,C1.B1,C0.B2,C1.B2
GPBC_INLINE F12 f12_frob(const F12 &x, int j) {
    const bool odd = j & 1;
    F2 c[6] = {x.c0.b0, x.c1.b0, x.c0.b1, x.c1.b1, x.c0.b2, x.c1.b2};
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if (odd) c[k] = f2_conj(c[k]);
        if (k) c[k] = f2_mul(c[k], gamma29(j, k));
    }
    return F12{F6{c[0], c[2], c[4]}, F6{c[1], c[3], c[5]}};
}
// Granger-Scott squaring in the cyclotomic subgroup.  x enters the result linearly (3t -+ 2x), so without a value
// reduction the worst-case magnitude would double with every squaring of a chain; REDUCE = false skips it and may be
// used for at most one squaring in a row (tools/bounds_check.cpp proves the int64 columns still cannot overflow).
template <bool REDUCE> GPBC_INLINE F2 cyclo_out(const F2 &t, const F2 &x, bool plus) {
    F2 d = f2_norm(plus ? f2_add(t, x) : f2_sub(t, x));
    F2 r = f2_norm(f2_add(f2_dbl(d), t));
    return REDUCE ? f2_reduce(r) : r;
}
template <bool REDUCE> GPBC_INLINE F12 f12_cyclo_sqr_t(const F12 &x) {
    F2 t0 = f2_sqr(x.c1.b1), t1 = f2_sqr(x.c0.b0);
    F2 t6 = f2_sub(f2_sub(f2_sqr_n(f2_add(x.c1.b1, x.c0.b0)), t0), t1);       // (-2^30, 2^29): cyclo_out normalises the sum it enters
    F2 t2 = f2_sqr(x.c0.b2), t3 = f2_sqr(x.c1.b0);
    F2 t7 = f2_sub(f2_sub(f2_sqr_n(f2_add(x.c0.b2, x.c1.b0)), t2), t3);
    F2 t4 = f2_sqr(x.c1.b2), t5 = f2_sqr(x.c0.b1);
    F2 t8 = f2_mul_xi_n(f2_norm(f2_sub(f2_sub(f2_sqr_n(f2_add(x.c1.b2, x.c0.b1)), t4), t5)));
    t0 = f2_norm(f2_add(f2_mul_xi_n(t0), t1));
    t2 = f2_norm(f2_add(f2_mul_xi_n(t2), t3));
    t4 = f2_norm(f2_add(f2_mul_xi_n(t4), t5));
    F12 r;
    r.c0.b0 = cyclo_out<REDUCE>(t0, x.c0.b0, false);
    r.c0.b1 = cyclo_out<REDUCE>(t2, x.c0.b1, false);
    r.c0.b2 = cyclo_out<REDUCE>(t4, x.c0.b2, false);
    r.c1.b0 = cyclo_out<REDUCE>(t8, x.c1.b0, true);
    r.c1.b1 = cyclo_out<REDUCE>(t6, x.c1.b1, true);
    r.c1.b2 = cyclo_out<REDUCE>(t7, x.c1.b2, true);
    return r;
}
GPBC_INLINE F12 f12_cyclo_sqr(const F12 &x) { return f12_cyclo_sqr_t<true>(x); }
// z = x * (c0 + c3 w + c4 v w): the sparse line element of the Miller loop.  Karatsuba over F6 with the sparsity
// kept: a*l0 (3 F2 products), b*l1 (5), (a+b)*(l0+l1) (5) = 13 F2 products.
GPBC_INLINE F12 f12_mul_034(const F12 &x, const F2 &c0, const F2 &c3, const F2 &c4) {
    F2 s34 = f2_norm(f2_add(c3, c4));
    F2 c03 = f2_norm(f2_add(c0, c3));
    F2 s034 = f2_norm(f2_add(c03, c4));
    F6 t0 = f6_mul_f2(x.c0, c0);
    F6 t1 = f6_mul_01(x.c1, c3, c4, s34);
    F6 t2 = f6_mul_01(f6_norm(f6_add(x.c0, x.c1)), c03, c4, s034);
    F6 r0 = f6_add(t0, f6_mul_v(t1));
    F6 r1 = f6_sub(f6_sub(t2, t0), t1);
    return F12{f6_norm(r0), f6_reduce(f6_norm(r1))};     // r1 is a three-term sum of F6 products: reset its value bound
}
GPBC_INLINE void f12_load(F12 &z, const uint8_t *p) {
    z.c0.b0 = f2_load(p); z.c0.b1 = f2_load(p + 64); z.c0.b2 = f2_load(p + 128);
    z.c1.b0 = f2_load(p + 192); z.c1.b1 = f2_load(p + 256); z.c1.b2 = f2_load(p + 320);
}
GPBC_INLINE void f12_store(uint8_t *p, const F12 &z) {
    f2_store(p, z.c0.b0); f2_store(p + 64, z.c0.b1); f2_store(p + 128, z.c0.b2);
    f2_store(p + 192, z.c1.b0); f2_store(p + 256, z.c1.b1); f2_store(p + 320, z.c1.b2);
}

}  // namespace gpbc
#endif
