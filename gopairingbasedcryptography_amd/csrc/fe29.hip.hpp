// BN254 base field for gfx950, second-generation layout: nine SIGNED 29-bit limbs per element
// (value = sum v[i] * 2^(29 i)), Montgomery radix R' = 2^261.
//
// Why this shape (measured, profiles/r01_microbench_valu.txt): on MI355X v_mad_i64_i32 / v_mad_u64_u32 issue at the
// same ~4.5 cycles per wave-instruction as a carry-chain add (v_addc_co_u32), while a plain v_add_u32 / v_sub_u32
// issues at 2.4.  So the cheapest big-integer product is one MAD per limb pair into a 64-bit column accumulator with
// NO carry handling (29-bit limbs leave 6 spare bits: 18 products of 2^58 plus the reduction terms fit in int64),
// and field add / sub are nine independent full-rate VALU ops with no carries and no conditional subtract.
// Limbs and values are allowed to drift (lazy reduction); `fe_norm` is a carry-free re-normalisation; only values
// leaving the kernel are made canonical in [0,p) and converted to gnark's R = 2^256 Montgomery bytes — which is
// what makes the outputs bit-identical to the reference's CPU path.
//
// Static safety: compiled with -DGPBC_BOUNDS on the host (tools/bounds_check.cpp) every Fe carries data-independent
// magnitude bounds and every multiplication asserts that its int64 columns cannot overflow.
#ifndef GPBC_FE29_HIP_HPP
#define GPBC_FE29_HIP_HPP
#include <stdint.h>
#include <math.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GPBC_INLINE __host__ __device__ __forceinline__
#define GPBC_NOINLINE inline __host__ __device__ __noinline__   /* inline: one definition per translation unit */
#else
#define GPBC_INLINE inline
#define GPBC_NOINLINE
#endif
#include "bn254_constants29.hip.hpp"

#ifdef GPBC_BOUNDS
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <execinfo.h>
#define GPBC_B(...) __VA_ARGS__
#else
#define GPBC_B(...)
#endif

namespace gpbc {

constexpr int NL = 9;
constexpr int LB = 29;
constexpr int32_t LMASK = (1 << LB) - 1;

struct Fe {
    int32_t v[NL];
#ifdef GPBC_BOUNDS
    double lo[NL], hi[NL];   // lo[i] <= v[i] <= hi[i]   (data-independent signed intervals)
    double vb;               // |value| <= vb * p
#endif
};

#ifdef GPBC_BOUNDS
struct BoundStats { double max_col = 0, max_limb = 0, max_vb = 0; long muls = 0, norms = 0, muls2 = 0, reduces = 0, mads = 0; };   // mads: the v_mad_i64_i32 the DEVICE form of the same calls executes
inline BoundStats &bound_stats() { static thread_local BoundStats s; return s; }   // per thread (the pair harness runs two)
inline void bound_stats_merge(BoundStats &into, BoundStats &from) {
    if (from.max_col > into.max_col) into.max_col = from.max_col;
    if (from.max_limb > into.max_limb) into.max_limb = from.max_limb;
    if (from.max_vb > into.max_vb) into.max_vb = from.max_vb;
    into.muls += from.muls; into.norms += from.norms; into.muls2 += from.muls2; into.reduces += from.reduces; into.mads += from.mads;
    from = BoundStats();
}
inline void bounds_fail(const char *what, double got, double lim) {
    fprintf(stderr, "BOUNDS VIOLATION: %s: %.6g exceeds %.6g\n", what, got, lim);
    void *bt[24];
    int n = backtrace(bt, 24);
    backtrace_symbols_fd(bt, n, 2);       // build with -O0 -g -rdynamic to get one frame per tower function
    abort();
}
constexpr double P_OVER_2_232 = 3171407.0;     // ceil(p / 2^232): top-limb magnitude per unit of p
constexpr double P_OVER_RP = 0.0059081;         // p / 2^261 (rounded up)
inline double limb_mag(const Fe &r, int i) { return std::fmax(std::fabs(r.lo[i]), std::fabs(r.hi[i])); }
inline void set_class_n(Fe &r, double vb) {     // limbs 0..7 masked to [0,2^29), top limb holds the rest
    for (int i = 0; i < NL - 1; i++) { r.lo[i] = 0; r.hi[i] = (double)LMASK; }
    r.hi[NL - 1] = vb * P_OVER_2_232 + 2;
    r.lo[NL - 1] = -r.hi[NL - 1];
    r.vb = vb;
    if (r.hi[NL - 1] > bound_stats().max_limb) bound_stats().max_limb = r.hi[NL - 1];
    if (vb > bound_stats().max_vb) bound_stats().max_vb = vb;
}
inline void check_limbs(const Fe &r, const char *what) {
    for (int i = 0; i < NL; i++) {
        if (r.hi[i] >= 2147483648.0) bounds_fail(what, r.hi[i], 2147483648.0);
        if (r.lo[i] < -2147483648.0) bounds_fail(what, r.lo[i], -2147483648.0);
        if ((double)r.v[i] > r.hi[i] || (double)r.v[i] < r.lo[i]) bounds_fail("tracked interval does not hold the actual value", (double)r.v[i], r.v[i] > r.hi[i] ? r.hi[i] : r.lo[i]);
        if (limb_mag(r, i) > bound_stats().max_limb) bound_stats().max_limb = limb_mag(r, i);
    }
}
// interval of the column sums of a product sum_k x_k * y_k, limb intervals given: lo / hi of sum_{i+j=col} x_i y_j
inline void prod_interval(double xl, double xh, double yl, double yh, double &pl, double &ph) {
    double a = xl * yl, b = xl * yh, c = xh * yl, d = xh * yh;
    pl = std::fmin(std::fmin(a, b), std::fmin(c, d));
    ph = std::fmax(std::fmax(a, b), std::fmax(c, d));
}
inline void check_columns(double lo, double hi, const char *what) {
    // + reduction terms m_i p_j (both factors in [0, 2^29): nine of them) and the running carry (|carry| < 2^35)
    double top = hi + 9.0 * 536870912.0 * 536870912.0 + 34359738368.0, bot = lo - 34359738368.0;
    if (top >= 9223372036854775808.0) bounds_fail(what, top, 9223372036854775808.0);
    if (bot <= -9223372036854775808.0) bounds_fail(what, bot, -9223372036854775808.0);
    double m = std::fmax(top, -bot);
    if (m > bound_stats().max_col) bound_stats().max_col = m;
}
#endif

GPBC_INLINE constexpr int32_t f29_p(int i) { constexpr int32_t P[NL] = F29_P; return P[i]; }

GPBC_INLINE Fe fe_const(const int32_t (&c)[NL]) {
    Fe r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = c[i];
    GPBC_B(set_class_n(r, 1.0);)
    return r;
}
GPBC_INLINE Fe fe_zero() {
    Fe r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = 0;
    GPBC_B(for (int i = 0; i < NL; i++) { r.lo[i] = 0; r.hi[i] = 0; } r.vb = 0;)
    return r;
}
GPBC_INLINE Fe fe_one() { constexpr int32_t C[NL] = F29_ONE; return fe_const(C); }

// ------------------------------------------------------------------------------------------------ carry-free linear ops
GPBC_INLINE Fe fe_add(const Fe &a, const Fe &b) {
    Fe r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = a.v[i] + b.v[i];
    GPBC_B(for (int i = 0; i < NL; i++) { r.lo[i] = a.lo[i] + b.lo[i]; r.hi[i] = a.hi[i] + b.hi[i]; } r.vb = a.vb + b.vb; check_limbs(r, "fe_add limb");)
    return r;
}
GPBC_INLINE Fe fe_sub(const Fe &a, const Fe &b) {
    Fe r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = a.v[i] - b.v[i];
    GPBC_B(for (int i = 0; i < NL; i++) { r.lo[i] = a.lo[i] - b.hi[i]; r.hi[i] = a.hi[i] - b.lo[i]; } r.vb = a.vb + b.vb; check_limbs(r, "fe_sub limb");)
    return r;
}
GPBC_INLINE Fe fe_neg(const Fe &a) {
    Fe r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = -a.v[i];
    GPBC_B(for (int i = 0; i < NL; i++) { r.lo[i] = -a.hi[i]; r.hi[i] = -a.lo[i]; } r.vb = a.vb; check_limbs(r, "fe_neg limb");)
    return r;
}
GPBC_INLINE Fe fe_dbl(const Fe &a) { return fe_add(a, a); }

// Carry-free ("parallel") normalisation: every limb keeps its low 29 bits and receives the carry of the limb
// below; the top limb is not masked.  Result limbs 0..7 lie in [-2^k, 2^29 + 2^k) for inputs below 2^(29+k).
GPBC_INLINE Fe fe_norm(const Fe &a) {
    Fe r;
    r.v[0] = a.v[0] & LMASK;
#pragma unroll
    for (int i = 1; i < NL - 1; i++) r.v[i] = (a.v[i] & LMASK) + (a.v[i - 1] >> LB);
    r.v[NL - 1] = a.v[NL - 1] + (a.v[NL - 2] >> LB);
#ifdef GPBC_BOUNDS
    r.lo[0] = 0; r.hi[0] = (double)LMASK;
    // (x & M) is in [0, M]; (y >> 29) = floor(y / 2^29) is monotone in y
    for (int i = 1; i < NL - 1; i++) { r.lo[i] = std::floor(a.lo[i - 1] / 536870912.0); r.hi[i] = (double)LMASK + std::floor(a.hi[i - 1] / 536870912.0); }
    r.lo[NL - 1] = a.lo[NL - 1] + std::floor(a.lo[NL - 2] / 536870912.0);
    r.hi[NL - 1] = a.hi[NL - 1] + std::floor(a.hi[NL - 2] / 536870912.0);
    r.vb = a.vb;
    bound_stats().norms++;
    check_limbs(r, "fe_norm limb");
#endif
    return r;
}

// ------------------------------------------------------------------------------------------------ Montgomery products
// (a*b [+ c*d]) / 2^261 mod p, one v_mad_i64_i32 per limb pair.
// Output: limbs 0..7 in [0,2^29), limb 8 signed and small; value in (-eps p, (1+eps) p).
template <bool TWO>
GPBC_INLINE Fe fe_mul_core(const Fe &a, const Fe &b, const Fe &c, const Fe &d) {
#ifdef GPBC_BOUNDS
    {
        for (int k = 0; k < 2 * NL - 1; k++) {
            double sl = 0, sh = 0;
            for (int i = 0; i < NL; i++) {
                int j = k - i;
                if (j < 0 || j >= NL) continue;
                double pl, ph;
                prod_interval(a.lo[i], a.hi[i], b.lo[j], b.hi[j], pl, ph); sl += pl; sh += ph;
                if (TWO) { prod_interval(c.lo[i], c.hi[i], d.lo[j], d.hi[j], pl, ph); sl += pl; sh += ph; }
            }
            check_columns(sl, sh, "fe_mul column");
        }
        bound_stats().muls++;
        if (TWO) bound_stats().muls2++;
        bound_stats().mads += (TWO ? 2 : 1) * NL * NL + NL * NL;      // limb products + reduction terms
    }
#endif
    // Column-wise (product-scanning) Montgomery: one running 64-bit accumulator; column k collects its limb
    // products and the reduction terms m_j * p_(k-j), the low columns each produce m_k and are cleared by m_k * p_0,
    // the high columns each emit one 29-bit limb; the carry into the next column is the accumulator shifted by 29
    // (it simply stays in the accumulator, so there is no separate carry addition).
    int32_t m[NL];
    Fe r;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int j = k - i;
            if (j < 0 || j >= NL) continue;
            acc += (int64_t)a.v[i] * (int64_t)b.v[j];
            if (TWO) acc += (int64_t)c.v[i] * (int64_t)d.v[j];
        }
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int j = k - i;            // m_i * p_j, only for already known m_i (i < k for low columns)
            if (j < 1 || j >= NL) continue;
            acc += (int64_t)m[i] * (int64_t)f29_p(j);
        }
        if (k < NL) {
            m[k] = (int32_t)(((uint32_t)acc * (uint32_t)F29_PINV) & (uint32_t)LMASK);
            acc += (int64_t)m[k] * (int64_t)f29_p(0);
        } else {
            r.v[k - NL] = (int32_t)(acc & LMASK);
        }
        acc >>= LB;
    }
    r.v[NL - 1] = (int32_t)acc;
#ifdef GPBC_BOUNDS
    double vb = (a.vb * b.vb + (TWO ? c.vb * d.vb : 0.0)) * P_OVER_RP + 1.0;
    set_class_n(r, vb);
    if (r.hi[NL - 1] >= 268435456.0) bounds_fail("fe_mul output top limb", r.hi[NL - 1], 268435456.0);
    check_limbs(r, "fe_mul output");
#endif
    return r;
}

// Two independent products-with-reduction advanced column by column in lockstep: r0 = (a0*b0 + c0*d0)/R', r1 = (a1*b1 + c1*d1)/R'.
// Same arithmetic as two fe_mul_core<true> calls; interleaving the two accumulator chains gives the instruction
// scheduler independent MADs to alternate between (a dependent v_mad_i64_i32 chain issues at ~7 cycles per instruction
// at two waves per SIMD, independent ones at ~4.6).
GPBC_INLINE void fe_mul2_dual(Fe &r0, Fe &r1, const Fe &a0, const Fe &b0, const Fe &c0, const Fe &d0,
                              const Fe &a1, const Fe &b1, const Fe &c1, const Fe &d1) {
#ifdef GPBC_BOUNDS
    r0 = fe_mul_core<true>(a0, b0, c0, d0);
    r1 = fe_mul_core<true>(a1, b1, c1, d1);
#else
    int32_t m0[NL], m1[NL];
    int64_t acc0 = 0, acc1 = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int j = k - i;
            if (j < 0 || j >= NL) continue;
            acc0 += (int64_t)a0.v[i] * (int64_t)b0.v[j];
            acc1 += (int64_t)a1.v[i] * (int64_t)b1.v[j];
            acc0 += (int64_t)c0.v[i] * (int64_t)d0.v[j];
            acc1 += (int64_t)c1.v[i] * (int64_t)d1.v[j];
        }
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int j = k - i;
            if (j < 1 || j >= NL) continue;
            acc0 += (int64_t)m0[i] * (int64_t)f29_p(j);
            acc1 += (int64_t)m1[i] * (int64_t)f29_p(j);
        }
        if (k < NL) {
            m0[k] = (int32_t)(((uint32_t)acc0 * (uint32_t)F29_PINV) & (uint32_t)LMASK);
            m1[k] = (int32_t)(((uint32_t)acc1 * (uint32_t)F29_PINV) & (uint32_t)LMASK);
            acc0 += (int64_t)m0[k] * (int64_t)f29_p(0);
            acc1 += (int64_t)m1[k] * (int64_t)f29_p(0);
        } else {
            r0.v[k - NL] = (int32_t)(acc0 & LMASK);
            r1.v[k - NL] = (int32_t)(acc1 & LMASK);
        }
        acc0 >>= LB;
        acc1 >>= LB;
    }
    r0.v[NL - 1] = (int32_t)acc0;
    r1.v[NL - 1] = (int32_t)acc1;
#endif
}

// Same lockstep pairing for two single products r0 = a0*b0/R', r1 = a1*b1/R' (the two halves of an F2 squaring)
GPBC_INLINE void fe_mul_dual(Fe &r0, Fe &r1, const Fe &a0, const Fe &b0, const Fe &a1, const Fe &b1) {
#ifdef GPBC_BOUNDS
    r0 = fe_mul_core<false>(a0, b0, a0, b0);
    r1 = fe_mul_core<false>(a1, b1, a1, b1);
#else
    int32_t m0[NL], m1[NL];
    int64_t acc0 = 0, acc1 = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int j = k - i;
            if (j < 0 || j >= NL) continue;
            acc0 += (int64_t)a0.v[i] * (int64_t)b0.v[j];
            acc1 += (int64_t)a1.v[i] * (int64_t)b1.v[j];
        }
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int j = k - i;
            if (j < 1 || j >= NL) continue;
            acc0 += (int64_t)m0[i] * (int64_t)f29_p(j);
            acc1 += (int64_t)m1[i] * (int64_t)f29_p(j);
        }
        if (k < NL) {
            m0[k] = (int32_t)(((uint32_t)acc0 * (uint32_t)F29_PINV) & (uint32_t)LMASK);
            m1[k] = (int32_t)(((uint32_t)acc1 * (uint32_t)F29_PINV) & (uint32_t)LMASK);
            acc0 += (int64_t)m0[k] * (int64_t)f29_p(0);
            acc1 += (int64_t)m1[k] * (int64_t)f29_p(0);
        } else {
            r0.v[k - NL] = (int32_t)(acc0 & LMASK);
            r1.v[k - NL] = (int32_t)(acc1 & LMASK);
        }
        acc0 >>= LB;
        acc1 >>= LB;
    }
    r0.v[NL - 1] = (int32_t)acc0;
    r1.v[NL - 1] = (int32_t)acc1;
#endif
}

// The two halves of an F2 squaring, lazily and without forming a0 +- a1:  r0 = (a0^2 - a1^2)/R' from the symmetric halves of
// the two squares (45 + 45 product MADs, one reduction), r1 = (2 a0 a1)/R' (81 + reduction) — 333 MADs against 324 for the
// (a0+a1)(a0-a1) form, but no operand sums to normalise (~70 instructions fewer per squaring).  Operands normalised.
GPBC_INLINE void fe_sqrdiff_mul_dual(Fe &r0, Fe &r1, const Fe &a0, const Fe &a1) {
#ifdef GPBC_BOUNDS
    r0 = fe_mul_core<true>(a0, a0, fe_neg(a1), a1);
    r1 = fe_mul_core<false>(fe_dbl(a0), a1, a0, a1);
    bound_stats().mads -= 2 * NL * NL - 2 * (NL * (NL + 1) / 2);       // the device form takes the symmetric halves of the two squares: 45 + 45, not 81 + 81
#else
    int32_t d0[NL], n1[NL], dn1[NL], m0[NL], m1[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) { d0[i] = a0.v[i] * 2; n1[i] = -a1.v[i]; dn1[i] = -2 * a1.v[i]; }
    int64_t acc0 = 0, acc1 = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int j = k - i;
            if (j < 0 || j >= NL) continue;
            acc1 += (int64_t)d0[i] * (int64_t)a1.v[j];
            if (i > j) continue;
            if (i == j) {
                acc0 += (int64_t)a0.v[i] * (int64_t)a0.v[i];
                acc0 += (int64_t)n1[i] * (int64_t)a1.v[i];
            } else {
                acc0 += (int64_t)d0[i] * (int64_t)a0.v[j];
                acc0 += (int64_t)dn1[i] * (int64_t)a1.v[j];
            }
        }
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int j = k - i;
            if (j < 1 || j >= NL) continue;
            acc0 += (int64_t)m0[i] * (int64_t)f29_p(j);
            acc1 += (int64_t)m1[i] * (int64_t)f29_p(j);
        }
        if (k < NL) {
            m0[k] = (int32_t)(((uint32_t)acc0 * (uint32_t)F29_PINV) & (uint32_t)LMASK);
            m1[k] = (int32_t)(((uint32_t)acc1 * (uint32_t)F29_PINV) & (uint32_t)LMASK);
            acc0 += (int64_t)m0[k] * (int64_t)f29_p(0);
            acc1 += (int64_t)m1[k] * (int64_t)f29_p(0);
        } else {
            r0.v[k - NL] = (int32_t)(acc0 & LMASK);
            r1.v[k - NL] = (int32_t)(acc1 & LMASK);
        }
        acc0 >>= LB;
        acc1 >>= LB;
    }
    r0.v[NL - 1] = (int32_t)acc0;
    r1.v[NL - 1] = (int32_t)acc1;
#endif
}

#define GPBC_ARGS9(x) int32_t x##0, int32_t x##1, int32_t x##2, int32_t x##3, int32_t x##4, int32_t x##5, int32_t x##6, int32_t x##7, int32_t x##8
#define GPBC_PASS9(x) x.v[0], x.v[1], x.v[2], x.v[3], x.v[4], x.v[5], x.v[6], x.v[7], x.v[8]
#define GPBC_PACK9(x) Fe{{x##0, x##1, x##2, x##3, x##4, x##5, x##6, x##7, x##8}}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(GPBC_BOUNDS)
// Leaf functions with every limb passed as a scalar argument: scalars go in VGPRs v0..v31 (aggregates larger than
// 16 dwords would travel through scratch), so the tower above can stay inlined while the ~200 / ~290-instruction
// multipliers exist once in the code object (I-cache).
__device__ __noinline__ Fe fe_mul_leaf(GPBC_ARGS9(a), GPBC_ARGS9(b)) {
    Fe a = GPBC_PACK9(a), b = GPBC_PACK9(b);
    return fe_mul_core<false>(a, b, a, b);
}
__device__ __noinline__ Fe fe_mul2_leaf(GPBC_ARGS9(a), GPBC_ARGS9(b), GPBC_ARGS9(c), GPBC_ARGS9(d)) {
    Fe a = GPBC_PACK9(a), b = GPBC_PACK9(b), c = GPBC_PACK9(c), d = GPBC_PACK9(d);
    return fe_mul_core<true>(a, b, c, d);
}
GPBC_INLINE Fe fe_mul(const Fe &a, const Fe &b) { return fe_mul_leaf(GPBC_PASS9(a), GPBC_PASS9(b)); }
GPBC_INLINE Fe fe_mul2(const Fe &a, const Fe &b, const Fe &c, const Fe &d) {
    return fe_mul2_leaf(GPBC_PASS9(a), GPBC_PASS9(b), GPBC_PASS9(c), GPBC_PASS9(d));
}
#else
GPBC_INLINE Fe fe_mul(const Fe &a, const Fe &b) { return fe_mul_core<false>(a, b, a, b); }
GPBC_INLINE Fe fe_mul2(const Fe &a, const Fe &b, const Fe &c, const Fe &d) { return fe_mul_core<true>(a, b, c, d); }
#endif
// a^2 / 2^261 mod p with the symmetric half of the product only: column k = sum_{i<j, i+j=k} (2 a_i) a_j + a_(k/2)^2,
// 45 product MADs instead of 81 (the reduction half is unchanged).
GPBC_INLINE Fe fe_sqr_core(const Fe &a) {
#ifdef GPBC_BOUNDS
    {
        for (int k = 0; k < 2 * NL - 1; k++) {
            double sl = 0, sh = 0;
            for (int i = 0; i < NL; i++) { int j = k - i; if (j < 0 || j >= NL) continue; double pl, ph; prod_interval(a.lo[i], a.hi[i], a.lo[j], a.hi[j], pl, ph); sl += pl; sh += ph; }
            check_columns(sl, sh, "fe_sqr column");
        }
        for (int i = 0; i < NL; i++) if (2 * limb_mag(a, i) >= 2147483648.0) bounds_fail("fe_sqr doubled limb", 2 * limb_mag(a, i), 2147483648.0);
        bound_stats().muls++;
        bound_stats().mads += NL * (NL + 1) / 2 + NL * NL;
    }
#endif
    int32_t d[NL], m[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) d[i] = a.v[i] * 2;
    Fe r;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int j = k - i;
            if (j < 0 || j >= NL || i > j) continue;
            acc += (i == j) ? (int64_t)a.v[i] * (int64_t)a.v[i] : (int64_t)d[i] * (int64_t)a.v[j];
        }
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int j = k - i;
            if (j < 1 || j >= NL) continue;
            acc += (int64_t)m[i] * (int64_t)f29_p(j);
        }
        if (k < NL) {
            m[k] = (int32_t)(((uint32_t)acc * (uint32_t)F29_PINV) & (uint32_t)LMASK);
            acc += (int64_t)m[k] * (int64_t)f29_p(0);
        } else {
            r.v[k - NL] = (int32_t)(acc & LMASK);
        }
        acc >>= LB;
    }
    r.v[NL - 1] = (int32_t)acc;
#ifdef GPBC_BOUNDS
    set_class_n(r, a.vb * a.vb * P_OVER_RP + 1.0);
    if (r.hi[NL - 1] >= 268435456.0) bounds_fail("fe_sqr output top limb", r.hi[NL - 1], 268435456.0);
    check_limbs(r, "fe_sqr output");
#endif
    return r;
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GPBC_BOUNDS)
__device__ __noinline__ Fe fe_sqr_leaf(GPBC_ARGS9(a)) { Fe a = GPBC_PACK9(a); return fe_sqr_core(a); }
GPBC_INLINE Fe fe_sqr(const Fe &a) { return fe_sqr_leaf(GPBC_PASS9(a)); }
#else
GPBC_INLINE Fe fe_sqr(const Fe &a) { return fe_sqr_core(a); }
#endif

// ------------------------------------------------------------------------------------------------ small multiples
// 8a with the limbs re-split on the fly (no limb grows beyond 2^29 + 2^(k+3) for |a limbs| < 2^(29+k))
GPBC_INLINE Fe fe_mul8_norm(const Fe &a) {
    Fe r;
    constexpr int32_t M26 = (1 << (LB - 3)) - 1;
    r.v[0] = (a.v[0] & M26) << 3;
#pragma unroll
    for (int i = 1; i < NL - 1; i++) r.v[i] = ((a.v[i] & M26) << 3) + (a.v[i - 1] >> (LB - 3));
    r.v[NL - 1] = a.v[NL - 1] * 8 + (a.v[NL - 2] >> (LB - 3));
#ifdef GPBC_BOUNDS
    r.lo[0] = 0; r.hi[0] = (double)(LMASK - 7);
    for (int i = 1; i < NL - 1; i++) { r.lo[i] = std::floor(a.lo[i - 1] / 67108864.0); r.hi[i] = (double)(LMASK - 7) + std::floor(a.hi[i - 1] / 67108864.0); }
    r.lo[NL - 1] = a.lo[NL - 1] * 8 + std::floor(a.lo[NL - 2] / 67108864.0);
    r.hi[NL - 1] = a.hi[NL - 1] * 8 + std::floor(a.hi[NL - 2] / 67108864.0);
    r.vb = a.vb * 8;
    check_limbs(r, "fe_mul8_norm limb");
#endif
    return r;
}
// a/2 mod p: make the value even with p (parity of the value = parity of limb 0), then shift across limbs
GPBC_INLINE Fe fe_halve(const Fe &a) {
    int32_t odd = -(a.v[0] & 1);
    int32_t t[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) t[i] = a.v[i] + (f29_p(i) & odd);
    Fe r;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) r.v[i] = (t[i] >> 1) + ((t[i + 1] & 1) << (LB - 1));
    r.v[NL - 1] = t[NL - 1] >> 1;
#ifdef GPBC_BOUNDS
    for (int i = 0; i < NL; i++) if (a.hi[i] + (double)LMASK >= 2147483648.0) bounds_fail("fe_halve input limb", a.hi[i], 2147483648.0 - LMASK);
    for (int i = 0; i < NL - 1; i++) { r.lo[i] = std::floor(a.lo[i] / 2); r.hi[i] = std::floor((a.hi[i] + (double)f29_p(i)) / 2) + 268435456.0; }
    r.lo[NL - 1] = std::floor(a.lo[NL - 1] / 2); r.hi[NL - 1] = std::floor((a.hi[NL - 1] + (double)f29_p(NL - 1)) / 2);
    r.vb = (a.vb + 1) / 2;
    check_limbs(r, "fe_halve limb");
#endif
    return r;
}

// Value reduction without a multiplication: subtract the table row k*p with k = round(top limb / p_8).  Input: weakly normalised,
// |value| < 256p.  Output: |value| < 0.51p, limbs 0..7 within +-(2^29 + 2^9).  Used where a small-constant multiple
// (the non-residue 9+i) would otherwise let the worst-case magnitude compound through the tower.
// k * p for k in [-256, 256] (row k + 256): limbs 0..7 in [0, 2^29), top limb signed (tools/gen_constants.py).  21 KB,
// resident in the caches: k is almost always within a few units of zero.
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ const int32_t F29_KP[513][12] = F29_KP_ROWS;
#else
static const int32_t F29_KP[513][12] = F29_KP_ROWS;
#endif
GPBC_INLINE Fe fe_reduce(const Fe &a) {
#ifdef GPBC_BOUNDS
    if (a.vb >= 256.0) bounds_fail("fe_reduce input value", a.vb, 256.0);
    for (int i = 0; i < NL - 1; i++) if (limb_mag(a, i) > 536870912.0 + 1024) bounds_fail("fe_reduce input limb (normalise first)", limb_mag(a, i), 536870912.0 + 1024);
#endif
    constexpr int32_t P8 = f29_p(NL - 1);
    int32_t k = (int32_t)rintf((float)a.v[NL - 1] * (1.0f / (float)P8));
    k = k < -256 ? -256 : (k > 256 ? 256 : k);          // never taken for in-bound inputs; keeps the row index in the table
    const int32_t *row = F29_KP[k + 256];
    Fe r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = a.v[i] - row[i];
#ifdef GPBC_BOUNDS
    for (int i = 0; i < NL - 1; i++) { r.lo[i] = a.lo[i] - (double)LMASK - 257; r.hi[i] = a.hi[i] + 257; }   // minus a row limb in [0, 2^29) (+-256 for the arithmetic form's carries)
    r.hi[NL - 1] = (double)P8 / 2 + 270; r.lo[NL - 1] = -r.hi[NL - 1];
    r.vb = 0.51;
    bound_stats().reduces++;
    check_limbs(r, "fe_reduce limb");
#endif
    return r;
}

// out = norm(3 t - 2 x - k p): the output step of a cyclotomic squaring (tower29_pair.hip.hpp) with the value reduction folded in
// BEFORE the one normalisation.  t: carry-free normalised (limbs 0..7 within [-2^4, 2^29 + 2^4]); x: normalised the same way (so
// -2x cannot push a limb below -2^30 - 2^5); the row k p has limbs in [0, 2^29): every limb of 3t - 2x - row stays inside
// (-2^31, 2^31).  k comes from the un-normalised top limb — the carries still sitting in limb 7 are worth less than 2^-20 p.
GPBC_INLINE Fe fe_cyclo_out(const Fe &t, const Fe &x) {
    Fe w;
#pragma unroll
    for (int i = 0; i < NL; i++) w.v[i] = 3 * t.v[i] - 2 * x.v[i];
    constexpr int32_t P8 = f29_p(NL - 1);
    int32_t k = (int32_t)rintf((float)w.v[NL - 1] * (1.0f / (float)P8));
    k = k < -256 ? -256 : (k > 256 ? 256 : k);
    const int32_t *row = F29_KP[k + 256];
#pragma unroll
    for (int i = 0; i < NL; i++) w.v[i] -= row[i];
#ifdef GPBC_BOUNDS
    if (3 * t.vb + 2 * x.vb >= 256.0) bounds_fail("fe_cyclo_out input value", 3 * t.vb + 2 * x.vb, 256.0);
    for (int i = 0; i < NL - 1; i++) {
        w.lo[i] = 3 * t.lo[i] - 2 * x.hi[i] - (double)LMASK;
        w.hi[i] = 3 * t.hi[i] - 2 * x.lo[i];
    }
    w.hi[NL - 1] = (double)P8 / 2 + 270; w.lo[NL - 1] = -w.hi[NL - 1];
    w.vb = 0.51;
    bound_stats().reduces++;
    check_limbs(w, "fe_cyclo_out limb");
#endif
    return fe_norm(w);
}

// The same reduction with k*p formed arithmetically (nine 64-bit multiplies with splits, no memory access): used by the
// Miller accumulator, whose memory pipeline is busy streaming the lines — there the table loads cost more than they save
// (measured: k_miller_accumulate 14.6 -> 15.1 ms with the table, k_final_exp 77.6 -> 76.0 ms).
GPBC_INLINE Fe fe_reduce_arith(const Fe &a) {
#ifdef GPBC_BOUNDS
    if (a.vb >= 256.0) bounds_fail("fe_reduce input value", a.vb, 256.0);
    for (int i = 0; i < NL - 1; i++) if (limb_mag(a, i) > 536870912.0 + 1024) bounds_fail("fe_reduce input limb (normalise first)", limb_mag(a, i), 536870912.0 + 1024);
#endif
    constexpr int32_t P8 = f29_p(NL - 1);
    int32_t k = (int32_t)rintf((float)a.v[NL - 1] * (1.0f / (float)P8));
    GPBC_B(bound_stats().mads += NL;)
    Fe r;
    int32_t hi_prev = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int64_t t = (int64_t)k * (int64_t)f29_p(i);
        int32_t lo = (int32_t)(t & LMASK), hi = (int32_t)(t >> LB);
        r.v[i] = (i < NL - 1) ? a.v[i] - lo - hi_prev : a.v[i] - (int32_t)t - hi_prev;
        hi_prev = hi;
    }
#ifdef GPBC_BOUNDS
    for (int i = 0; i < NL - 1; i++) { r.lo[i] = a.lo[i] - (double)LMASK - 257; r.hi[i] = a.hi[i] + 257; }   // minus a row limb in [0, 2^29) (+-256 for the arithmetic form's carries)
    r.hi[NL - 1] = (double)P8 / 2 + 270; r.lo[NL - 1] = -r.hi[NL - 1];
    r.vb = 0.51;
    bound_stats().reduces++;
    check_limbs(r, "fe_reduce limb");
#endif
    return r;
}

// norm(a - k p) with k p formed arithmetically: the value reduction IN FRONT of the normalisation, so that the result comes out of
// fe_norm (limbs 0..7 non-negative, "positive-normalised": what the subtractive Karatsuba forms of the towers need) instead of out
// of a subtraction (signed limbs).  a: any limbs for which a_i - 2^29 - carry stays inside int32 (the interval harness checks);
// k from the un-normalised top limb — what still sits in the lower limbs as carries is worth less than 2^-20 p.
GPBC_INLINE Fe fe_reduce_arith_norm(const Fe &a) {
    constexpr int32_t P8 = f29_p(NL - 1);
    int32_t k = (int32_t)rintf((float)a.v[NL - 1] * (1.0f / (float)P8));
    GPBC_B(bound_stats().mads += NL;)
    Fe r;
    int32_t hi_prev = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int64_t t = (int64_t)k * (int64_t)f29_p(i);
        int32_t lo = (int32_t)(t & LMASK), hi = (int32_t)(t >> LB);
        r.v[i] = (i < NL - 1) ? a.v[i] - lo - hi_prev : a.v[i] - (int32_t)t - hi_prev;
        hi_prev = hi;
    }
#ifdef GPBC_BOUNDS
    if (a.vb >= 256.0) bounds_fail("fe_reduce_arith_norm input value", a.vb, 256.0);
    for (int i = 0; i < NL - 1; i++) { r.lo[i] = a.lo[i] - (double)LMASK - 257; r.hi[i] = a.hi[i] + 257; }
    r.hi[NL - 1] = (double)P8 / 2 + 270; r.lo[NL - 1] = -r.hi[NL - 1];
    r.vb = 0.51;
    bound_stats().reduces++;
    check_limbs(r, "fe_reduce_arith_norm limb");
#endif
    return fe_norm(r);
}

// ------------------------------------------------------------------------------------------------ canonical form, I/O
// Fully reduce to [0,p) with limbs in [0,2^29) (limb 8 < 2^24). Input value must lie in (-4p, 4p).
GPBC_INLINE Fe fe_canonical(const Fe &a) {
    int32_t t[NL];
    int64_t c = 0;
    // add 4p so the value is positive while propagating carries, then subtract p while >= p (at most 8 times)
#pragma unroll
    for (int i = 0; i < NL - 1; i++) { int64_t s = (int64_t)a.v[i] + 4 * (int64_t)f29_p(i) + c; t[i] = (int32_t)(s & LMASK); c = s >> LB; }
    t[NL - 1] = (int32_t)((int64_t)a.v[NL - 1] + 4 * (int64_t)f29_p(NL - 1) + c);
    for (int rep = 0; rep < 8; rep++) {
        int32_t d[NL];
        int32_t b = 0;
#pragma unroll
        for (int i = 0; i < NL - 1; i++) { int32_t s = t[i] - f29_p(i) + b; d[i] = s & LMASK; b = s >> LB; }
        d[NL - 1] = t[NL - 1] - f29_p(NL - 1) + b;
        bool ge = d[NL - 1] >= 0;
#pragma unroll
        for (int i = 0; i < NL; i++) t[i] = ge ? d[i] : t[i];
    }
    Fe r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = t[i];
#ifdef GPBC_BOUNDS
    if (a.vb >= 4.0) bounds_fail("fe_canonical input value", a.vb, 4.0);
    set_class_n(r, 1.0);
#endif
    return r;
}
// x == 0 (mod p) for a weakly normalised x with |value| < 128p.  If x = k p then its top limb lies within |k| + 2
// of k * p_8, so one float quotient and one integer remainder rule out all but ~2^-13 of the non-zero values; the
// survivors take the exact (slow, divergent) path: one Montgomery product by R' mod p, canonical form, compare.
GPBC_INLINE bool fe_is_zero(const Fe &a) {
#ifdef GPBC_BOUNDS
    if (a.vb >= 128.0) bounds_fail("fe_is_zero input value", a.vb, 128.0);
    for (int i = 0; i < NL - 1; i++) if (limb_mag(a, i) > 536870912.0 + 64) bounds_fail("fe_is_zero input limb (normalise first)", limb_mag(a, i), 536870912.0 + 64);
#endif
    constexpr int32_t P8 = f29_p(NL - 1);
    int32_t top = a.v[NL - 1];
    int32_t k = (int32_t)rintf((float)top * (1.0f / (float)P8));
    int32_t r = top - k * P8;
    if (r > 136 || r < -136) return false;
    Fe c = fe_canonical(fe_mul(a, fe_one()));
    int32_t o = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) o |= c.v[i];
    return o == 0;
}

// gnark fp.Element bytes (8 x u32 little-endian, Montgomery R = 2^256, canonical) <-> internal
GPBC_INLINE Fe fe_from_words(const uint32_t w[8]) {
    Fe x;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int bit = LB * i, wi = bit >> 5, sh = bit & 31;
        uint64_t two = (uint64_t)w[wi] | ((wi + 1 < 8) ? ((uint64_t)w[wi + 1] << 32) : 0);
        x.v[i] = (int32_t)((two >> sh) & (uint64_t)LMASK);
    }
    GPBC_B(set_class_n(x, 1.0);)
    constexpr int32_t C[NL] = F29_TO_INTERNAL;
    return fe_mul(x, fe_const(C));
}
GPBC_INLINE void fe_to_words(uint32_t w[8], const Fe &a) {
    constexpr int32_t C[NL] = F29_FROM_INTERNAL;
    Fe x = fe_canonical(fe_mul(a, fe_const(C)));
    uint64_t acc = 0;
    int have = 0, wi = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        acc |= (uint64_t)(uint32_t)x.v[i] << have;
        have += LB;
        if (have >= 32 && wi < 8) { w[wi++] = (uint32_t)acc; acc >>= 32; have -= 32; }
    }
    if (wi < 8) w[wi] = (uint32_t)acc;
}
GPBC_INLINE Fe fe_load(const uint8_t *p) {
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = q[i];
    return fe_from_words(w);
}
GPBC_INLINE void fe_store(uint8_t *p, const Fe &a) {
    uint32_t w[8];
    fe_to_words(w, a);
    uint32_t *q = reinterpret_cast<uint32_t *>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = w[i];
}
GPBC_INLINE bool bytes_all_zero(const uint8_t *p, int n_words) {
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
    uint32_t o = 0;
    for (int i = 0; i < n_words; i++) o |= q[i];
    return o == 0;
}

// x^(p-2) (0 -> 0), kept as the independent check of fe_inv: fixed 4-bit windows over the constant exponent, left to right (254 squarings + <=64 products + 14
// for the table, against 127 products for the plain binary method)
GPBC_NOINLINE Fe fe_inv_fermat(const Fe &x) {
    constexpr int32_t E[NL] = F29_P;              // p - 2: subtract 2 from limb 0 (p's limb 0 is >= 2)
    Fe tab[16];
    tab[0] = fe_one();
    tab[1] = x;
    for (int i = 2; i < 16; i++) tab[i] = fe_mul(tab[i - 1], x);
    auto ebit = [&](int i) -> int {
        if (i < 0 || i >= 254) return 0;
        int limb = i / LB, bit = i % LB;
        int32_t e = E[limb] - (limb == 0 ? 2 : 0);
        return (e >> bit) & 1;
    };
    Fe r = fe_one();
    for (int w = 63; w >= 0; w--) {               // 64 windows cover bits 255..0 (the top two bits are zero)
        if (w != 63) { r = fe_sqr(r); r = fe_sqr(r); r = fe_sqr(r); r = fe_sqr(r); }
        int d = ebit(4 * w) | (ebit(4 * w + 1) << 1) | (ebit(4 * w + 2) << 2) | (ebit(4 * w + 3) << 3);
        if (d) r = fe_mul(r, tab[d]);
    }
    return r;
}

// ------------------------------------------------------------------------------------------------ inversion
// Provenance: the inv30_* routines below restate, for nine 30-bit limbs and the BN254 modulus, the public constant-time
// "modinv32" algorithm of libsecp256k1 (src/modinv32_impl.h, Pieter Wuille, after Bernstein-Yang "Fast constant-time gcd
// computation and modular inversion" 2019 and Thomas Pornin's half-delta variant; MIT licence, Copyright (c) 2020 Peter
// Dettman / Pieter Wuille).  Nothing here comes from the reference repository.
// x^-1 (0 -> 0) by the Bernstein-Yang "safegcd" divsteps in the constant-time half-delta form (Pornin / Wuille:
// zeta = -(delta + 1/2), 590 divsteps suffice below 2^256; here 20 batches of 30 = 600).  Every lane runs the same
// straight-line code — no data-dependent branch — which is what a 64-lane wave needs; a batch is 30 divsteps on the low
// words of (f, g) collecting a 2x2 transition matrix with entries up to 2^30, then one matrix application to the full
// (f, g) and, modulo p, to (d, e), on nine signed 30-bit limbs.  ~14 k instructions against ~70 k for the Fermat power
// below (254 squarings + ~75 products).  The arithmetic is plain two's-complement integer work on canonical values, so
// the magnitude-bound harness has nothing to track here; tests compare it with fe_inv_fermat and with the oracle.
struct Inv30 { int32_t v[9]; };
GPBC_INLINE int32_t inv30_divsteps(int32_t zeta, uint32_t f0, uint32_t g0, int32_t &tu, int32_t &tv, int32_t &tq, int32_t &tr) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 6
    for (int i = 0; i < 30; i++) {
        uint32_t c1 = (uint32_t)(zeta >> 31);                  // zeta < 0
        uint32_t c2 = 0u - (g & 1u);                           // g odd
        uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
        g += x & c2; q += y & c2; r += z & c2;
        c1 &= c2;
        zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;
        f += g & c1; u += q & c1; v += r & c1;
        g >>= 1; u <<= 1; v <<= 1;
    }
    tu = (int32_t)u; tv = (int32_t)v; tq = (int32_t)q; tr = (int32_t)r;
    return zeta;
}
GPBC_INLINE void inv30_update_de(Inv30 &d, Inv30 &e, int32_t u, int32_t v, int32_t q, int32_t r) {
    constexpr int32_t M30 = 0x3fffffff;
    constexpr int32_t MOD[9] = INV30_P;
    const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
    int64_t cd = (int64_t)u * d.v[0] + (int64_t)v * e.v[0];
    int64_t ce = (int64_t)q * d.v[0] + (int64_t)r * e.v[0];
    md -= (int32_t)((INV30_PINV * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
    me -= (int32_t)((INV30_PINV * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
    cd += (int64_t)MOD[0] * md;
    ce += (int64_t)MOD[0] * me;
    cd >>= 30; ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cd += (int64_t)u * d.v[i] + (int64_t)v * e.v[i] + (int64_t)MOD[i] * md;
        ce += (int64_t)q * d.v[i] + (int64_t)r * e.v[i] + (int64_t)MOD[i] * me;
        d.v[i - 1] = (int32_t)cd & M30; cd >>= 30;
        e.v[i - 1] = (int32_t)ce & M30; ce >>= 30;
    }
    d.v[8] = (int32_t)cd;
    e.v[8] = (int32_t)ce;
}
GPBC_INLINE void inv30_update_fg(Inv30 &f, Inv30 &g, int32_t u, int32_t v, int32_t q, int32_t r) {
    constexpr int32_t M30 = 0x3fffffff;
    int64_t cf = (int64_t)u * f.v[0] + (int64_t)v * g.v[0];
    int64_t cg = (int64_t)q * f.v[0] + (int64_t)r * g.v[0];
    cf >>= 30; cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cf += (int64_t)u * f.v[i] + (int64_t)v * g.v[i];
        cg += (int64_t)q * f.v[i] + (int64_t)r * g.v[i];
        f.v[i - 1] = (int32_t)cf & M30; cf >>= 30;
        g.v[i - 1] = (int32_t)cg & M30; cg >>= 30;
    }
    f.v[8] = (int32_t)cf;
    g.v[8] = (int32_t)cg;
}
// d in (-2p, p) -> [0, p), negated first if `sign` is negative (f ended as -1)
GPBC_INLINE void inv30_normalize(Inv30 &r, int32_t sign) {
    constexpr int32_t M30 = 0x3fffffff;
    constexpr int32_t MOD[9] = INV30_P;
    int32_t add = r.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] += MOD[i] & add;
    const int32_t neg = sign >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = (r.v[i] ^ neg) - neg;
#pragma unroll
    for (int i = 0; i < 8; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= M30; }
    add = r.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] += MOD[i] & add;
#pragma unroll
    for (int i = 0; i < 8; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= M30; }
}
GPBC_NOINLINE Fe fe_inv(const Fe &x) {
    // canonical integer A = x * 2^261 mod p, in 30-bit limbs
    Fe c = fe_canonical(fe_mul(x, fe_one()));
    Inv30 g, f, d, e;
    {
        uint64_t acc = 0;
        int have = 0, wi = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            acc |= (uint64_t)(uint32_t)c.v[i] << have;
            have += LB;
            if (have >= 30 && wi < 9) { g.v[wi++] = (int32_t)(acc & 0x3fffffffu); acc >>= 30; have -= 30; }
        }
        if (wi < 9) g.v[wi] = (int32_t)acc;       // 9 x 29 = 261 bits: eight full 30-bit limbs + 21 bits
    }
    constexpr int32_t MOD[9] = INV30_P;
#pragma unroll
    for (int i = 0; i < 9; i++) { f.v[i] = MOD[i]; d.v[i] = 0; e.v[i] = i == 0 ? 1 : 0; }
    int32_t zeta = -1;
    GPBC_B(bound_stats().mads += 20 * (6 * 9 + 4 * 9);)          // the 64-bit limb products of the 20 matrix applications
    for (int it = 0; it < 20; it++) {
        int32_t u, v, q, r;
        zeta = inv30_divsteps(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], u, v, q, r);
        inv30_update_de(d, e, u, v, q, r);
        inv30_update_fg(f, g, u, v, q, r);
    }
    inv30_normalize(d, f.v[8]);
    // back to 29-bit limbs; d = A^-1 in [0, p): one product by 2^783 gives x^-1 * 2^261
    Fe y;
    {
        uint64_t acc = 0;
        int have = 0, wi = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            acc |= (uint64_t)(uint32_t)d.v[i] << have;
            have += 30;
            while (have >= LB && wi < NL) { y.v[wi++] = (int32_t)(acc & (uint64_t)LMASK); acc >>= LB; have -= LB; }
        }
        if (wi < NL) y.v[wi] = (int32_t)acc;
    }
    GPBC_B(set_class_n(y, 1.0);)
    constexpr int32_t RC[NL] = F29_RCUBE;
    return fe_mul(y, fe_const(RC));
}

// Legendre symbol (x / p) by the same machinery: the "posdivsteps" of libsecp256k1's secp256k1_jacobi32_maybe_var (modinv32_impl.h,
// Pieter Wuille, MIT) — divsteps that ADD instead of subtract, so f and g stay non-negative and the symbol can be tracked through
// the quadratic-reciprocity rules (halving g flips it when f = 3, 5 mod 8; swapping flips it when f = g = 3 mod 4) — restated
// here bit by bit with masks: batches of 30 steps with a 2 x 2 transition matrix applied to the full (f, g) per batch, until f = 1.
// The number of steps a pair (p, x) needs has no proven bound of this size (measured on 20 000 random x: 690 .. 844); the function
// answers 0 = "not determined" when f has not reached one within 40 batches (or x = 0), and the caller then takes the power
// x^((p-1)/2).  ~25 k instructions against ~65 k for the power.  Any power of two is a square modulo p (p = 7 mod 8), so the
// Montgomery factors of the representation do not change the symbol.
GPBC_INLINE int32_t jac30_steps(int32_t eta, uint32_t f0, uint32_t g0, int32_t &tu, int32_t &tv, int32_t &tq, int32_t &tr, uint32_t &jac) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 6
    for (int i = 0; i < 30; i++) {
        const uint32_t c1 = 0u - (g & 1u);                     // g odd
        const uint32_t c2 = (uint32_t)(eta >> 31) & c1;         // ... and eta < 0: swap
        jac ^= ((f & g) >> 1) & c2;
        uint32_t t = (f ^ g) & c2; f ^= t; g ^= t;
        t = (u ^ q) & c2; u ^= t; q ^= t;
        t = (v ^ r) & c2; v ^= t; r ^= t;
        eta = (int32_t)(((uint32_t)eta ^ c2) - c2);
        g += f & c1; q += u & c1; r += v & c1;
        g >>= 1; u <<= 1; v <<= 1; eta -= 1;
        jac ^= (f >> 1) ^ (f >> 2);
    }
    tu = (int32_t)u; tv = (int32_t)v; tq = (int32_t)q; tr = (int32_t)r;
    return eta;
}
GPBC_NOINLINE int fe_legendre(const Fe &x) {
    Fe c = fe_canonical(fe_mul(x, fe_one()));
    Inv30 g, f;
    {
        uint64_t acc = 0;
        int have = 0, wi = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            acc |= (uint64_t)(uint32_t)c.v[i] << have;
            have += LB;
            if (have >= 30 && wi < 9) { g.v[wi++] = (int32_t)(acc & 0x3fffffffu); acc >>= 30; have -= 30; }
        }
        if (wi < 9) g.v[wi] = (int32_t)acc;
    }
    constexpr int32_t MOD[9] = INV30_P;
#pragma unroll
    for (int i = 0; i < 9; i++) f.v[i] = MOD[i];
    int32_t eta = -1;
    uint32_t jac = 0;
    // (f, g) = (1, 1) is where every run with gcd 1 ends and stays; on the way there f can pass through 1 and leave it again, so a
    // lane stops at the first batch boundary that finds f = 1 — the symbol is settled at that moment — and the wave runs until its
    // last lane has stopped (27-29 batches for random x) or the budget of 40 is spent
    auto f_is_one = [&]() {
        int32_t rest = f.v[0] ^ 1;
#pragma unroll
        for (int i = 1; i < 9; i++) rest |= f.v[i];
        return rest == 0;
    };
    for (int it = 0; it < 40 && !f_is_one(); it++) {
        int32_t u, v, q, r;
        eta = jac30_steps(eta, (uint32_t)f.v[0] | ((uint32_t)f.v[1] << 30), (uint32_t)g.v[0] | ((uint32_t)g.v[1] << 30), u, v, q, r, jac);
        inv30_update_fg(f, g, u, v, q, r);
    }
    return f_is_one() ? 1 - 2 * (int)(jac & 1u) : 0;
}

}  // namespace gpbc
#endif
