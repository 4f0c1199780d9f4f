// Bucket (Pippenger) multi-scalar multiplication  sum_i [s_i] P_i  over variable bases, G1 and G2 — the verifier's sums of
// BLS aggregate verification (BASELINE config 3: sum rho_i pk_i, sum rho_i sigma_i) and the commitment sums
// SURVEY §8f-3 names ("a real G1 MSM (Pippenger) on GPU instead of B separate scalar mults").  The reference computes such sums
// as loops of ScalarMultiplication + Add (gka/agka09/asbb.go:193-220, bibe/afp25_bibe/afp25_bibe_utils.go:44-55); the affine
// result is the unique point, so any correct evaluation order is bit-identical.
//
// Shape of the computation (c-bit windows, W = ceil(256 / c) windows, digits d_{i,w} of the plain 256-bit scalars):
//   1. counting sort of the (window, digit) keys of all non-zero digits            -> idx[] grouped by bucket
//   2. bucket sums      B_{w,d} = sum_{i : d_{i,w} = d} P_i                        one lane per bucket, mixed additions
//   3. group reduction  for G consecutive digits [lo, lo+G):  sum_d d B_{w,d} = sum_d (d - lo + 1) B_{w,d} + (lo - 1) sum_d B_{w,d}
//                       (running sums + one multiplication by the small integer lo - 1)   one lane per (window, group)
//   4. tree sum of the groups of a window, then  result = sum_w 2^(c w) S_w  by Horner (c doublings per window), affine output
// This header holds the per-lane pieces; csrc/gpbc_msm.hip has the kernels and the sort, tools/bounds_check.cpp runs the same
// pieces on the host under the interval harness.
#ifndef GPBC_MSM29_HIP_HPP
#define GPBC_MSM29_HIP_HPP
#include "curve29.hip.hpp"

namespace gpbc {

constexpr int MSM_GROUP = 8;                     // digits per group of step 3 (2^c / 8 lanes per window: a full round of the chip at c = 16)

// digit w (c bits, c <= 16) of a 256-bit little-endian scalar held as eight 32-bit words
GPBC_INLINE uint32_t msm_digit(const uint32_t (&k)[8], int w, int c) {
    const int bit = w * c;
    if (bit >= 256) return 0;
    const int wi = bit >> 5, sh = bit & 31;
    uint64_t two = (uint64_t)k[wi] | ((wi + 1 < 8) ? ((uint64_t)k[wi + 1] << 32) : 0);
    return (uint32_t)((two >> sh) & ((1u << c) - 1u));
}

// Jacobian points in HBM between the steps: internal limbs, one 128-byte (G1) / 256-byte (G2) row per point.  (Device only: the
// bounds harness keeps the points as objects so that their tracked intervals flow from step to step.)
template <class F> struct JacRow { static constexpr int DWORDS = sizeof(F) == sizeof(Fe) ? 32 : 64; };
#ifndef GPBC_BOUNDS
GPBC_INLINE void limbs_store(int32_t *p, const Fe &a) { for (int i = 0; i < NL; i++) p[i] = a.v[i]; }
GPBC_INLINE void limbs_store(int32_t *p, const F2 &a) { limbs_store(p, a.a0); limbs_store(p + NL, a.a1); }
GPBC_INLINE void limbs_load(Fe &a, const int32_t *p) { for (int i = 0; i < NL; i++) a.v[i] = p[i]; }
GPBC_INLINE void limbs_load(F2 &a, const int32_t *p) { limbs_load(a.a0, p); limbs_load(a.a1, p + NL); }
template <class F> GPBC_INLINE void jac_row_store(int32_t *row, const JacP<F> &p) {
    constexpr int E = sizeof(F) / sizeof(Fe) * NL;
    limbs_store(row, p.x); limbs_store(row + E, p.y); limbs_store(row + 2 * E, p.z);
    row[3 * E] = p.inf ? 1 : 0;
}
template <class F> GPBC_INLINE void jac_row_load(JacP<F> &p, const int32_t *row) {
    constexpr int E = sizeof(F) / sizeof(Fe) * NL;
    p.inf = row[3 * E] != 0;
    if (p.inf) { jac_set_inf(p); return; }
    limbs_load(p.x, row); limbs_load(p.y, row + E); limbs_load(p.z, row + 2 * E);
}
#endif

// step 2: the sum of one bucket's points; load(j) yields the j-th affine point of the bucket
template <class F, class Load> GPBC_INLINE void msm_bucket_sum(JacP<F> &acc, size_t lo, size_t hi, Load &&load) {
    jac_set_inf(acc);
    for (size_t j = lo; j < hi; j++) jac_add_mixed(acc, acc, load(j));
}

// [k] p for a small non-negative integer k (binary, most significant bit first)
template <class F> GPBC_INLINE void jac_mul_small(JacP<F> &r, const JacP<F> &p, uint32_t k) {
    jac_set_inf(r);
    if (p.inf || k == 0) return;
    int top = 31;
    while (!((k >> top) & 1)) top--;
    r = p;
    for (int b = top - 1; b >= 0; b--) {
        JacP<F> t;
        jac_dbl(t, r);
        r = t;
        if ((k >> b) & 1) { jac_add(t, r, p); r = t; }
    }
}

// step 3 for the digits [lo, hi] of one window (lo >= 1): out = sum_{d = lo..hi} d * B_d; bucket(d) yields B_d
template <class F, class Bucket> GPBC_INLINE void msm_group_reduce(JacP<F> &out, uint32_t lo, uint32_t hi, Bucket &&bucket) {
    JacP<F> running, total, t;
    jac_set_inf(running); jac_set_inf(total);
    for (uint32_t d = hi; d >= lo; d--) {
        jac_add(t, running, bucket(d)); running = t;          // running = sum_{e >= d} B_e
        jac_add(t, total, running); total = t;                // total   = sum_e (e - d + 1) B_e
        if (d == lo) break;
    }
    if (lo > 1) {
        jac_mul_small(t, running, lo - 1);
        JacP<F> s;
        jac_add(s, total, t);
        total = s;
    }
    out = total;
}

// step 4, Horner over the windows: acc <- [2^c] acc + s
template <class F> GPBC_INLINE void msm_horner_step(JacP<F> &acc, const JacP<F> &s, int c) {
    JacP<F> t;
    for (int i = 0; i < c; i++) { jac_dbl(t, acc); acc = t; }
    jac_add(t, acc, s);
    acc = t;
}

}  // namespace gpbc
#endif
