// Miller accumulator and final exponentiation with one Fp12 value per LANE PAIR (tower29_pair.hip.hpp): even lane = C0,
// odd lane = C1.  Same mathematics and operation order as pairing29.hip.hpp (which remains the single-lane form used by
// the line phase, the GT kernels and the host harness); see tower29_pair.hip.hpp for why the state is split.
#ifndef GPBC_PAIRING29_PAIR_HIP_HPP
#define GPBC_PAIRING29_PAIR_HIP_HPP
#include "pairing29.hip.hpp"
#include "tower29_pair.hip.hpp"

namespace gpbc {

// Phase B of the Miller loop on a lane pair; next() yields the 88 lines in order (both lanes read the same line).
// (Folding the tangent and the chord of a non-zero NAF digit into one full product, as miller_accumulate_multi does for the lines
// of two pairs, was measured on this kernel and is SLOWER here: k_miller_accumulate 56.9 -> 67.1 ms per 2^20 pairings — the 23
// extra inlined full products and value reductions cost more in registers and code than the 4 F2 products they save.  Round 3 tried
// it again with the full product as a CALL through memory, like final_exp_pair's: 894 k instead of 922 k VALU instructions per wave,
// but the operands' round trips through private memory doubled the wait share (14 -> 29 % of wave-cycles) and the kernel went from
// 56.1 to 66.5 ms on the same box — profiles/r03_variant_line_pairs.txt.  Two F6 operands, the parked product and the operand sums
// do not fit 256 registers, and this kernel's LDS holds the line stage.)
template <class X, class Src> GPBC_INLINE F6 miller_accumulate_pair(const X &x, Src &&next) {
    // the accumulator stays positive-normalised throughout (PN forms: every step ends in a normalisation), so the F6 products inside
    // run in the subtractive Karatsuba form with no operand normalisations
    LineS l0 = next();
    F6 h = f6_norm(f6_sel(x.odd, F6{l0.c3, l0.c4, f2_zero()}, F6{l0.c0, f2_zero(), f2_zero()}));
    for (int i = BN254_ATE_NAF_LEN - 2; i >= 0; i--) {
        if (i != BN254_ATE_NAF_LEN - 2) {
            h = f12p_sqr<true>(x, h);
            LineS l = next();
            h = f12p_mul_034<true>(x, h, l.c0, l.c3, l.c4);
        }
        if (ate_naf_digit(i) != 0) {
            LineS l = next();
            h = f12p_mul_034<true>(x, h, l.c0, l.c3, l.c4);
        }
    }
    for (int k = 0; k < 2; k++) {
        LineS l = next();
        h = f12p_mul_034<true>(x, h, l.c0, l.c3, l.c4);
    }
    return h;
}

// Phase B for a PRODUCT of Miller functions on one lane pair: F = prod_p f_p obeys F <- F^2 * prod_p l_p, so the m pairs of
// a chunk share the 64 squarings (gnark's multi-pairing does the same; the product of Fp12 values is exact, so the result
// equals multiplying m separately accumulated values bit for bit).  line(p, li) yields line li (0..87, generation order)
// of pair p; m >= 1 (the caller drops pairs with a point at infinity: their Miller value is one).
template <class X, class LineAt> GPBC_INLINE F6 miller_accumulate_multi(const X &x, int m, LineAt &&line) {
    int li = 0;
    // h <- h * prod_{p >= from} l_p(li): the lines of two pairs are multiplied together first (3 F2 products per lane) and
    // enter h by ONE full product (9) — two sparse products would be 8 + 8; an odd one out is a sparse product
    auto mul_lines = [&](F6 h, int from) {
        int p = from;
        for (; p + 1 < m; p += 2) {
            LineS a = line(p, li), b = line(p + 1, li);
            h = f12p_mul(x, h, f12p_mul_034_by_034(x, a.c0, a.c3, a.c4, b.c0, b.c3, b.c4));
        }
        if (p < m) { LineS l = line(p, li); h = f12p_mul_034(x, h, l.c0, l.c3, l.c4); }
        li++;
        return h;
    };
    LineS l0 = line(0, li);
    F6 h = f6_sel(x.odd, F6{l0.c3, l0.c4, f2_zero()}, F6{l0.c0, f2_zero(), f2_zero()});
    h = mul_lines(h, 1);
    for (int i = BN254_ATE_NAF_LEN - 2; i >= 0; i--) {
        if (i != BN254_ATE_NAF_LEN - 2) {
            h = f12p_sqr(x, h);
            if (m >= 2) h = f6_reduce_arith(h);             // the full product's sums need a value-reduced operand (the sparse one does not)
            h = mul_lines(h, 0);
        }
        if (ate_naf_digit(i) != 0) h = mul_lines(h, 0);
    }
    for (int k = 0; k < 2; k++) h = mul_lines(h, 0);
    return h;
}

// The same for lines scaled to c0 = 1 (fixed-Q table: Line34 = the w and vw coefficients).  With five F2 products per lane a
// sparse product is cheaper than half of (line product + full product), so the lines enter one by one.
struct Line34 { F2 c3, c4; };
template <class X, class LineAt> GPBC_INLINE F6 miller_accumulate_multi_34(const X &x, int m, LineAt &&line) {
    int li = 0;
    auto mul_lines = [&](F6 h, int from) {
        for (int p = from; p < m; p++) { Line34 l = line(p, li); h = f12p_mul_34<true>(x, h, l.c3, l.c4); }
        li++;
        return h;
    };
    Line34 l0 = line(0, li);
    F6 h = f6_norm(f6_sel(x.odd, F6{l0.c3, l0.c4, f2_zero()}, F6{f2_one(), f2_zero(), f2_zero()}));   // positive-normalised from here on (PN forms)
    h = mul_lines(h, 1);
    for (int i = BN254_ATE_NAF_LEN - 2; i >= 0; i--) {
        if (i != BN254_ATE_NAF_LEN - 2) h = mul_lines(f12p_sqr<true>(x, h), 0);
        if (ate_naf_digit(i) != 0) h = mul_lines(h, 0);
    }
    for (int k = 0; k < 2; k++) h = mul_lines(h, 0);
    return h;
}

// x^u over the dictionary {x^3, x^15, x^75} (see f12_expt_to).  The squarings run in the alternating-sign form (f12p_cyclo_sqr_run): after
// an odd number of them the lane pair holds the CONJUGATE of the running value; conj(r) conj(t) = conj(r t), so a product takes the
// table entry conjugated exactly when (digit negative) differs from (r conjugated), and one conjugation at the end restores the sign.
template <class X> GPBC_NOINLINE void f12p_expt_to(const X &x, F6 &z, const F6 &b) {
    constexpr int8_t D[GPBC_U_CHAIN_LEN] = GPBC_U_CHAIN;
    // the table entries are kept positive-normalised (one f6_norm each): the products then run in the subtractive-Karatsuba form
    // (f12p_mul<true>) — their other operand comes out of a squaring run, which ends in a normalisation
    F6 tab[3];
    bool flipped = false;
    F6 r = f12p_cyclo_sqr<true>(x, b);
    tab[0] = f6_norm(f12p_mul(x, f6_norm(b), r));                       // x^3
#pragma unroll 1
    for (int k = 1; k < 3; k++) {                                       // x^15 = (x^3)^4 x^3, x^75 = (x^15)^4 x^15 (even runs: not conjugated)
        r = f12p_cyclo_sqr_run(x, tab[k - 1], 2, flipped);
        tab[k] = f6_norm(f12p_mul<true>(x, r, tab[k - 1]));
    }
    r = tab[u_chain_entry(D[GPBC_U_CHAIN_LEN - 1])];
    int run = 0;
    for (int i = GPBC_U_CHAIN_LEN - 2; i >= 0; i--) {
        run++;
        int d = D[i];
        if (d != 0) {
            r = f12p_cyclo_sqr_run(x, r, run, flipped);
            run = 0;
            F6 t = tab[u_chain_entry(d)];
            if ((d < 0) != flipped) t = f6_norm(f12p_conj(x, t));       // (the negated half normalises to non-negative limbs again)
            r = f12p_mul<true>(x, r, t);
        }
    }
    if (run) r = f12p_cyclo_sqr_run(x, r, run, flipped);
    z = flipped ? f12p_conj(x, r) : r;
}

// The sixteen products OUTSIDE x^u as calls of one function: inlined, their ~2 500 glue instructions each made k_final_exp 45 k
// instructions long with 1 141 spilled registers and 4.6 KB of private memory per lane; through memory (two F6 in, one out: ~160
// dwords per call) the kernel body is a few thousand instructions and the frame a third of that.
template <class X> GPBC_NOINLINE void f12p_mul_to(const X &x, F6 &z, const F6 &a, const F6 &b) { z = f12p_mul(x, a, b); }
// x^(s (p^12-1)/r) on a lane pair; operation order of final_exp29.  Eight named F6 values are the whole working set.
template <class X> GPBC_INLINE F6 final_exp_pair(const X &x, const F6 &in) {
    F6 r, t0, t1, t2, t3, t4;                                  // t3 / t4 / t1 double as the short-lived operands (one frame slot less each)
    t3 = f12p_conj(x, in); t4 = f12p_inv(x, in);
    f12p_mul_to(x, t0, t3, t4);
    t3 = f12p_frob(x, t0, 2);
    f12p_mul_to(x, r, t3, t0);
    f12p_expt_to(x, t0, r); t0 = f12p_conj(x, t0);
    t0 = f12p_cyclo_sqr<true>(x, t0);
    t1 = f12p_cyclo_sqr<true>(x, t0);
    f12p_mul_to(x, t1, t0, t1);
    f12p_expt_to(x, t2, t1); t2 = f12p_conj(x, t2);
    t3 = f12p_conj(x, t1);
    f12p_mul_to(x, t1, t2, t3);
    t3 = f12p_cyclo_sqr<true>(x, t2);
    f12p_expt_to(x, t4, t3);
    f12p_mul_to(x, t4, t1, t4);
    f12p_mul_to(x, t3, t0, t4);
    f12p_mul_to(x, t0, t2, t4);
    f12p_mul_to(x, t0, r, t0);
    t2 = f12p_frob(x, t3, 1);
    f12p_mul_to(x, t0, t2, t0);
    t2 = f12p_frob(x, t4, 2);
    f12p_mul_to(x, t0, t2, t0);
    t1 = f12p_conj(x, r);
    f12p_mul_to(x, t2, t1, t3);
    t2 = f12p_frob(x, t2, 3);
    f12p_mul_to(x, t1, t2, t0);
    return t1;
}

// b^k for a 256-bit k (GT.Exp: generic squarings, so any Fp12 element is handled, not only the cyclotomic subgroup;
// no reduction of k): fixed 4-bit windows, left to right.  The table b^0..b^15 (this lane's half of each power) lives in a
// caller-provided block of GT_EXP_TAB_DWORDS int32, one contiguous 256-byte row per entry — not in a private array, whose
// dword-swizzled layout would make every one of the 54 dwords of a per-lane-indexed entry a separate cache line.
constexpr int GT_EXP_ROW_DWORDS = 64, GT_EXP_TAB_DWORDS = 16 * GT_EXP_ROW_DWORDS;
GPBC_INLINE void f6_row_store(int32_t *row, const F6 &v) {
    const Fe *fe[6] = {&v.b0.a0, &v.b0.a1, &v.b1.a0, &v.b1.a1, &v.b2.a0, &v.b2.a1};
    int32_t w[56];
#pragma unroll
    for (int e = 0; e < 6; e++)
#pragma unroll
        for (int i = 0; i < NL; i++) w[e * NL + i] = fe[e]->v[i];
    w[54] = 0; w[55] = 0;
    TabQuad *q = reinterpret_cast<TabQuad *>(row);
#pragma unroll
    for (int j = 0; j < 14; j++) q[j] = TabQuad{w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]};
}
GPBC_INLINE F6 f6_row_load(const int32_t *row) {
    const TabQuad *q = reinterpret_cast<const TabQuad *>(row);
    int32_t w[56];
#pragma unroll
    for (int j = 0; j < 14; j++) { TabQuad t = q[j]; w[4 * j] = t.a; w[4 * j + 1] = t.b; w[4 * j + 2] = t.c; w[4 * j + 3] = t.d; }
    F6 v;
    Fe *fe[6] = {&v.b0.a0, &v.b0.a1, &v.b1.a0, &v.b1.a1, &v.b2.a0, &v.b2.a1};
#pragma unroll
    for (int e = 0; e < 6; e++) {
#pragma unroll
        for (int i = 0; i < NL; i++) fe[e]->v[i] = w[e * NL + i];
#ifdef GPBC_BOUNDS
        // table entries are one, the base (normalised, limbs in [0, 2^29)) or outputs of f12p_mul (value-reduced: limbs
        // within +-(2^29 + 2^10), |value| < 0.51 p): the union of those intervals
        set_class_n(*fe[e], 1.5);
        for (int i = 0; i < NL - 1; i++) { fe[e]->lo[i] = -((double)LMASK + 1024); fe[e]->hi[i] = (double)LMASK + 1024; }
        check_limbs(*fe[e], "GT.Exp table entry");
#endif
    }
    return v;
}
template <class X> GPBC_NOINLINE F6 f12p_exp256(const X &x, const F6 &b, const uint32_t (&k)[8], int32_t *tab) {
    F6 cur = f12p_one(x);
    f6_row_store(tab, cur);
    cur = b;
    f6_row_store(tab + GT_EXP_ROW_DWORDS, cur);
    for (int i = 2; i < 16; i++) { cur = f12p_mul(x, cur, b); f6_row_store(tab + i * GT_EXP_ROW_DWORDS, cur); }
    // a wavefront whose bases all lie in the cyclotomic subgroup (pairing values: what the reference's call sites raise) squares by
    // Granger-Scott, 3.5 k instead of 6 k instructions per squaring; one base outside it (a Miller value, a random Fp12 element —
    // gnark's Exp takes those too) and the whole wavefront squares generically, so that no lane pair waits for another's path
    const bool cyclotomic = x.all(f12p_is_cyclotomic(x, b));
    F6 r = f6_row_load(tab + ((k[7] >> 28) & 15) * GT_EXP_ROW_DWORDS);
    for (int w = 62; w >= 0; w--) {
        if (cyclotomic) { bool flipped = false; r = f12p_cyclo_sqr_run(x, r, 4, flipped); }
        else for (int s = 0; s < 4; s++) r = f6_reduce(f12p_sqr(x, r));
        int d = (k[w >> 3] >> (4 * (w & 7))) & 15;
        r = f12p_mul(x, r, f6_row_load(tab + d * GT_EXP_ROW_DWORDS));      // d = 0 multiplies by one: no divergence
    }
    return r;
}

GPBC_INLINE F6 f6_load(const uint8_t *p) { return F6{f2_load(p), f2_load(p + 64), f2_load(p + 128)}; }
GPBC_INLINE void f6_store(uint8_t *p, const F6 &z) { f2_store(p, z.b0); f2_store(p + 64, z.b1); f2_store(p + 128, z.b2); }

}  // namespace gpbc
#endif
