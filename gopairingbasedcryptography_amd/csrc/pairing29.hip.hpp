// Optimal-ate pairing on BN254 over the 29-bit-limb tower: Miller loop + final exponentiation, one pairing per lane.
// Replaces the inside of gnark-crypto's bn254.Pair / PairingCheck as called by the reference at
// cpabe/bsw07/bsw07_cpabe.go:75,184, access/tree/access_tree_node.go:106,110,119,
// bibe/afp25_bibe/afp25_bibe.go:227,395,399,403, signature/bls01_signature/bls_signature.go:81.
// The GT value is the unique residue e(P,Q)^s (s = 2u(6u^2+3u+1)), so any correct
// evaluation order gives gnark's bytes once the result is made canonical.
#ifndef GPBC_PAIRING29_HIP_HPP
#define GPBC_PAIRING29_HIP_HPP
#include "bn254_constants.hip.hpp"
#include "tower29.hip.hpp"

namespace gpbc {

struct G1A { Fe x, y; };
struct G2A { F2 x, y; };
struct G2P { F2 x, y, z; };     // homogeneous projective
struct LineE { F2 r0, r1, r2; };  // l = r0*yP + r1*xP*w + r2*w^3

GPBC_INLINE int ate_naf_digit(int i) { constexpr int8_t D[BN254_ATE_NAF_LEN] = BN254_ATE_NAF; return D[i]; }

// (9 - i)(a0 + a1 i) = (9 a0 + a1) + (9 a1 - a0) i: the curve coefficient of the isomorphic twist (below), value-reduced in front of
// its one normalisation (|value| < 0.51 p, limbs 0..7 non-negative like a product's: the sums B + 3E of the doubling sit one unit
// below 2^31 and have no room for signed limbs) — what the product by b' returned, without the product.  Input N-class.
GPBC_INLINE F2 f2_mul_9mi_n(const F2 &x) {
    return F2{fe_reduce_arith_norm(fe_add(fe_add(fe_mul8_norm(x.a0), x.a0), x.a1)), fe_reduce_arith_norm(fe_sub(fe_add(fe_mul8_norm(x.a1), x.a1), x.a0))};
}
// Tangent line at T and T <- 2T (Costello-Lange-Naehrig, ePrint 2013/722 §4.3, a = 0 twist). T in/out N-class.
// ISO: the walk runs on the isomorphic twist y^2 = x^3 + (9 - i) (miller_lines), whose coefficient costs additions, not an F2 product.
template <bool ISO = false> GPBC_INLINE void g2_double_step(G2P &t, LineE &l) {
    F2 A = f2_halve(f2_mul(t.x, t.y));
    F2 B = f2_sqr(t.y);
    F2 C = f2_sqr(t.z);
    F2 C3 = f2_norm(f2_add(f2_dbl(C), C));
    F2 E = ISO ? f2_mul_9mi_n(C3) : f2_mul(C3, b_twist29());
    F2 F = f2_add(f2_dbl(E), E);
    // G = (B + 3E) / 2.  A product's limbs are masked (below 2^29) and B + 3E fits int32 by four units; E out of fe_norm may sit a carry
    // above 2^29, so the ISO form takes G = E + (B + E) / 2 (f2_sqr_n normalises its operand either way)
    F2 G = ISO ? f2_add(E, f2_halve(f2_norm(f2_add(B, E)))) : f2_halve(f2_norm(f2_add(B, F)));
    F2 H = f2_sub(f2_sqr_n(f2_add(t.y, t.z)), f2_add(B, C));
    F2 J = f2_sqr(t.x);
    F2 EE = f2_sqr(E);
    t.x = f2_mul(A, f2_norm(f2_sub(B, F)));
    t.y = f2_norm(f2_sub(f2_sqr_n(G), f2_add(f2_dbl(EE), EE)));
    t.z = f2_mul(B, f2_norm(H));
    l.r0 = f2_neg(H);                        // (-2^29, 2^30): a line coefficient only enters products with normalised partners
    l.r1 = f2_norm(f2_add(f2_dbl(J), J));
    l.r2 = f2_sub(E, B);
}
// Chord through T and affine Q, T <- T + Q
GPBC_INLINE void g2_add_step(G2P &t, LineE &l, const G2A &q) {
    F2 O = f2_sub(t.y, f2_mul(q.y, t.z));     // differences of two normalised values (limbs within +-2^29) go into the
    F2 L = f2_sub(t.x, f2_mul(q.x, t.z));     // products as they are: signed columns stay below 27 * 2^58 (interval harness)
    F2 C = f2_sqr(O);
    F2 D = f2_sqr(L);
    F2 E = f2_mul(L, D);
    F2 F = f2_mul(t.z, C);
    F2 G = f2_mul(t.x, D);
    F2 H = f2_norm(f2_sub(f2_add(E, F), f2_dbl(G)));
    F2 t1 = f2_mul(t.y, E);
    t.x = f2_mul(L, H);
    t.y = f2_norm(f2_sub(f2_mul(f2_sub(G, H), O), t1));
    t.z = f2_mul(E, t.z);
    l.r0 = L;
    l.r1 = f2_neg(O);
    l.r2 = f2_sub(f2_mul(q.x, O), f2_mul(L, q.y));
}
// The chord alone (the last step of the loop: T is not needed afterwards) — 4 of the 13 products of g2_add_step
GPBC_INLINE void g2_line_step(const G2P &t, LineE &l, const G2A &q) {
    F2 O = f2_sub(t.y, f2_mul(q.y, t.z));
    F2 L = f2_sub(t.x, f2_mul(q.x, t.z));
    l.r0 = L;
    l.r1 = f2_neg(O);
    l.r2 = f2_sub(f2_mul(q.x, O), f2_mul(L, q.y));
}
// A Miller step's line already evaluated at P: l = c0 + c3 w + c4 v w with c0 = r0*yP, c3 = r1*xP, c4 = r2
struct LineS { F2 c0, c3, c4; };
constexpr int MILLER_LINES = 88;      // 65 tangent lines + 21 chords (non-zero NAF digits below the top) + 2 Frobenius chords
constexpr int LINE_WORDS = 6 * NL;    // 54 int32 per line in internal limb form

GPBC_INLINE LineS line_scale(const LineE &l, const G1A &p) { return LineS{f2_mul_fe(l.r0, p.y), f2_mul_fe(l.r1, p.x), l.r2}; }

// Phase A of the Miller loop — the G2 arithmetic.  The sequence of lines depends only on (P, Q), never on the
// accumulator f, so it runs on its own (own kernel, own register budget) and hands the 88 lines to phase B.
// sink(LineS) is called once per line, in evaluation order.  Caller has checked neither point is infinity.
// The walk runs on an ISOMORPHIC curve pair: (x, y) -> (t^2 x, t^3 y), t in Fp with t^6 = 82/3, takes E: y^2 = x^3 + 3 and the twist
// E': y^2 = x^3 + 3/(9+i) to y^2 = x^3 + 82 and y^2 = x^3 + (9 - i), where the doubling's product by the twist coefficient is a few
// additions (f2_mul_9mi_n) instead of one of its four F2 products.  P and Q are mapped once (six Fp products); the Frobenius
// endpoints commute with the map (t is in Fp).  Every line then equals gnark's times a power of t that depends on the step alone —
// an Fp factor, which the final exponentiation removes — and the LAST line carries the constant t^-N (F29_ISO_KFIX, derived in
// tools/gen_constants.py iso_twist_constants) that cancels their product, so that the Miller value itself, not only the pairing, is
// the one the other forms and gnark's MillerLoop give.
template <class Sink> GPBC_INLINE void miller_lines(const G1A &p0, const G2A &q0, Sink &&sink) {
    constexpr int32_t T2[NL] = F29_ISO_T2, T3[NL] = F29_ISO_T3, KF[NL] = F29_ISO_KFIX;
    const Fe t2 = fe_const(T2), t3 = fe_const(T3);
    const G1A p{fe_mul(p0.x, t2), fe_mul(p0.y, t3)};
    const G2A q{f2_mul_fe(q0.x, t2), f2_mul_fe(q0.y, t3)};
    G2P t{q.x, q.y, f2_one()};
    const F2 ny = f2_neg(q.y);
    LineE l;
    for (int i = BN254_ATE_NAF_LEN - 2; i >= 0; i--) {
        g2_double_step<true>(t, l);
        sink(line_scale(l, p));
        int d = ate_naf_digit(i);
        if (d != 0) {
            g2_add_step(t, l, G2A{q.x, f2_sel(d > 0, q.y, ny)});
            sink(line_scale(l, p));
        }
    }
    G2A q1{f2_mul(f2_conj(q.x), gamma29(1, 2)), f2_mul(f2_conj(q.y), gamma29(1, 3))};
    G2A q2{f2_mul(q.x, gamma29(2, 2)), f2_neg(f2_mul(q.y, gamma29(2, 3)))};
    g2_add_step(t, l, q1);
    sink(line_scale(l, p));
    g2_line_step(t, l, q2);
    const Fe k = fe_const(KF);
    sink(LineS{f2_mul_fe(l.r0, fe_mul(p.y, k)), f2_mul_fe(l.r1, fe_mul(p.x, k)), f2_mul_fe(l.r2, k)});
}

// The same walk WITHOUT the evaluation point: the raw coefficients (r0, r1, r2) of the 88 lines depend on Q alone, so a
// Q that is paired with many P's (a decryption key against many ciphertexts, a fixed public key) needs them once.
template <class Sink> GPBC_INLINE void miller_lines_raw(const G2A &q, Sink &&sink) {
    G2P t{q.x, q.y, f2_one()};
    const F2 ny = f2_neg(q.y);
    LineE l;
    for (int i = BN254_ATE_NAF_LEN - 2; i >= 0; i--) {
        g2_double_step(t, l);
        sink(l);
        int d = ate_naf_digit(i);
        if (d != 0) {
            g2_add_step(t, l, G2A{q.x, f2_sel(d > 0, q.y, ny)});
            sink(l);
        }
    }
    G2A q1{f2_mul(f2_conj(q.x), gamma29(1, 2)), f2_mul(f2_conj(q.y), gamma29(1, 3))};
    G2A q2{f2_mul(q.x, gamma29(2, 2)), f2_neg(f2_mul(q.y, gamma29(2, 3)))};
    g2_add_step(t, l, q1);
    sink(l);
    g2_line_step(t, l, q2);
    sink(l);
}

// 1 / y_j for cnt <= G values by ONE inversion (Montgomery's trick: prefix products, invert, walk back); a zero takes the place
// of a one in the chain so that it cannot spoil its neighbours (its own result is then 1 — the caller flags such points).
// y(j) yields the j-th value (canonical), emit(j, inverse) receives the results, last first.
template <int G, class LoadY, class Emit> GPBC_INLINE void fe_batch_inverse(int cnt, LoadY &&y_at, Emit &&emit) {
    Fe y[G], pre[G];
#pragma unroll
    for (int j = 0; j < G; j++) {
        y[j] = fe_one();
        if (j < cnt) {
            Fe v = y_at(j);
            if (!fe_is_zero(v)) y[j] = v;
        }
        pre[j] = j ? fe_mul(pre[j - 1], y[j]) : y[j];
    }
    Fe inv = fe_inv(pre[G - 1]);
#pragma unroll
    for (int j = G - 1; j >= 0; j--) {
        Fe r = j ? fe_mul(inv, pre[j - 1]) : inv;
        inv = fe_mul(inv, y[j]);
        if (j < cnt) emit(j, r);
    }
}

GPBC_INLINE F12 f12_from_line(const LineS &l) {
    return F12{F6{l.c0, f2_zero(), f2_zero()}, F6{l.c3, l.c4, f2_zero()}};
}
// Phase B — the accumulator: f <- f^2 * l per doubling, f <- f * l per chord.  next() yields the lines in order.
template <class Src> GPBC_INLINE F12 miller_accumulate(Src &&next) {
    F12 f = f12_from_line(next());                 // first doubling: 1^2 * l
    for (int i = BN254_ATE_NAF_LEN - 2; i >= 0; i--) {
        if (i != BN254_ATE_NAF_LEN - 2) {
            f = f12_sqr(f);
            LineS l = next();
            f = f12_mul_034(f, l.c0, l.c3, l.c4);
        }
        if (ate_naf_digit(i) != 0) {
            LineS l = next();
            f = f12_mul_034(f, l.c0, l.c3, l.c4);
        }
    }
    for (int k = 0; k < 2; k++) {
        LineS l = next();
        f = f12_mul_034(f, l.c0, l.c3, l.c4);
    }
    return f;
}

// Both phases in one call on one lane.  The kernels do not use this form (phase B and the final exponentiation run on
// lane pairs, pairing29_pair.hip.hpp); it is the single-lane statement of the same mathematics that tools/bounds_check.cpp
// runs under bound instrumentation and compares with the oracle.
GPBC_INLINE F12 miller_loop29(const G1A &p, const G2A &q) {
    LineS lines[MILLER_LINES];
    int n = 0;
    miller_lines(p, q, [&](const LineS &l) { lines[n++] = l; });
    int k = 0;
    return miller_accumulate([&]() -> LineS { return lines[k++]; });
}

// ------------------------------------------------------------------------------------------- final exponentiation
// F12-level steps (kept inline: measured on MI355X, staging F12 values through private memory between real function
// calls was slower — 140 ms vs 118 ms per 2^20 final exponentiations).
GPBC_INLINE void f12_mul_to(F12 &z, const F12 &x, const F12 &y) { z = f12_mul(x, y); }
// n cyclotomic squarings in place; the value reduction runs on every second squaring (and always on the last)
GPBC_INLINE void f12_cyclo_sqr_n(F12 &z, int n) {
    F12 r = z;
    if (n & 1) r = f12_cyclo_sqr_t<true>(r);
    for (int i = 0; i < n / 2; i++) {
        r = f12_cyclo_sqr_t<false>(r);
        r = f12_cyclo_sqr_t<true>(r);
    }
    z = r;
}
GPBC_INLINE void f12_frob_to(F12 &z, const F12 &x, int j) { z = f12_frob(x, j); }
GPBC_INLINE void f12_inv_to(F12 &z, const F12 &x) { z = f12_inv(x); }

// z = x^u, u = 0x44e992b44a6909f1, x in the cyclotomic subgroup (inverse = conjugate, free).  Signed digits over the dictionary
// {x^3, x^15, x^75} (GPBC_U_CHAIN, tools/gen_constants.py: dictionary_digits): x^3 = x^2 x, x^15 = (x^3)^4 x^3, x^75 = (x^15)^4 x^15, then
// 56 squarings and 10 products — 61 squarings + 13 products in all.  Width-4 windows (rounds 1-3) took 63 + 16, gnark's unsigned chain
// 62 + 17; a product costs 2.3 cyclotomic squarings, and tools/u_chain_search.py found nothing below 13 products.
// A real function: it is used three times by the hard part and its table lives in private memory.  (THREE entries, 648 B, on purpose:
// a table of two — {x^17, x^35}: 62 + 13 — is below the compiler's promote-alloca budget of 512 B, becomes 108 spilled registers, and
// tab[e] becomes loads of both entries and 54 selects each behind its own wait: k_final_exp's memory waits went from 14 to 28 % of the
// wave cycles and the kernel was SLOWER with 7 % fewer instructions, profiles/r04_variant_u_chain.txt.)
GPBC_INLINE int u_chain_entry(int d) { const int a = d < 0 ? -d : d; return (a > 3) + (a > 15); }
GPBC_NOINLINE void f12_expt_to(F12 &z, const F12 &x) {
    constexpr int8_t D[GPBC_U_CHAIN_LEN] = GPBC_U_CHAIN;    // (sum d_i 2^i == u is asserted in tests/test_device_math_bounds.py)
    F12 tab[3];
    F12 r = f12_cyclo_sqr(x);
    tab[0] = f12_mul(r, x);
    for (int k = 1; k < 3; k++) {
        r = tab[k - 1];
        f12_cyclo_sqr_n(r, 2);
        tab[k] = f12_mul(r, tab[k - 1]);
    }
    r = tab[u_chain_entry(D[GPBC_U_CHAIN_LEN - 1])];         // top digit is positive
    int run = 0;
    for (int i = GPBC_U_CHAIN_LEN - 2; i >= 0; i--) {
        run++;
        int d = D[i];
        if (d != 0) {
            f12_cyclo_sqr_n(r, run);
            run = 0;
            F12 t = tab[u_chain_entry(d)];
            if (d < 0) t = f12_conj(t);
            r = f12_mul(r, t);
        }
    }
    if (run) f12_cyclo_sqr_n(r, run);
    z = r;
}

// x^(s (p^12-1)/r): easy part, then the Fuentes-Castaneda hard part (gnark's operation order).
// gnark returns early when the easy part gives 1; the hard part maps 1 to 1, so the early exit is not needed here.
GPBC_INLINE F12 final_exp29(const F12 &x) {
    F12 r, t0, t1, t2, t3, t4;
    f12_inv_to(t1, x);
    t0 = f12_conj(x);
    f12_mul_to(t0, t0, t1);
    f12_frob_to(r, t0, 2);
    f12_mul_to(r, r, t0);
    f12_expt_to(t0, r); t0 = f12_conj(t0);
    f12_cyclo_sqr_n(t0, 1);
    t1 = t0; f12_cyclo_sqr_n(t1, 1);
    f12_mul_to(t1, t0, t1);
    f12_expt_to(t2, t1); t2 = f12_conj(t2);
    t3 = f12_conj(t1);
    f12_mul_to(t1, t2, t3);
    t3 = t2; f12_cyclo_sqr_n(t3, 1);
    f12_expt_to(t4, t3);
    f12_mul_to(t4, t1, t4);
    f12_mul_to(t3, t0, t4);
    f12_mul_to(t0, t2, t4);
    f12_mul_to(t0, r, t0);
    f12_frob_to(t2, t3, 1);
    f12_mul_to(t0, t2, t0);
    f12_frob_to(t2, t4, 2);
    f12_mul_to(t0, t2, t0);
    t2 = f12_conj(r);
    f12_mul_to(t2, t2, t3);
    f12_frob_to(t2, t2, 3);
    f12_mul_to(t0, t2, t0);
    return t0;
}

}  // namespace gpbc
#endif
