// LATENCY form of the pairing: ONE pairing per WAVEFRONT, the 64 lanes working on the F2 products INSIDE it.
//
// Every reference call site is a single bn254.Pair / PairingCheck.  In the throughput kernels such a call is one
// lane pair walking a chain of ~2.1 M dependent instructions (accumulator 0.93 M + final exponentiation 1.2 M) while a lone wave
// issues one instruction every 5-9 cycles: ~5.8 ms whatever the batch size up to a few thousand pairs.  Here the
// chain is cut by spending lanes: an Fp12 value lives in LDS as six F2 slots, a product of two values is 36 F2 products on 36
// lanes (schoolbook in the w-basis, w^6 = xi: no operand sums to form) followed by six lanes that add up one output coefficient
// each; a cyclotomic squaring is nine squarings on nine lanes; the G2 doubling / addition steps are three / four rounds of
// independent F2 products.  The chain per Fp12 product is one F2 leaf + one recombination (~1.4 k instructions instead of ~8.3 k).
//
// Same arithmetic leaves, same tower conventions, same canonical outputs as the throughput path (parity tests compare the two
// and the oracle); the code is written once over a memory policy M — LDS on the device, an array with interval-carrying values
// under the host harness (tools/bounds_check.cpp), where the lanes of a phase run one after the other.
//   M::ld(slot) -> F2        M::st(slot, F2)        M::run(n, body): body(l) for lanes l < n, then a barrier
//   M::ldh(slot, half) -> Fe    M::sth(slot, half, Fe): one half (0: a0, 1: a1) of a slot
// A lone wave issues one v_mad_i64_i32 per ~9 cycles, so wherever the lanes are there the two HALVES of an F2 product go to two lanes
// (r0 = a0 b0 - a1 b1, r1 = a0 b1 + a1 b0: one 243-MAD leaf each instead of one 486-MAD leaf), and the recombinations run per Fp
// component as well: sums on 24 lanes, the xi step and the reduction on 12.
#ifndef GPBC_WIDE29_HIP_HPP
#define GPBC_WIDE29_HIP_HPP
#include "pairing29.hip.hpp"

namespace gpbc {

// ---- slot map (one F2 per slot).  An Fp12 value is six consecutive slots in tower order C0.b0 C0.b1 C0.b2 C1.b0 C1.b1 C1.b2.
constexpr int W_VALUES = 12;                         // values 0..11: the accumulator / final-exponentiation registers
constexpr int W_PROD = 6 * W_VALUES;                 // 36 product slots
constexpr int W_TX = W_PROD + 36, W_TY = W_TX + 1, W_TZ = W_TX + 2;              // G2 accumulator T (projective)
constexpr int W_QX = W_TX + 3, W_QY = W_TX + 4, W_NQY = W_TX + 5;                // Q, -Q.y
constexpr int W_Q1X = W_TX + 6, W_Q1Y = W_TX + 7, W_Q2X = W_TX + 8, W_Q2Y = W_TX + 9;   // pi(Q), -pi^2(Q)
constexpr int W_XP = W_TX + 10, W_YP = W_TX + 11;                                // P as (xP, 0), (yP, 0)
constexpr int W_L0 = W_TX + 12, W_L3 = W_TX + 13, W_L4 = W_TX + 14;              // the current line c0, c3, c4
constexpr int W_G = W_TX + 15;                       // 12 scratch slots of the point steps
constexpr int W_CL = W_G + 12;                       // the accumulator's copy of the line it is multiplying by (c0, c3, c4)
// operands the linear phases of the point steps leave, normalised, for the product rounds (the G2 walk's own: the accumulator wave never touches them)
constexpr int W_YZ = W_CL + 3;                       // y + z of T
constexpr int W_HN = W_YZ + 1, W_C3 = W_YZ + 2, W_J3 = W_YZ + 3, W_AH = W_YZ + 4, W_BF = W_YZ + 5, W_GG = W_YZ + 6;   // doubling: H, 3C, 3J, XY/2, B - F, (B + F)/2
constexpr int W_HH = W_YZ + 7, W_GH = W_YZ + 8;      // addition: H, G - H
constexpr int W_SLOTS = W_YZ + 9;
constexpr int wv(int n) { return 6 * n; }
GPBC_INLINE int w2t(int k) { return (k >> 1) + 3 * (k & 1); }      // coefficient of w^k -> tower slot offset (0,3,1,4,2,5)

// ---- linear phases, ONE LIMB PER LANE.  A lone wave pays ~5 cycles per instruction whatever it is, so sums, normalisations and
// halvings done on whole elements (27-40 instructions each, on a handful of lanes) cost as much as the products between them.  A linear
// phase is written once as a function of an operations object `o` and a component number c (= output half: slot * 2 + half):
//     m.limbs(n_components, [&](auto &o, int c) { ... o.ld(slot, h) ... o.add / sub / dbl / neg / sel / norm / halve ... o.st(slot, h, v); });
// On the device `o` works on ONE limb (lane = c * NL + limb; a normalisation takes the carry from the lane below through a DPP
// wave shift, a halving the parity bit from the lane above): WideLds::LimbOps in gpbc_pairing.hip.  Under the host harness `o` is
// LimbOpsFe below: the same calls on whole elements with their tracked intervals — limb for limb the same values, so what the harness
// proves holds for the device form.  At most 7 components per phase (63 lanes).
template <class M> struct LimbOpsFe {
    using V = Fe;
    M &m;
    V ld(int slot, int h) const { return m.ldh(slot, h); }
    void st(int slot, int h, const V &v) const { m.sth(slot, h, v); }
    static V add(const V &a, const V &b) { return fe_add(a, b); }
    static V sub(const V &a, const V &b) { return fe_sub(a, b); }
    static V dbl(const V &a) { return fe_dbl(a); }
    static V neg(const V &a) { return fe_neg(a); }
    // a choice by COMPONENT NUMBER (which output this is): the host knows the component, so the bound is the chosen value's — fe_sel's
    // "either lane" bound would price every output at the largest candidate.  (The device picks per lane; same values.)
    static V sel(bool c, const V &a, const V &b) { return c ? a : b; }
    static V norm(const V &a) { return fe_norm(a); }
    static V halve(const V &a) { return fe_halve(a); }
};

// ---- half products: lane (product p, half h) of a product phase
// h = 0: a0 b0 - a1 b1      h = 1: a0 b1 + a1 b0          (a, b N-class)
GPBC_INLINE Fe wide_half_mul(const F2 &a, const F2 &b, bool h) {
    return fe_mul2_l(a.a0, fe_sel(h, b.a1, b.a0), fe_sel(h, a.a1, fe_neg(a.a1)), fe_sel(h, b.a0, b.a1));
}
// h = 0: (x0 + x1)(x0 - x1)      h = 1: (2 x0) x1          (x normalised: a single Fp product takes one operand with limbs up to 2^30)
GPBC_INLINE Fe wide_half_sqr(const F2 &x, bool h) {
    return fe_mul(fe_sel(h, fe_dbl(x.a0), fe_add(x.a0, x.a1)), fe_sel(h, x.a1, fe_sub(x.a0, x.a1)));
}
// one component of (9 + i)(m + o i) seen from the lane that holds `mine`: 9 a0 - a1 (h = 0) or 9 a1 + a0 (h = 1); not normalised
GPBC_INLINE Fe wide_xi_half(const Fe &mine, const Fe &other, bool h) {
    return fe_add(fe_add(fe_mul8_norm(mine), mine), fe_sel(h, other, fe_neg(other)));
}
// Recombination of product halves into the six coefficients of dst.  term(k, t) -> product slot and whether it wraps (i + j >= 6:
// multiplied by xi).  Per output half (k, h):  lo = sum of the un-wrapped terms' half h,  hi = sum of the wrapped terms (both halves:
// the xi step mixes them), each normalised after every third term and at the end;  out = reduce(norm(lo + norm(xi hi))).
// Device: ONE LIMB PER LANE (lane = output half * NL + limb, two passes of 54 lanes: k < 3, k >= 3) — a lone wave pays ~5 cycles per
// instruction, and per limb a normalisation is five instructions instead of 27 (see wide_cyclo_out_limbs).  Host harness: the same
// sequence on whole elements, limb for limb the same values.
template <class M, class Term> GPBC_INLINE void wide_recombine(M &m, int dst, int n_terms, Term &&term) {
    if constexpr (M::LIMB_PARALLEL) {
        for (int pass = 0; pass < 2; pass++)
            m.run(6 * NL, [&](int) {
                const int o = m.comp(), i = m.limb(), k = 3 * pass + (o >> 1), h = o & 1;
                const int32_t low = i == NL - 1 ? -1 : LMASK, low26 = i == NL - 1 ? -1 : (1 << (LB - 3)) - 1, carry_in = i == 0 ? 0 : -1;
                auto norm = [&](int32_t v) { return (v & low) + ((m.below(v) >> LB) & carry_in); };
                auto mul8n = [&](int32_t v) { return ((v & low26) << 3) + ((m.below(v) >> (LB - 3)) & carry_in); };
                int32_t lo = 0, hi = 0, ho = 0;
                for (int t = 0; t < n_terms; t++) {
                    int slot;
                    bool wraps;
                    term(k, t, slot, wraps);
                    const bool there = slot >= 0;                              // (a coefficient may have fewer terms than n_terms: slot -1)
                    const int32_t mine = there ? m.ldw(there ? slot : 0, h, i) : 0, other = there ? m.ldw(there ? slot : 0, 1 - h, i) : 0;
                    lo += wraps ? 0 : mine; hi += wraps ? mine : 0; ho += wraps ? other : 0;
                    if (t % 3 == 2) { lo = norm(lo); hi = norm(hi); ho = norm(ho); }
                }
                lo = norm(lo); hi = norm(hi); ho = norm(ho);
                const int32_t w = norm(lo + norm(mul8n(hi) + hi + (h ? ho : -ho)));
                const int32_t kp = (int32_t)rintf((float)m.from_top(w) * (1.0f / (float)f29_p(NL - 1)));
                const int64_t prod = (int64_t)kp * (int64_t)m.p_limb();
                const int32_t plo = (int32_t)(prod & LMASK), phi = (int32_t)(prod >> LB);
                m.stw(dst + w2t(k), h, i, w - (i == NL - 1 ? (int32_t)prod : plo) - (m.below(phi) & carry_in));
            });
        return;
    }
    m.run(12, [&](int L) {
        const int k = L >> 1;
        const bool h = L & 1;
        Fe lo = fe_zero(), hi = fe_zero(), ho = fe_zero();
        for (int t = 0; t < n_terms; t++) {
            int slot;
            bool wraps;
            term(k, t, slot, wraps);
            const Fe mine = slot >= 0 ? m.ldh(slot, h) : fe_zero(), other = slot >= 0 ? m.ldh(slot, 1 - h) : fe_zero();
            lo = fe_add(lo, fe_sel(wraps, fe_zero(), mine)); hi = fe_add(hi, fe_sel(wraps, mine, fe_zero())); ho = fe_add(ho, fe_sel(wraps, other, fe_zero()));
            if (t % 3 == 2) { lo = fe_norm(lo); hi = fe_norm(hi); ho = fe_norm(ho); }
        }
        lo = fe_norm(lo); hi = fe_norm(hi); ho = fe_norm(ho);
        const Fe x = fe_norm(wide_xi_half(hi, ho, h));
        m.sth(dst + w2t(k), h, fe_reduce_arith(fe_norm(fe_add(lo, x))));
    });
}

// ---- Fp12 operations on slots
// dst = a * b.  Schoolbook over the w-basis: c_k = sum_{i+j=k} a_i b_j + xi sum_{i+j=k+6} a_i b_j.  (36 products: 72 halves would
// need a second wave, so this one keeps whole F2 products on 36 lanes.)
template <class M> GPBC_INLINE void wide_mul(M &m, int dst, int a, int b) {
    if (m.waves() == 2)      // with the helper wave: 72 half products on 72 lanes (the same limbs as the whole products: r0 = a0 b0 - a1 b1, r1 = a0 b1 + a1 b0)
        m.run2(72, [&](int L) { const int l = L >> 1, i = l / 6, j = l % 6; m.sth(W_PROD + l, L & 1, wide_half_mul(m.ld(a + w2t(i)), m.ld(b + w2t(j)), L & 1)); });
    else
        m.run(36, [&](int l) { const int i = l / 6, j = l % 6; m.st(W_PROD + l, f2_mul(m.ld(a + w2t(i)), m.ld(b + w2t(j)))); });
    wide_recombine(m, dst, 6, [](int k, int i, int &slot, bool &wraps) {
        int j = k - i;
        wraps = j < 0;
        if (wraps) j += 6;
        slot = W_PROD + i * 6 + j;
    });
}
// dst = a^2 for any a.  Of the 36 products a_i a_j only the 21 with i <= j are distinct; those with i < j count twice.  21 products as 42
// halves fit ONE wave (where the general product's 72 need the helper), each lane a 243-MAD leaf: half the product phase of wide_mul —
// this is the accumulator's f <- f^2 of every Miller step.  A product with i < j is stored doubled and normalised by the lane that made
// it (36 instructions beside a leaf of ~300), so the recombination is the general one over at most four terms per coefficient.
template <class M> GPBC_INLINE void wide_sqr(M &m, int dst, int a) {
    m.run(42, [&](int L) {
        const int p = L >> 1;
        // pair number p -> (i, j), i <= j, rows of 6, 5, 4, 3, 2, 1 pairs
        const int i = p < 6 ? 0 : p < 11 ? 1 : p < 15 ? 2 : p < 18 ? 3 : p < 20 ? 4 : 5;
        const int j = p - (i == 0 ? 0 : i == 1 ? 5 : i == 2 ? 9 : i == 3 ? 12 : i == 4 ? 14 : 15);
        const Fe r = wide_half_mul(m.ld(a + w2t(i)), m.ld(a + w2t(j)), L & 1);
        m.sth(W_PROD + p, L & 1, fe_sel(i < j, fe_norm(fe_dbl(r)), r));
    });
    wide_recombine(m, dst, 4, [](int k, int t, int &slot, bool &wraps) {
        // the t-th pair (i <= j) with i + j = k or k + 6, in order of i, as its number among the 21 (+ 64 when it wraps, -1: none) —
        // a table by (t, k) written as selections: k differs from lane to lane, t is the unrolled loop's constant
        auto by_k = [&](int c0, int c1, int c2, int c3, int c4, int c5) { return k == 0 ? c0 : k == 1 ? c1 : k == 2 ? c2 : k == 3 ? c3 : k == 4 ? c4 : c5; };
        const int code = t == 0 ? k                                                       // (0, k)
                       : t == 1 ? by_k(64 + 10, 64 + 14, 6, 7, 8, 9)                       // (1,5)' (2,5)' (1,1) (1,2) (1,3) (1,4)
                       : t == 2 ? by_k(64 + 13, 64 + 16, 64 + 17, 64 + 19, 11, 12)         // (2,4)' (3,4)' (3,5)' (4,5)' (2,2) (2,3)
                                : by_k(64 + 15, -1, 64 + 18, -1, 64 + 20, -1);             // (3,3)' none (4,4)' none (5,5)' none
        wraps = code >= 64;
        slot = code < 0 ? -1 : W_PROD + (code & 63);
    });
}
// dst = a * (c0 + c3 w + c4 w^3): the line of a Miller step (three slots at `line`).  18 products as 36 halves, three terms per coefficient.
template <class M> GPBC_INLINE void wide_mul_line(M &m, int dst, int a, int line) {
    m.run(36, [&](int L) {
        const int l = L >> 1, i = l / 3, t = l % 3;
        m.sth(W_PROD + l, L & 1, wide_half_mul(m.ld(a + w2t(i)), m.ld(line + t), L & 1));
    });
    wide_recombine(m, dst, 3, [](int k, int t, int &slot, bool &wraps) {
        int i = k - (t == 0 ? 0 : t == 1 ? 1 : 3);
        wraps = i < 0;
        if (wraps) i += 6;
        slot = W_PROD + i * 3 + t;
    });
}
// The output step of wide_cyclo_sqr with one limb per lane (device only; see there).  C0: the pass for k < 3, else k >= 3.
template <bool C0, class M> GPBC_INLINE void wide_cyclo_out_limbs(M &m, int dst, int a) {
    if constexpr (M::LIMB_PARALLEL) {
        m.run(6 * NL, [&](int L) {
            const int o = m.comp(), i = m.limb(), kk = o >> 1, h = o & 1, k = (C0 ? 0 : 3) + kk;
            const int A = C0 ? (kk == 0 ? 4 : kk == 1 ? 2 : 5) : (kk == 0 ? 5 : kk == 1 ? 4 : 2);
            const int B = C0 ? (kk == 0 ? 0 : kk == 1 ? 3 : 1) : (kk == 0 ? 1 : kk == 1 ? 0 : 3);
            const int32_t low = i == NL - 1 ? -1 : LMASK, low26 = i == NL - 1 ? -1 : (1 << (LB - 3)) - 1, carry_in = i == 0 ? 0 : -1;
            auto norm = [&](int32_t v) { return (v & low) + ((m.below(v) >> LB) & carry_in); };                       // fe_norm
            auto mul8n = [&](int32_t v) { return ((v & low26) << 3) + ((m.below(v) >> (LB - 3)) & carry_in); };       // fe_mul8_norm
            int32_t mine, other, t;
            if constexpr (C0) {
                mine = m.ldw(W_PROD + A, h, i); other = m.ldw(W_PROD + A, 1 - h, i);
            } else {
                const int S = kk == 0 ? 8 : kk == 1 ? 6 : 7;
                mine = norm(m.ldw(W_PROD + S, h, i) - m.ldw(W_PROD + A, h, i) - m.ldw(W_PROD + B, h, i));              // d, this half
                other = norm(m.ldw(W_PROD + S, 1 - h, i) - m.ldw(W_PROD + A, 1 - h, i) - m.ldw(W_PROD + B, 1 - h, i)); // d, the other half
            }
            const int32_t X = norm(mul8n(mine) + mine + (h ? other : -other));                                        // one component of xi (mine + other i)
            if constexpr (C0) t = norm(X + m.ldw(W_PROD + B, h, i));
            else t = kk == 0 ? X : mine;
            const int32_t x = m.ldw(a + k, h, i);
            const int32_t dd = norm(t + (C0 ? -x : x));
            const int32_t w = norm(2 * dd + t);
            // fe_reduce_arith: k p with k from the top limb of w, this lane's limb of it from one product, the high part from the limb below
            const int32_t kp = (int32_t)rintf((float)m.from_top(w) * (1.0f / (float)f29_p(NL - 1)));
            const int64_t prod = (int64_t)kp * (int64_t)m.p_limb();
            const int32_t lo = (int32_t)(prod & LMASK), hi = (int32_t)(prod >> LB);
            m.stw(dst + k, h, i, w - (i == NL - 1 ? (int32_t)prod : lo) - (m.below(hi) & carry_in));
        });
    }
}
// dst = a^2 for a in the cyclotomic subgroup (Granger-Scott; the formulas of f12_cyclo_sqr_t): nine squarings as 18 halves, six
// outputs as 12.
template <class M> GPBC_INLINE void wide_cyclo_sqr(M &m, int dst, int a) {
    m.run(18, [&](int L) {
        // squarings 0..5: the coefficients themselves; 6: (C0.b0 + C1.b1), 7: (C0.b2 + C1.b0), 8: (C1.b2 + C0.b1)
        const int l = L >> 1;
        const int u = l < 6 ? l : l == 6 ? 0 : l == 7 ? 2 : 5, v = l < 6 ? l : l == 6 ? 4 : l == 7 ? 3 : 1;
        const F2 x = m.ld(a + u), y = m.ld(a + v);
        m.sth(W_PROD + l, L & 1, wide_half_sqr(f2_norm(f2_sel(l < 6, x, f2_add(x, y))), L & 1));
    });
    // twelve output halves:
    //   k < 3:  C0.b_k' = 3 (xi S[A] + S[B]) - 2 x          k >= 3:  C1.b' = 3 [xi] (S[sum] - S[A] - S[B]) + 2 x   (xi for k = 3 only)
    if constexpr (M::LIMB_PARALLEL) {
        // Everything after the squares is LINEAR, and a lone wave pays ~5 cycles per instruction whatever it is: with one output half
        // per lane this phase was ~350 instructions on 12 lanes — as long as the squarings themselves (profiles/r04_latency_phases.txt).
        // One LIMB per lane instead: lane (o, i) holds limb i of output half o; a normalisation is "keep the low bits, add the carry of
        // the limb below", i.e. one neighbour exchange (m.below) and four instructions, the value reduction takes its multiplier k from
        // the lane of the top limb (m.from_top) and its limb of k p from one 64-bit product.  Limb for limb the values of the Fe-level
        // form in the other branch (fe_norm, fe_mul8_norm, fe_reduce_arith written out per limb) — that branch is what the interval
        // harness runs.  Two passes of 54 lanes: k < 3, then k >= 3 (no per-lane choice between the two formulas).
        wide_cyclo_out_limbs<true>(m, dst, a);
        wide_cyclo_out_limbs<false>(m, dst, a);
        return;
    }
    // ONE instruction stream (a per-lane branch would run both forms one after the other)
    m.run(12, [&](int L) {
        const int k = L >> 1;
        const bool h = L & 1, c0 = k < 3;
        const int S = k == 3 ? 8 : k == 4 ? 6 : 7;
        const int A = k == 0 ? 4 : k == 1 ? 2 : k == 2 ? 5 : k == 3 ? 5 : k == 4 ? 4 : 2;
        const int B = k == 0 ? 0 : k == 1 ? 3 : k == 2 ? 1 : k == 3 ? 1 : k == 4 ? 0 : 3;
        const F2 sA = m.ld(W_PROD + A), sB = m.ld(W_PROD + B);
        const F2 d = f2_norm(f2_sub(f2_sub(m.ld(W_PROD + S), sA), sB));           // (k >= 3; both halves: the xi step mixes them)
        const F2 U = f2_sel(c0, sA, d);
        const Fe X = fe_norm(wide_xi_half(h ? U.a1 : U.a0, h ? U.a0 : U.a1, h));  // xi S[A]  |  xi (cross term)
        const Fe sBm = h ? sB.a1 : sB.a0, dm = h ? d.a1 : d.a0;
        const Fe t = fe_sel(c0, fe_norm(fe_add(X, sBm)), fe_sel(k == 3, X, dm));
        // 3 t -+ 2 x with the value reduction at the end
        const Fe x = m.ldh(a + k, h);
        const Fe dd = fe_norm(fe_add(t, fe_sel(c0, fe_neg(x), x)));
        m.sth(dst + k, h, fe_reduce_arith(fe_norm(fe_add(fe_dbl(dd), t))));
    });
}
template <class M> GPBC_INLINE void wide_copy(M &m, int dst, int a) { m.run(6, [&](int k) { m.st(dst + k, m.ld(a + k)); }); }
template <class M> GPBC_INLINE void wide_conj(M &m, int dst, int a) {
    m.run(6, [&](int k) { const F2 x = m.ld(a + k); m.st(dst + k, f2_sel(k >= 3, f2_neg(x), x)); });
}
// dst = a^(p^j): coefficient of w^k -> (conjugate if j odd) times gamma_j[k]
template <class M> GPBC_INLINE void wide_frob(M &m, int dst, int a, int j) {
    m.run(12, [&](int L) {
        const int k = L >> 1;
        F2 x = m.ld(a + w2t(k));
        if (j & 1) x = f2_conj(x);
        const F2 g = k ? gamma29(j, k) : f2_one();
        m.sth(dst + w2t(k), L & 1, wide_half_mul(x, g, L & 1));
    });
}
// dst = 1 / a = conj(a) / (a conj(a)): the norm N = a conj(a) lies in Fp6 (its C1 half is zero), so one wide product, ONE lane for the
// Fp6 inversion (nine F2 products and one Fp inversion: ~25 k instructions, where a whole single-lane Fp12 inversion is ~70 k) and a
// second wide product.  tmp: one value of scratch.
template <class M> GPBC_INLINE void wide_inv(M &m, int dst, int a, int tmp) {
    wide_conj(m, tmp, a);
    wide_mul(m, dst, a, tmp);                                // N = (c0^2 - v c1^2, 0)
    m.run(1, [&](int) {
        const F6 n{m.ld(dst), m.ld(dst + 1), m.ld(dst + 2)};
        const F6 z = f6_inv(n);
        m.st(dst, z.b0); m.st(dst + 1, z.b1); m.st(dst + 2, z.b2);
        m.st(dst + 3, f2_zero()); m.st(dst + 4, f2_zero()); m.st(dst + 5, f2_zero());
    });
    wide_mul(m, dst, dst, tmp);
}

// ---- G2 steps on slots (formulas and names of g2_double_step / g2_add_step; every round is a set of independent F2 products, each
// lane picking its operand SLOTS by its number — one instruction stream for all of them)
template <class M> GPBC_INLINE void wide_double_step(M &m) {
    // round 1: G0 = XY = x y, G1 = B = y^2, G2 = C = z^2, G3 = J = x^2, G4 = YY = (y + z)^2 — y + z stands normalised in W_YZ (left by
    // the step before), so every lane loads its two operands and multiplies
    m.run(10, [&](int L) {
        const int l = L >> 1;
        const int sa = l == 0 || l == 3 ? W_TX : l == 2 ? W_TZ : l == 4 ? W_YZ : W_TY, sb = l == 3 ? W_TX : l == 2 ? W_TZ : l == 4 ? W_YZ : W_TY;
        m.sth(W_G + l, L & 1, wide_half_mul(m.ld(sa), m.ld(sb), L & 1));
    });
    // linear: H = YY - (B + C), 3 C, 3 J
    m.limbs(6, [&](auto &o, int c) {
        const int w = c >> 1, h = c & 1;
        const auto X = o.ld(w == 1 ? W_G + 2 : W_G + 3, h);
        const auto vH = o.sub(o.ld(W_G + 4, h), o.add(o.ld(W_G + 1, h), o.ld(W_G + 2, h))), v3 = o.add(o.dbl(X), X);
        o.st(w == 0 ? W_HN : w == 1 ? W_C3 : W_J3, h, o.norm(o.sel(w == 0, vH, v3)));
    });
    // round 2: G5 = E = 3 C b',  c3 = 3 J xP,  c0 = -H yP
    m.run(6, [&](int L) {
        const int l = L >> 1;
        const F2 v = m.ld(l == 0 ? W_C3 : l == 1 ? W_J3 : W_HN);
        const F2 rhs = f2_sel(l == 0, b_twist29(), m.ld(l == 1 ? W_XP : W_YP));
        m.sth(l == 0 ? W_G + 5 : l == 1 ? W_L3 : W_L0, L & 1, wide_half_mul(f2_sel(l == 2, f2_neg(v), v), rhs, L & 1));
    });
    // linear: A = XY / 2,  B - F,  G = (B + F) / 2      (F = 3 E)
    m.limbs(6, [&](auto &o, int c) {
        const int w = c >> 1, h = c & 1;
        const auto B = o.ld(W_G + 1, h), E = o.ld(W_G + 5, h);
        const auto F = o.add(o.dbl(E), E);
        const auto halved = o.halve(o.sel(w == 0, o.ld(W_G, h), o.norm(o.add(B, F))));
        o.st(w == 0 ? W_AH : w == 1 ? W_BF : W_GG, h, o.norm(o.sel(w == 1, o.sub(B, F), halved)));
    });
    // round 3: x' = A (B - F), G7 = G^2, G8 = E^2, z' = B H
    m.run(8, [&](int L) {
        const int l = L >> 1;
        const F2 lhs = m.ld(l == 0 ? W_AH : l == 1 ? W_GG : l == 2 ? W_G + 5 : W_G + 1), rhs = m.ld(l == 0 ? W_BF : l == 1 ? W_GG : l == 2 ? W_G + 5 : W_HN);
        m.sth(l == 0 ? W_TX : l == 1 ? W_G + 7 : l == 2 ? W_G + 8 : W_TZ, L & 1, wide_half_mul(lhs, rhs, L & 1));
    });
    // linear: y' = G^2 - 3 E^2,  c4 = E - B,  and y' + z' for the next step
    m.limbs(6, [&](auto &o, int c) {
        const int w = c >> 1, h = c & 1;
        const auto EE = o.ld(W_G + 8, h);
        const auto y = o.norm(o.sub(o.ld(W_G + 7, h), o.add(o.dbl(EE), EE)));
        const auto c4 = o.norm(o.sub(o.ld(W_G + 5, h), o.ld(W_G + 1, h)));
        const auto yz = o.norm(o.add(y, o.ld(W_TZ, h)));
        o.st(w == 0 ? W_TY : w == 1 ? W_L4 : W_YZ, h, o.sel(w == 0, y, o.sel(w == 1, c4, yz)));
    });
}
// T <- T + Q' and the chord (Q' = slots qx, qy); with_point = false: the chord alone (last step of the loop)
template <class M> GPBC_INLINE void wide_add_step(M &m, int qx, int qy, bool with_point) {
    // round 1: G0 = O = y - qy z,  G1 = L = x - qx z
    m.run(4, [&](int L) {
        const int l = L >> 1, h = L & 1;
        const Fe p = wide_half_mul(m.ld(l == 0 ? qy : qx), m.ld(W_TZ), h);
        m.sth(W_G + l, h, fe_norm(fe_sub(m.ldh(l == 0 ? W_TY : W_TX, h), p)));
    });
    // round 2: G2 = C = O^2, G3 = D = L^2, G4 = M1 = O qx, G5 = M2 = L qy, c0 = L yP, c3 = -O xP
    m.run(12, [&](int L) {
        const int l = L >> 1;
        const F2 a = m.ld(l == 0 || l == 2 || l == 5 ? W_G : W_G + 1);
        const F2 b = m.ld(l == 0 ? W_G : l == 1 ? W_G + 1 : l == 2 ? qx : l == 3 ? qy : l == 4 ? W_YP : W_XP);
        m.sth(l < 4 ? W_G + 2 + l : l == 4 ? W_L0 : W_L3, L & 1, wide_half_mul(f2_sel(l == 5, f2_neg(a), a), b, L & 1));
    });
    if (!with_point) {
        m.limbs(2, [&](auto &o, int c) { o.st(W_L4, c, o.norm(o.sub(o.ld(W_G + 4, c), o.ld(W_G + 5, c)))); });
        return;
    }
    // round 3: G6 = E = L D, G7 = F = z C, G8 = G = x D
    m.run(6, [&](int L) {
        const int l = L >> 1;
        m.sth(W_G + 6 + l, L & 1, wide_half_mul(m.ld(l == 0 ? W_G + 1 : l == 1 ? W_TZ : W_TX), m.ld(l == 1 ? W_G + 2 : W_G + 3), L & 1));
    });
    // linear: H = E + F - 2 G,  G - H
    m.limbs(4, [&](auto &o, int c) {
        const int w = c >> 1, h = c & 1;
        const auto G = o.ld(W_G + 8, h);
        const auto H = o.norm(o.sub(o.add(o.ld(W_G + 6, h), o.ld(W_G + 7, h)), o.dbl(G)));
        o.st(w == 0 ? W_HH : W_GH, h, o.sel(w == 0, H, o.norm(o.sub(G, H))));
    });
    // round 4: x' = L H, G10 = U = (G - H) O, G11 = t1 = y E, z' = E z   (x' and z' go straight to T: every operand of the round is
    // read before any lane stores)
    m.run(8, [&](int L) {
        const int l = L >> 1;
        const F2 a = m.ld(l == 0 ? W_G + 1 : l == 1 ? W_GH : l == 2 ? W_TY : W_G + 6), b = m.ld(l == 0 ? W_HH : l == 1 ? W_G : l == 2 ? W_G + 6 : W_TZ);
        m.sth(l == 0 ? W_TX : l == 1 ? W_G + 10 : l == 2 ? W_G + 11 : W_TZ, L & 1, wide_half_mul(a, b, L & 1));
    });
    // linear: y' = U - t1,  c4 = M1 - M2,  and y' + z' for the next step
    m.limbs(6, [&](auto &o, int c) {
        const int w = c >> 1, h = c & 1;
        const auto y = o.norm(o.sub(o.ld(W_G + 10, h), o.ld(W_G + 11, h)));
        const auto c4 = o.norm(o.sub(o.ld(W_G + 4, h), o.ld(W_G + 5, h)));
        const auto yz = o.norm(o.add(y, o.ld(W_TZ, h)));
        o.st(w == 0 ? W_TY : w == 1 ? W_L4 : W_YZ, h, o.sel(w == 0, y, o.sel(w == 1, c4, yz)));
    });
}

// ---- the Miller loop in two halves that share nothing but the 88 lines: the G2 walk (producer: slots W_TX .. W_G) and the
// accumulator (consumer: the value slots, W_PROD, W_CL).  On the device they run as two waves of one workgroup, the lines passing
// through an LDS ring (gpbc_pairing.hip: k_miller_wide); under the host harness one after the other.  P, Q not at infinity.
//   emit(j): line j is in W_L0, W_L3, W_L4          fetch(j): put line j into W_CL .. W_CL + 2
template <class M, class Emit> GPBC_INLINE void wide_miller_lines(M &m, const G1A &p, const G2A &q, Emit &&emit) {
    m.run(1, [&](int) {
        m.st(W_TX, q.x); m.st(W_TY, q.y); m.st(W_TZ, f2_one()); m.st(W_YZ, f2_norm(f2_add(q.y, f2_one())));
        m.st(W_QX, q.x); m.st(W_QY, q.y); m.st(W_NQY, f2_neg(q.y));
        m.st(W_Q1X, f2_mul(f2_conj(q.x), gamma29(1, 2))); m.st(W_Q1Y, f2_mul(f2_conj(q.y), gamma29(1, 3)));
        m.st(W_Q2X, f2_mul(q.x, gamma29(2, 2))); m.st(W_Q2Y, f2_norm(f2_neg(f2_mul(q.y, gamma29(2, 3)))));
        m.st(W_XP, F2{p.x, fe_zero()}); m.st(W_YP, F2{p.y, fe_zero()});
    });
    int j = 0;
    for (int i = BN254_ATE_NAF_LEN - 2; i >= 0; i--) {
        wide_double_step(m);
        emit(j++);
        const int d = ate_naf_digit(i);
        if (d != 0) {
            wide_add_step(m, W_QX, d > 0 ? W_QY : W_NQY, true);
            emit(j++);
        }
    }
    wide_add_step(m, W_Q1X, W_Q1Y, true);
    emit(j++);
    wide_add_step(m, W_Q2X, W_Q2Y, false);
    emit(j++);
}
template <class M, class Fetch> GPBC_INLINE void wide_miller_accumulate(M &m, int f, Fetch &&fetch) {
    int j = 0;
    for (int i = BN254_ATE_NAF_LEN - 2; i >= 0; i--) {
        fetch(j++);
        if (i == BN254_ATE_NAF_LEN - 2) {
            // the first doubling: f = 1^2 * l = c0 + c3 w + c4 w^3
            m.run(6, [&](int k) { m.st(f + k, f2_sel(k == 0 || k == 3 || k == 4, m.ld(k == 0 ? W_CL : k == 3 ? W_CL + 1 : W_CL + 2), f2_zero())); });
        } else {
            wide_sqr(m, f, f);
            wide_mul_line(m, f, f, W_CL);
        }
        if (ate_naf_digit(i) != 0) {
            fetch(j++);
            wide_mul_line(m, f, f, W_CL);
        }
    }
    for (int k = 0; k < 2; k++) {
        fetch(j++);
        wide_mul_line(m, f, f, W_CL);
    }
}

// ---- final exponentiation (operation order of final_exp29 / final_exp_pair)
template <class M> GPBC_INLINE void wide_cyclo_sqr_n(M &m, int v, int n) { for (int i = 0; i < n; i++) wide_cyclo_sqr(m, v, v); }
// z = x^u over the dictionary {x^3, x^15, x^75} (f12_expt_to); tab = three consecutive values, tmp = one value
template <class M> GPBC_INLINE void wide_expt(M &m, int z, int x, int tab, int tmp) {
    constexpr int8_t D[GPBC_U_CHAIN_LEN] = GPBC_U_CHAIN;
    wide_cyclo_sqr(m, tmp, x);
    wide_mul(m, tab, tmp, x);                                  // x^3
    for (int k = 1; k < 3; k++) {                              // x^15, x^75
        wide_cyclo_sqr(m, tmp, tab + 6 * (k - 1));
        wide_cyclo_sqr(m, tmp, tmp);
        wide_mul(m, tab + 6 * k, tmp, tab + 6 * (k - 1));
    }
    wide_copy(m, z, tab + 6 * u_chain_entry(D[GPBC_U_CHAIN_LEN - 1]));
    int run = 0;
    for (int i = GPBC_U_CHAIN_LEN - 2; i >= 0; i--) {
        run++;
        const int d = D[i];
        if (d != 0) {
            wide_cyclo_sqr_n(m, z, run);
            run = 0;
            const int e = tab + 6 * u_chain_entry(d);
            if (d < 0) { wide_conj(m, tmp, e); wide_mul(m, z, z, tmp); }
            else wide_mul(m, z, z, e);
        }
    }
    if (run) wide_cyclo_sqr_n(m, z, run);
}
// value slot `f` <- f^(s (p^12 - 1) / r); uses values 1..11 as registers (f must be value 0)
template <class M> GPBC_INLINE void wide_final_exp(M &m) {
    constexpr int F = wv(0), R = wv(1), T0 = wv(2), T1 = wv(3), T2 = wv(4), T3 = wv(5), T4 = wv(6), TAB = wv(7), TMP = wv(11);
    wide_inv(m, T1, F, T0);                                  // (T0 = conj(f) on the way out)
    wide_mul(m, T0, T0, T1);
    wide_frob(m, R, T0, 2);
    wide_mul(m, R, R, T0);
    wide_expt(m, T0, R, TAB, TMP); wide_conj(m, T0, T0);
    wide_cyclo_sqr(m, T0, T0);
    wide_cyclo_sqr(m, T1, T0);
    wide_mul(m, T1, T0, T1);
    wide_expt(m, T2, T1, TAB, TMP); wide_conj(m, T2, T2);
    wide_conj(m, T3, T1);
    wide_mul(m, T1, T2, T3);
    wide_cyclo_sqr(m, T3, T2);
    wide_expt(m, T4, T3, TAB, TMP);
    wide_mul(m, T4, T1, T4);
    wide_mul(m, T3, T0, T4);
    wide_mul(m, T0, T2, T4);
    wide_mul(m, T0, R, T0);
    wide_frob(m, T2, T3, 1);
    wide_mul(m, T0, T2, T0);
    wide_frob(m, T2, T4, 2);
    wide_mul(m, T0, T2, T0);
    wide_conj(m, T2, R);
    wide_mul(m, T2, T2, T3);
    wide_frob(m, T2, T2, 3);
    wide_mul(m, F, T2, T0);
}

// ---- GT.Exp for the latency path: value 0 <- x^k for a 256-bit plain exponent (k = 0 -> one), x any Fp12 element (gnark's Exp is the
// generic square-and-multiply, and callers hand it values outside the cyclotomic subgroup too).  Three-bit windows from the top over
// the table x^1 .. x^7 in values 1..7; the exponent is the same for every lane of the wavefront, so the digit branches are uniform.
// Is value `a` in the cyclotomic subgroup (a^(p^4) a == a^(p^2), see f12p_is_cyclotomic)?  t2, t4: two scratch values.  The same
// answer on every lane.
template <class M> GPBC_INLINE bool wide_is_cyclotomic(M &m, int a, int t2, int t4) {
    wide_frob(m, t2, a, 2);
    wide_frob(m, t4, t2, 2);
    wide_mul(m, t4, t4, a);
    m.run(6, [&](int c) {
        const F2 d = f2_norm(f2_sub(m.ld(t4 + c), m.ld(t2 + c)));
        m.st(W_G + c, f2_sel(f2_is_zero(d), f2_one(), f2_zero()));
    });
    m.sync_waves();                                          // a helper wave (it sat the phase out) must not read the flags before they are there: it takes the same branch
    bool ok = true;
    for (int c = 0; c < 6; c++) ok = ok && !f2_is_zero(m.ld(W_G + c));
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GPBC_BOUNDS)
    ok = __builtin_amdgcn_readfirstlane((int)ok) != 0;       // every lane read the same six flags: say so to the compiler
#endif
    return ok;
}
template <class M> GPBC_INLINE void wide_exp256(M &m, const uint32_t (&k)[8]) {
    constexpr int Z = wv(0);
    // pairing values (what the reference raises) square by Granger-Scott: 18 half products instead of 36 products per squaring
    const bool cyclotomic = wide_is_cyclotomic(m, wv(1), wv(8), wv(9));
    for (int e = 2; e < 8; e++) wide_mul(m, wv(e), wv(e - 1), wv(1));
    bool started = false;
    for (int w = 85; w >= 0; w--) {
        const int bit = 3 * w, word = bit >> 5, sh = bit & 31;
        uint32_t d = k[word] >> sh;
        if (sh > 29 && word < 7) d |= k[word + 1] << (32 - sh);
        d &= 7u;
        if (started) for (int q = 0; q < 3; q++) { if (cyclotomic) wide_cyclo_sqr(m, Z, Z); else wide_sqr(m, Z, Z); }
        if (d) {
            if (started) wide_mul(m, Z, Z, wv((int)d));
            else { wide_copy(m, Z, wv((int)d)); started = true; }
        }
    }
    if (!started) m.run(6, [&](int c) { m.st(Z + c, f2_sel(c == 0, f2_one(), f2_zero())); });
}

}  // namespace gpbc
#endif
