// Fp12 arithmetic with ONE Fp12 VALUE SPREAD OVER TWO ADJACENT LANES: the even lane of a pair holds C0, the odd lane
// holds C1 (each an F6 = 54 registers).  Why: with one pairing per lane the Fp12 accumulator plus the temporaries of a
// Karatsuba F6 product need ~350 live registers, the 256-register budget of a 2-waves-per-SIMD kernel spills ~140 KB
// per pairing to scratch, and the kernels stall ~45 % of the time on that traffic (profiles/r01_v3_pmc_summary.txt).
// Split over a lane pair, every F12 operation becomes one F6 product per lane with the halves swapped through DPP
// (v_mov_b32_dpp quad_perm [1,0,3,2], full rate): the work stays balanced and the state per lane halves.
//
//   f12 square  (complex method)   even: m = c0*c1           odd: st = (c0+c1)(c0+v c1)      -> 6 F2 products / lane
//   f12 * line  (034-sparse)       both: h*l0 and h*l1 on the lane's own half                 -> 8 F2 products / lane
//   f12 * f12   (Karatsuba)        even: a0*b0 + diagonal part of the third product,
//                                  odd:  a1*b1 + cross part of the third product              -> 9 F2 products / lane
//   cyclotomic square              4.5 F2 squarings / lane (the ninth is split: real part on one lane, imaginary on the other)
//
// X is the exchange policy: X::odd (lane parity) and X::swap(F2/F6) -> the partner lane's value.  The device policy
// uses DPP; tools/bounds_check.cpp runs the two lanes as two host threads with a rendezvous.
#ifndef GPBC_TOWER29_PAIR_HIP_HPP
#define GPBC_TOWER29_PAIR_HIP_HPP
#include "tower29.hip.hpp"

namespace gpbc {

GPBC_INLINE F6 f6_sel(bool c, const F6 &a, const F6 &b) { return F6{f2_sel(c, a.b0, b.b0), f2_sel(c, a.b1, b.b1), f2_sel(c, a.b2, b.b2)}; }
GPBC_INLINE F6 f6_dbl(const F6 &x) { return f6_add(x, x); }

#if defined(__HIPCC__) && !defined(GPBC_BOUNDS)
struct PairDpp {
    bool odd;
    __device__ __forceinline__ static int32_t sw(int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
#else
        return v;                                                      // host pass of hipcc: never executed
#endif
    }
    __device__ __forceinline__ Fe swap(const Fe &a) const {
        Fe r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.v[i] = sw(a.v[i]);
        return r;
    }
    __device__ __forceinline__ F2 swap(const F2 &a) const { return F2{swap(a.a0), swap(a.a1)}; }
    __device__ __forceinline__ F6 swap(const F6 &a) const { return F6{swap(a.b0), swap(a.b1), swap(a.b2)}; }
    // a on the odd lane, -a on the even lane: ONE multiplication by the lane's +-1 per limb instead of a negation and a select
    __device__ __forceinline__ Fe sgn(const Fe &a) const {
        int32_t s = odd ? 1 : -1;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(s));                 // keep it a multiplication (the optimiser would turn it back into negate + select)
#endif
        Fe r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.v[i] = a.v[i] * s;
        return r;
    }
    // a + swap(b): written limb by limb on a single-use swap, which the compiler folds into one v_add_u32_dpp per limb (it does so
    // for Fe / F2-sized values; an F6-sized swap goes through memory first and is not folded).
    // NEVER write a - swap(b) with a single-use swap: the compiler folds it into v_subrev_u32_dpp, and on gfx950 that instruction
    // computes dpp(src1) - src0, not src1 - dpp(src0) as the compiler assumes (tools/subdpp_probe.hip: 64 of 64 lanes wrong, for the
    // compiler's own output and for the hand-written instruction alike; v_sub_u32_dpp and v_add_u32_dpp are fine).  Subtract on the
    // SENDING lane instead — a + swap(c - b) — and _build.py refuses a library whose ISA contains v_subrev_*_dpp.
    __device__ __forceinline__ Fe add_swap(const Fe &a, const Fe &b) const {
        Fe r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.v[i] = a.v[i] + sw(b.v[i]);
        return r;
    }
    // true when `c` holds on every active lane of the wavefront (a uniform value: branches on it do not diverge)
    __device__ __forceinline__ static bool all(bool c) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_ballot_w64(!c) == 0;
#else
        return c;
#endif
    }
};
#endif
// policy helpers on the tower types; the host policy of tools/bounds_check.cpp provides the same two Fe primitives
template <class X> GPBC_INLINE F2 f2p_sgn(const X &x, const F2 &a) { return F2{x.sgn(a.a0), x.sgn(a.a1)}; }
template <class X> GPBC_INLINE F6 f6p_sgn(const X &x, const F6 &a) { return F6{f2p_sgn(x, a.b0), f2p_sgn(x, a.b1), f2p_sgn(x, a.b2)}; }
template <class X> GPBC_INLINE F2 f2p_add_swap(const X &x, const F2 &a, const F2 &b) { return F2{x.add_swap(a.a0, b.a0), x.add_swap(a.a1, b.a1)}; }
template <class X> GPBC_INLINE F6 f6p_add_swap(const X &x, const F6 &a, const F6 &b) { return F6{f2p_add_swap(x, a.b0, b.b0), f2p_add_swap(x, a.b1, b.b1), f2p_add_swap(x, a.b2, b.b2)}; }

template <class X> GPBC_INLINE F6 f6p_add_partner(const X &x, const F6 &a) {
    return F6{f2_add(a.b0, x.swap(a.b0)), f2_add(a.b1, x.swap(a.b1)), f2_add(a.b2, x.swap(a.b2))};
}

// (C0 on even, C1 on odd) <- one()
template <class X> GPBC_INLINE F6 f12p_one(const X &x) { return f6_sel(x.odd, F6{f2_zero(), f2_zero(), f2_zero()}, F6{f2_one(), f2_zero(), f2_zero()}); }
template <class X> GPBC_INLINE F6 f12p_conj(const X &x, const F6 &h) { return f6_sel(x.odd, f6_neg(h), h); }

// f^2, complex method: (c0 + c1 w)^2 = (st - m - v m) + 2m w with m = c0 c1, st = (c0 + c1)(c0 + v c1)
// PN = true: h is positive-normalised (straight out of fe_norm: the outputs of this function and of the PN sparse products are), and
// the F6 product takes the subtractive Karatsuba form without operand normalisations (f6_mul_pn_t).
template <bool PN = false, class X> GPBC_INLINE F6 f12p_sqr(const X &x, const F6 &h) {
    F6 p = x.swap(h);
    // even lane (h = c0, p = c1): m = h * p.   odd lane (h = c1, p = c0): st = (h + p) * (p + v h).
    F6 s = f6_norm(f6_add(h, p));
    F6 t = f6_norm(f6_add(p, f6_mul_v_t<false>(h)));
    F6 r = PN ? f6_mul_pn_t<false>(f6_sel(x.odd, s, h), f6_sel(x.odd, t, p)) : f6_mul_t<false>(f6_sel(x.odd, s, h), f6_sel(x.odd, t, p));   // even: m, odd: st
    F6 pr = x.swap(r);                                                   // even: st, odd: m
    F6 even_out = f6_sub(f6_sub(pr, r), f6_mul_v_t<false>(r));
    F6 odd_out = f6_dbl(pr);
    return f6_norm(f6_sel(x.odd, odd_out, even_out));                    // value-reduced by the sparse product that follows
}

// f * (l0 + l1 w), l0 = (c0,0,0), l1 = (c3,c4,0):  C0' = a l0 + v (b l1),  C1' = a l1 + b l0.
// Each lane multiplies its own half by l0 and by l1 (3 + 5 F2 products), then the l1-products are swapped.
// PN = true: h positive-normalised in, positive-normalised out (the value reduction goes in front of the normalisation), and the
// five-product half in the subtractive form (f6_mul_01_pn_t).
template <bool PN = false, class X> GPBC_INLINE F6 f12p_mul_034(const X &x, const F6 &h, const F2 &c0, const F2 &c3, const F2 &c4) {
    F2 s34 = f2_norm(PN ? f2_sub(c3, c4) : f2_add(c3, c4));
    F6 r0 = f6_mul_f2(h, c0);
    F6 r1 = PN ? f6_mul_01_pn_t<false>(h, c3, c4, s34) : f6_mul_01_t<false, false>(h, c3, c4, s34);       // b0, b1 un-normalised: they only enter the sum below
    // each lane sends what its partner adds — the odd lane v (b l1), the even lane a l1 — so the received value has one use and
    // rides on the addition (v_add_u32_dpp) instead of costing a move per limb
    F6 send = f6_sel(x.odd, f6_mul_v_t<false>(r1), r1);
    F6 sum = f6p_add_swap(x, r0, send);
    return PN ? f6_reduce_arith_norm(sum) : f6_reduce_arith(f6_norm(sum));   // the one value reduction of this step (no table loads
                                                                         // here: this runs beside the lines stream, see fe29.hip.hpp)
}

// f * (1 + l1 w), l1 = (c3, c4, 0) — a line scaled to c0 = 1 (what the fixed-Q line table holds; the scaling factor lies in Fp2
// and the final exponentiation removes it):  C0' = a + v (b l1),  C1' = b + a l1.  Five F2 products per lane instead of eight.
template <bool PN = false, class X> GPBC_INLINE F6 f12p_mul_34(const X &x, const F6 &h, const F2 &c3, const F2 &c4) {
    F2 s34 = f2_norm(PN ? f2_sub(c3, c4) : f2_add(c3, c4));
    F6 r1 = PN ? f6_mul_01_pn_t<false>(h, c3, c4, s34) : f6_mul_01_t<false, false>(h, c3, c4, s34);
    F6 send = f6_sel(x.odd, f6_mul_v_t<false>(r1), r1);       // as in f12p_mul_034
    F6 sum = f6p_add_swap(x, h, send);
    return PN ? f6_reduce_arith_norm(sum) : f6_reduce_arith(f6_norm(sum));
}

// Product of two lines, (c0 + (c3 + c4 v) w)(d0 + (d3 + d4 v) w), as a lane-pair value:
//   C0 = (c0 d0 + xi c4 d4,  c3 d3,  c3 d4 + c4 d3),   C1 = (c0 d3 + c3 d0,  c0 d4 + c4 d0,  0)
// (gnark's Mul034By034).  Six F2 products with Karatsuba for the three cross sums, three per lane:
//   even  A = c0 d0,  B = c3 d3,  C = (c3+c4)(d3+d4);     odd  D = c4 d4,  E = (c0+c3)(d0+d3),  F = (c0+c4)(d0+d4)
// then one swap.  Multiplying f by the product (f12p_mul) replaces two sparse multiplications: 3 + 9 F2 products per lane
// instead of 8 + 8.
template <class X> GPBC_INLINE F6 f12p_mul_034_by_034(const X &x, const F2 &c0, const F2 &c3, const F2 &c4, const F2 &d0, const F2 &d3, const F2 &d4) {
    F2 x1 = f2_sel(x.odd, c4, c0), y1 = f2_sel(x.odd, d4, d0);
    F2 x2 = f2_sel(x.odd, f2_norm(f2_add(c0, c3)), c3), y2 = f2_sel(x.odd, f2_norm(f2_add(d0, d3)), d3);
    F2 x3 = f2_norm(f2_add(f2_sel(x.odd, c0, c3), c4)), y3 = f2_norm(f2_add(f2_sel(x.odd, d0, d3), d4));
    F6 mine{f2_mul(x1, y1), f2_mul(x2, y2), f2_mul(x3, y3)};           // even (A, B, C)   odd (D, E, F)
    F6 other = x.swap(mine);
    // even: b0 = A + xi D, b1 = B, b2 = C - B - D        odd: b0 = E - A - B, b1 = F - A - D, b2 = 0
    F2 t = f2_sub(f2_sub(mine.b2, other.b0), f2_sel(x.odd, mine.b0, mine.b1));
    F2 u = f2_sel(x.odd, f2_sub(f2_sub(mine.b1, other.b0), other.b1), f2_add(mine.b0, f2_mul_xi(other.b0)));
    return f6_reduce_arith(f6_norm(F6{u, f2_sel(x.odd, t, mine.b1), f2_sel(x.odd, f2_zero(), t)}));
}

// One F6 value per lane parked in LDS (216 of a lane's 320-byte share at two waves per SIMD): a register spill costs an HBM
// transaction on this machine, LDS does not.  Used where a value is computed early and needed only at the end of a
// long stretch of leaf calls, during which just ~158 VGPRs survive a call (the leaf itself takes 98).  Layout [chunk of 4 limbs]
// [lane]: 14 ds_write_b128 / ds_read_b128, conflict-free.  The host build of the bounds harness keeps the object (with its
// intervals) as it is.  One slot: parked values must not nest.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GPBC_BOUNDS)
__shared__ i32x4 g_f6_park[14][64];
struct F6Park {
    GPBC_INLINE explicit F6Park(const F6 &v) {
        const Fe *fe[6] = {&v.b0.a0, &v.b0.a1, &v.b1.a0, &v.b1.a1, &v.b2.a0, &v.b2.a1};
        int32_t w[56];
#pragma unroll
        for (int e = 0; e < 6; e++)
#pragma unroll
            for (int i = 0; i < NL; i++) w[e * NL + i] = fe[e]->v[i];
        w[54] = 0; w[55] = 0;
        const unsigned lane = f2_arg_lane();
#pragma unroll
        for (int c = 0; c < 14; c++) g_f6_park[c][lane] = i32x4{w[4 * c], w[4 * c + 1], w[4 * c + 2], w[4 * c + 3]};
    }
    GPBC_INLINE F6 get() const {
        int32_t w[56];
        const unsigned lane = f2_arg_lane();
#pragma unroll
        for (int c = 0; c < 14; c++) { i32x4 t = g_f6_park[c][lane]; w[4 * c] = t.x; w[4 * c + 1] = t.y; w[4 * c + 2] = t.z; w[4 * c + 3] = t.w; }
        F6 v;
        Fe *fe[6] = {&v.b0.a0, &v.b0.a1, &v.b1.a0, &v.b1.a1, &v.b2.a0, &v.b2.a1};
#pragma unroll
        for (int e = 0; e < 6; e++)
#pragma unroll
            for (int i = 0; i < NL; i++) fe[e]->v[i] = w[e * NL + i];
        return v;
    }
};
#else
struct F6Park {                                               // host / A-B builds: the value itself (with its intervals)
    F6 v;
    GPBC_INLINE explicit F6Park(const F6 &x) : v(x) {}
    GPBC_INLINE F6 get() const { return v; }
};
#endif

// full product (Karatsuba over F6).  The third product (a0+a1)(b0+b1) = sx * sy is itself split over the pair, 2 + 2 operand sums:
//   even lane  u0 = sx0 sy0,  u1 = sx1 sy1,  m01 = (sx0+sx1)(sy0+sy1)        odd lane  u2 = sx2 sy2,  m12 = (sx1+sx2)(sy1+sy2),  m02 = (sx0+sx2)(sy0+sy2)
// (with all three cross products on one lane that lane needed three operand combinations per side and the other none — but both
// lanes execute every instruction, so the split that minimises the work of the busier lane wins).
// (Leaving the outputs un-reduced for the squarings that follow in x^u does not work: the squares of an un-reduced value break the
// 2^28 top-limb guard of the product routine — the interval harness rejects it.)
// PN = true: both operands are positive-normalised (straight out of fe_norm / f6_norm; see f6_mul_pn_t) and the own-half product takes
// the subtractive form as well.
template <bool PN = false, class X> GPBC_INLINE F6 f12p_mul(const X &x, const F6 &hx, const F6 &hy) {
    const F6Park parked(PN ? f6_mul_pn_t<false>(hx, hy) : f6_mul_t<false>(hx, hy));   // even: t0 = a0 b0, odd: t1 = a1 b1 — needed again after the three products below
    // own + partner's half, coefficient by coefficient: with the swap taken per F2 the compiler folds the DPP move into the addition
    // (v_add_u32_dpp) — on a whole F6 it does not (the 216-byte aggregate goes through memory first), and hand-written
    // v_add_u32_dpp statements had measured 3 % slower (profiles/r02_microbench_pair.txt)
    F6 sx = f6_norm(f6p_add_partner(x, hx));
    F6 sy = f6_norm(f6p_add_partner(x, hy));
    // The cross products in SUBTRACTIVE Karatsuba form, m_ij' = (x_i - x_j)(y_i - y_j) = u_i + u_j - (x_i y_j + x_j y_i): sx and sy were
    // just normalised (limbs 0..7 non-negative), so a DIFFERENCE of two coefficients stays within +-(2^29 + small) and needs no
    // normalisation of its own — the sums (x_i + x_j) of the additive form did (four per product, ~190 instructions).
    F2 xa = f2_sel(x.odd, sx.b2, sx.b0), ya = f2_sel(x.odd, sy.b2, sy.b0);                                   // u0 | u2
    F2 xb = f2_sub(sx.b1, f2_sel(x.odd, sx.b2, f2_zero())), yb = f2_sub(sy.b1, f2_sel(x.odd, sy.b2, f2_zero()));   // u1 | m12' = (x1 - x2)(y1 - y2)
    F2 xc = f2_sub(sx.b0, f2_sel(x.odd, sx.b2, sx.b1)), yc = f2_sub(sy.b0, f2_sel(x.odd, sy.b2, sy.b1));           // m01' = (x0 - x1)(y0 - y1) | m02' = (x0 - x2)(y0 - y2)
    F6 mine{f2_mul(xa, ya), f2_mul(xb, yb), f2_mul(xc, yc)};  // even: (u0, u1, m01')   odd: (u2, m12', m02')
    F6 other = x.swap(mine);
    // (u0, u1, u2) and (m12, m01, m02) AS THE ODD LANE SEES THEM — only the odd lane's output uses the third product (the even
    // lane's is t0 + v t1), so no select is spent on making the even lane's copy right; its values are products like any other
    // and go through the same arithmetic unused
    F6 dg{other.b0, other.b1, mine.b0};
    F6 cr{mine.b1, other.b2, mine.b2};
    F6 t = parked.get();
    F6 pt = x.swap(t);                                        // even: t1, odd: t0
    // one shared xi-multiplication: the odd lane needs xi (m12 - u1 - u2) for m, the even lane xi t1.b2 for v t1
    // (cross terms: x_i y_j + x_j y_i = u_i + u_j - m_ij')
    F2 xi1 = f2_mul_xi_nn(f2_sel(x.odd, f2_norm(f2_sub(f2_add(dg.b1, dg.b2), cr.b0)), pt.b2));
    // odd lane: (a0+a1)(b0+b1) - t0 - t1, coefficient by coefficient; additions and subtractions alternate so that every intermediate
    // stays inside int32 (products have limbs 0..7 in [0, 2^29); the interval harness checks the order)
    F2 o0 = f2_sub(f2_add(f2_sub(xi1, pt.b0), dg.b0), t.b0);
    F2 o1 = f2_sub(f2_add(f2_sub(f2_add(f2_sub(dg.b0, cr.b1), dg.b1), pt.b1), f2_mul_xi_nn(dg.b2)), t.b1);
    F2 o2 = f2_sub(f2_add(f2_sub(f2_add(f2_sub(dg.b0, cr.b2), dg.b2), pt.b2), dg.b1), t.b2);
    F6 odd_out{o0, o1, o2};                                   // m - t0 - t1
    F6 even_out = f6_add(t, F6{xi1, pt.b0, pt.b1});           // t0 + v t1
    return f6_reduce(f6_norm(f6_sel(x.odd, odd_out, even_out)));
}

// Granger-Scott cyclotomic squaring.  In tower names the nine squarings are t1=C0.b0^2, t5=C0.b1^2, t2=C0.b2^2,
// t3=C1.b0^2, t0=C1.b1^2, t4=C1.b2^2, s6=(C0.b0+C1.b1)^2, s7=(C0.b2+C1.b0)^2, s8=(C1.b2+C0.b1)^2 and the result is
//   C0' = 3(xi t0 + t1) - 2 C0.b0,  3(xi t2 + t3) - 2 C0.b1,  3(xi t4 + t5) - 2 C0.b2
//   C1' = 3 xi (s8-t4-t5) + 2 C1.b0,  3 (s6-t0-t1) + 2 C1.b1,  3 (s7-t2-t3) + 2 C1.b2.
// Each lane squares its own three coefficients and one or two of the mixed sums (5 / 4 squarings), and the four
// multiplications by xi are shared two per lane: the even lane forms xi t2 and xi (s8-t4-t5), the odd lane xi t4, xi t0.
template <bool REDUCE, class X> GPBC_INLINE F6 f12p_cyclo_sqr(const X &x, const F6 &h) {
    F6 p = x.swap(h);
    F2 q0 = f2_sqr(h.b0), q1 = f2_sqr(h.b1), q2 = f2_sqr(h.b2);        // even: t1,t5,t2   odd: t3,t0,t4
    F2 sa = f2_sqr_n(f2p_add_swap(x, f2_sel(x.odd, h.b2, h.b0), h.b1));   // even: s6   odd: s8   (own + partner's b1)
    // s7 = (C0.b2 + C1.b0)^2 is the ninth squaring: its two Fe products are shared, the even lane forms the real part
    // (u0+u1)(u0-u1), the odd lane the imaginary part 2 u0 u1, so each lane carries 4.5 squarings.
    F2 u = f2_norm(f2_add(f2_sel(x.odd, p.b2, h.b2), f2_sel(x.odd, h.b0, p.b0)));        // C0.b2 + C1.b0 on both lanes
    // (a single Fe product has nine limb products per column, not eighteen: one operand may carry limbs up to 2^30, so the
    // sum u0 + u1 goes in as it is and the difference — limbs within +-2^29 — needs no normalisation either)
    Fe half = fe_mul(fe_sel(x.odd, fe_dbl(u.a0), fe_add(u.a0, u.a1)), fe_sel(x.odd, u.a1, fe_sub(u.a0, u.a1)));
    Fe phalf = x.swap(half);
    F2 sb{fe_sel(x.odd, phalf, half), fe_sel(x.odd, half, phalf)};        // s7 on both lanes
    F2 psa = x.swap(sa), pq0 = x.swap(q0), pq2 = x.swap(q2);
    // (the xi products travel UN-normalised — limbs within (-2^29, 2^30) — and are normalised once, inside the sums they
    // enter; the value reduction happens once, on the outputs)
    F2 A = f2_mul_xi(q2);                                                 // even: xi t2   odd: xi t4
    F2 B = f2_mul_xi(f2_sel(x.odd, q1, f2_norm(f2_sub(f2_sub(psa, pq2), q1))));          // even: xi (s8-t4-t5)   odd: xi t0
    F2 pA = x.swap(A), pB = x.swap(B);
    const F2 &psb = sb;
    F2 tt0 = f2_norm(f2_sel(x.odd, pB, f2_add(pB, q0)));
    F2 tt1 = f2_norm(f2_sel(x.odd, f2_sub(f2_sub(psa, q1), pq0), f2_add(A, pq0)));       // select first: one normalisation, not two
    F2 tt2 = f2_norm(f2_sel(x.odd, f2_sub(f2_sub(psb, pq2), q0), f2_add(pA, q1)));
    // out = 3 tt -+ 2 own coefficient  (minus on the even lane, plus on the odd lane)
    F6 sgn = f6p_sgn(x, h);
    F6 r{f2_norm(f2_add(f2_dbl(f2_norm(f2_add(tt0, sgn.b0))), tt0)),
         f2_norm(f2_add(f2_dbl(f2_norm(f2_add(tt1, sgn.b1))), tt1)),
         f2_norm(f2_add(f2_dbl(f2_norm(f2_add(tt2, sgn.b2))), tt2))};
    return REDUCE ? f6_reduce(r) : r;
}

// The same squaring for RUNS of squarings (x^u: 62 of them in runs of ~4), with the odd lane's sign ALTERNATING: the caller passes
// C1 on the odd lane as s = sigma C1 and gets back -sigma C1' — the conjugate of the square when sigma = +1.  Why: the Granger-Scott
// output is 3 T(x) - 2 conj(x), minus on the C0 half and plus on the C1 half; the C1 half of T is bilinear in (C0, C1), so with
// s in place of C1 the same code yields sigma T, its negative costs nothing (the differences are formed in the other order), and
// 3 (-sigma T) - 2 s = -sigma (3 T + 2 C1): BOTH lanes evaluate 3 tt - 2 h.  That removes the per-lane sign (54 multiplications
// by +-1 per squaring) and, with the operand normalised non-negative, lets the value reduction go in front of ONE normalisation
// (fe_cyclo_out) instead of two normalisations and a reduction after them: ~260 of ~3 700 instructions per squaring.
// Conjugation is a ring automorphism, so products are taken with the operand conjugated alike (f12p_expt_to tracks the parity).
// h: normalised by fe_norm as its last step (limbs 0..7 within [-2^4, 2^29 + 2^4]) and value-reduced — outputs of this function
// qualify, anything else goes through f6_norm first (f12p_cyclo_sqr_run).
template <class X> GPBC_INLINE F6 f12p_cyclo_sqr_alt(const X &x, const F6 &h) {
    F2 q0 = f2_sqr(h.b0), q1 = f2_sqr(h.b1), q2 = f2_sqr(h.b2);        // even: t1,t5,t2   odd: t3,t0,t4  (squares: sign-free)
    F2 sa = f2_sqr_n(f2p_add_swap(x, f2_sel(x.odd, h.b2, h.b0), h.b1));   // even: s6   odd: s8   (own + partner's b1)
    // C0.b2 + s.b0 on both lanes: each lane's own term is exactly what the partner needs (even: b2, odd: b0) — one select, one
    // swap-and-add
    F2 own = f2_sel(x.odd, h.b0, h.b2);
    F2 u = f2_norm(f2p_add_swap(x, own, own));
    // the NEGATED ninth square, -s7 = -(u0^2 - u1^2) - 2 u0 u1 i: the even lane forms (u0 + u1)(u1 - u0), the odd lane (2 u0)(-u1).  Negated
    // because only the odd lane's tt2 = t2 + t3 - s7 uses it and the partner's half must enter that sum by an ADDITION (see add_swap)
    Fe nhalf = fe_mul(fe_sel(x.odd, fe_dbl(u.a0), fe_add(u.a0, u.a1)), fe_sel(x.odd, fe_neg(u.a1), fe_sub(u.a1, u.a0)));
    F2 A = f2_mul_xi(q2);                                                 // even: xi t2   odd: xi t4
    // Every exchange below is "each lane sends what its partner needs", so that the received value has ONE use and rides on an
    // addition (f2p_add_swap) instead of costing a move of its own; where the partner must subtract, the SENDER subtracts:
    //   Y = q1 + partner's ((even lane receives q2 | odd lane receives q0) - sa)    even: t5 + t4 - s8 = -sigma x the cross term of the odd
    //                                                                                 lane's first coefficient (before xi)   odd: t0 + t1 - s6 = tt1
    //   Z = (even: A | odd: t3 - s7) + partner's (even receives q0 | odd receives q2) even: xi t2 + t3 = tt1                 odd: t2 + t3 - s7 = tt2
    F2 Y = f2p_add_swap(x, q1, f2_sub(f2_sel(x.odd, q2, q0), sa));
    F2 ns7{x.add_swap(q0.a0, nhalf), fe_add(q0.a1, nhalf)};               // odd lane: t3 - s7 (real part of -s7 from the partner, own imaginary part)
    F2 Z = f2p_add_swap(x, f2_sel(x.odd, ns7, A), f2_sel(x.odd, q0, q2));
    F2 B = f2_mul_xi(f2_sel(x.odd, q1, f2_norm(Y)));                      // even: xi (t4 + t5 - s8)   odd: xi t0
    F2 tt0 = f2_norm(f2p_add_swap(x, f2_sel(x.odd, f2_zero(), q0), B));   // even: xi t0 + t1          odd: -sigma xi (cross)
    F2 tt1 = f2_norm(f2_sel(x.odd, Y, Z));
    F2 tt2 = f2_norm(f2_sel(x.odd, Z, f2p_add_swap(x, q1, A)));          // even: xi t4 + t5
    return F6{F2{fe_cyclo_out(tt0.a0, h.b0.a0), fe_cyclo_out(tt0.a1, h.b0.a1)},
              F2{fe_cyclo_out(tt1.a0, h.b1.a0), fe_cyclo_out(tt1.a1, h.b1.a1)},
              F2{fe_cyclo_out(tt2.a0, h.b2.a0), fe_cyclo_out(tt2.a1, h.b2.a1)}};
}
// n squarings in a row.  `flipped` is the parity the caller tracks: false = the odd lane holds C1, true = it holds -C1 (the value
// is the conjugate of what the pair nominally represents).  Any N-class input is accepted (one normalisation up front).
template <class X> GPBC_INLINE F6 f12p_cyclo_sqr_run(const X &x, F6 r, int n, bool &flipped) {
    if (n <= 0) return r;
    r = f6_norm(r);
    for (int i = 0; i < n; i++) r = f12p_cyclo_sqr_alt(x, r);
    flipped ^= (bool)(n & 1);
    return r;
}

// x^(p^j): coefficient of w^k -> (conj if j odd)(g_k) * gamma_j[k]; even lane k = 0,2,4, odd lane k = 1,3,5
template <class X> GPBC_INLINE F6 f12p_frob(const X &x, const F6 &h, int j) {
    const bool cj = j & 1;
    F2 g[3] = {h.b0, h.b1, h.b2};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (cj) g[i] = f2_conj(g[i]);
        // even: k = 2i (gamma index 2i, none for i = 0); odd: k = 2i+1
        F2 ge = i ? gamma29(j, 2 * i) : f2_one();
        F2 go = gamma29(j, 2 * i + 1);
        F2 prod = f2_mul(g[i], f2_sel(x.odd, go, ge));
        g[i] = (i == 0) ? f2_sel(x.odd, prod, g[i]) : prod;
    }
    return F6{g[0], g[1], g[2]};
}

// Is the pair's value in the cyclotomic subgroup, x^(p^4 - p^2 + 1) = 1 ?  Tested as x^(p^4) x == x^(p^2): two Frobenius maps and one
// product.  The same answer on both lanes.  (Granger-Scott squarings are right exactly there, and so is conj = inverse; every pairing
// value is in it, a Miller value or a random Fp12 element is not.)
template <class X> GPBC_INLINE bool f12p_is_cyclotomic(const X &x, const F6 &h) {
    F6 t2 = f12p_frob(x, h, 2);
    F6 t4 = f12p_frob(x, f6_reduce(f6_norm(t2)), 2);
    F6 d = f6_norm(f6_sub(f12p_mul(x, f6_reduce(f6_norm(t4)), h), t2));
    const bool mine = f2_is_zero(d.b0) && f2_is_zero(d.b1) && f2_is_zero(d.b2);
    const Fe flag = x.swap(fe_sel(mine, fe_one(), fe_zero()));
    return mine && !fe_is_zero(flag);
}

// 1 / (c0 + c1 w) = (c0 - c1 w) / (c0^2 - v c1^2)
template <class X> GPBC_INLINE F6 f12p_inv(const X &x, const F6 &h) {
    F6 s = f6_sqr(h);                                         // even: c0^2, odd: c1^2
    F6 ps = x.swap(s);
    F6 a = f6_sel(x.odd, ps, s), b = f6_sel(x.odd, s, ps);
    F6 d = f6_inv(f6_norm(f6_sub(a, f6_mul_v(b))));          // both lanes (redundant, once per final exponentiation)
    F6 r = f6_mul(h, d);
    return f6_sel(x.odd, f6_neg(r), r);
}

}  // namespace gpbc
#endif
