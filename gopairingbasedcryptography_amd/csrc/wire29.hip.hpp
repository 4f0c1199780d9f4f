// Wire formats of G1 / G2 / GT, one element per lane: gnark-crypto ecc/bn254 marshal.go as published
// [EXT, parity unpinned].  Coordinates are big-endian canonical (non-Montgomery) integers; the two most significant
// bits of the first byte carry the form (p < 2^254 leaves them free):
//   00 uncompressed (X || Y, infinity = all zero)      10 compressed, Y is the smaller of {Y, -Y}
//   01 compressed infinity (rest zero)                 11 compressed, Y is the larger ("LexicographicallyLargest")
// G2 writes X.A1 || X.A0 [|| Y.A1 || Y.A0]; GT.Bytes() writes the twelve coefficients from C1.B2.A1 down to C0.B0.A0.
// Replaces (G1Affine|G2Affine|GT).Marshal / Unmarshal (reference serialization/serialization_curve.go:5-33),
// G1Affine.Bytes / GT.Bytes (ibe/gentry06_ibe/gentry06_ibe.go:322-324, hash/hash_from_gt.go:5-8).
// Decoding follows SetBytes: canonical-range check of every coordinate, square root for the compressed forms (error
// if none), curve / subgroup check (G1: on the curve, cofactor 1; G2: on the twist and in the order-r subgroup).
#ifndef GPBC_WIRE29_HIP_HPP
#define GPBC_WIRE29_HIP_HPP
#include "curve29.hip.hpp"

namespace gpbc {

constexpr uint8_t WIRE_MASK = 0xC0, WIRE_UNCOMPRESSED = 0x00, WIRE_INFINITY = 0x40, WIRE_SMALLEST = 0x80, WIRE_LARGEST = 0xC0;

GPBC_INLINE uint32_t wire_bswap(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24); }

// 32 big-endian bytes -> eight 32-bit words, least significant first; `first_mask` clears the flag bits of byte 0
GPBC_INLINE void wire_load_words(uint32_t w[8], const uint8_t *p, uint8_t first_mask) {
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) w[7 - i] = wire_bswap(q[i]);
    w[7] &= ((uint32_t)first_mask << 24) | 0x00ffffffu;
}
GPBC_INLINE void wire_store_words(uint8_t *p, const uint32_t w[8], uint8_t first_or) {
    uint32_t *q = reinterpret_cast<uint32_t *>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = wire_bswap(i == 0 ? (w[7] | ((uint32_t)first_or << 24)) : w[7 - i]);
}
GPBC_INLINE bool wire_words_zero(const uint32_t w[8]) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= w[i];
    return o == 0;
}
// w < p  (fp.Element.SetBytesCanonical rejects anything else)
GPBC_INLINE bool wire_words_canonical(const uint32_t w[8]) {
    constexpr uint32_t P32[8] = F29_P32;
    bool lt = false, decided = false;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (!decided && w[i] != P32[i]) { lt = w[i] < P32[i]; decided = true; }
    }
    return lt;
}
// plain integer words -> internal (x * 2^261), and back to canonical plain words
GPBC_INLINE Fe fe_from_plain_words(const uint32_t w[8]) {
    Fe x;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int bit = LB * i, wi = bit >> 5, sh = bit & 31;
        uint64_t two = (uint64_t)w[wi] | ((wi + 1 < 8) ? ((uint64_t)w[wi + 1] << 32) : 0);
        x.v[i] = (int32_t)((two >> sh) & (uint64_t)LMASK);
    }
    GPBC_B(set_class_n(x, 1.0);)
    constexpr int32_t C[NL] = F29_PLAIN_TO_INTERNAL;
    return fe_mul(x, fe_const(C));
}
GPBC_INLINE void fe_to_plain_words(uint32_t w[8], const Fe &a) {
    constexpr int32_t C[NL] = F29_PLAIN_ONE;
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
// canonical plain words > (p-1)/2
GPBC_INLINE bool wire_words_lex_largest(const uint32_t w[8]) {
    constexpr uint32_t P32[8] = F29_P32;
    // (p-1)/2 = p >> 1 (p odd)
    bool gt = false, decided = false;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        uint32_t h = (P32[i] >> 1) | (i < 7 ? (P32[i + 1] << 31) : 0u);
        if (!decided && w[i] != h) { gt = w[i] > h; decided = true; }
    }
    return gt;
}

// x^e for a constant exponent given as nine 29-bit limbs (e < 2^256): fixed 4-bit windows, left to right
template <class F> GPBC_NOINLINE F g_pow_limbs(const F &x, const int32_t (&E)[NL]) {
    F tab[16];
    g_set_one(tab[0]);
    tab[1] = x;
    for (int i = 2; i < 16; i++) tab[i] = g_mul(tab[i - 1], x);
    auto ebit = [&](int i) -> int {
        if (i < 0 || i >= 256) return 0;
        return (E[i / LB] >> (i % LB)) & 1;
    };
    F r;
    g_set_one(r);
    for (int w = 63; w >= 0; w--) {
        if (w != 63) { r = g_sqr(r); r = g_sqr(r); r = g_sqr(r); r = g_sqr(r); }
        int d = ebit(4 * w) | (ebit(4 * w + 1) << 1) | (ebit(4 * w + 2) << 2) | (ebit(4 * w + 3) << 3);
        if (d) r = g_mul(r, tab[d]);
    }
    return r;
}
GPBC_INLINE bool g_equal(const Fe &a, const Fe &b) { return fe_is_zero(fe_norm(fe_sub(a, b))); }
GPBC_INLINE bool g_equal(const F2 &a, const F2 &b) { return g_equal(a.a0, b.a0) && g_equal(a.a1, b.a1); }

// square roots (p = 3 mod 4); ok = false when a is not a square
GPBC_INLINE Fe fe_sqrt(const Fe &a, bool &ok) {
    constexpr int32_t E[NL] = F29_EXP_SQRT;
    Fe y = g_pow_limbs(a, E);
    ok = g_equal(fe_sqr(y), a);
    return y;
}
// is_square in Fp: the divstep-based Legendre symbol (fe29.hip.hpp), the power a^((p-1)/2) where that does not decide; 0 counts as
// a square (RFC 9380)
GPBC_INLINE bool fe_is_square(const Fe &a) {
    const int j = fe_legendre(a);
    if (j) return j > 0;
    constexpr int32_t E12[NL] = F29_EXP_P12;
    Fe l = g_pow_limbs(a, E12);
    return !fe_is_zero(fe_norm(fe_add(l, fe_one())));
}
// Fp2 by the norm: for a = a0 + a1 i with a1 != 0, s = sqrt(a0^2 + a1^2) exists exactly when a is a square; exactly one of
// (a0 + s) / 2 and (a0 - s) / 2 is a square in Fp (their product is -a1^2 / 4), the Legendre symbol says which, x0 is its root
// and x1 = a1 / (2 x0).  Two Fp powers, one symbol and one safegcd inversion (~170 k instructions) in place of the two Fp2 powers
// of the complex method below (~340 k).  Either root serves: every caller fixes the sign afterwards (sgn0 / the wire flag).
// (profiles/r02_variant_f2_sqrt.txt has the A/B run against the complex method.)
GPBC_INLINE F2 f2_sqrt(const F2 &a, bool &ok) {
    F2 x;
    if (fe_is_zero(fe_norm(a.a1))) {                             // a in Fp: sqrt(a0), or i sqrt(-a0) (-1 is not a square)
        bool ok0;
        const bool sq = fe_is_square(a.a0);
        Fe r = fe_sqrt(sq ? a.a0 : fe_neg(a.a0), ok0);
        x = sq ? F2{r, fe_zero()} : F2{fe_zero(), r};
    } else {
        bool ok0;
        const Fe n = fe_norm(fe_add(fe_sqr(a.a0), fe_sqr(a.a1)));
        const Fe s = fe_sqrt(n, ok0);
        Fe d = fe_halve(fe_norm(fe_add(a.a0, s)));
        if (!fe_is_square(d)) d = fe_norm(fe_sub(d, s));
        const Fe x0 = fe_sqrt(d, ok0);
        x = F2{x0, fe_mul(a.a1, fe_inv(fe_norm(fe_dbl(x0))))};
    }
    ok = g_equal(f2_sqr(x), a);
    return x;
}

// ---- coordinates <-> bytes
GPBC_INLINE bool fe_wire_load(Fe &r, const uint8_t *p, uint8_t first_mask, bool &zero) {
    uint32_t w[8];
    wire_load_words(w, p, first_mask);
    zero = wire_words_zero(w);
    if (!wire_words_canonical(w)) { r = fe_zero(); return false; }
    r = fe_from_plain_words(w);
    return true;
}
GPBC_INLINE void fe_wire_store(uint8_t *p, const Fe &a, uint8_t first_or) {
    uint32_t w[8];
    fe_to_plain_words(w, a);
    wire_store_words(p, w, first_or);
}
GPBC_INLINE bool fe_lex_largest(const Fe &a) {
    uint32_t w[8];
    fe_to_plain_words(w, a);
    return wire_words_lex_largest(w);
}
GPBC_INLINE bool f2_lex_largest(const F2 &a) {
    uint32_t w1[8], w0[8];
    fe_to_plain_words(w1, a.a1);
    if (!wire_words_zero(w1)) return wire_words_lex_largest(w1);
    fe_to_plain_words(w0, a.a0);
    return wire_words_lex_largest(w0);
}
GPBC_INLINE void wire_zero_bytes(uint8_t *p, int n_bytes) {
    uint32_t *q = reinterpret_cast<uint32_t *>(p);
    for (int i = 0; i < n_bytes / 4; i++) q[i] = 0;
}

// ---- G1
// in: gnark G1Affine (64 B, Montgomery); out: 64 B (Marshal / RawBytes) or 32 B (Bytes)
GPBC_INLINE void g1_wire_encode(uint8_t *out, const uint8_t *in, bool compressed) {
    if (bytes_all_zero(in, 16)) {
        wire_zero_bytes(out, compressed ? 32 : 64);
        if (compressed) out[0] = WIRE_INFINITY;
        return;
    }
    Fe x = fe_load(in), y = fe_load(in + 32);
    if (compressed) fe_wire_store(out, x, fe_lex_largest(y) ? WIRE_LARGEST : WIRE_SMALLEST);
    else { fe_wire_store(out, x, 0); fe_wire_store(out + 32, y, 0); }
}
// in: one element buffer of elem_bytes (32 or 64); out: gnark G1Affine; returns gnark's "no error"
GPBC_INLINE bool g1_wire_decode(uint8_t *out, const uint8_t *in, int elem_bytes) {
    wire_zero_bytes(out, 64);
    const uint8_t flag = in[0] & WIRE_MASK;
    constexpr int32_t B3[NL] = F29_B_G1;
    Fe x, y;
    bool zx, zy;
    if (flag == WIRE_UNCOMPRESSED) {
        if (elem_bytes < 64) return false;                              // io.ErrShortBuffer
        if (!fe_wire_load(x, in, 0xff, zx) || !fe_wire_load(y, in + 32, 0xff, zy)) return false;
        if (zx && zy) return true;                                      // infinity
        Fe rhs = fe_norm(fe_add(fe_mul(fe_sqr(x), x), fe_const(B3)));
        if (!g_equal(fe_sqr(y), rhs)) return false;                     // subgroup check = on the curve (cofactor 1)
    } else if (flag == WIRE_INFINITY) {
        uint32_t w[8];
        wire_load_words(w, in, (uint8_t)~WIRE_MASK);
        return wire_words_zero(w);                                      // ErrInvalidInfinityEncoding otherwise
    } else {
        if (!fe_wire_load(x, in, (uint8_t)~WIRE_MASK, zx)) return false;
        bool ok;
        y = fe_sqrt(fe_norm(fe_add(fe_mul(fe_sqr(x), x), fe_const(B3))), ok);
        if (!ok) return false;                                          // "square root doesn't exist"
        if (fe_lex_largest(y) != (flag == WIRE_LARGEST)) y = fe_neg(y);
    }
    fe_store(out, x);
    fe_store(out + 32, y);
    return true;
}

// ---- G2
GPBC_INLINE void g2_wire_encode(uint8_t *out, const uint8_t *in, bool compressed) {
    if (bytes_all_zero(in, 32)) {
        wire_zero_bytes(out, compressed ? 64 : 128);
        if (compressed) out[0] = WIRE_INFINITY;
        return;
    }
    F2 x = f2_load(in), y = f2_load(in + 64);
    const uint8_t fl = compressed ? (f2_lex_largest(y) ? WIRE_LARGEST : WIRE_SMALLEST) : 0;
    fe_wire_store(out, x.a1, fl);
    fe_wire_store(out + 32, x.a0, 0);
    if (!compressed) { fe_wire_store(out + 64, y.a1, 0); fe_wire_store(out + 96, y.a0, 0); }
}
// Q on the twist lies in the order-r subgroup  <=>  [x+1]Q + psi([x]Q) + psi^2([x]Q) = psi^3([2x]Q)   (x = the BN parameter u;
// El Housni, Guillevic, Piellard, ePrint 2022/348 §3 and §5.1 — the test gnark-crypto's G2 IsInSubGroup makes).  One 63-bit
// plain double-and-add instead of the 254-bit [r]Q (the GLV path of scalar_mul29 is not usable here: it presumes
// membership).  The oracle keeps the definition [r]Q = infinity; tests compare the two on subgroup points, random twist
// points, points of the cofactor group (including its small prime order 10069) and mixtures.
GPBC_INLINE JacP<F2> jac_psi_tw(const JacP<F2> &p, int j) {
    if (p.inf) return p;
    const bool cj = j & 1;
    return JacP<F2>{f2_mul(cj ? f2_conj(p.x) : p.x, gamma29(j, 2)), f2_mul(cj ? f2_conj(p.y) : p.y, gamma29(j, 3)), cj ? f2_conj(p.z) : p.z, false};
}
GPBC_NOINLINE bool g2_in_subgroup29(const F2 &x, const F2 &y) {
    constexpr uint64_t X = 4965661367192848881ull;
    AffP<F2> q{x, y, false};
    JacP<F2> xq;
    jac_set_inf(xq);
    for (int i = 62; i >= 0; i--) {
        jac_dbl(xq, xq);
        if ((X >> i) & 1) jac_add_mixed(xq, xq, q);
    }
    JacP<F2> lhs, t, rhs;
    jac_add_mixed(lhs, xq, q);                                   // [x+1]Q
    jac_add(lhs, lhs, jac_psi_tw(xq, 1));
    jac_add(lhs, lhs, jac_psi_tw(xq, 2));
    jac_dbl(t, xq);
    rhs = jac_psi_tw(t, 3);
    if (!rhs.inf) rhs.y = f2_neg(rhs.y);
    jac_add(t, lhs, rhs);                                        // lhs - rhs
    return t.inf;
}
// `in_subgroup(x, y)`: the membership test to use (g2_in_subgroup29, or its one-point-per-quad form in small calls: gpbc_wire.hip)
template <class Sub> GPBC_INLINE bool g2_wire_decode(uint8_t *out, const uint8_t *in, int elem_bytes, Sub &&in_subgroup) {
    wire_zero_bytes(out, 128);
    const uint8_t flag = in[0] & WIRE_MASK;
    constexpr int32_t BT[2][NL] = F29_B_G2;
    F2 x, y;
    bool z0, z1, z2, z3;
    if (flag == WIRE_UNCOMPRESSED) {
        if (elem_bytes < 128) return false;
        if (!fe_wire_load(x.a1, in, 0xff, z0) || !fe_wire_load(x.a0, in + 32, 0xff, z1) ||
            !fe_wire_load(y.a1, in + 64, 0xff, z2) || !fe_wire_load(y.a0, in + 96, 0xff, z3)) return false;
        if (z0 && z1 && z2 && z3) return true;
        F2 rhs = f2_norm(f2_add(f2_mul(f2_sqr(x), x), f2_const(BT)));
        if (!g_equal(f2_sqr(y), rhs)) return false;
    } else if (flag == WIRE_INFINITY) {
        uint32_t w[8], v[8];
        wire_load_words(w, in, (uint8_t)~WIRE_MASK);
        wire_load_words(v, in + 32, 0xff);
        return wire_words_zero(w) && wire_words_zero(v);
    } else {
        if (!fe_wire_load(x.a1, in, (uint8_t)~WIRE_MASK, z0) || !fe_wire_load(x.a0, in + 32, 0xff, z1)) return false;
        bool ok;
        y = f2_sqrt(f2_norm(f2_add(f2_mul(f2_sqr(x), x), f2_const(BT))), ok);
        if (!ok) return false;
        if (f2_lex_largest(y) != (flag == WIRE_LARGEST)) y = f2_neg(y);
    }
    if (!in_subgroup(x, y)) return false;                               // "subgroup check failed"
    f2_store(out, x);
    f2_store(out + 64, y);
    return true;
}
GPBC_INLINE bool g2_wire_decode(uint8_t *out, const uint8_t *in, int elem_bytes) {
    return g2_wire_decode(out, in, elem_bytes, [](const F2 &x, const F2 &y) { return g2_in_subgroup29(x, y); });
}

// ---- GT: coefficient k of the memory order C0.B0.A0 ... C1.B2.A1 sits at wire position 11 - k
GPBC_INLINE void gt_wire_encode(uint8_t *out, const uint8_t *in) {
    for (int k = 0; k < 12; k++) fe_wire_store(out + 32 * (11 - k), fe_load(in + 32 * k), 0);
}
GPBC_INLINE bool gt_wire_decode(uint8_t *out, const uint8_t *in) {
    bool good = true;
    for (int k = 0; k < 12; k++) {
        Fe c;
        bool z;
        good = fe_wire_load(c, in + 32 * (11 - k), 0xff, z) && good;
        fe_store(out + 32 * k, c);
    }
    if (!good) wire_zero_bytes(out, 384);
    return good;
}

}  // namespace gpbc
#endif
