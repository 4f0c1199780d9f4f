"""BN254 big-integer oracle (TEST INFRASTRUCTURE ONLY — never imported by the product path).

PARITY UNPINNED: the arithmetic the reference calls lives in the un-vendored dependency
github.com/consensys/gnark-crypto v0.19.0 (reference go.mod:5); its source and a Go toolchain are
absent from the build container and the reference's tests hold no known-answer vectors
(SURVEY.md §4, §8c).  This file restates the *published* definition of that library's BN254
pairing from first principles:

  * curve  y^2 = x^3 + 3 over Fp, D-type twist y^2 = x^3 + 3/(9+i) over Fp2=Fp[i]/(i^2+1)
  * tower  Fp6 = Fp2[v]/(v^3-(9+i)), Fp12 = Fp6[w]/(w^2-v)          (gnark E2/E6/E12)
  * optimal-ate Miller function f_{6u+2,Q}(P) * l_{pi(Q)} * l_{-pi^2(Q)}
  * final exponent  s*(p^12-1)/r  with s = 2u(6u^2+3u+1)   (gnark's Fuentes-Castaneda hard part;
    see SURVEY.md §8a-2 — this fixes the GT bit pattern)
  * in-memory layouts: fp.Element = 4 little-endian u64 limbs of x*2^256 mod p (Montgomery)

It deliberately uses the *textbook* route (affine coordinates, plain binary double-and-add Miller
loop, final exponentiation by one big `pow`) so that it shares no algorithmic shortcut with the C
restatement (oracle/bn254_oracle.c: projective lines, NAF loop, addition chains) or with the HIP
kernels.  All three must agree bit for bit.

Call sites in the reference that this path serves: bn254.Pair (cpabe/bsw07/bsw07_cpabe.go:75,184;
access/tree/access_tree_node.go:106,110,119; bibe/afp25_bibe/afp25_bibe.go:227,395,399,403),
bn254.PairingCheck (signature/bls01_signature/bls_signature.go:81), ScalarMultiplication
(signature/bls01_signature/bls_signature.go:45,63), GT.Exp (access/tree/access_tree_node.go:156).
"""
import hashlib

# ----------------------------------------------------------------------------- constants
U = 4965661367192848881
P = 36 * U**4 + 36 * U**3 + 24 * U**2 + 6 * U + 1
R = 36 * U**4 + 36 * U**3 + 18 * U**2 + 6 * U + 1
assert P == 21888242871839275222246405745257275088696311157297823662689037894645226208583
assert R == 21888242871839275222246405745257275088548364400416034343698204186575808495617
ATE_LOOP = 6 * U + 2
S_COFACTOR = 2 * U * (6 * U * U + 3 * U + 1)
FINAL_EXP = S_COFACTOR * ((P**12 - 1) // R)
HARD_EXP = S_COFACTOR * ((P**4 - P**2 + 1) // R)
assert (P**12 - 1) % R == 0 and (P**4 - P**2 + 1) % R == 0
MONT_R = 1 << 256
B_G1 = 3

G1_GEN = (1, 2)
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)


# ----------------------------------------------------------------------------- Fp2
def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_neg(a): return ((-a[0]) % P, (-a[1]) % P)
def f2_conj(a): return (a[0], (-a[1]) % P)
def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2_sqr(a): return f2_mul(a, a)
def f2_scal(a, k): return (a[0] * k % P, a[1] * k % P)
def f2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * n % P, (-a[1]) * n % P)
def f2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_sqr(a)
        e >>= 1
    return r

F2_ZERO, F2_ONE = (0, 0), (1, 0)
XI = (9, 1)
def f2_mul_xi(a): return f2_mul(a, XI)

B_G2 = f2_mul((3, 0), f2_inv(XI))  # twist coefficient 3/(9+i)


# ----------------------------------------------------------------------------- Fp6 = Fp2[v]/(v^3 - xi)
def f6_add(a, b): return tuple(f2_add(x, y) for x, y in zip(a, b))
def f6_sub(a, b): return tuple(f2_sub(x, y) for x, y in zip(a, b))
def f6_neg(a): return tuple(f2_neg(x) for x in a)
def f6_mul(a, b):
    a0, a1, a2 = a
    b0, b1, b2 = b
    c0 = f2_add(f2_mul(a0, b0), f2_mul_xi(f2_add(f2_mul(a1, b2), f2_mul(a2, b1))))
    c1 = f2_add(f2_add(f2_mul(a0, b1), f2_mul(a1, b0)), f2_mul_xi(f2_mul(a2, b2)))
    c2 = f2_add(f2_add(f2_mul(a0, b2), f2_mul(a1, b1)), f2_mul(a2, b0))
    return (c0, c1, c2)
def f6_mul_v(a):  # multiply by v
    return (f2_mul_xi(a[2]), a[0], a[1])
def f6_inv(a):
    a0, a1, a2 = a
    t0 = f2_sub(f2_sqr(a0), f2_mul_xi(f2_mul(a1, a2)))
    t1 = f2_sub(f2_mul_xi(f2_sqr(a2)), f2_mul(a0, a1))
    t2 = f2_sub(f2_sqr(a1), f2_mul(a0, a2))
    d = f2_add(f2_mul(a0, t0), f2_mul_xi(f2_add(f2_mul(a2, t1), f2_mul(a1, t2))))
    di = f2_inv(d)
    return (f2_mul(t0, di), f2_mul(t1, di), f2_mul(t2, di))

F6_ZERO = (F2_ZERO, F2_ZERO, F2_ZERO)
F6_ONE = (F2_ONE, F2_ZERO, F2_ZERO)


# ----------------------------------------------------------------------------- Fp12 = Fp6[w]/(w^2 - v)
def f12_mul(a, b):
    a0, a1 = a
    b0, b1 = b
    return (f6_add(f6_mul(a0, b0), f6_mul_v(f6_mul(a1, b1))),
            f6_add(f6_mul(a0, b1), f6_mul(a1, b0)))
def f12_sqr(a): return f12_mul(a, a)
def f12_conj(a): return (a[0], f6_neg(a[1]))
def f12_inv(a):
    a0, a1 = a
    d = f6_sub(f6_mul(a0, a0), f6_mul_v(f6_mul(a1, a1)))
    di = f6_inv(d)
    return (f6_mul(a0, di), f6_neg(f6_mul(a1, di)))
def f12_pow(a, e):
    if e < 0:
        return f12_pow(f12_inv(a), -e)
    r = F12_ONE
    for bit in bin(e)[2:] if e else "":
        r = f12_sqr(r)
        if bit == "1":
            r = f12_mul(r, a)
    return r

F12_ONE = (F6_ONE, F6_ZERO)


def f12_from_w_coeffs(c):
    """c[k] in Fp2 is the coefficient of w^k, k=0..5 (w^2=v): w^0,w^2,w^4 -> C0.B0..B2; w^1,w^3,w^5 -> C1."""
    return ((c[0], c[2], c[4]), (c[1], c[3], c[5]))


# Frobenius constants, recomputed (not copied): gamma_k = xi^((p^k-1)/6)
GAMMA1 = f2_pow(XI, (P - 1) // 6)
GAMMA1_2 = f2_sqr(GAMMA1)
GAMMA1_3 = f2_mul(GAMMA1_2, GAMMA1)
NGAMMA = f2_pow(XI, (P * P - 1) // 6)       # = gamma^(p+1), lies in Fp
assert NGAMMA[1] == 0
NGAMMA_2 = f2_sqr(NGAMMA)
NGAMMA_3 = f2_mul(NGAMMA_2, NGAMMA)


def f12_frobenius(a, k=1):
    """x -> x^(p^k) computed coefficient-wise on the w-basis (independent of the pow-based FE)."""
    for _ in range(k):
        (b0, b1, b2), (d0, d1, d2) = a
        w = [b0, d0, b1, d1, b2, d2]          # coefficients of w^0..w^5
        g = F2_ONE
        out = []
        for c in w:
            out.append(f2_mul(f2_conj(c), g))
            g = f2_mul(g, GAMMA1)
        a = f12_from_w_coeffs(out)
    return a


# ----------------------------------------------------------------------------- curves (affine; None = infinity)
def g1_is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - B_G1) % P == 0

def g1_add(a, b):
    if a is None: return b
    if b is None: return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, P) % P
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x = (lam * lam - a[0] - b[0]) % P
    return (x, (lam * (a[0] - x) - a[1]) % P)

def g1_neg(a): return None if a is None else (a[0], (-a[1]) % P)

def g1_mul(a, k):
    k %= R
    r = None
    while k:
        if k & 1:
            r = g1_add(r, a)
        a = g1_add(a, a)
        k >>= 1
    return r

def g2_is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return f2_sub(f2_sqr(y), f2_add(f2_mul(f2_sqr(x), x), B_G2)) == F2_ZERO

def g2_add(a, b):
    if a is None: return b
    if b is None: return a
    if a[0] == b[0]:
        if f2_add(a[1], b[1]) == F2_ZERO:
            return None
        lam = f2_mul(f2_scal(f2_sqr(a[0]), 3), f2_inv(f2_scal(a[1], 2)))
    else:
        lam = f2_mul(f2_sub(b[1], a[1]), f2_inv(f2_sub(b[0], a[0])))
    x = f2_sub(f2_sub(f2_sqr(lam), a[0]), b[0])
    return (x, f2_sub(f2_mul(lam, f2_sub(a[0], x)), a[1]))

def g2_neg(a): return None if a is None else (a[0], f2_neg(a[1]))

def g2_mul(a, k):
    k %= R
    r = None
    while k:
        if k & 1:
            r = g2_add(r, a)
        a = g2_add(a, a)
        k >>= 1
    return r

def g2_frobenius(q):
    if q is None:
        return None
    return (f2_mul(f2_conj(q[0]), GAMMA1_2), f2_mul(f2_conj(q[1]), GAMMA1_3))

def g2_frobenius2(q):
    if q is None:
        return None
    return (f2_mul(q[0], NGAMMA_2), f2_mul(q[1], NGAMMA_3))


# ----------------------------------------------------------------------------- pairing
def _line(t, q, p):
    """Line through twist points t,q (tangent if equal) evaluated at P in G1; returns (line, t+q)."""
    if t[0] == q[0] and t[1] == q[1]:
        lam = f2_mul(f2_scal(f2_sqr(t[0]), 3), f2_inv(f2_scal(t[1], 2)))
    else:
        lam = f2_mul(f2_sub(q[1], t[1]), f2_inv(f2_sub(q[0], t[0])))
    x3 = f2_sub(f2_sub(f2_sqr(lam), t[0]), q[0])
    y3 = f2_sub(f2_mul(lam, f2_sub(t[0], x3)), t[1])
    # l = yP - lam*xP*w + (lam*xT - yT)*w^3   (untwist (x,y)->(x w^2, y w^3), w^6 = xi)
    c = [F2_ZERO] * 6
    c[0] = (p[1], 0)
    c[1] = f2_neg(f2_scal(lam, p[0]))
    c[3] = f2_sub(f2_mul(lam, t[0]), t[1])
    return f12_from_w_coeffs(c), (x3, y3)


def miller_loop(p, q):
    """Textbook optimal-ate Miller function (binary expansion of 6u+2), p in G1, q in G2 (affine)."""
    if p is None or q is None:
        return F12_ONE
    f = F12_ONE
    t = q
    for bit in bin(ATE_LOOP)[3:]:
        l, t = _line(t, t, p)
        f = f12_mul(f12_sqr(f), l)
        if bit == "1":
            l, t = _line(t, q, p)
            f = f12_mul(f, l)
    q1 = g2_frobenius(q)
    q2 = g2_neg(g2_frobenius2(q))
    l, t = _line(t, q1, p)
    f = f12_mul(f, l)
    l, t = _line(t, q2, p)
    f = f12_mul(f, l)
    return f


def final_exp_direct(f):
    """f^(s*(p^12-1)/r) by one square-and-multiply."""
    return f12_pow(f, FINAL_EXP)


def final_exp(f):
    """Same value, cheaper: easy part with conj/inverse/Frobenius^2, hard part by pow."""
    t = f12_mul(f12_conj(f), f12_inv(f))          # f^(p^6-1)
    t = f12_mul(f12_frobenius(t, 2), t)           # ^(p^2+1)
    return f12_pow(t, HARD_EXP)


def pair(ps, qs):
    """gnark bn254.Pair semantics: product of pairings, pairs with infinity skipped."""
    if len(ps) != len(qs) or len(ps) == 0:
        raise ValueError("invalid inputs sizes")
    f = F12_ONE
    for p, q in zip(ps, qs):
        f = f12_mul(f, miller_loop(p, q))
    return final_exp(f)


def gt_exp(x, k):
    """gnark GT.Exp: k=0 -> 1, k<0 -> inverse then |k|."""
    return f12_pow(x, k)


# ----------------------------------------------------------------------------- gnark in-memory layouts
def fp_to_mont_bytes(x): return (x * MONT_R % P).to_bytes(32, "little")
def fp_from_mont_bytes(b): return int.from_bytes(b, "little") * pow(MONT_R, -1, P) % P
def fr_to_mont_bytes(x): return (x * MONT_R % R).to_bytes(32, "little")
def fr_from_mont_bytes(b): return int.from_bytes(b, "little") * pow(MONT_R, -1, R) % R
def scalar_to_bytes(k): return (k % (1 << 256)).to_bytes(32, "little")

def g1_to_bytes(pt):
    if pt is None:
        return bytes(64)
    return fp_to_mont_bytes(pt[0]) + fp_to_mont_bytes(pt[1])

def g1_from_bytes(b):
    if b == bytes(64):
        return None
    return (fp_from_mont_bytes(b[0:32]), fp_from_mont_bytes(b[32:64]))

def f2_to_bytes(a): return fp_to_mont_bytes(a[0]) + fp_to_mont_bytes(a[1])
def f2_from_bytes(b): return (fp_from_mont_bytes(b[0:32]), fp_from_mont_bytes(b[32:64]))

def g2_to_bytes(pt):
    if pt is None:
        return bytes(128)
    return f2_to_bytes(pt[0]) + f2_to_bytes(pt[1])

def g2_from_bytes(b):
    if b == bytes(128):
        return None
    return (f2_from_bytes(b[0:64]), f2_from_bytes(b[64:128]))

def gt_to_bytes(a):
    """E12{C0,C1 E6{B0,B1,B2 E2{A0,A1}}} in declaration order = 12 x 32 B."""
    return b"".join(f2_to_bytes(c) for half in a for c in half)

def gt_from_bytes(b):
    cs = [f2_from_bytes(b[64 * i:64 * i + 64]) for i in range(6)]
    return ((cs[0], cs[1], cs[2]), (cs[3], cs[4], cs[5]))

def gt_to_canonical_bytes(a):
    """gnark GT.Bytes(): 12 x 32-byte big-endian canonical values, order C1.B2.A1 ... C0.B0.A0."""
    flat = [x for half in a for c in half for x in c]
    return b"".join(x.to_bytes(32, "big") for x in reversed(flat))


# ----------------------------------------------------------------------------- wire formats
# gnark-crypto ecc/bn254 marshal.go as published [EXT, parity unpinned]: big-endian canonical (non-Montgomery)
# coordinates; the two most significant bits of the first byte carry the form (p < 2^254 leaves them free):
#   00 uncompressed (X || Y; the point at infinity is all zero)     10 compressed, Y is the smaller of {Y, -Y}
#   01 compressed infinity (rest zero)                              11 compressed, Y is the larger
# G2 writes X.A1 || X.A0 [|| Y.A1 || Y.A0].  "Larger" = LexicographicallyLargest: for Fp y > (p-1)/2; for Fp2 the
# test is made on A1 unless A1 = 0, then on A0.  Reference call sites: serialization/serialization_curve.go:5-33
# (Marshal / Unmarshal), ibe/gentry06_ibe/gentry06_ibe.go:322-324 (G1 Bytes, GT Bytes), hash/hash_from_gt.go:5-8.
M_MASK, M_UNCOMPRESSED, M_INFINITY, M_SMALLEST, M_LARGEST = 0xC0, 0x00, 0x40, 0x80, 0xC0

def fp_lex_largest(y): return y > (P - 1) // 2
def f2_lex_largest(y): return fp_lex_largest(y[1]) if y[1] != 0 else fp_lex_largest(y[0])

def fp_sqrt(a):
    """A square root of a in Fp (p = 3 mod 4) or None."""
    y = pow(a, (P + 1) // 4, P)
    return y if y * y % P == a % P else None

def f2_sqrt(a):
    """A square root of a in Fp2 or None (norm method; either root may be returned, callers fix the sign)."""
    a0, a1 = a[0] % P, a[1] % P
    if a1 == 0:
        y = fp_sqrt(a0)
        if y is not None:
            return (y, 0)
        y = fp_sqrt((-a0) % P)               # sqrt(-|a0|) = i sqrt(|a0|)
        return None if y is None else (0, y)
    n = fp_sqrt((a0 * a0 + a1 * a1) % P)
    if n is None:
        return None
    half = pow(2, -1, P)
    t = (a0 + n) * half % P
    x0 = fp_sqrt(t)
    if x0 is None:
        t = (a0 - n) * half % P
        x0 = fp_sqrt(t)
        if x0 is None:
            return None
    x1 = a1 * pow(2 * x0, -1, P) % P
    r = (x0, x1)
    return r if f2_sqr(r) == (a0, a1) else None

def g1_marshal(pt, compressed=False):
    """G1Affine.Marshal()/RawBytes() (64 B) or Bytes() (32 B)."""
    if compressed:
        if pt is None:
            return bytes([M_INFINITY]) + bytes(31)
        b = bytearray(pt[0].to_bytes(32, "big"))
        b[0] |= M_LARGEST if fp_lex_largest(pt[1]) else M_SMALLEST
        return bytes(b)
    if pt is None:
        return bytes(64)
    return pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")

def g2_marshal(pt, compressed=False):
    if compressed:
        if pt is None:
            return bytes([M_INFINITY]) + bytes(63)
        b = bytearray(pt[0][1].to_bytes(32, "big") + pt[0][0].to_bytes(32, "big"))
        b[0] |= M_LARGEST if f2_lex_largest(pt[1]) else M_SMALLEST
        return bytes(b)
    if pt is None:
        return bytes(128)
    return b"".join(v.to_bytes(32, "big") for v in (pt[0][1], pt[0][0], pt[1][1], pt[1][0]))

def g2_in_subgroup(pt):
    return pt is None or (g2_is_on_curve(pt) and g2_mul_plain(pt, R) is None)

def g2_mul_plain(a, k):
    """double-and-add without reducing k mod r (the point may lie outside the order-r subgroup)."""
    acc = None
    for bit in bin(k)[2:]:
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, a)
    return acc

def g1_unmarshal(buf):
    """G1Affine.SetBytes on one element buffer (32 or 64 B): (point, ok).  ok = False where gnark returns an error
    (short buffer, non-canonical coordinate, bad infinity encoding, no square root, not on the curve)."""
    if len(buf) < 32:
        return None, False
    flag = buf[0] & M_MASK
    if flag == M_UNCOMPRESSED:
        if len(buf) < 64:
            return None, False
        x, y = int.from_bytes(buf[0:32], "big"), int.from_bytes(buf[32:64], "big")
        if x >= P or y >= P:
            return None, False
        if x == 0 and y == 0:
            return None, True
        return ((x, y), True) if g1_is_on_curve((x, y)) else (None, False)
    if flag == M_INFINITY:
        ok = (buf[0] & ~M_MASK & 0xFF) == 0 and not any(buf[1:32])
        return None, ok
    x = int.from_bytes(bytes([buf[0] & ~M_MASK & 0xFF]) + bytes(buf[1:32]), "big")
    if x >= P:
        return None, False
    y = fp_sqrt((x * x * x + B_G1) % P)
    if y is None:
        return None, False
    if fp_lex_largest(y) != (flag == M_LARGEST):
        y = (-y) % P
    return (x, y), True

def g2_unmarshal(buf):
    """G2Affine.SetBytes on one element buffer (64 or 128 B): (point, ok); includes the subgroup check."""
    if len(buf) < 64:
        return None, False
    flag = buf[0] & M_MASK
    rd = lambda i: int.from_bytes(buf[32 * i:32 * i + 32], "big")
    if flag == M_UNCOMPRESSED:
        if len(buf) < 128:
            return None, False
        x1, x0, y1, y0 = rd(0), rd(1), rd(2), rd(3)
        if max(x1, x0, y1, y0) >= P:
            return None, False
        if x0 == x1 == y0 == y1 == 0:
            return None, True
        pt = ((x0, x1), (y0, y1))
        return (pt, True) if g2_in_subgroup(pt) else (None, False)
    if flag == M_INFINITY:
        ok = (buf[0] & ~M_MASK & 0xFF) == 0 and not any(buf[1:64])
        return None, ok
    x1 = int.from_bytes(bytes([buf[0] & ~M_MASK & 0xFF]) + bytes(buf[1:32]), "big")
    x0 = rd(1)
    if x1 >= P or x0 >= P:
        return None, False
    x = (x0, x1)
    y = f2_sqrt(f2_add(f2_mul(f2_sqr(x), x), B_G2))
    if y is None:
        return None, False
    if f2_lex_largest(y) != (flag == M_LARGEST):
        y = f2_neg(y)
    pt = (x, y)
    return (pt, True) if g2_in_subgroup(pt) else (None, False)

def gt_marshal(a): return gt_to_canonical_bytes(a)

def gt_unmarshal(buf):
    """GT.SetBytes: 12 canonical big-endian coefficients, C1.B2.A1 first; (value, ok)."""
    if len(buf) < 384:
        return None, False
    vals = [int.from_bytes(buf[32 * i:32 * i + 32], "big") for i in range(12)]
    if max(vals) >= P:
        return None, False
    flat = list(reversed(vals))                       # C0.B0.A0, C0.B0.A1, ..., C1.B2.A1
    cs = [(flat[2 * i], flat[2 * i + 1]) for i in range(6)]
    return ((cs[0], cs[1], cs[2]), (cs[3], cs[4], cs[5])), True


# ----------------------------------------------------------------------------- hash to curve
# bn254.HashToG1 / HashToG2 (reference hash/hash_to.go:113-119,169-175,204-210,271-277) [EXT, parity unpinned]:
# RFC 9380 hash_to_curve with expand_message_xmd(SHA-256), L = 48, two field elements, the Shallue-van de Woestijne map
# (RFC 9380 §6.6.1 / Appendix F.1 straight-line version; gnark-crypto's MapToCurve1 / MapToCurve2 are that listing), point
# addition, and for G2 the cofactor clearing of Fuentes-Castaneda et al. §6.1:  [x]P + psi([3x]P) + psi^2([x]P) + psi^3(P).
# Z is the first value accepted by RFC 9380 Appendix H.1 find_z_svdw (candidates 1, -1, 2, -2, ...): Z = 1 for both
# y^2 = x^3 + 3 and the twist (computed below, not assumed).
def expand_message_xmd(msg, dst, n):
    """RFC 9380 §5.3.1 with SHA-256 (b = 32, s = 64)."""
    if len(dst) > 255:
        dst = hashlib.sha256(b"H2C-OVERSIZE-DST-" + dst).digest()
    ell = (n + 31) // 32
    if ell > 255 or n > 65535:
        raise ValueError("expand_message_xmd: length too large")
    dst_prime = dst + bytes([len(dst)])
    b0 = hashlib.sha256(bytes(64) + msg + n.to_bytes(2, "big") + b"\x00" + dst_prime).digest()
    bi = hashlib.sha256(b0 + b"\x01" + dst_prime).digest()
    out = bi
    for i in range(2, ell + 1):
        bi = hashlib.sha256(bytes(a ^ b for a, b in zip(b0, bi)) + bytes([i]) + dst_prime).digest()
        out += bi
    return out[:n]

H2C_L = 48

def hash_to_field_fp(msg, dst, count):
    u = expand_message_xmd(msg, dst, count * H2C_L)
    return [int.from_bytes(u[H2C_L * i:H2C_L * (i + 1)], "big") % P for i in range(count)]

def hash_to_field_fp2(msg, dst, count):
    u = expand_message_xmd(msg, dst, 2 * count * H2C_L)
    e = [int.from_bytes(u[H2C_L * i:H2C_L * (i + 1)], "big") % P for i in range(2 * count)]
    return [(e[2 * i], e[2 * i + 1]) for i in range(count)]

class _FpOps:
    zero, one = 0, 1
    b = B_G1
    @staticmethod
    def add(a, b): return (a + b) % P
    @staticmethod
    def sub(a, b): return (a - b) % P
    @staticmethod
    def neg(a): return (-a) % P
    @staticmethod
    def mul(a, b): return a * b % P
    @staticmethod
    def inv0(a): return pow(a, P - 2, P)
    @staticmethod
    def is_square(a): return pow(a, (P - 1) // 2, P) in (0, 1)
    @staticmethod
    def sqrt(a): return fp_sqrt(a)
    @staticmethod
    def sgn0(a): return a & 1
    @staticmethod
    def small(k): return k % P

class _Fp2Ops:
    zero, one = F2_ZERO, F2_ONE
    b = B_G2
    add, sub, neg, mul = staticmethod(f2_add), staticmethod(f2_sub), staticmethod(f2_neg), staticmethod(f2_mul)
    @staticmethod
    def inv0(a): return F2_ZERO if a == F2_ZERO else f2_inv(a)
    @staticmethod
    def is_square(a): return pow((a[0] * a[0] + a[1] * a[1]) % P, (P - 1) // 2, P) in (0, 1)
    @staticmethod
    def sqrt(a): return f2_sqrt(a)
    @staticmethod
    def sgn0(a): return (a[0] & 1) | ((a[0] == 0) & (a[1] & 1))       # RFC 9380 §4.1, m = 2
    @staticmethod
    def small(k): return (k % P, 0)

def find_z_svdw(F):
    """RFC 9380 Appendix H.1 (A = 0)."""
    g = lambda x: F.add(F.mul(F.mul(x, x), x), F.b)
    ctr = 1
    while True:
        for z in (F.small(ctr), F.small(-ctr)):
            gz = g(z)
            if gz == F.zero:
                continue
            h = F.neg(F.mul(F.mul(F.small(3), F.mul(z, z)), F.inv0(F.mul(F.small(4), gz))))
            if h == F.zero or not F.is_square(h):
                continue
            if F.is_square(gz) or F.is_square(g(F.mul(F.neg(z), F.inv0(F.small(2))))):
                return z
        ctr += 1

def svdw_constants(F):
    z = find_z_svdw(F)
    gz = F.add(F.mul(F.mul(z, z), z), F.b)
    c1 = gz
    c2 = F.mul(F.neg(z), F.inv0(F.small(2)))
    three_z2 = F.mul(F.small(3), F.mul(z, z))
    c3 = F.sqrt(F.mul(F.neg(gz), three_z2))
    if F.sgn0(c3) == 1:
        c3 = F.neg(c3)                                   # sgn0(c3) MUST equal 0
    c4 = F.mul(F.mul(F.small(-4), gz), F.inv0(three_z2))
    return z, c1, c2, c3, c4

def map_to_curve_svdw(F, consts, u):
    """RFC 9380 Appendix F.1, steps 1-35 (A = 0)."""
    z, c1, c2, c3, c4 = consts
    g = lambda x: F.add(F.mul(F.mul(x, x), x), F.b)
    tv1 = F.mul(F.mul(u, u), c1)
    tv2 = F.add(F.one, tv1)
    tv1 = F.sub(F.one, tv1)
    tv3 = F.inv0(F.mul(tv1, tv2))
    tv4 = F.mul(F.mul(F.mul(u, tv1), tv3), c3)
    x1 = F.sub(c2, tv4)
    gx1 = g(x1)
    e1 = F.is_square(gx1)
    x2 = F.add(c2, tv4)
    gx2 = g(x2)
    e2 = F.is_square(gx2) and not e1
    x3 = F.mul(F.mul(tv2, tv2), tv3)
    x3 = F.add(F.mul(F.mul(x3, x3), c4), z)
    x = x1 if e1 else x3
    x = x2 if e2 else x
    y = F.sqrt(g(x))
    if F.sgn0(u) != F.sgn0(y):
        y = F.neg(y)
    return (x, y)

SVDW_G1 = svdw_constants(_FpOps)
SVDW_G2 = svdw_constants(_Fp2Ops)

def g2_clear_cofactor(q):
    """[x]Q + psi([3x]Q) + psi^2([x]Q) + psi^3(Q), x = u (the curve parameter)."""
    xq = g2_mul_plain(q, U)
    t1 = g2_frobenius(g2_add(g2_add(xq, xq), xq))
    t2 = g2_frobenius(g2_frobenius(xq))
    t3 = g2_frobenius(g2_frobenius(g2_frobenius(q)))
    return g2_add(g2_add(g2_add(xq, t1), t2), t3)

def map_fields_to_g1(u0, u1):
    """The tail of HashToG1: both field elements through the map, then one addition (cofactor 1)."""
    return g1_add(map_to_curve_svdw(_FpOps, SVDW_G1, u0), map_to_curve_svdw(_FpOps, SVDW_G1, u1))

def map_fields_to_g2(u0, u1):
    return g2_clear_cofactor(g2_add(map_to_curve_svdw(_Fp2Ops, SVDW_G2, u0), map_to_curve_svdw(_Fp2Ops, SVDW_G2, u1)))

def hash_to_g1(msg, dst):
    u = hash_to_field_fp(msg, dst, 2)
    return map_fields_to_g1(u[0], u[1])

def hash_to_g2(msg, dst):
    u = hash_to_field_fp2(msg, dst, 2)
    return map_fields_to_g2(u[0], u[1])


# ----------------------------------------------------------------------------- deterministic synthetic inputs
SEED = 0x424E323534

def bench_scalar(tag, i):
    """k(tag,i) = SHA-256("gpbc-bench/v1/" || tag || LE64(seed) || LE64(i)) mod r   (SURVEY.md §8d)."""
    h = hashlib.sha256(b"gpbc-bench/v1/" + tag.encode() + SEED.to_bytes(8, "little")
                       + i.to_bytes(8, "little")).digest()
    return int.from_bytes(h, "big") % R


# ----------------------------------------------------------------------------- self checks
def self_check(verbose=False):
    assert P % 4 == 3 and P.bit_length() == 254 and R.bit_length() == 254
    assert g1_is_on_curve(G1_GEN) and g2_is_on_curve(G2_GEN)
    assert g1_mul(G1_GEN, R - 1) == g1_neg(G1_GEN) and g1_add(g1_mul(G1_GEN, R - 1), G1_GEN) is None
    assert g2_add(g2_mul(G2_GEN, R - 1), G2_GEN) is None
    assert B_G2 == (19485874751759354771024239261021720505790618469301721065564631296452457478373,
                    266929791119991161246907387137283842545076965332900288569378510910307636690)
    # lambda-chain exponent identity
    l0 = 1 + 6 * U + 12 * U**2 + 12 * U**3
    l1 = 4 * U + 6 * U**2 + 12 * U**3
    l2 = 6 * U + 6 * U**2 + 12 * U**3
    l3 = -1 + 4 * U + 6 * U**2 + 12 * U**3
    assert l0 + l1 * P + l2 * P**2 + l3 * P**3 == HARD_EXP
    e = pair([G1_GEN], [G2_GEN])
    assert e != F12_ONE
    assert f12_pow(e, R) == F12_ONE
    # Frobenius coefficient map == pow
    assert f12_frobenius(e) == f12_pow(e, P)
    # direct pow == split final exp
    f = miller_loop(G1_GEN, G2_GEN)
    assert final_exp_direct(f) == e
    # bilinearity
    a, b = bench_scalar("selfcheck-a", 0), bench_scalar("selfcheck-b", 0)
    lhs = pair([g1_mul(G1_GEN, a)], [g2_mul(G2_GEN, b)])
    assert lhs == f12_pow(e, a * b % R)
    # product / inverse identities used by multi-pairing restructuring
    assert pair([g1_neg(G1_GEN)], [G2_GEN]) == f12_inv(e)
    assert pair([G1_GEN, g1_neg(G1_GEN)], [G2_GEN, G2_GEN]) == F12_ONE
    if verbose:
        print("self-check OK")
    return True


if __name__ == "__main__":
    self_check(verbose=True)
