// BN254 base field Fp on gfx950: 8 x 32-bit limbs, Montgomery form with R = 2^256 — the same bytes
// as gnark-crypto's fp.Element ([4]uint64 little-endian), so ABI buffers are used without conversion.
// Values are kept fully reduced in [0,p) after every operation (gnark's invariant), which is what makes
// every output bit-identical to the reference's CPU path regardless of the algorithm above it.
//
// Multiplication is CIOS over v_mad_u64_u32 (one 32x32+64 MAC per lane per instruction; measured peak
// 4.5 cycles per wave-instruction per SIMD, profiles/r01_microbench_valu.txt).
#ifndef GPBC_FP_CUH
#define GPBC_FP_CUH
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bn254_constants.cuh"

namespace gpbc {

typedef uint32_t u32;
typedef uint64_t u64;

struct Fp { u32 v[8]; };

__device__ __forceinline__ constexpr Fp fp_from_u64(u64 a, u64 b, u64 c, u64 d) {
    return Fp{{(u32)a, (u32)(a >> 32), (u32)b, (u32)(b >> 32), (u32)c, (u32)(c >> 32), (u32)d, (u32)(d >> 32)}};
}

// modulus limbs as immediates (the compiler folds them into instruction literals / SGPRs)
__device__ __forceinline__ constexpr u32 fp_p(int i) {
    constexpr u64 P[4] = BN254_P_LIMBS;
    return (u32)(P[i >> 1] >> ((i & 1) * 32));
}
constexpr u32 FP_PINV32 = BN254_P_INV_NEG32;

__device__ __forceinline__ Fp fp_zero() { return Fp{{0, 0, 0, 0, 0, 0, 0, 0}}; }
__device__ __forceinline__ Fp fp_one() {
    constexpr u64 L[4] = BN254_FP_ONE;
    return fp_from_u64(L[0], L[1], L[2], L[3]);
}

__device__ __forceinline__ bool fp_is_zero(const Fp &a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}
__device__ __forceinline__ bool fp_eq(const Fp &a, const Fp &b) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}

// t -= p if t >= p  (t < 2p, given as 8 limbs + optional carry bit)
__device__ __forceinline__ void fp_reduce_once(u32 t[8], u32 carry) {
    u32 d[8];
    u64 b = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 s = (u64)t[i] - fp_p(i) - b;
        d[i] = (u32)s;
        b = (s >> 63);
    }
    bool ge = carry || !b;
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = ge ? d[i] : t[i];
}

__device__ __forceinline__ Fp fp_add(const Fp &x, const Fp &y) {
    Fp z;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)x.v[i] + y.v[i];
        z.v[i] = (u32)c;
        c >>= 32;
    }
    fp_reduce_once(z.v, (u32)c);
    return z;
}
__device__ __forceinline__ Fp fp_sub(const Fp &x, const Fp &y) {
    Fp z;
    u64 b = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 s = (u64)x.v[i] - y.v[i] - b;
        z.v[i] = (u32)s;
        b = s >> 63;
    }
    u32 mask = (u32)0 - (u32)b;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)z.v[i] + (fp_p(i) & mask);
        z.v[i] = (u32)c;
        c >>= 32;
    }
    return z;
}
__device__ __forceinline__ Fp fp_neg(const Fp &x) { return fp_sub(fp_zero(), x); }
__device__ __forceinline__ Fp fp_dbl(const Fp &x) { return fp_add(x, x); }

// Montgomery product, CIOS. 64 + 64 + 8 multiply instructions.
__device__ __noinline__ Fp fp_mul(const Fp &x, const Fp &y) {
    u32 t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u64 acc = (u64)x.v[j] * y.v[i] + t[j] + c;
            t[j] = (u32)acc;
            c = acc >> 32;
        }
        u64 top = (u64)t[8] + c;           // < 2^33
        u32 m = t[0] * FP_PINV32;
        c = ((u64)m * fp_p(0) + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            u64 acc = (u64)m * fp_p(j) + t[j] + c;
            t[j - 1] = (u32)acc;
            c = acc >> 32;
        }
        top += c;
        t[7] = (u32)top;
        t[8] = (u32)(top >> 32);
    }
    fp_reduce_once(t, t[8]);
    Fp z;
#pragma unroll
    for (int i = 0; i < 8; i++) z.v[i] = t[i];
    return z;
}
__device__ __forceinline__ Fp fp_sqr(const Fp &x) { return fp_mul(x, x); }

// x^(p-2); 0 -> 0
__device__ __noinline__ Fp fp_inv(const Fp &x) {
    constexpr u64 E[4] = BN254_P_MINUS_2;
    Fp r = fp_one(), b = x;
    for (int i = 0; i < 254; i++) {
        if ((E[i >> 6] >> (i & 63)) & 1) r = fp_mul(r, b);
        b = fp_sqr(b);
    }
    return r;
}
__device__ __forceinline__ Fp fp_halve(const Fp &x) {
    // (x + (x odd ? p : 0)) >> 1
    u32 mask = (u32)0 - (x.v[0] & 1);
    u32 t[9];
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)x.v[i] + (fp_p(i) & mask);
        t[i] = (u32)c;
        c >>= 32;
    }
    t[8] = (u32)c;
    Fp z;
#pragma unroll
    for (int i = 0; i < 8; i++) z.v[i] = (t[i] >> 1) | (t[i + 1] << 31);
    return z;
}

__device__ __forceinline__ Fp fp_load(const uint8_t *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    return Fp{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
__device__ __forceinline__ void fp_store(uint8_t *p, const Fp &x) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    q[1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}

}  // namespace gpbc
#endif
