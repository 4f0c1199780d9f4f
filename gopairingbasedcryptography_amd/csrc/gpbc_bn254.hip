// libgpbc_bn254.so — kernels and C ABI (include/gpbc_bn254.h) of the MI355X BN254 engine.
// gfx950 only. One batch element per lane; ABI buffers are gnark in-memory structs (AoS).
// Arithmetic: csrc/fe29.cuh (9 x 29-bit signed limbs, lazy reduction) -> tower29 -> curve29 / pairing29.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>
#include "../../include/gpbc_bn254.h"
#include "curve29.cuh"
#include "pairing29.cuh"
#include "pairing29_pair.cuh"
#include "wire29.cuh"
#include "h2c29.cuh"

using namespace gpbc;

// =============================================================================================== kernels
// One batch element per lane.  Inputs/outputs are gnark structs (Montgomery R = 2^256, canonical); each kernel
// converts to the internal 9 x 29-bit signed-limb form on load and back to canonical bytes on store.
constexpr int BLOCK = 64;
#ifndef GPBC_WAVES_PER_SIMD
#define GPBC_WAVES_PER_SIMD 2
#endif
#define GPBC_KERNEL __global__ void __launch_bounds__(BLOCK, GPBC_WAVES_PER_SIMD)
// G1 arithmetic is light enough on registers for three waves per SIMD (168 VGPRs): measured +12 % over two, while the Fp2 /
// Fp12 kernels lose 15-75 % to the extra spills (profiles/r01_microbench_valu2.txt explains the gain: a wave issues
// at most one VALU instruction per ~4.5 cycles, so the 2.4-cycle VOP2 glue only gets cheaper with more waves).
#ifndef GPBC_WAVES_G1
#define GPBC_WAVES_G1 3
#endif
#define GPBC_KERNEL_G1 __global__ void __launch_bounds__(BLOCK, GPBC_WAVES_G1)

__device__ __forceinline__ bool g1_bytes_inf(const uint8_t *p) { return bytes_all_zero(p, 16); }
__device__ __forceinline__ bool g2_bytes_inf(const uint8_t *p) { return bytes_all_zero(p, 32); }

// ---- Miller loop, two kernels.  The 88 lines of a pairing depend only on (P, Q), so the G2 arithmetic (phase A)
// and the Fp12 accumulator (phase B) run as separate kernels, each with its own register budget; the lines travel
// through an HBM workspace in internal limb form, word-major so that a wave's accesses are contiguous:
//   lines[(step * 54 + word) * stride + lane]        (19 KB per pairing, streamed once each way)
__device__ __forceinline__ void line_store(int32_t *__restrict__ buf, size_t stride, size_t lane, int step, const LineS &l) {
    int32_t *b = buf + (size_t)step * LINE_WORDS * stride + lane;
    const Fe *fe[6] = {&l.c0.a0, &l.c0.a1, &l.c3.a0, &l.c3.a1, &l.c4.a0, &l.c4.a1};
#pragma unroll
    for (int e = 0; e < 6; e++)
#pragma unroll
        for (int i = 0; i < NL; i++) b[(size_t)(e * NL + i) * stride] = fe[e]->v[i];
}
__device__ __forceinline__ LineS line_load(const int32_t *__restrict__ buf, size_t stride, size_t lane, int step) {
    const int32_t *b = buf + (size_t)step * LINE_WORDS * stride + lane;
    LineS l;
    Fe *fe[6] = {&l.c0.a0, &l.c0.a1, &l.c3.a0, &l.c3.a1, &l.c4.a0, &l.c4.a1};
#pragma unroll
    for (int e = 0; e < 6; e++)
#pragma unroll
        for (int i = 0; i < NL; i++) fe[e]->v[i] = b[(size_t)(e * NL + i) * stride];
    return l;
}

GPBC_KERNEL k_miller_lines(const uint8_t *__restrict__ P, const uint8_t *__restrict__ Q, int32_t *__restrict__ lines, size_t n, size_t stride) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint8_t *p = P + i * GPBC_G1_BYTES, *q = Q + i * GPBC_G2_BYTES;
    if (g1_bytes_inf(p) || g2_bytes_inf(q)) return;          // phase B skips this pair as well
    G1A a{fe_load(p), fe_load(p + 32)};
    G2A b{f2_load(q), f2_load(q + 64)};
    int step = 0;
    miller_lines(a, b, [&](const LineS &l) { line_store(lines, stride, i, step++, l); });
}

// Phase B and the final exponentiation run with one pairing per LANE PAIR (even lane: C0, odd lane: C1 of every Fp12
// value, halves swapped by DPP — tower29_pair.cuh), so a batch of n pairings is a grid of 2n lanes.
GPBC_KERNEL k_miller_accumulate(const uint8_t *__restrict__ P, const uint8_t *__restrict__ Q, const int32_t *__restrict__ lines,
                                uint8_t *__restrict__ f_out, size_t n, size_t stride) {
    size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t i = lane >> 1;
    if (i >= n) return;
    PairDpp x{(bool)(lane & 1)};
    const uint8_t *p = P + i * GPBC_G1_BYTES, *q = Q + i * GPBC_G2_BYTES;
    F6 h;
    if (g1_bytes_inf(p) || g2_bytes_inf(q)) h = f12p_one(x);
    else {
        int step = 0;
        h = miller_accumulate_pair(x, [&]() -> LineS { return line_load(lines, stride, i, step++); });
    }
    f6_store(f_out + i * GPBC_GT_BYTES + (x.odd ? 192 : 0), h);
}

// Multi-pairing form of the two phases (host entry gpbc_multi_pair / gpbc_pairing_check): the pairs of a segment are cut
// into chunks of at most MULTI_CHUNK pairs, one lane pair accumulates a whole chunk with SHARED squarings
// (miller_accumulate_multi), and the lines workspace is laid out by slot = i * n_chunks + c (pair i of chunk c) so that
// adjacent lane pairs read adjacent words whatever the chunk lengths are.
constexpr int MULTI_CHUNK = 8;
constexpr size_t MULTI_GROUP = 65536;
GPBC_KERNEL k_miller_lines_chunks(const uint8_t *__restrict__ P, const uint8_t *__restrict__ Q, int32_t *__restrict__ lines,
                                  const uint64_t *__restrict__ chunk_off, size_t n_chunks, size_t n_slots) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n_slots) return;
    const size_t i = t / n_chunks, c = t % n_chunks;
    const uint64_t pair = chunk_off[c] + i;
    if (pair >= chunk_off[c + 1]) return;                      // slot beyond this chunk's length
    const uint8_t *p = P + pair * GPBC_G1_BYTES, *q = Q + pair * GPBC_G2_BYTES;
    if (g1_bytes_inf(p) || g2_bytes_inf(q)) return;
    G1A a{fe_load(p), fe_load(p + 32)};
    G2A b{f2_load(q), f2_load(q + 64)};
    int step = 0;
    miller_lines(a, b, [&](const LineS &l) { line_store(lines, n_slots, t, step++, l); });
}
GPBC_KERNEL k_miller_accumulate_chunks(const uint8_t *__restrict__ P, const uint8_t *__restrict__ Q, const int32_t *__restrict__ lines,
                                       const uint64_t *__restrict__ chunk_off, uint8_t *__restrict__ f_out, size_t n_chunks, size_t n_slots) {
    size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t c = lane >> 1;
    if (c >= n_chunks) return;
    PairDpp x{(bool)(lane & 1)};
    const uint64_t lo = chunk_off[c], hi = chunk_off[c + 1];
    int vi[MULTI_CHUNK], m = 0;                                // positions of the pairs that have no point at infinity
    for (uint64_t i = 0; i < hi - lo && i < (uint64_t)MULTI_CHUNK; i++)
        if (!g1_bytes_inf(P + (lo + i) * GPBC_G1_BYTES) && !g2_bytes_inf(Q + (lo + i) * GPBC_G2_BYTES)) vi[m++] = (int)i;
    F6 h;
    if (m == 0) h = f12p_one(x);
    else h = miller_accumulate_multi(x, m, [&](int p, int li) -> LineS { return line_load(lines, n_slots, (size_t)vi[p] * n_chunks + c, li); });
    f6_store(f_out + c * GPBC_GT_BYTES + (x.odd ? 192 : 0), h);
}

// Fixed-Q multi-pairing (gpbc_multi_pair_fixed_q): k segments pair their own m points P[j*m + i] with ONE shared list
// Q[0..m) — a BSW07 key against k ciphertexts, a public key against k signatures.  The raw line coefficients of every Q_i
// are computed once (k_q_lines: 88 x 54 int32 per Q_i, laid out [line][word][i]), the P's are converted to internal form
// once (k_g1_internal), and the accumulator kernel evaluates a line at its own P (two Fp x Fp2 products) right before the
// sparse multiplication.  Lane pairs are numbered chunk-major (t = c * k + j): the 32 lane pairs of a wave then work on
// the same Q_i at the same time, so their line loads are one broadcast transaction.
GPBC_KERNEL k_q_lines(const uint8_t *__restrict__ Q, int32_t *__restrict__ qlines, size_t m) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= m) return;
    const uint8_t *q = Q + i * GPBC_G2_BYTES;
    if (g2_bytes_inf(q)) return;
    G2A b{f2_load(q), f2_load(q + 64)};
    int step = 0;
    miller_lines_raw(b, [&](const LineE &l) { line_store(qlines, m, i, step++, LineS{l.r0, l.r1, l.r2}); });
}
GPBC_KERNEL_G1 k_g1_internal(const uint8_t *__restrict__ P, int32_t *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint8_t *pb = P + i * GPBC_G1_BYTES;
    AffP<Fe> a{fe_load(pb), fe_load(pb + 32), g1_bytes_inf(pb)};
    int32_t *o = out + i * 20;                                    // x (9), y (9), infinity flag, pad
#pragma unroll
    for (int w = 0; w < NL; w++) { o[w] = a.x.v[w]; o[NL + w] = a.y.v[w]; }
    o[18] = a.inf ? 1 : 0;
    o[19] = 0;
}
GPBC_KERNEL k_miller_accumulate_fixed_q(const int32_t *__restrict__ Pint, const uint8_t *__restrict__ Q, const int32_t *__restrict__ qlines,
                                        uint8_t *__restrict__ f_out, size_t m, size_t k, size_t L, size_t n_c) {
    size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t t = lane >> 1;
    if (t >= n_c * k) return;
    PairDpp x{(bool)(lane & 1)};
    const size_t c = t / k, j = t % k;
    const size_t lo = c * L, hi = (c + 1) * L < m ? (c + 1) * L : m;
    int vi[MULTI_CHUNK], n = 0;
    for (size_t i = lo; i < hi; i++)
        if (!Pint[(j * m + i) * 20 + 18] && !g2_bytes_inf(Q + i * GPBC_G2_BYTES)) vi[n++] = (int)i;
    F6 h;
    if (n == 0) h = f12p_one(x);
    else h = miller_accumulate_multi(x, n, [&](int p, int li) -> LineS {
        const size_t i = (size_t)vi[p];
        LineS r = line_load(qlines, m, i, li);                    // raw (r0, r1, r2) of Q_i: the same address for the whole wave
        const int32_t *pp = Pint + (j * m + i) * 20;
        Fe px, py;
#pragma unroll
        for (int w = 0; w < NL; w++) { px.v[w] = pp[w]; py.v[w] = pp[NL + w]; }
        return LineS{f2_mul_fe(r.c0, py), f2_mul_fe(r.c3, px), r.c4};
    });
    f6_store(f_out + (j * n_c + c) * GPBC_GT_BYTES + (x.odd ? 192 : 0), h);     // segment-major: chunks of a segment are adjacent
}

GPBC_KERNEL k_final_exp(const uint8_t *f_in, uint8_t *gt_out, size_t n) {
    size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t i = lane >> 1;
    if (i >= n) return;
    PairDpp x{(bool)(lane & 1)};
    size_t off = i * GPBC_GT_BYTES + (x.odd ? 192 : 0);
    F6 h = f6_load(f_in + off);
    f6_store(gt_out + off, final_exp_pair(x, h));
}

// product of the Miller functions of each segment: thread j multiplies f[seg_off[j] .. seg_off[j+1])
// (the segment table lives in device memory and cannot be validated by the host without a copy: offsets are clamped to
// the number of pairs so that a malformed table can never read outside the Miller-value buffer)
GPBC_KERNEL k_segment_product(const uint8_t *__restrict__ f, const uint64_t *__restrict__ seg_off, uint8_t *__restrict__ out, size_t k, size_t n_pairs) {
    size_t j = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= k) return;
    F12 acc = f12_one(), t;
    uint64_t lo = seg_off[j], hi = seg_off[j + 1];
    if (hi > n_pairs) hi = n_pairs;
    if (lo > hi) lo = hi;
    for (uint64_t i = lo; i < hi; i++) {
        f12_load(t, f + i * GPBC_GT_BYTES);
        acc = f12_mul(acc, t);
    }
    f12_store(out + j * GPBC_GT_BYTES, acc);
}

// GT one in gnark bytes: C0.B0.A0 = R mod p, everything else zero
__global__ void __launch_bounds__(BLOCK) k_gt_is_one(const uint8_t *__restrict__ gt, uint8_t *__restrict__ ok, size_t n) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    constexpr uint64_t ONE[4] = BN254_FP_ONE;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(gt + i * GPBC_GT_BYTES);
    uint32_t diff = 0;
    for (int j = 0; j < 8; j++) diff |= w[j] ^ (uint32_t)(ONE[j >> 1] >> ((j & 1) * 32));
    for (int j = 8; j < 96; j++) diff |= w[j];
    ok[i] = diff == 0 ? 1 : 0;
}

__device__ __forceinline__ void load_scalar(uint32_t k[8], const uint8_t *p) {
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = q[i];
}
__device__ __forceinline__ AffP<Fe> g1_load_aff(const uint8_t *p) { return AffP<Fe>{fe_load(p), fe_load(p + 32), g1_bytes_inf(p)}; }
__device__ __forceinline__ AffP<F2> g2_load_aff(const uint8_t *p) { return AffP<F2>{f2_load(p), f2_load(p + 64), g2_bytes_inf(p)}; }
__device__ __forceinline__ void g1_store_aff(uint8_t *p, const AffP<Fe> &r) { fe_store(p, r.x); fe_store(p + 32, r.y); }
__device__ __forceinline__ void g2_store_aff(uint8_t *p, const AffP<F2> &r) { f2_store(p, r.x); f2_store(p + 64, r.y); }

// Scalar multiplication: a lane owns SMUL_K points (t, t+T, t+2T, ...; T = ceil(n / SMUL_K)) whose Jacobian results share
// one field inversion.  Measured on MI355X (2^20 points): K = 1 -> 36.9 M G1 / 15.0 M G2 per second, K = 2 -> 36.9 / 13.7,
// K = 4 -> 35.9 / 13.3: holding K results costs more in registers and scratch than the shared inversion saves, so K = 1.
#ifndef GPBC_SMUL_K
#define GPBC_SMUL_K 1
#endif
constexpr int SMUL_K = GPBC_SMUL_K;
GPBC_KERNEL_G1 k_g1_scalar_mul(const uint8_t *__restrict__ bases, int shared_base, const uint8_t *__restrict__ scalars, uint8_t *__restrict__ out, size_t n, int32_t *__restrict__ tabws) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t T = (n + SMUL_K - 1) / SMUL_K;
    if (t >= T) return;
    JacP<Fe> res[SMUL_K];
    for (int j = 0; j < SMUL_K; j++) {
        size_t i = t + (size_t)j * T;
        if (i >= n) { jac_set_inf(res[j]); continue; }
        AffP<Fe> b = g1_load_aff(bases + (shared_base ? 0 : i * GPBC_G1_BYTES));
        uint32_t k[8];
        load_scalar(k, scalars + i * GPBC_SCALAR_BYTES);
        scalar_mul29_jac<Fe>(res[j], b, k, tabws + t * (size_t)glv_table_dwords<Fe>());
    }
    AffP<Fe> aff[SMUL_K];
    jac_to_affine_batch<Fe, SMUL_K>(aff, res);
    for (int j = 0; j < SMUL_K; j++) {
        size_t i = t + (size_t)j * T;
        if (i < n) g1_store_aff(out + i * GPBC_G1_BYTES, aff[j]);
    }
}
GPBC_KERNEL k_g2_scalar_mul(const uint8_t *__restrict__ bases, int shared_base, const uint8_t *__restrict__ scalars, uint8_t *__restrict__ out, size_t n, int32_t *__restrict__ tabws) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t T = (n + SMUL_K - 1) / SMUL_K;
    if (t >= T) return;
    JacP<F2> res[SMUL_K];
    for (int j = 0; j < SMUL_K; j++) {
        size_t i = t + (size_t)j * T;
        if (i >= n) { jac_set_inf(res[j]); continue; }
        AffP<F2> b = g2_load_aff(bases + (shared_base ? 0 : i * GPBC_G2_BYTES));
        uint32_t k[8];
        load_scalar(k, scalars + i * GPBC_SCALAR_BYTES);
        scalar_mul29_jac<F2>(res[j], b, k, tabws + t * (size_t)glv_table_dwords<F2>());
    }
    AffP<F2> aff[SMUL_K];
    jac_to_affine_batch<F2, SMUL_K>(aff, res);
    for (int j = 0; j < SMUL_K; j++) {
        size_t i = t + (size_t)j * T;
        if (i < n) g2_store_aff(out + i * GPBC_G2_BYTES, aff[j]);
    }
}

// one level of the point-sum tree: thread t adds in[t], in[t+n_out], in[t+2 n_out], ... -> out[t] (affine)
GPBC_KERNEL_G1 k_g1_sum_level(const uint8_t *__restrict__ in, size_t n_in, uint8_t *__restrict__ out, size_t n_out) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n_out) return;
    JacP<Fe> acc;
    jac_set_inf(acc);
    for (size_t i = t; i < n_in; i += n_out) jac_add_mixed(acc, acc, g1_load_aff(in + i * GPBC_G1_BYTES));
    AffP<Fe> r;
    jac_to_affine(r, acc);
    g1_store_aff(out + t * GPBC_G1_BYTES, r);
}
GPBC_KERNEL k_g2_sum_level(const uint8_t *__restrict__ in, size_t n_in, uint8_t *__restrict__ out, size_t n_out) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n_out) return;
    JacP<F2> acc;
    jac_set_inf(acc);
    for (size_t i = t; i < n_in; i += n_out) jac_add_mixed(acc, acc, g2_load_aff(in + i * GPBC_G2_BYTES));
    AffP<F2> r;
    jac_to_affine(r, acc);
    g2_store_aff(out + t * GPBC_G2_BYTES, r);
}

// GT.Exp: left-to-right square-and-multiply on a 256-bit plain exponent (k = 0 -> one)
GPBC_KERNEL k_gt_exp(const uint8_t *__restrict__ x, const uint8_t *__restrict__ kk, uint8_t *__restrict__ out, size_t n) {
    size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t i = lane >> 1;                                   // one Fp12 per lane pair
    if (i >= n) return;
    PairDpp px{(bool)(lane & 1)};
    size_t off = i * GPBC_GT_BYTES + (px.odd ? 192 : 0);
    uint32_t k[8];
    load_scalar(k, kk + i * GPBC_SCALAR_BYTES);
    f6_store(out + off, f12p_exp256(px, f6_load(x + off), k));
}

// op 0: a*b   1: a*b^-1   2: a^-1
GPBC_KERNEL k_gt_binary(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint8_t *__restrict__ out, size_t n, int op) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    F12 x, y, z;
    f12_load(x, a + i * GPBC_GT_BYTES);
    if (op == 2) z = f12_inv(x);
    else {
        f12_load(y, b + i * GPBC_GT_BYTES);
        if (op == 1) y = f12_inv(y);
        z = f12_mul(x, y);
    }
    f12_store(out + i * GPBC_GT_BYTES, z);
}

__global__ void __launch_bounds__(BLOCK) k_fp_mul(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint8_t *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    fe_store(out + i * 32, fe_mul(fe_load(a + i * 32), fe_load(b + i * 32)));
}

// ---- wire formats (csrc/wire29.cuh): one element per lane
GPBC_KERNEL k_g1_encode(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, size_t n, int compressed) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    g1_wire_encode(out + i * (compressed ? GPBC_G1_COMPRESSED_BYTES : GPBC_G1_RAW_BYTES), in + i * GPBC_G1_BYTES, compressed != 0);
}
GPBC_KERNEL k_g2_encode(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, size_t n, int compressed) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    g2_wire_encode(out + i * (compressed ? GPBC_G2_COMPRESSED_BYTES : GPBC_G2_RAW_BYTES), in + i * GPBC_G2_BYTES, compressed != 0);
}
GPBC_KERNEL k_gt_encode(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    gt_wire_encode(out + i * GPBC_GT_BYTES, in + i * GPBC_GT_BYTES);
}
GPBC_KERNEL k_g1_decode(const uint8_t *__restrict__ in, int elem_bytes, uint8_t *__restrict__ out, uint8_t *__restrict__ ok, size_t n) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    ok[i] = g1_wire_decode(out + i * GPBC_G1_BYTES, in + i * (size_t)elem_bytes, elem_bytes) ? 1 : 0;
}
GPBC_KERNEL k_g2_decode(const uint8_t *__restrict__ in, int elem_bytes, uint8_t *__restrict__ out, uint8_t *__restrict__ ok, size_t n) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    ok[i] = g2_wire_decode(out + i * GPBC_G2_BYTES, in + i * (size_t)elem_bytes, elem_bytes) ? 1 : 0;
}
GPBC_KERNEL k_gt_decode(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint8_t *__restrict__ ok, size_t n) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    ok[i] = gt_wire_decode(out + i * GPBC_GT_BYTES, in + i * GPBC_GT_BYTES) ? 1 : 0;
}

// ---- hash to curve, group part (csrc/h2c29.cuh): u = n x 2 field elements -> n points
GPBC_KERNEL k_g1_map_fields(const uint8_t *__restrict__ u, uint8_t *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    AffP<Fe> r;
    g1_map_fields(r, fe_load(u + i * 64), fe_load(u + i * 64 + 32));
    g1_store_aff(out + i * GPBC_G1_BYTES, r);
}
GPBC_KERNEL k_g2_map_fields(const uint8_t *__restrict__ u, uint8_t *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    AffP<F2> r;
    g2_map_fields(r, f2_load(u + i * 128), f2_load(u + i * 128 + 64));
    g2_store_aff(out + i * GPBC_G2_BYTES, r);
}

// ---- fixed-base tables (8-bit windows): entry ((b * 32 + w) * 255 + d - 1) = [d * 2^(8w)] base_b, affine, internal limb
// form in the 128-byte-aligned row layout of curve29.cuh (tab_store / tab_load).  1 MB per G1 base, 2 MB per G2 base.
constexpr int FB_WINDOWS = 32, FB_DIGITS = 255, FB_ENTRIES = FB_WINDOWS * FB_DIGITS;
template <class F> __device__ __forceinline__ void fb_build_lane(const uint8_t *bases, size_t nbase, int32_t *table, uint8_t *base_inf, int32_t *tabws, size_t first, size_t count) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= count) return;
    const size_t e = first + t, b = e / FB_ENTRIES;
    const int rem = (int)(e % FB_ENTRIES), w = rem / FB_DIGITS, d = rem % FB_DIGITS + 1;
    constexpr size_t PT = sizeof(F) == sizeof(Fe) ? GPBC_G1_BYTES : GPBC_G2_BYTES;
    const uint8_t *bp = bases + b * PT;
    AffP<F> base;
    if constexpr (sizeof(F) == sizeof(Fe)) base = g1_load_aff(bp); else base = g2_load_aff(bp);
    if (rem == 0) base_inf[b] = base.inf ? 1 : 0;
    uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    k[w >> 2] = (uint32_t)d << (8 * (w & 3));
    JacP<F> r;
    scalar_mul29_jac<F>(r, base, k, tabws + t * (size_t)glv_table_dwords<F>());
    AffP<F> a;
    jac_to_affine(a, r);
    tab_store(table + e * (size_t)TabLayout<F>::ENTRY_DWORDS, 0, a);
}
GPBC_KERNEL_G1 k_g1_fb_build(const uint8_t *__restrict__ bases, size_t nbase, int32_t *__restrict__ table, uint8_t *__restrict__ base_inf, int32_t *__restrict__ tabws, size_t first, size_t count) {
    fb_build_lane<Fe>(bases, nbase, table, base_inf, tabws, first, count);
}
GPBC_KERNEL k_g2_fb_build(const uint8_t *__restrict__ bases, size_t nbase, int32_t *__restrict__ table, uint8_t *__restrict__ base_inf, int32_t *__restrict__ tabws, size_t first, size_t count) {
    fb_build_lane<F2>(bases, nbase, table, base_inf, tabws, first, count);
}
// multi-scalar multiplication over the tables: lane (c, m) adds the terms of MSM m for bases [c*C, (c+1)*C): 32 mixed
// additions per term, no doublings.  Partials are written chunk-major (partial[c * n_msm + m]) so that ONE launch of the
// strided point-sum kernel adds the chunks of every MSM.
template <class F> __device__ __forceinline__ void fb_msm_lane(const int32_t *table, const uint8_t *base_inf, size_t nbase, const uint8_t *scalars,
                                                               size_t n_msm, size_t C, size_t n_chunks, uint8_t *partial) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n_msm * n_chunks) return;
    const size_t m = t % n_msm, c = t / n_msm;
    JacP<F> acc;
    jac_set_inf(acc);
    const size_t j1 = (c + 1) * C < nbase ? (c + 1) * C : nbase;
    for (size_t j = c * C; j < j1; j++) {
        if (base_inf[j]) continue;
        uint32_t k[8];
        load_scalar(k, scalars + (m * nbase + j) * GPBC_SCALAR_BYTES);
        const int32_t *tb = table + j * (size_t)FB_ENTRIES * TabLayout<F>::ENTRY_DWORDS;
        for (int w = 0; w < FB_WINDOWS; w++) {
            const int d = (int)((k[w >> 2] >> (8 * (w & 3))) & 255u);
            if (d) {
                AffP<F> e;
                tab_load(tb + (size_t)(w * FB_DIGITS + d - 1) * TabLayout<F>::ENTRY_DWORDS, 0, e);
                jac_add_mixed(acc, acc, e);
            }
        }
    }
    AffP<F> a;
    jac_to_affine(a, acc);
    if constexpr (sizeof(F) == sizeof(Fe)) g1_store_aff(partial + t * GPBC_G1_BYTES, a); else g2_store_aff(partial + t * GPBC_G2_BYTES, a);
}
GPBC_KERNEL_G1 k_g1_fb_msm(const int32_t *__restrict__ table, const uint8_t *__restrict__ base_inf, size_t nbase, const uint8_t *__restrict__ scalars,
                           size_t n_msm, size_t C, size_t n_chunks, uint8_t *__restrict__ partial) {
    fb_msm_lane<Fe>(table, base_inf, nbase, scalars, n_msm, C, n_chunks, partial);
}
GPBC_KERNEL k_g2_fb_msm(const int32_t *__restrict__ table, const uint8_t *__restrict__ base_inf, size_t nbase, const uint8_t *__restrict__ scalars,
                        size_t n_msm, size_t C, size_t n_chunks, uint8_t *__restrict__ partial) {
    fb_msm_lane<F2>(table, base_inf, nbase, scalars, n_msm, C, n_chunks, partial);
}

// =============================================================================================== host side
static thread_local char g_err[512] = "";
static std::atomic<int> g_device{-1};
static void free_workspaces();

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(GPBC_ERR_HIP, "%s failed: %s", #x, hipGetErrorString(e_)); } while (0)

static int bind_device() {
    int d = g_device.load();
    if (d < 0) return fail(GPBC_ERR_NO_DEVICE, "gpbc_init() has not bound a HIP device (no CPU fallback exists)");
    HIP_TRY(hipSetDevice(d));
    return GPBC_OK;
}
static inline unsigned grid_for(size_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }
static int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GPBC_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return GPBC_OK;
}
#define TRY(x) do { int rc_ = (x); if (rc_ != GPBC_OK) return rc_; } while (0)

// RAII device buffer for the host-pointer entry points
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
        if (e != hipSuccess) { p = nullptr; return fail(GPBC_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
        return GPBC_OK;
    }
    int upload(const void *src, size_t bytes) {
        TRY(alloc(bytes));
        if (bytes) HIP_TRY(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        return GPBC_OK;
    }
    int download(void *dst, size_t bytes) const {
        if (bytes) HIP_TRY(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost));
        return GPBC_OK;
    }
    uint8_t *u8() const { return static_cast<uint8_t *>(p); }
};

extern "C" {

int gpbc_abi_version(void) { return 4; }
const char *gpbc_last_error(void) { return g_err; }

int gpbc_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(GPBC_ERR_NO_DEVICE, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return n;
}

int gpbc_init(int device) {
    int n = gpbc_device_count();
    if (n <= 0) return fail(GPBC_ERR_NO_DEVICE, "no HIP device visible (this engine has no CPU fallback)");
    if (device < 0 || device >= n) return fail(GPBC_ERR_INVALID_ARG, "device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(GPBC_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
    g_device.store(device);
    return GPBC_OK;
}

int gpbc_shutdown(void) {
    free_workspaces();
    g_device.store(-1);
    return GPBC_OK;
}

// ----------------------------------------------------------------------------------------------- device-pointer API
// Internal workspace: the Miller lines (88 x 54 int32 per pairing) and the per-point GLV tables of the scalar
// multiplications (2 KB / 4 KB per point).  One grow-only buffer per (device, stream) — calls on one stream are ordered and
// may share it, calls on different streams get different buffers — and batches are processed in chunks so the footprint
// stays bounded.
constexpr size_t MILLER_CHUNK = 262144;                               // pairings per chunk: 4.98 GB of lines
constexpr size_t SMUL_CHUNK = 262144;                                 // points per chunk: 0.5 GB (G1) / 1 GB (G2) of tables
constexpr size_t LINE_BYTES_PER_PAIR = (size_t)MILLER_LINES * LINE_WORDS * sizeof(int32_t);
struct StreamWs { int device; hipStream_t stream; void *ptr; size_t bytes; };
static std::mutex g_ws_mu;
// Held while a call enqueues the kernels that share the stream's workspace (lines then accumulate; table build and loop
// in one kernel): two host threads launching on the same stream must not interleave such sequences.  Enqueueing is
// asynchronous, so the lock is held for microseconds.
static std::mutex g_ws_seq_mu;
static std::vector<StreamWs> g_ws;
static int stream_workspace(hipStream_t stream, size_t bytes, int32_t **out) {
    int dev = g_device.load();
    std::lock_guard<std::mutex> lk(g_ws_mu);
    for (auto &w : g_ws)
        if (w.device == dev && w.stream == stream) {
            if (w.bytes < bytes) {
                HIP_TRY(hipStreamSynchronize(stream));
                HIP_TRY(hipFree(w.ptr));
                w.ptr = nullptr; w.bytes = 0;
                HIP_TRY(hipMalloc(&w.ptr, bytes));
                w.bytes = bytes;
            }
            *out = (int32_t *)w.ptr;
            return GPBC_OK;
        }
    void *ptr = nullptr;
    HIP_TRY(hipMalloc(&ptr, bytes));
    g_ws.push_back(StreamWs{dev, stream, ptr, bytes});
    *out = (int32_t *)ptr;
    return GPBC_OK;
}
static int lines_workspace(hipStream_t stream, size_t pairs, int32_t **out) { return stream_workspace(stream, pairs * LINE_BYTES_PER_PAIR, out); }
static void free_workspaces() {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    for (auto &w : g_ws) if (w.ptr) { (void)hipSetDevice(w.device); (void)hipFree(w.ptr); }
    g_ws.clear();
}

int gpbc_miller_loop_dev(const void *dP, const void *dQ, size_t n, void *d_f_out, void *stream) {
    if (!n) return GPBC_OK;
    if (!dP || !dQ || !d_f_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    hipStream_t st = (hipStream_t)stream;
    size_t chunk = n < MILLER_CHUNK ? n : MILLER_CHUNK;
    std::lock_guard<std::mutex> seq(g_ws_seq_mu);
    int32_t *lines = nullptr;
    TRY(lines_workspace(st, chunk, &lines));
    for (size_t off = 0; off < n; off += chunk) {
        size_t m = n - off < chunk ? n - off : chunk;
        const uint8_t *p = (const uint8_t *)dP + off * GPBC_G1_BYTES, *q = (const uint8_t *)dQ + off * GPBC_G2_BYTES;
        k_miller_lines<<<grid_for(m), BLOCK, 0, st>>>(p, q, lines, m, chunk);
        TRY(check_launch("k_miller_lines"));
        k_miller_accumulate<<<grid_for(2 * m), BLOCK, 0, st>>>(p, q, lines, (uint8_t *)d_f_out + off * GPBC_GT_BYTES, m, chunk);
        TRY(check_launch("k_miller_accumulate"));
    }
    return GPBC_OK;
}
int gpbc_final_exp_dev(const void *d_f, size_t n, void *d_gt_out, void *stream) {
    if (!n) return GPBC_OK;
    if (!d_f || !d_gt_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    k_final_exp<<<grid_for(2 * n), BLOCK, 0, (hipStream_t)stream>>>((const uint8_t *)d_f, (uint8_t *)d_gt_out, n);
    return check_launch("k_final_exp");
}
int gpbc_pair_batch_dev(const void *dP, const void *dQ, size_t n, void *d_gt_out, void *stream) {
    if (!n) return fail(GPBC_ERR_INVALID_ARG, "invalid inputs sizes");
    TRY(gpbc_miller_loop_dev(dP, dQ, n, d_gt_out, stream));        // f staged in the output buffer
    return gpbc_final_exp_dev(d_gt_out, n, d_gt_out, stream);      // each lane rewrites its own 384 B
}
size_t gpbc_multi_pair_workspace_bytes(size_t n_pairs, size_t k) { (void)k; return n_pairs * GPBC_GT_BYTES; }
int gpbc_multi_pair_dev(const void *dP, const void *dQ, const uint64_t *d_seg_off, size_t n_pairs, size_t k,
                        void *d_gt_out, void *d_workspace, size_t workspace_bytes, void *stream) {
    if (!k) return fail(GPBC_ERR_INVALID_ARG, "invalid inputs sizes");
    if (!d_seg_off || !d_gt_out || (n_pairs && (!dP || !dQ || !d_workspace))) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    if (workspace_bytes < gpbc_multi_pair_workspace_bytes(n_pairs, k)) return fail(GPBC_ERR_WORKSPACE, "workspace too small");
    TRY(gpbc_miller_loop_dev(dP, dQ, n_pairs, d_workspace, stream));
    k_segment_product<<<grid_for(k), BLOCK, 0, (hipStream_t)stream>>>((const uint8_t *)d_workspace, d_seg_off, (uint8_t *)d_gt_out, k, n_pairs);
    TRY(check_launch("k_segment_product"));
    return gpbc_final_exp_dev(d_gt_out, k, d_gt_out, stream);
}
constexpr size_t FB_AUTO_MIN = 16384;
static int shared_base_mul_dev(bool g2, const void *d_base, const void *d_scalars, size_t n, void *d_out, hipStream_t st) {
    const size_t row_dwords = g2 ? (size_t)TabLayout<F2>::ENTRY_DWORDS : (size_t)TabLayout<Fe>::ENTRY_DWORDS;
    const size_t table_bytes = (size_t)FB_ENTRIES * row_dwords * sizeof(int32_t);
    void *mem = nullptr;
    if (hipMallocAsync(&mem, table_bytes + 256, st) != hipSuccess) { (void)hipGetLastError(); return GPBC_ERR_WORKSPACE; }
    int32_t *table = (int32_t *)mem;
    uint8_t *base_inf = (uint8_t *)mem + table_bytes;
    const size_t tab_bytes = sizeof(int32_t) * (g2 ? (size_t)glv_table_dwords<F2>() : (size_t)glv_table_dwords<Fe>());
    int rc;
    {
        std::lock_guard<std::mutex> seq(g_ws_seq_mu);
        int32_t *tabws = nullptr;
        rc = stream_workspace(st, (size_t)FB_ENTRIES * tab_bytes, &tabws);
        if (rc == GPBC_OK) {
            if (g2) k_g2_fb_build<<<grid_for(FB_ENTRIES), BLOCK, 0, st>>>((const uint8_t *)d_base, 1, table, base_inf, tabws, 0, FB_ENTRIES);
            else k_g1_fb_build<<<grid_for(FB_ENTRIES), BLOCK, 0, st>>>((const uint8_t *)d_base, 1, table, base_inf, tabws, 0, FB_ENTRIES);
            rc = check_launch("k_fb_build");
        }
    }
    if (rc == GPBC_OK) {
        if (g2) k_g2_fb_msm<<<grid_for(n), BLOCK, 0, st>>>(table, base_inf, 1, (const uint8_t *)d_scalars, n, 1, 1, (uint8_t *)d_out);
        else k_g1_fb_msm<<<grid_for(n), BLOCK, 0, st>>>(table, base_inf, 1, (const uint8_t *)d_scalars, n, 1, 1, (uint8_t *)d_out);
        rc = check_launch("k_fb_msm");
    }
    (void)hipFreeAsync(mem, st);                                     // stream-ordered: released after the kernels above
    return rc;
}
static int scalar_mul_dev(bool g2, const void *d_bases, size_t nbase, const void *d_scalars, size_t n, void *d_out, void *stream) {
    if (!n) return GPBC_OK;
    if (!d_bases || !d_scalars || !d_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    if (nbase != 1 && nbase != n) return fail(GPBC_ERR_INVALID_ARG, "nbase must be 1 or n");
    TRY(bind_device());
    static_assert(SMUL_K == 1, "the table workspace is laid out for one point per lane");
    if (nbase == 1 && n >= FB_AUTO_MIN) {
        // One base for a large batch (ScalarMultiplicationBase-style calls): a transient fixed-base window table (8 160 rows,
        // ~0.1 ms to build) turns every multiplication into 32 mixed additions.  Same canonical affine results.
        int rc = shared_base_mul_dev(g2, d_bases, d_scalars, n, d_out, (hipStream_t)stream);
        if (rc != GPBC_ERR_WORKSPACE) return rc;                      // only "could not allocate the table" falls through
    }
    const int shared = (nbase == 1 && n != 1) ? 1 : 0;
    const size_t pt = g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES;
    const size_t tab_bytes = sizeof(int32_t) * (g2 ? (size_t)glv_table_dwords<F2>() : (size_t)glv_table_dwords<Fe>());
    hipStream_t st = (hipStream_t)stream;
    const size_t chunk = n < SMUL_CHUNK ? n : SMUL_CHUNK;
    std::lock_guard<std::mutex> seq(g_ws_seq_mu);
    int32_t *tabws = nullptr;
    TRY(stream_workspace(st, chunk * tab_bytes, &tabws));
    for (size_t off = 0; off < n; off += chunk) {
        const size_t m = n - off < chunk ? n - off : chunk;
        const uint8_t *b = (const uint8_t *)d_bases + (shared ? 0 : off * pt), *k = (const uint8_t *)d_scalars + off * GPBC_SCALAR_BYTES;
        uint8_t *o = (uint8_t *)d_out + off * pt;
        if (g2) k_g2_scalar_mul<<<grid_for(m), BLOCK, 0, st>>>(b, shared, k, o, m, tabws);
        else k_g1_scalar_mul<<<grid_for(m), BLOCK, 0, st>>>(b, shared, k, o, m, tabws);
        TRY(check_launch(g2 ? "k_g2_scalar_mul" : "k_g1_scalar_mul"));
    }
    return GPBC_OK;
}
int gpbc_g1_scalar_mul_batch_dev(const void *b, size_t nb, const void *s, size_t n, void *o, void *st) { return scalar_mul_dev(false, b, nb, s, n, o, st); }
int gpbc_g2_scalar_mul_batch_dev(const void *b, size_t nb, const void *s, size_t n, void *o, void *st) { return scalar_mul_dev(true, b, nb, s, n, o, st); }

constexpr size_t SUM_FANIN = 32;
size_t gpbc_sum_workspace_bytes(size_t n, int is_g2) {
    size_t pt = is_g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES, total = 0;
    while (n > 1) { n = (n + SUM_FANIN - 1) / SUM_FANIN; total += n * pt; }
    return total + pt;
}
static int sum_dev(bool g2, const void *d_pts, size_t n, void *d_out, void *d_ws, size_t ws_bytes, void *stream) {
    if (!d_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    size_t pt = g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES;
    if (!n) { HIP_TRY(hipMemsetAsync(d_out, 0, pt, (hipStream_t)stream)); return GPBC_OK; }
    if (!d_pts || !d_ws) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    if (ws_bytes < gpbc_sum_workspace_bytes(n, g2)) return fail(GPBC_ERR_WORKSPACE, "workspace too small");
    const uint8_t *in = (const uint8_t *)d_pts;
    uint8_t *ws = (uint8_t *)d_ws;
    size_t n_in = n;
    for (;;) {
        size_t n_out = (n_in + SUM_FANIN - 1) / SUM_FANIN;
        uint8_t *out = n_out == 1 ? (uint8_t *)d_out : ws;
        if (g2) k_g2_sum_level<<<grid_for(n_out), BLOCK, 0, (hipStream_t)stream>>>(in, n_in, out, n_out);
        else k_g1_sum_level<<<grid_for(n_out), BLOCK, 0, (hipStream_t)stream>>>(in, n_in, out, n_out);
        TRY(check_launch("k_sum_level"));
        if (n_out == 1) break;
        in = out; ws += n_out * pt; n_in = n_out;
    }
    return GPBC_OK;
}
int gpbc_g1_sum_dev(const void *p, size_t n, void *o, void *w, size_t wb, void *s) { return sum_dev(false, p, n, o, w, wb, s); }
int gpbc_g2_sum_dev(const void *p, size_t n, void *o, void *w, size_t wb, void *s) { return sum_dev(true, p, n, o, w, wb, s); }

int gpbc_gt_exp_batch_dev(const void *d_x, const void *d_k, size_t n, void *d_out, void *stream) {
    if (!n) return GPBC_OK;
    if (!d_x || !d_k || !d_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    k_gt_exp<<<grid_for(2 * n), BLOCK, 0, (hipStream_t)stream>>>((const uint8_t *)d_x, (const uint8_t *)d_k, (uint8_t *)d_out, n);
    return check_launch("k_gt_exp");
}
static int gt_binary_dev(int op, const void *a, const void *b, size_t n, void *out, void *stream) {
    if (!n) return GPBC_OK;
    if (!a || (op != 2 && !b) || !out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    k_gt_binary<<<grid_for(n), BLOCK, 0, (hipStream_t)stream>>>((const uint8_t *)a, (const uint8_t *)b, (uint8_t *)out, n, op);
    return check_launch("k_gt_binary");
}
int gpbc_gt_mul_batch_dev(const void *a, const void *b, size_t n, void *o, void *s) { return gt_binary_dev(0, a, b, n, o, s); }
int gpbc_gt_div_batch_dev(const void *a, const void *b, size_t n, void *o, void *s) { return gt_binary_dev(1, a, b, n, o, s); }
int gpbc_gt_inverse_batch_dev(const void *a, size_t n, void *o, void *s) { return gt_binary_dev(2, a, nullptr, n, o, s); }

// ----------------------------------------------------------------------------------------------- host-pointer API
static int sync_default() { HIP_TRY(hipStreamSynchronize(nullptr)); return GPBC_OK; }

int gpbc_miller_loop(const void *P, const void *Q, size_t n, void *f_out) {
    if (!n) return GPBC_OK;
    if (!P || !Q || !f_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    DevBuf dP, dQ, dF;
    TRY(dP.upload(P, n * GPBC_G1_BYTES)); TRY(dQ.upload(Q, n * GPBC_G2_BYTES)); TRY(dF.alloc(n * GPBC_GT_BYTES));
    TRY(gpbc_miller_loop_dev(dP.p, dQ.p, n, dF.p, nullptr));
    TRY(sync_default());
    return dF.download(f_out, n * GPBC_GT_BYTES);
}
int gpbc_final_exp(const void *f, size_t n, void *gt_out) {
    if (!n) return GPBC_OK;
    if (!f || !gt_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    DevBuf dF;
    TRY(dF.upload(f, n * GPBC_GT_BYTES));
    TRY(gpbc_final_exp_dev(dF.p, n, dF.p, nullptr));
    TRY(sync_default());
    return dF.download(gt_out, n * GPBC_GT_BYTES);
}
int gpbc_pair_batch(const void *P, const void *Q, size_t n, void *gt_out) {
    if (!n) return fail(GPBC_ERR_INVALID_ARG, "invalid inputs sizes");
    if (!P || !Q || !gt_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    DevBuf dP, dQ, dG;
    TRY(dP.upload(P, n * GPBC_G1_BYTES)); TRY(dQ.upload(Q, n * GPBC_G2_BYTES)); TRY(dG.alloc(n * GPBC_GT_BYTES));
    TRY(gpbc_pair_batch_dev(dP.p, dQ.p, n, dG.p, nullptr));
    TRY(sync_default());
    return dG.download(gt_out, n * GPBC_GT_BYTES);
}
static int check_segments(const uint64_t *seg_off, size_t k, size_t *n_pairs) {
    if (!seg_off) return fail(GPBC_ERR_INVALID_ARG, "null segment table");
    if (seg_off[0] != 0) return fail(GPBC_ERR_INVALID_ARG, "seg_off[0] must be 0");
    for (size_t j = 0; j < k; j++)
        if (seg_off[j + 1] < seg_off[j]) return fail(GPBC_ERR_INVALID_ARG, "segment table not monotone at %zu", j);
    *n_pairs = (size_t)seg_off[k];
    return GPBC_OK;
}
// Core of the multi-pairing with the segment table on the HOST and the points in device memory: every segment is cut into
// chunks of at most L <= MULTI_CHUNK pairs, one lane pair runs the Miller accumulator of a whole chunk with shared
// squarings (k_miller_accumulate_chunks), the chunk values of each segment are multiplied (one lane per segment) and one
// final exponentiation per segment follows.  Every pair beyond the first of a chunk saves its 64 Fp12 squarings (~40 % of
// its accumulator work); a single bn254.Pair call with hundreds of pairs (ibe/bb04_ibe/bb04_ibe.go:213-225: 257; a
// 256-attribute BSW07 decrypt: 513) still spreads over many lanes.  Synchronises `st` before it returns (its tables and
// chunk values are released on return).
static std::atomic<int> g_multi_chunk{0};
int gpbc_set_multi_pair_chunk(int pairs_per_chunk) {
    if (pairs_per_chunk < 0 || pairs_per_chunk > MULTI_CHUNK) return fail(GPBC_ERR_INVALID_ARG, "chunk length must be 0 (automatic) .. %d", MULTI_CHUNK);
    g_multi_chunk.store(pairs_per_chunk);
    return GPBC_OK;
}
static int multi_pair_core(const uint8_t *dP, const uint8_t *dQ, const uint64_t *seg_off, size_t k, size_t n_pairs, uint8_t *dG, uint8_t *dOk, hipStream_t st) {
    if (n_pairs < 4 * k && g_multi_chunk.load() <= 0) {
        // Short segments (BLS checks: 2 pairs; AFP25: 3): sharing squarings among two or three pairs saves ~1 ms per 200 000
        // pairs, less than the extra chunk-product pass costs; one Miller loop per lane pair and one product per segment.
        DevBuf dSeg, dW;
        TRY(dSeg.upload(seg_off, (k + 1) * sizeof(uint64_t)));
        const size_t wsb = gpbc_multi_pair_workspace_bytes(n_pairs, k);
        TRY(dW.alloc(wsb));
        TRY(gpbc_multi_pair_dev(dP, dQ, (const uint64_t *)dSeg.p, n_pairs, k, dG, dW.p, wsb, st));
        if (dOk) {
            k_gt_is_one<<<grid_for(k), BLOCK, 0, st>>>(dG, dOk, k);
            TRY(check_launch("k_gt_is_one"));
        }
        HIP_TRY(hipStreamSynchronize(st));
        return GPBC_OK;
    }
    // Chunk length L: as long as possible (more shared squarings) while ~131072 lane pairs stay in flight, and at most
    // MULTI_CHUNK.  Chunks are launched in groups of MULTI_GROUP = 65536 (131072 lanes = 2048 waves: exactly one full round
    // of two waves per SIMD on 256 CUs — a lane pair here runs for tens of milliseconds, so a partially filled second round
    // would cost as much as a full one); the slot grid of a group, L x 65536 lines rows, is at most 10 GB.
    uint64_t L = (n_pairs + 131071) / 131072;
    if (g_multi_chunk.load() > 0) L = (uint64_t)g_multi_chunk.load();
    if (L < 1) L = 1;
    if (L > (uint64_t)MULTI_CHUNK) L = MULTI_CHUNK;
    std::vector<uint64_t> chunk_off(1, 0), seg_chunk(1, 0);
    for (size_t j = 0; j < k; j++) {
        for (uint64_t a = seg_off[j]; a < seg_off[j + 1]; a += L)
            chunk_off.push_back(a + L < seg_off[j + 1] ? a + L : seg_off[j + 1]);
        seg_chunk.push_back(chunk_off.size() - 1);
    }
    const size_t n_chunks = chunk_off.size() - 1;
    DevBuf dChunkOff, dSegChunk, dPart;
    TRY(dChunkOff.upload(chunk_off.data(), chunk_off.size() * sizeof(uint64_t)));
    TRY(dSegChunk.upload(seg_chunk.data(), seg_chunk.size() * sizeof(uint64_t)));
    TRY(dPart.alloc(n_chunks * GPBC_GT_BYTES));
    {
        std::lock_guard<std::mutex> seq(g_ws_seq_mu);
        for (size_t cb = 0; cb < n_chunks; cb += MULTI_GROUP) {
            const size_t g = n_chunks - cb < MULTI_GROUP ? n_chunks - cb : MULTI_GROUP;
            size_t max_len = 0;
            for (size_t c = cb; c < cb + g; c++) { size_t len = (size_t)(chunk_off[c + 1] - chunk_off[c]); if (len > max_len) max_len = len; }
            const size_t n_slots = max_len * g;
            int32_t *lines = nullptr;
            TRY(lines_workspace(st, n_slots, &lines));
            const uint64_t *co = (const uint64_t *)dChunkOff.p + cb;
            k_miller_lines_chunks<<<grid_for(n_slots), BLOCK, 0, st>>>(dP, dQ, lines, co, g, n_slots);
            TRY(check_launch("k_miller_lines_chunks"));
            k_miller_accumulate_chunks<<<grid_for(2 * g), BLOCK, 0, st>>>(dP, dQ, lines, co, dPart.u8() + cb * GPBC_GT_BYTES, g, n_slots);
            TRY(check_launch("k_miller_accumulate_chunks"));
        }
    }
    k_segment_product<<<grid_for(k), BLOCK, 0, st>>>(dPart.u8(), (const uint64_t *)dSegChunk.p, dG, k, n_chunks);
    TRY(check_launch("k_segment_product (segments)"));
    TRY(gpbc_final_exp_dev(dG, k, dG, st));
    if (dOk) {
        k_gt_is_one<<<grid_for(k), BLOCK, 0, st>>>(dG, dOk, k);
        TRY(check_launch("k_gt_is_one"));
    }
    HIP_TRY(hipStreamSynchronize(st));
    return GPBC_OK;
}
// out[j] = Pair(P[j*m .. (j+1)*m), Q[0 .. m)), j < k.  Synchronises the stream before returning.
int gpbc_multi_pair_fixed_q_dev(const void *dP, const void *dQ, size_t m, size_t k, void *d_gt_out, void *stream) {
    if (!k || !m) return fail(GPBC_ERR_INVALID_ARG, "invalid inputs sizes");
    if (!dP || !dQ || !d_gt_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    hipStream_t st = (hipStream_t)stream;
    // chunk of the Q list per lane pair: long enough to share squarings, short enough that >= ~65536 lane pairs exist
    size_t L = (m * k + 65535) / 65536;
    if (g_multi_chunk.load() > 0) L = (size_t)g_multi_chunk.load();
    if (L < 1) L = 1;
    if (L > (size_t)MULTI_CHUNK) L = MULTI_CHUNK;
    if (L > m) L = m;
    const size_t n_c = (m + L - 1) / L;
    DevBuf dLines, dPint, dPart, dSegChunk;
    TRY(dLines.alloc(m * LINE_BYTES_PER_PAIR));
    TRY(dPint.alloc(m * k * 20 * sizeof(int32_t)));
    TRY(dPart.alloc(n_c * k * GPBC_GT_BYTES));
    std::vector<uint64_t> seg_chunk(k + 1);
    for (size_t j = 0; j <= k; j++) seg_chunk[j] = j * n_c;
    TRY(dSegChunk.upload(seg_chunk.data(), seg_chunk.size() * sizeof(uint64_t)));
    k_q_lines<<<grid_for(m), BLOCK, 0, st>>>((const uint8_t *)dQ, (int32_t *)dLines.p, m);
    TRY(check_launch("k_q_lines"));
    k_g1_internal<<<grid_for(m * k), BLOCK, 0, st>>>((const uint8_t *)dP, (int32_t *)dPint.p, m * k);
    TRY(check_launch("k_g1_internal"));
    k_miller_accumulate_fixed_q<<<grid_for(2 * n_c * k), BLOCK, 0, st>>>((const int32_t *)dPint.p, (const uint8_t *)dQ, (const int32_t *)dLines.p, dPart.u8(), m, k, L, n_c);
    TRY(check_launch("k_miller_accumulate_fixed_q"));
    k_segment_product<<<grid_for(k), BLOCK, 0, st>>>(dPart.u8(), (const uint64_t *)dSegChunk.p, (uint8_t *)d_gt_out, k, n_c * k);
    TRY(check_launch("k_segment_product"));
    TRY(gpbc_final_exp_dev(d_gt_out, k, d_gt_out, st));
    HIP_TRY(hipStreamSynchronize(st));
    return GPBC_OK;
}
int gpbc_multi_pair_fixed_q(const void *P, const void *Q, size_t m, size_t k, void *gt_out) {
    if (!k || !m) return fail(GPBC_ERR_INVALID_ARG, "invalid inputs sizes");
    if (!P || !Q || !gt_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    DevBuf dP, dQ, dG;
    TRY(dP.upload(P, m * k * GPBC_G1_BYTES)); TRY(dQ.upload(Q, m * GPBC_G2_BYTES)); TRY(dG.alloc(k * GPBC_GT_BYTES));
    TRY(gpbc_multi_pair_fixed_q_dev(dP.p, dQ.p, m, k, dG.p, nullptr));
    return dG.download(gt_out, k * GPBC_GT_BYTES);
}
int gpbc_multi_pair_hostseg_dev(const void *dP, const void *dQ, const uint64_t *seg_off, size_t k, void *d_gt_out, void *stream) {
    if (!k) return fail(GPBC_ERR_INVALID_ARG, "invalid inputs sizes");
    size_t n_pairs = 0;
    TRY(check_segments(seg_off, k, &n_pairs));
    if ((n_pairs && (!dP || !dQ)) || !d_gt_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    return multi_pair_core((const uint8_t *)dP, (const uint8_t *)dQ, seg_off, k, n_pairs, (uint8_t *)d_gt_out, nullptr, (hipStream_t)stream);
}
static int multi_pair_host(const void *P, const void *Q, const uint64_t *seg_off, size_t k, void *gt_out, uint8_t *ok_out) {
    if (!k) return fail(GPBC_ERR_INVALID_ARG, "invalid inputs sizes");
    size_t n_pairs = 0;
    TRY(check_segments(seg_off, k, &n_pairs));
    if ((n_pairs && (!P || !Q)) || (!gt_out && !ok_out)) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    DevBuf dP, dQ, dG, dOk;
    TRY(dP.upload(P, n_pairs * GPBC_G1_BYTES)); TRY(dQ.upload(Q, n_pairs * GPBC_G2_BYTES));
    TRY(dG.alloc(k * GPBC_GT_BYTES));
    if (ok_out) TRY(dOk.alloc(k));
    TRY(multi_pair_core(dP.u8(), dQ.u8(), seg_off, k, n_pairs, dG.u8(), ok_out ? dOk.u8() : nullptr, nullptr));
    if (gt_out) TRY(dG.download(gt_out, k * GPBC_GT_BYTES));
    if (ok_out) TRY(dOk.download(ok_out, k));
    return GPBC_OK;
}
int gpbc_multi_pair(const void *P, const void *Q, const uint64_t *seg_off, size_t k, void *gt_out) {
    if (!gt_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    return multi_pair_host(P, Q, seg_off, k, gt_out, nullptr);
}
int gpbc_pairing_check(const void *P, const void *Q, const uint64_t *seg_off, size_t k, uint8_t *ok_out) {
    if (!ok_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    return multi_pair_host(P, Q, seg_off, k, nullptr, ok_out);
}
static int scalar_mul_host(bool g2, const void *bases, size_t nbase, const void *scalars, size_t n, void *out) {
    if (!n) return GPBC_OK;
    if (!bases || !scalars || !out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    if (nbase != 1 && nbase != n) return fail(GPBC_ERR_INVALID_ARG, "nbase must be 1 or n");
    TRY(bind_device());
    size_t pt = g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES;
    DevBuf dB, dS, dO;
    TRY(dB.upload(bases, nbase * pt)); TRY(dS.upload(scalars, n * GPBC_SCALAR_BYTES)); TRY(dO.alloc(n * pt));
    TRY(scalar_mul_dev(g2, dB.p, nbase, dS.p, n, dO.p, nullptr));
    TRY(sync_default());
    return dO.download(out, n * pt);
}
int gpbc_g1_scalar_mul_batch(const void *b, size_t nb, const void *s, size_t n, void *o) { return scalar_mul_host(false, b, nb, s, n, o); }
int gpbc_g2_scalar_mul_batch(const void *b, size_t nb, const void *s, size_t n, void *o) { return scalar_mul_host(true, b, nb, s, n, o); }

static int sum_host(bool g2, const void *pts, size_t n, void *out) {
    if (!out || (n && !pts)) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    size_t pt = g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES;
    DevBuf dP, dO, dW;
    TRY(dP.upload(pts, n * pt)); TRY(dO.alloc(pt));
    size_t wsb = gpbc_sum_workspace_bytes(n, g2);
    TRY(dW.alloc(wsb));
    TRY(sum_dev(g2, dP.p, n, dO.p, dW.p, wsb, nullptr));
    TRY(sync_default());
    return dO.download(out, pt);
}
int gpbc_g1_sum(const void *p, size_t n, void *o) { return sum_host(false, p, n, o); }
int gpbc_g2_sum(const void *p, size_t n, void *o) { return sum_host(true, p, n, o); }

int gpbc_gt_exp_batch(const void *x, const void *k, size_t n, void *out) {
    if (!n) return GPBC_OK;
    if (!x || !k || !out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    DevBuf dX, dK, dO;
    TRY(dX.upload(x, n * GPBC_GT_BYTES)); TRY(dK.upload(k, n * GPBC_SCALAR_BYTES)); TRY(dO.alloc(n * GPBC_GT_BYTES));
    TRY(gpbc_gt_exp_batch_dev(dX.p, dK.p, n, dO.p, nullptr));
    TRY(sync_default());
    return dO.download(out, n * GPBC_GT_BYTES);
}
static int gt_binary_host(int op, const void *a, const void *b, size_t n, void *out) {
    if (!n) return GPBC_OK;
    if (!a || (op != 2 && !b) || !out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    DevBuf dA, dB, dO;
    TRY(dA.upload(a, n * GPBC_GT_BYTES));
    if (op != 2) TRY(dB.upload(b, n * GPBC_GT_BYTES));
    TRY(dO.alloc(n * GPBC_GT_BYTES));
    TRY(gt_binary_dev(op, dA.p, dB.p, n, dO.p, nullptr));
    TRY(sync_default());
    return dO.download(out, n * GPBC_GT_BYTES);
}
int gpbc_gt_mul_batch(const void *a, const void *b, size_t n, void *o) { return gt_binary_host(0, a, b, n, o); }
int gpbc_gt_div_batch(const void *a, const void *b, size_t n, void *o) { return gt_binary_host(1, a, b, n, o); }
int gpbc_gt_inverse_batch(const void *a, size_t n, void *o) { return gt_binary_host(2, a, nullptr, n, o); }

int gpbc_fp_mul_batch(const void *a, const void *b, size_t n, void *out) {
    if (!n) return GPBC_OK;
    if (!a || !b || !out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    DevBuf dA, dB, dO;
    TRY(dA.upload(a, n * 32)); TRY(dB.upload(b, n * 32)); TRY(dO.alloc(n * 32));
    k_fp_mul<<<grid_for(n), BLOCK>>>(dA.u8(), dB.u8(), dO.u8(), n);
    TRY(check_launch("k_fp_mul"));
    TRY(sync_default());
    return dO.download(out, n * 32);
}

// ----------------------------------------------------------------------------------------------- wire formats
// kind 0 = G1, 1 = G2, 2 = GT
static size_t wire_mem_bytes(int kind) { return kind == 0 ? GPBC_G1_BYTES : kind == 1 ? GPBC_G2_BYTES : GPBC_GT_BYTES; }
static size_t wire_enc_bytes(int kind, int compressed) {
    return kind == 0 ? (compressed ? GPBC_G1_COMPRESSED_BYTES : GPBC_G1_RAW_BYTES)
         : kind == 1 ? (compressed ? GPBC_G2_COMPRESSED_BYTES : GPBC_G2_RAW_BYTES) : GPBC_GT_BYTES;
}
static int marshal_dev(int kind, const void *d_in, size_t n, int compressed, void *d_out, void *stream) {
    if (!n) return GPBC_OK;
    if (!d_in || !d_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    if (d_in == d_out) return fail(GPBC_ERR_INVALID_ARG, "marshal cannot run in place");
    TRY(bind_device());
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) k_g1_encode<<<grid_for(n), BLOCK, 0, st>>>((const uint8_t *)d_in, (uint8_t *)d_out, n, compressed);
    else if (kind == 1) k_g2_encode<<<grid_for(n), BLOCK, 0, st>>>((const uint8_t *)d_in, (uint8_t *)d_out, n, compressed);
    else k_gt_encode<<<grid_for(n), BLOCK, 0, st>>>((const uint8_t *)d_in, (uint8_t *)d_out, n);
    return check_launch("wire encode");
}
static int unmarshal_dev(int kind, const void *d_in, size_t elem_bytes, size_t n, void *d_out, uint8_t *d_ok, void *stream) {
    if (kind == 0 && elem_bytes != GPBC_G1_COMPRESSED_BYTES && elem_bytes != GPBC_G1_RAW_BYTES)
        return fail(GPBC_ERR_INVALID_ARG, "G1 element size must be 32 or 64");
    if (kind == 1 && elem_bytes != GPBC_G2_COMPRESSED_BYTES && elem_bytes != GPBC_G2_RAW_BYTES)
        return fail(GPBC_ERR_INVALID_ARG, "G2 element size must be 64 or 128");
    if (!n) return GPBC_OK;
    if (!d_in || !d_out || !d_ok) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    if (d_in == d_out) return fail(GPBC_ERR_INVALID_ARG, "unmarshal cannot run in place");
    TRY(bind_device());
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) k_g1_decode<<<grid_for(n), BLOCK, 0, st>>>((const uint8_t *)d_in, (int)elem_bytes, (uint8_t *)d_out, d_ok, n);
    else if (kind == 1) k_g2_decode<<<grid_for(n), BLOCK, 0, st>>>((const uint8_t *)d_in, (int)elem_bytes, (uint8_t *)d_out, d_ok, n);
    else k_gt_decode<<<grid_for(n), BLOCK, 0, st>>>((const uint8_t *)d_in, (uint8_t *)d_out, d_ok, n);
    return check_launch("wire decode");
}
static int marshal_host(int kind, const void *in, size_t n, int compressed, void *out) {
    if (!n) return GPBC_OK;
    if (!in || !out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    DevBuf dI, dO;
    TRY(dI.upload(in, n * wire_mem_bytes(kind))); TRY(dO.alloc(n * wire_enc_bytes(kind, compressed)));
    TRY(marshal_dev(kind, dI.p, n, compressed, dO.p, nullptr));
    TRY(sync_default());
    return dO.download(out, n * wire_enc_bytes(kind, compressed));
}
static int unmarshal_host(int kind, const void *in, size_t elem_bytes, size_t n, void *out, uint8_t *ok) {
    if (n && (!in || !out || !ok)) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    DevBuf dI, dO, dK;
    if (n) {
        TRY(bind_device());
        TRY(dI.upload(in, n * elem_bytes)); TRY(dO.alloc(n * wire_mem_bytes(kind))); TRY(dK.alloc(n));
    }
    TRY(unmarshal_dev(kind, dI.p, elem_bytes, n, dO.p, dK.u8(), nullptr));
    if (!n) return GPBC_OK;
    TRY(sync_default());
    TRY(dO.download(out, n * wire_mem_bytes(kind)));
    return dK.download(ok, n);
}
int gpbc_g1_marshal_batch(const void *p, size_t n, int c, void *o) { return marshal_host(0, p, n, c != 0, o); }
int gpbc_g2_marshal_batch(const void *p, size_t n, int c, void *o) { return marshal_host(1, p, n, c != 0, o); }
int gpbc_gt_marshal_batch(const void *g, size_t n, void *o) { return marshal_host(2, g, n, 0, o); }
int gpbc_g1_marshal_batch_dev(const void *p, size_t n, int c, void *o, void *st) { return marshal_dev(0, p, n, c != 0, o, st); }
int gpbc_g2_marshal_batch_dev(const void *p, size_t n, int c, void *o, void *st) { return marshal_dev(1, p, n, c != 0, o, st); }
int gpbc_gt_marshal_batch_dev(const void *g, size_t n, void *o, void *st) { return marshal_dev(2, g, n, 0, o, st); }
int gpbc_g1_unmarshal_batch(const void *in, size_t eb, size_t n, void *o, uint8_t *ok) { return unmarshal_host(0, in, eb, n, o, ok); }
int gpbc_g2_unmarshal_batch(const void *in, size_t eb, size_t n, void *o, uint8_t *ok) { return unmarshal_host(1, in, eb, n, o, ok); }
int gpbc_gt_unmarshal_batch(const void *in, size_t n, void *o, uint8_t *ok) { return unmarshal_host(2, in, GPBC_GT_BYTES, n, o, ok); }
int gpbc_g1_unmarshal_batch_dev(const void *in, size_t eb, size_t n, void *o, uint8_t *ok, void *st) { return unmarshal_dev(0, in, eb, n, o, ok, st); }
int gpbc_g2_unmarshal_batch_dev(const void *in, size_t eb, size_t n, void *o, uint8_t *ok, void *st) { return unmarshal_dev(1, in, eb, n, o, ok, st); }
int gpbc_gt_unmarshal_batch_dev(const void *in, size_t n, void *o, uint8_t *ok, void *st) { return unmarshal_dev(2, in, GPBC_GT_BYTES, n, o, ok, st); }

// ----------------------------------------------------------------------------------------------- hash to curve (group part)
static int map_fields_dev(bool g2, const void *d_u, size_t n, void *d_out, void *stream) {
    if (!n) return GPBC_OK;
    if (!d_u || !d_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    if (g2) k_g2_map_fields<<<grid_for(n), BLOCK, 0, (hipStream_t)stream>>>((const uint8_t *)d_u, (uint8_t *)d_out, n);
    else k_g1_map_fields<<<grid_for(n), BLOCK, 0, (hipStream_t)stream>>>((const uint8_t *)d_u, (uint8_t *)d_out, n);
    return check_launch(g2 ? "k_g2_map_fields" : "k_g1_map_fields");
}
static int map_fields_host(bool g2, const void *u, size_t n, void *out) {
    if (!n) return GPBC_OK;
    if (!u || !out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    size_t pt = g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES;               // two field elements occupy as many bytes as one point
    DevBuf dU, dO;
    TRY(dU.upload(u, n * pt)); TRY(dO.alloc(n * pt));
    TRY(map_fields_dev(g2, dU.p, n, dO.p, nullptr));
    TRY(sync_default());
    return dO.download(out, n * pt);
}
int gpbc_g1_map_to_curve_batch(const void *u, size_t n, void *o) { return map_fields_host(false, u, n, o); }
int gpbc_g2_map_to_curve_batch(const void *u, size_t n, void *o) { return map_fields_host(true, u, n, o); }
int gpbc_g1_map_to_curve_batch_dev(const void *u, size_t n, void *o, void *st) { return map_fields_dev(false, u, n, o, st); }
int gpbc_g2_map_to_curve_batch_dev(const void *u, size_t n, void *o, void *st) { return map_fields_dev(true, u, n, o, st); }

// ----------------------------------------------------------------------------------------------- fixed-base tables / MSM
struct gpbc_fixed_base { int device; int is_g2; size_t nbase; int32_t *table; uint8_t *base_inf; };
static size_t fb_table_bytes(size_t nbase, int is_g2) {
    return nbase * (size_t)FB_ENTRIES * sizeof(int32_t) * (is_g2 ? (size_t)TabLayout<F2>::ENTRY_DWORDS : (size_t)TabLayout<Fe>::ENTRY_DWORDS);
}
size_t gpbc_fixed_base_table_bytes(size_t nbase, int is_g2) { return fb_table_bytes(nbase, is_g2); }
int gpbc_fixed_base_create_dev(int is_g2, const void *d_bases, size_t nbase, void *stream, gpbc_fixed_base **out) {
    if (!out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    *out = nullptr;
    if (!nbase || !d_bases) return fail(GPBC_ERR_INVALID_ARG, "fixed-base table needs at least one base");
    TRY(bind_device());
    hipStream_t st = (hipStream_t)stream;
    gpbc_fixed_base *h = new gpbc_fixed_base{g_device.load(), is_g2 ? 1 : 0, nbase, nullptr, nullptr};
    hipError_t e1 = hipMalloc((void **)&h->table, fb_table_bytes(nbase, is_g2));
    hipError_t e2 = e1 == hipSuccess ? hipMalloc((void **)&h->base_inf, nbase) : e1;
    if (e1 != hipSuccess || e2 != hipSuccess) {
        if (h->table) (void)hipFree(h->table);
        delete h;
        return fail(GPBC_ERR_HIP, "hipMalloc of a %zu-byte fixed-base table failed", fb_table_bytes(nbase, is_g2));
    }
    const size_t total = nbase * (size_t)FB_ENTRIES;
    const size_t tab_bytes = sizeof(int32_t) * (is_g2 ? (size_t)glv_table_dwords<F2>() : (size_t)glv_table_dwords<Fe>());
    const size_t chunk = total < SMUL_CHUNK ? total : SMUL_CHUNK;
    int rc = GPBC_OK;
    {
        std::lock_guard<std::mutex> seq(g_ws_seq_mu);
        int32_t *tabws = nullptr;
        rc = stream_workspace(st, chunk * tab_bytes, &tabws);
        for (size_t off = 0; rc == GPBC_OK && off < total; off += chunk) {
            const size_t m = total - off < chunk ? total - off : chunk;
            if (is_g2) k_g2_fb_build<<<grid_for(m), BLOCK, 0, st>>>((const uint8_t *)d_bases, nbase, h->table, h->base_inf, tabws, off, m);
            else k_g1_fb_build<<<grid_for(m), BLOCK, 0, st>>>((const uint8_t *)d_bases, nbase, h->table, h->base_inf, tabws, off, m);
            rc = check_launch("k_fb_build");
        }
    }
    if (rc != GPBC_OK) { (void)hipFree(h->table); (void)hipFree(h->base_inf); delete h; return rc; }
    *out = h;
    return GPBC_OK;
}
static int fb_create_host(int is_g2, const void *bases, size_t nbase, gpbc_fixed_base **out) {
    if (!out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    *out = nullptr;
    if (!nbase || !bases) return fail(GPBC_ERR_INVALID_ARG, "fixed-base table needs at least one base");
    TRY(bind_device());
    DevBuf dB;
    TRY(dB.upload(bases, nbase * (is_g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES)));
    TRY(gpbc_fixed_base_create_dev(is_g2, dB.p, nbase, nullptr, out));
    return sync_default();                                           // the bases buffer is freed on return
}
int gpbc_g1_fixed_base_create(const void *bases, size_t nbase, gpbc_fixed_base **out) { return fb_create_host(0, bases, nbase, out); }
int gpbc_g2_fixed_base_create(const void *bases, size_t nbase, gpbc_fixed_base **out) { return fb_create_host(1, bases, nbase, out); }
int gpbc_fixed_base_destroy(gpbc_fixed_base *h) {
    if (!h) return GPBC_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(h->table);
    (void)hipFree(h->base_inf);
    delete h;
    return GPBC_OK;
}
static void fb_shape(const gpbc_fixed_base *h, size_t n_msm, size_t *C, size_t *n_chunks) {
    // enough lanes to fill the chip (>= 131072 = 2048 waves) before a lane takes more than one term; at most 16 terms per lane
    size_t c = (h->nbase * n_msm) / 131072;
    if (c < 1) c = 1;
    if (c > 16) c = 16;
    if (c > h->nbase) c = h->nbase;
    *C = c;
    *n_chunks = (h->nbase + c - 1) / c;
}
// partial sums of every level of the fan-in-16 reduction over the chunks
size_t gpbc_fixed_base_msm_workspace_bytes(const gpbc_fixed_base *h, size_t n_msm) {
    if (!h || !n_msm) return 0;
    size_t C, n_chunks, total = 0;
    fb_shape(h, n_msm, &C, &n_chunks);
    for (size_t c = n_chunks; c > 1; c = (c + 15) / 16) total += c * n_msm;
    return total * (h->is_g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES);
}
int gpbc_fixed_base_msm_dev(const gpbc_fixed_base *h, const void *d_scalars, size_t n_msm, void *d_out, void *d_workspace, size_t workspace_bytes, void *stream) {
    if (!h) return fail(GPBC_ERR_INVALID_ARG, "null table handle");
    if (!n_msm) return GPBC_OK;
    if (!d_scalars || !d_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    if (g_device.load() != h->device) return fail(GPBC_ERR_INVALID_ARG, "table was built on device %d", h->device);
    size_t C, n_chunks;
    fb_shape(h, n_msm, &C, &n_chunks);
    if (n_chunks > 1 && (!d_workspace || workspace_bytes < gpbc_fixed_base_msm_workspace_bytes(h, n_msm))) return fail(GPBC_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const size_t pt = h->is_g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES;
    uint8_t *partial = n_chunks > 1 ? (uint8_t *)d_workspace : (uint8_t *)d_out;
    const size_t lanes = n_msm * n_chunks;
    if (h->is_g2) k_g2_fb_msm<<<grid_for(lanes), BLOCK, 0, st>>>(h->table, h->base_inf, h->nbase, (const uint8_t *)d_scalars, n_msm, C, n_chunks, partial);
    else k_g1_fb_msm<<<grid_for(lanes), BLOCK, 0, st>>>(h->table, h->base_inf, h->nbase, (const uint8_t *)d_scalars, n_msm, C, n_chunks, partial);
    TRY(check_launch("k_fb_msm"));
    // partials are chunk-major (partial[c * n_msm + m]); the strided sum kernel with n_out = c' * n_msm adds, for every m,
    // the chunks c' + i * c'' — so each launch divides the number of chunks by 16 until one row per sum is left
    const uint8_t *in = partial;
    uint8_t *ws = partial + lanes * pt;
    for (size_t c = n_chunks; c > 1;) {
        const size_t c2 = (c + 15) / 16;
        uint8_t *out = c2 == 1 ? (uint8_t *)d_out : ws;
        if (h->is_g2) k_g2_sum_level<<<grid_for(c2 * n_msm), BLOCK, 0, st>>>(in, c * n_msm, out, c2 * n_msm);
        else k_g1_sum_level<<<grid_for(c2 * n_msm), BLOCK, 0, st>>>(in, c * n_msm, out, c2 * n_msm);
        TRY(check_launch("k_sum_level"));
        in = out; ws += c2 * n_msm * pt; c = c2;
    }
    return GPBC_OK;
}
int gpbc_fixed_base_msm(const gpbc_fixed_base *h, const void *scalars, size_t n_msm, void *out) {
    if (!h) return fail(GPBC_ERR_INVALID_ARG, "null table handle");
    if (!n_msm) return GPBC_OK;
    if (!scalars || !out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    const size_t pt = h->is_g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES;
    DevBuf dS, dO, dW;
    TRY(dS.upload(scalars, n_msm * h->nbase * GPBC_SCALAR_BYTES)); TRY(dO.alloc(n_msm * pt));
    const size_t wsb = gpbc_fixed_base_msm_workspace_bytes(h, n_msm);
    TRY(dW.alloc(wsb));
    TRY(gpbc_fixed_base_msm_dev(h, dS.p, n_msm, dO.p, dW.p, wsb, nullptr));
    TRY(sync_default());
    return dO.download(out, n_msm * pt);
}

}  // extern "C"
