// libgpbc_bn254.so, unit 4 of 5: gnark wire formats (csrc/wire29.hip.hpp) and hash to curve — hash_to_field
// (csrc/xmd29.hip.hpp) and the group part (csrc/h2c29.hip.hpp) — with their C-ABI entries (include/gpbc_bn254.h).  gfx950 only.
#include "gpbc_common.hpp"
#include "curve29_oct.hip.hpp"
#include "wire29.hip.hpp"
#include "h2c29.hip.hpp"
#include "xmd29.hip.hpp"

// ---- wire formats (csrc/wire29.hip.hpp): one element per lane
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
// G2 decoding with one element per QUAD of lanes (calls of up to WIRE_QUAD_MAX elements): the 63-bit double-and-add of the subgroup
// test — two thirds of a decode — three products wide (csrc/curve29_quad.hip.hpp); the four lanes decode the same element and take
// the same branches.
constexpr size_t WIRE_QUAD_MAX = 16384;
__device__ __forceinline__ bool g2_in_subgroup29_quad(const F2 &x, const F2 &y, int q) {
    constexpr uint64_t X = 4965661367192848881ull;
    AffP<F2> pt{x, y, false};
    JacP<F2> xq;
    jac_set_inf(xq);
    for (int i = 62; i >= 0; i--) {
        jac_dbl_quad(xq, q);
        if ((X >> i) & 1) jac_add_mixed_quad(xq, pt, q);
    }
    JacP<F2> lhs, t, rhs;
    jac_add_mixed(lhs, xq, pt);                                  // [x+1]Q
    jac_add(lhs, lhs, jac_psi_tw(xq, 1));
    jac_add(lhs, lhs, jac_psi_tw(xq, 2));
    jac_dbl(t, xq);
    rhs = jac_psi_tw(t, 3);
    if (!rhs.inf) rhs.y = f2_neg(rhs.y);
    jac_add(t, lhs, rhs);                                        // lhs - rhs
    return t.inf;
}
GPBC_KERNEL k_g2_decode_quad(const uint8_t *__restrict__ in, int elem_bytes, uint8_t *__restrict__ out, uint8_t *__restrict__ ok, size_t n) {
    const size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x, i = lane >> 2;
    if (i >= n) return;
    const int q = (int)(lane & 3);
    const bool good = g2_wire_decode(out + i * GPBC_G2_BYTES, in + i * (size_t)elem_bytes, elem_bytes, [&](const F2 &x, const F2 &y) { return g2_in_subgroup29_quad(x, y, q); });
    if (q == 0) ok[i] = good ? 1 : 0;                            // (the four lanes wrote the same point bytes)
}
// ... and with one element per OCTET (calls of up to WIRE_OCT_MAX elements: one Unmarshal): the same chain with the two halves of every
// Fp2 product on two lanes (csrc/curve29_oct.hip.hpp)
constexpr size_t WIRE_OCT_MAX = 2048;
__device__ __forceinline__ bool g2_in_subgroup29_oct(const F2 &x, const F2 &y, int q) {
    constexpr uint64_t X = 4965661367192848881ull;
    AffP<F2> pt{x, y, false};
    JacP<F2> xq;
    jac_set_inf(xq);
    for (int i = 62; i >= 0; i--) {
        jac_dbl_oct(xq, q);
        if ((X >> i) & 1) jac_add_mixed_oct(xq, pt, q);
    }
    JacP<F2> lhs = xq, t, rhs;
    jac_add_mixed_oct(lhs, pt, q);                               // [x+1]Q
    jac_add(lhs, lhs, jac_psi_tw(xq, 1));
    jac_add(lhs, lhs, jac_psi_tw(xq, 2));
    t = xq;
    jac_dbl_oct(t, q);
    rhs = jac_psi_tw(t, 3);
    if (!rhs.inf) rhs.y = f2_neg(rhs.y);
    jac_add(t, lhs, rhs);                                        // lhs - rhs
    return t.inf;
}
GPBC_KERNEL k_g2_decode_oct(const uint8_t *__restrict__ in, int elem_bytes, uint8_t *__restrict__ out, uint8_t *__restrict__ ok, size_t n) {
    const size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x, i = lane >> 3;
    if (i >= n) return;
    const int q = (int)(lane & 7);
    const bool good = g2_wire_decode(out + i * GPBC_G2_BYTES, in + i * (size_t)elem_bytes, elem_bytes, [&](const F2 &x, const F2 &y) { return g2_in_subgroup29_oct(x, y, q); });
    if (q == 0) ok[i] = good ? 1 : 0;
}
GPBC_KERNEL k_gt_decode(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint8_t *__restrict__ ok, size_t n) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    ok[i] = gt_wire_decode(out + i * GPBC_GT_BYTES, in + i * GPBC_GT_BYTES) ? 1 : 0;
}

// ---- hash to curve, group part (csrc/h2c29.hip.hpp): u = n x 2 field elements -> n points
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

// ---- hash to curve, whole (csrc/xmd29.hip.hpp + h2c29.hip.hpp): one message per lane.  Message i is msgs[off[i], off[i+1]);
// the offsets are clamped to [0, total] and made monotone, so a malformed device-resident table cannot read outside the buffer.
__device__ __forceinline__ void msg_range(const uint64_t *__restrict__ off, size_t total, size_t i, uint64_t &lo, uint64_t &len) {
    uint64_t a = off[i], b = off[i + 1];
    if (a > total) a = total;
    if (b > total) b = total;
    lo = a; len = b > a ? b - a : 0;
}
template <int COUNT> GPBC_KERNEL k_hash_to_field(const uint8_t *__restrict__ msgs, const uint64_t *__restrict__ off, size_t total, size_t n, XmdDst dst,
                                                 uint8_t *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint64_t lo, len;
    msg_range(off, total, i, lo, len);
    uint32_t u[COUNT * 12];
    expand_message_xmd<COUNT * 12 / 8>(u, msgs + lo, len, dst);
#pragma unroll
    for (int e = 0; e < COUNT; e++) fe_store(out + (i * COUNT + e) * 32, xmd_field(u, e));
}
GPBC_KERNEL k_g1_hash(const uint8_t *__restrict__ msgs, const uint64_t *__restrict__ off, size_t total, size_t n, XmdDst dst, uint8_t *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint64_t lo, len;
    msg_range(off, total, i, lo, len);
    uint32_t u[24];
    expand_message_xmd<3>(u, msgs + lo, len, dst);
    AffP<Fe> r;
    g1_map_fields(r, xmd_field(u, 0), xmd_field(u, 1));
    g1_store_aff(out + i * GPBC_G1_BYTES, r);
}
GPBC_KERNEL k_g2_hash(const uint8_t *__restrict__ msgs, const uint64_t *__restrict__ off, size_t total, size_t n, XmdDst dst, uint8_t *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint64_t lo, len;
    msg_range(off, total, i, lo, len);
    uint32_t u[48];
    expand_message_xmd<6>(u, msgs + lo, len, dst);
    AffP<F2> r;
    g2_map_fields(r, F2{xmd_field(u, 0), xmd_field(u, 1)}, F2{xmd_field(u, 2), xmd_field(u, 3)});
    g2_store_aff(out + i * GPBC_G2_BYTES, r);
}

// ---- the same with one message (or one pair of field elements) per QUAD of lanes, for calls of up to H2C_QUAD_MAX: lanes 0 and 1 map
// u0 and u1 side by side (each map is a chain of ~300 (G1) / ~650 (G2) dependent Fp products, most of it the square root), and the
// 63 doublings and 30 additions of G2's cofactor clearing run three products wide (csrc/curve29_quad.hip.hpp).  One call of
// HashToG1 / HashToG2 — what bls01's Sign and Verify make — 0.9 / 2.8 -> 0.6 / 1.4 ms.
constexpr size_t H2C_QUAD_MAX = 16384;
template <class F> __device__ __forceinline__ void map_pair_quad(AffP<F> &q0, AffP<F> &q1, const F &u0, const F &u1, int q) {
    AffP<F> m;
    map_to_curve_svdw<F>(m, g_sel<F>((q & 1) != 0, u1, u0));
    q0 = AffP<F>{quad_bcast<0>(m.x), quad_bcast<0>(m.y), false};
    q1 = AffP<F>{quad_bcast<1>(m.x), quad_bcast<1>(m.y), false};
}
__device__ __forceinline__ void g1_map_fields_quad(AffP<Fe> &out, const Fe &u0, const Fe &u1, int q) {
    AffP<Fe> q0, q1;
    map_pair_quad<Fe>(q0, q1, u0, u1, q);
    JacP<Fe> j{q0.x, q0.y, fe_one(), false}, s;
    jac_add_mixed(s, j, q1);
    jac_to_affine(out, s);
}
// g2_clear_cofactor29 with the [x]P chain on the quad
__device__ __forceinline__ void g2_clear_cofactor29_quad(JacP<F2> &out, const AffP<F2> &p, int q) {
    constexpr uint64_t X = 4965661367192848881ull;
    JacP<F2> xq, t;
    jac_set_inf(xq);
    for (int i = 62; i >= 0; i--) {
        jac_dbl_quad(xq, q);
        if ((X >> i) & 1) jac_add_mixed_quad(xq, p, q);
    }
    jac_dbl(t, xq);
    jac_add(t, t, xq);                                          // [3x]P
    JacP<F2> pj{p.x, p.y, f2_one(), p.inf};
    JacP<F2> acc;
    jac_add(acc, xq, jac_psi(t, 1));
    jac_add(acc, acc, jac_psi(xq, 2));
    jac_add(out, acc, jac_psi(pj, 3));
}
__device__ __forceinline__ void g2_map_fields_quad(AffP<F2> &out, const F2 &u0, const F2 &u1, int q) {
    AffP<F2> q0, q1, a;
    map_pair_quad<F2>(q0, q1, u0, u1, q);
    JacP<F2> j{q0.x, q0.y, f2_one(), false}, s, c;
    jac_add_mixed(s, j, q1);
    jac_to_affine(a, s);
    g2_clear_cofactor29_quad(c, a, q);
    jac_to_affine(out, c);
}
// ... and G2 with one message per OCTET (calls of up to H2C_OCT_MAX messages: the HashToG2 of one Sign or Verify): lanes q & 1 map u0 / u1,
// the [x]P chain of the cofactor clearing runs on half products (csrc/curve29_oct.hip.hpp)
constexpr size_t H2C_OCT_MAX = 2048;
__device__ __forceinline__ void g2_clear_cofactor29_oct(JacP<F2> &out, const AffP<F2> &p, int q) {
    constexpr uint64_t X = 4965661367192848881ull;
    JacP<F2> xq, t;
    jac_set_inf(xq);
    for (int i = 62; i >= 0; i--) {
        jac_dbl_oct(xq, q);
        if ((X >> i) & 1) jac_add_mixed_oct(xq, p, q);
    }
    t = xq;
    jac_dbl_oct(t, q);
    jac_add(t, t, xq);                                          // [3x]P
    JacP<F2> pj{p.x, p.y, f2_one(), p.inf};
    JacP<F2> acc;
    jac_add(acc, xq, jac_psi(t, 1));
    jac_add(acc, acc, jac_psi(xq, 2));
    jac_add(out, acc, jac_psi(pj, 3));
}
__device__ __forceinline__ void g2_map_fields_oct(AffP<F2> &out, const F2 &u0, const F2 &u1, int q) {
    AffP<F2> q0, q1, a;
    map_pair_quad<F2>(q0, q1, u0, u1, q & 3);                   // (each quad of the octet maps both: lanes 0 / 1 and 4 / 5)
    JacP<F2> s{q0.x, q0.y, f2_one(), false}, c;
    jac_add_mixed_oct(s, q1, q);
    jac_to_affine(a, s);
    g2_clear_cofactor29_oct(c, a, q);
    jac_to_affine(out, c);
}
GPBC_KERNEL k_g2_map_fields_oct(const uint8_t *__restrict__ u, uint8_t *__restrict__ out, size_t n) {
    const size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x, i = lane >> 3;
    if (i >= n) return;
    AffP<F2> r;
    g2_map_fields_oct(r, f2_load(u + i * 128), f2_load(u + i * 128 + 64), (int)(lane & 7));
    if ((lane & 7) == 0) g2_store_aff(out + i * GPBC_G2_BYTES, r);
}
GPBC_KERNEL k_g2_hash_oct(const uint8_t *__restrict__ msgs, const uint64_t *__restrict__ off, size_t total, size_t n, XmdDst dst, uint8_t *__restrict__ out) {
    const size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x, i = lane >> 3;
    if (i >= n) return;
    uint64_t lo, len;
    msg_range(off, total, i, lo, len);
    uint32_t u[48];
    expand_message_xmd<6>(u, msgs + lo, len, dst);
    AffP<F2> r;
    g2_map_fields_oct(r, F2{xmd_field(u, 0), xmd_field(u, 1)}, F2{xmd_field(u, 2), xmd_field(u, 3)}, (int)(lane & 7));
    if ((lane & 7) == 0) g2_store_aff(out + i * GPBC_G2_BYTES, r);
}
GPBC_KERNEL k_g1_map_fields_quad(const uint8_t *__restrict__ u, uint8_t *__restrict__ out, size_t n) {
    const size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x, i = lane >> 2;
    if (i >= n) return;
    AffP<Fe> r;
    g1_map_fields_quad(r, fe_load(u + i * 64), fe_load(u + i * 64 + 32), (int)(lane & 3));
    if ((lane & 3) == 0) g1_store_aff(out + i * GPBC_G1_BYTES, r);
}
GPBC_KERNEL k_g2_map_fields_quad(const uint8_t *__restrict__ u, uint8_t *__restrict__ out, size_t n) {
    const size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x, i = lane >> 2;
    if (i >= n) return;
    AffP<F2> r;
    g2_map_fields_quad(r, f2_load(u + i * 128), f2_load(u + i * 128 + 64), (int)(lane & 3));
    if ((lane & 3) == 0) g2_store_aff(out + i * GPBC_G2_BYTES, r);
}
GPBC_KERNEL k_g1_hash_quad(const uint8_t *__restrict__ msgs, const uint64_t *__restrict__ off, size_t total, size_t n, XmdDst dst, uint8_t *__restrict__ out) {
    const size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x, i = lane >> 2;
    if (i >= n) return;
    uint64_t lo, len;
    msg_range(off, total, i, lo, len);
    uint32_t u[24];
    expand_message_xmd<3>(u, msgs + lo, len, dst);
    AffP<Fe> r;
    g1_map_fields_quad(r, xmd_field(u, 0), xmd_field(u, 1), (int)(lane & 3));
    if ((lane & 3) == 0) g1_store_aff(out + i * GPBC_G1_BYTES, r);
}
GPBC_KERNEL k_g2_hash_quad(const uint8_t *__restrict__ msgs, const uint64_t *__restrict__ off, size_t total, size_t n, XmdDst dst, uint8_t *__restrict__ out) {
    const size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x, i = lane >> 2;
    if (i >= n) return;
    uint64_t lo, len;
    msg_range(off, total, i, lo, len);
    uint32_t u[48];
    expand_message_xmd<6>(u, msgs + lo, len, dst);
    AffP<F2> r;
    g2_map_fields_quad(r, F2{xmd_field(u, 0), xmd_field(u, 1)}, F2{xmd_field(u, 2), xmd_field(u, 3)}, (int)(lane & 3));
    if ((lane & 3) == 0) g2_store_aff(out + i * GPBC_G2_BYTES, r);
}

extern "C" {

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
    else if (kind == 1 && n <= WIRE_OCT_MAX) k_g2_decode_oct<<<grid_for(8 * n), BLOCK, 0, st>>>((const uint8_t *)d_in, (int)elem_bytes, (uint8_t *)d_out, d_ok, n);
    else if (kind == 1 && n <= WIRE_QUAD_MAX) k_g2_decode_quad<<<grid_for(4 * n), BLOCK, 0, st>>>((const uint8_t *)d_in, (int)elem_bytes, (uint8_t *)d_out, d_ok, n);
    else if (kind == 1) k_g2_decode<<<grid_for(n), BLOCK, 0, st>>>((const uint8_t *)d_in, (int)elem_bytes, (uint8_t *)d_out, d_ok, n);
    else k_gt_decode<<<grid_for(n), BLOCK, 0, st>>>((const uint8_t *)d_in, (uint8_t *)d_out, d_ok, n);
    return check_launch("wire decode");
}
static int marshal_one(int kind, const void *in, size_t n, int compressed, void *out) {
    TRY(bind_device());
    if (n <= LANE_CALL_MAX_UNITS)      // a small call: through a call lane (pinned block in and out, the lane's stream), gpbc_common.hpp
        return with_call_lane([&](CallLane &l) {
            const size_t ib = n * wire_mem_bytes(kind), ob = n * wire_enc_bytes(kind, compressed), o_out = Scratch::padded(ib);
            TRY(l.reserve(o_out + Scratch::padded(ob), 0));
            memcpy(l.pin, in, ib);
            TRY(marshal_dev(kind, l.d_pin, n, compressed, l.d_pin + o_out, l.stream));
            HIP_TRY(hipStreamSynchronize(l.stream));
            memcpy(out, l.pin + o_out, ob);
            return (int)GPBC_OK;
        });
    DevBuf dI, dO;
    TRY(dI.upload(in, n * wire_mem_bytes(kind))); TRY(dO.alloc(n * wire_enc_bytes(kind, compressed)));
    TRY(marshal_dev(kind, dI.p, n, compressed, dO.p, nullptr));
    TRY(sync_default());
    return dO.download(out, n * wire_enc_bytes(kind, compressed));
}
// host-pointer entries shard [0, n) over the bound devices (run_sharded, gpbc_core.hip)
constexpr size_t WIRE_SHARD_MIN = 65536;
static int marshal_host(int kind, const void *in, size_t n, int compressed, void *out) {
    if (!n) return GPBC_OK;
    if (!in || !out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    const size_t mb = wire_mem_bytes(kind), eb = wire_enc_bytes(kind, compressed);
    return run_sharded(n, WIRE_SHARD_MIN, [=](size_t lo, size_t hi) {
        return marshal_one(kind, (const uint8_t *)in + lo * mb, hi - lo, compressed, (uint8_t *)out + lo * eb);
    });
}
static int unmarshal_one(int kind, const void *in, size_t elem_bytes, size_t n, void *out, uint8_t *ok) {
    if (n && n <= LANE_CALL_MAX_UNITS)
        return with_call_lane([&](CallLane &l) {
            const size_t ib = n * elem_bytes, ob = n * wire_mem_bytes(kind), o_out = Scratch::padded(ib), o_ok = o_out + Scratch::padded(ob);
            TRY(l.reserve(o_ok + Scratch::padded(n), 0));
            memcpy(l.pin, in, ib);
            TRY(unmarshal_dev(kind, l.d_pin, elem_bytes, n, l.d_pin + o_out, l.d_pin + o_ok, l.stream));
            HIP_TRY(hipStreamSynchronize(l.stream));
            memcpy(out, l.pin + o_out, ob); memcpy(ok, l.pin + o_ok, n);
            return (int)GPBC_OK;
        });
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
static int unmarshal_host(int kind, const void *in, size_t elem_bytes, size_t n, void *out, uint8_t *ok) {
    if (n && (!in || !out || !ok)) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    const size_t mb = wire_mem_bytes(kind);
    return run_sharded(n, kind == 1 ? 8192 : WIRE_SHARD_MIN, [=](size_t lo, size_t hi) {      // G2 decoding carries the subgroup check
        return unmarshal_one(kind, n ? (const uint8_t *)in + lo * elem_bytes : nullptr, elem_bytes, hi - lo, n ? (uint8_t *)out + lo * mb : nullptr, n ? ok + lo : nullptr);
    });
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
    if (g2 && n <= H2C_OCT_MAX) {
        k_g2_map_fields_oct<<<grid_for(8 * n), BLOCK, 0, (hipStream_t)stream>>>((const uint8_t *)d_u, (uint8_t *)d_out, n);
        return check_launch("k_g2_map_fields_oct");
    }
    if (n <= H2C_QUAD_MAX) {
        if (g2) k_g2_map_fields_quad<<<grid_for(4 * n), BLOCK, 0, (hipStream_t)stream>>>((const uint8_t *)d_u, (uint8_t *)d_out, n);
        else k_g1_map_fields_quad<<<grid_for(4 * n), BLOCK, 0, (hipStream_t)stream>>>((const uint8_t *)d_u, (uint8_t *)d_out, n);
        return check_launch(g2 ? "k_g2_map_fields_quad" : "k_g1_map_fields_quad");
    }
    if (g2) k_g2_map_fields<<<grid_for(n), BLOCK, 0, (hipStream_t)stream>>>((const uint8_t *)d_u, (uint8_t *)d_out, n);
    else k_g1_map_fields<<<grid_for(n), BLOCK, 0, (hipStream_t)stream>>>((const uint8_t *)d_u, (uint8_t *)d_out, n);
    return check_launch(g2 ? "k_g2_map_fields" : "k_g1_map_fields");
}
static int map_fields_one(bool g2, const void *u, size_t n, void *out) {
    TRY(bind_device());
    size_t pt = g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES;               // two field elements occupy as many bytes as one point
    if (n <= LANE_CALL_MAX_UNITS)
        return with_call_lane([&](CallLane &l) {
            const size_t o_out = Scratch::padded(n * pt);
            TRY(l.reserve(2 * o_out, 0));
            memcpy(l.pin, u, n * pt);
            TRY(map_fields_dev(g2, l.d_pin, n, l.d_pin + o_out, l.stream));
            HIP_TRY(hipStreamSynchronize(l.stream));
            memcpy(out, l.pin + o_out, n * pt);
            return (int)GPBC_OK;
        });
    DevBuf dU, dO;
    TRY(dU.upload(u, n * pt)); TRY(dO.alloc(n * pt));
    TRY(map_fields_dev(g2, dU.p, n, dO.p, nullptr));
    TRY(sync_default());
    return dO.download(out, n * pt);
}
static int map_fields_host(bool g2, const void *u, size_t n, void *out) {
    if (!n) return GPBC_OK;
    if (!u || !out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    const size_t pt = g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES;
    return run_sharded(n, 8192, [=](size_t lo, size_t hi) { return map_fields_one(g2, (const uint8_t *)u + lo * pt, hi - lo, (uint8_t *)out + lo * pt); });
}
int gpbc_g1_map_to_curve_batch(const void *u, size_t n, void *o) { return map_fields_host(false, u, n, o); }
int gpbc_g2_map_to_curve_batch(const void *u, size_t n, void *o) { return map_fields_host(true, u, n, o); }
int gpbc_g1_map_to_curve_batch_dev(const void *u, size_t n, void *o, void *st) { return map_fields_dev(false, u, n, o, st); }
int gpbc_g2_map_to_curve_batch_dev(const void *u, size_t n, void *o, void *st) { return map_fields_dev(true, u, n, o, st); }

// ----------------------------------------------------------------------------------------------- hash to curve (whole) and hash to field
// what: 0 = G1 points, 1 = G2 points, 2 / 4 = that many field elements per message
static size_t hash_out_bytes(int what) { return what == 0 ? GPBC_G1_BYTES : what == 1 ? GPBC_G2_BYTES : (size_t)what * 32; }
static int hash_dev(int what, const void *d_msgs, const uint64_t *d_off, size_t msgs_bytes, size_t n, const void *dst, size_t dst_len, void *d_out, void *stream) {
    if (!n) return GPBC_OK;
    if (!d_off || !d_out || (msgs_bytes && !d_msgs) || (dst_len && !dst)) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    if (dst_len > 255) return fail(GPBC_ERR_INVALID_ARG, "DST longer than 255 bytes: hash it down first (RFC 9380 section 5.3.3)");
    TRY(bind_device());
    XmdDst d;
    memset(&d, 0, sizeof d);
    if (dst_len) memcpy(d.b, dst, dst_len);                      // dst == NULL with dst_len == 0 is a legal (empty) tag
    d.len = (uint32_t)dst_len;
    hipStream_t st = (hipStream_t)stream;
    const uint8_t *m = (const uint8_t *)d_msgs;
    uint8_t *o = (uint8_t *)d_out;
    switch (what) {
        case 0: if (n <= H2C_QUAD_MAX) k_g1_hash_quad<<<grid_for(4 * n), BLOCK, 0, st>>>(m, d_off, msgs_bytes, n, d, o);
                else k_g1_hash<<<grid_for(n), BLOCK, 0, st>>>(m, d_off, msgs_bytes, n, d, o);
                break;
        case 1: if (n <= H2C_OCT_MAX) k_g2_hash_oct<<<grid_for(8 * n), BLOCK, 0, st>>>(m, d_off, msgs_bytes, n, d, o);
                else if (n <= H2C_QUAD_MAX) k_g2_hash_quad<<<grid_for(4 * n), BLOCK, 0, st>>>(m, d_off, msgs_bytes, n, d, o);
                else k_g2_hash<<<grid_for(n), BLOCK, 0, st>>>(m, d_off, msgs_bytes, n, d, o);
                break;
        case 2: k_hash_to_field<2><<<grid_for(n), BLOCK, 0, st>>>(m, d_off, msgs_bytes, n, d, o); break;
        case 4: k_hash_to_field<4><<<grid_for(n), BLOCK, 0, st>>>(m, d_off, msgs_bytes, n, d, o); break;
        default: return fail(GPBC_ERR_INVALID_ARG, "count must be 2 or 4");
    }
    return check_launch("k_hash");
}
// Small calls of HashToG1 / HashToG2 (every BLS Sign and Verify starts with one: signature/bls01_signature/bls_signature.go:58-63,75-79) are
// COMBINED like the pairings (gpbc_common.hpp "Small host-pointer calls"): requests with the same domain-separation tag — the key —
// share one launch; in[0] = the call's message bytes, seg = its offsets (rebased to 0), units = messages.
static int small_hash_run(int what, CallLane &lane, SmallCall *const *calls, size_t nc) {
    size_t N = 0, B = 0;
    for (size_t c = 0; c < nc; c++) { N += calls[c]->units; B += (size_t)calls[c]->seg[calls[c]->units]; }
    const size_t ob = hash_out_bytes(what), o_off = Scratch::padded(B), o_out = o_off + Scratch::padded((N + 1) * sizeof(uint64_t));
    TRY(lane.reserve(o_out + Scratch::padded(N * ob), 0));
    uint64_t *off = (uint64_t *)(lane.pin + o_off);
    size_t n0 = 0, b0 = 0;
    for (size_t c = 0; c < nc; c++) {
        const SmallCall &r = *calls[c];
        const size_t bytes = (size_t)r.seg[r.units];
        if (bytes) memcpy(lane.pin + b0, r.in[0], bytes);
        for (size_t i = 0; i < r.units; i++) off[n0 + i] = b0 + r.seg[i];
        n0 += r.units; b0 += bytes;
    }
    off[N] = B;
    TRY(hash_dev(what, lane.d_pin, (const uint64_t *)(lane.d_pin + o_off), B, N, calls[0]->key, calls[0]->key_len, lane.d_pin + o_out, lane.stream));
    HIP_TRY(hipStreamSynchronize(lane.stream));
    n0 = 0;
    for (size_t c = 0; c < nc; c++) { memcpy(calls[c]->out[0], lane.pin + o_out + n0 * ob, calls[c]->units * ob); n0 += calls[c]->units; }
    return GPBC_OK;
}
static int small_hash_g1_run(CallLane &l, SmallCall *const *c, size_t n) { return small_hash_run(0, l, c, n); }
static int small_hash_g2_run(CallLane &l, SmallCall *const *c, size_t n) { return small_hash_run(1, l, c, n); }
static int hash_one(int what, const uint8_t *msgs, const uint64_t *off, size_t n, const void *dst, size_t dst_len, uint8_t *out) {
    TRY(bind_device());
    const uint64_t base = off[0], bytes = off[n] - base;
    std::vector<uint64_t> rel(n + 1);
    for (size_t i = 0; i <= n; i++) rel[i] = off[i] - base;
    if ((what == 0 || what == 1) && n <= SMALL_CALL_MAX_UNITS && bytes <= ((size_t)1 << 20) && dst_len <= 255) {
        SmallCall c;
        c.in[0] = msgs + base; c.seg = rel.data(); c.units = n; c.out[0] = out; c.key = dst; c.key_len = dst_len;
        return what == 0 ? small_call(CALL_HASH_G1, c, small_hash_g1_run) : small_call(CALL_HASH_G2, c, small_hash_g2_run);
    }
    if (n <= LANE_CALL_MAX_UNITS && bytes <= ((size_t)64 << 20))
        return with_call_lane([&](CallLane &l) {
            const size_t ob = n * hash_out_bytes(what), o_off = Scratch::padded(bytes), o_out = o_off + Scratch::padded((n + 1) * sizeof(uint64_t));
            TRY(l.reserve(o_out + Scratch::padded(ob), 0));
            if (bytes) memcpy(l.pin, msgs + base, bytes);
            memcpy(l.pin + o_off, rel.data(), (n + 1) * sizeof(uint64_t));
            TRY(hash_dev(what, l.d_pin, (const uint64_t *)(l.d_pin + o_off), bytes, n, dst, dst_len, l.d_pin + o_out, l.stream));
            HIP_TRY(hipStreamSynchronize(l.stream));
            memcpy(out, l.pin + o_out, ob);
            return (int)GPBC_OK;
        });
    DevBuf dM, dOff, dO;
    TRY(dM.upload(msgs + base, bytes)); TRY(dOff.upload(rel.data(), (n + 1) * sizeof(uint64_t))); TRY(dO.alloc(n * hash_out_bytes(what)));
    TRY(hash_dev(what, dM.p, (const uint64_t *)dOff.p, bytes, n, dst, dst_len, dO.p, nullptr));
    TRY(sync_default());
    return dO.download(out, n * hash_out_bytes(what));
}
static int hash_host(int what, const void *msgs, const uint64_t *off, size_t n, const void *dst, size_t dst_len, void *out) {
    if (!n) return GPBC_OK;
    if (!off || !out || (dst_len && !dst)) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    for (size_t i = 0; i < n; i++) if (off[i + 1] < off[i]) return fail(GPBC_ERR_INVALID_ARG, "message offsets must not decrease");
    if (off[n] > off[0] && !msgs) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    const size_t ob = hash_out_bytes(what);
    return run_sharded(n, 8192, [=](size_t lo, size_t hi) { return hash_one(what, (const uint8_t *)msgs, off + lo, hi - lo, dst, dst_len, (uint8_t *)out + lo * ob); });
}
int gpbc_hash_to_g1(const void *m, const uint64_t *off, size_t n, const void *dst, size_t dl, void *o) { return hash_host(0, m, off, n, dst, dl, o); }
int gpbc_hash_to_g2(const void *m, const uint64_t *off, size_t n, const void *dst, size_t dl, void *o) { return hash_host(1, m, off, n, dst, dl, o); }
int gpbc_hash_to_field(const void *m, const uint64_t *off, size_t n, const void *dst, size_t dl, int count, void *o) {
    if (count != 2 && count != 4) return fail(GPBC_ERR_INVALID_ARG, "count must be 2 or 4");
    return hash_host(count, m, off, n, dst, dl, o);
}
int gpbc_hash_to_g1_dev(const void *m, const uint64_t *off, size_t mb, size_t n, const void *dst, size_t dl, void *o, void *st) { return hash_dev(0, m, off, mb, n, dst, dl, o, st); }
int gpbc_hash_to_g2_dev(const void *m, const uint64_t *off, size_t mb, size_t n, const void *dst, size_t dl, void *o, void *st) { return hash_dev(1, m, off, mb, n, dst, dl, o, st); }
int gpbc_hash_to_field_dev(const void *m, const uint64_t *off, size_t mb, size_t n, const void *dst, size_t dl, int count, void *o, void *st) {
    if (count != 2 && count != 4) return fail(GPBC_ERR_INVALID_ARG, "count must be 2 or 4");
    return hash_dev(count, m, off, mb, n, dst, dl, o, st);
}

}  // extern "C"
