// libgpbc_bn254.so, unit 3 of 4: G1 / G2 scalar multiplication, point sums, fixed-base window tables and the sums over
// them, with their C-ABI entries (include/gpbc_bn254.h).  gfx950 only.
#include "gpbc_common.hpp"
#include "curve29_oct.hip.hpp"

// Scalar multiplication: a lane owns SMUL_K points (t, t+T, t+2T, ...; T = ceil(n / SMUL_K)) whose Jacobian results share
// one field inversion.  Measured on MI355X (2^20 points): K = 1 -> 36.9 M G1 / 15.0 M G2 per second, K = 2 -> 36.9 / 13.7,
// K = 4 -> 35.9 / 13.3: holding K results costs more in registers and scratch than the shared inversion saves, so K = 1.
constexpr int SMUL_K = 1;
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
        scalar_mul29_best(res[j], b, k, tabws + t * (size_t)glv_table_dwords<F2>());      // four-dimensional GLS loop
    }
    AffP<F2> aff[SMUL_K];
    jac_to_affine_batch<F2, SMUL_K>(aff, res);
    for (int j = 0; j < SMUL_K; j++) {
        size_t i = t + (size_t)j * T;
        if (i < n) g2_store_aff(out + i * GPBC_G2_BYTES, aff[j]);
    }
}

// The same for calls too small to fill the chip with one point per lane (up to SMUL_QUAD_MAX points): one point per QUAD, the loop's
// doublings and additions three products wide (csrc/curve29_quad.hip.hpp) — a lone lane walks the ~1 600 (G1) / ~2 300 (G2) dependent
// products of a scalar multiplication at a fixed pace whatever the batch, so the depth of the formulas is what such a call pays for.
constexpr size_t SMUL_QUAD_MAX = 16384;     // measured: 16 384 points 0.95 / 1.43 ms (G1 / G2) against 1.27 / 2.09 with one point per lane; 20 000: 1.26 / 2.24 — slower
GPBC_KERNEL_G1 k_g1_scalar_mul_quad(const uint8_t *__restrict__ bases, int shared_base, const uint8_t *__restrict__ scalars, uint8_t *__restrict__ out, size_t n, int32_t *__restrict__ tabws) {
    const size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x, i = lane >> 2;
    const int q = (int)(lane & 3);
    if (i >= n) return;
    AffP<Fe> b = g1_load_aff(bases + (shared_base ? 0 : i * GPBC_G1_BYTES));
    uint32_t k[8];
    load_scalar(k, scalars + i * GPBC_SCALAR_BYTES);
    JacP<Fe> res;
    scalar_mul29_jac_quad<Fe>(res, b, k, tabws + i * (size_t)glv_table_dwords<Fe>(), q);
    AffP<Fe> aff;
    jac_to_affine(aff, res);
    if (q == 0) g1_store_aff(out + i * GPBC_G1_BYTES, aff);
}
GPBC_KERNEL k_g2_scalar_mul_quad(const uint8_t *__restrict__ bases, int shared_base, const uint8_t *__restrict__ scalars, uint8_t *__restrict__ out, size_t n, int32_t *__restrict__ tabws) {
    const size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x, i = lane >> 2;
    const int q = (int)(lane & 3);
    if (i >= n) return;
    AffP<F2> b = g2_load_aff(bases + (shared_base ? 0 : i * GPBC_G2_BYTES));
    uint32_t k[8];
    load_scalar(k, scalars + i * GPBC_SCALAR_BYTES);
    JacP<F2> res;
    scalar_mul29_gls_quad(res, b, k, tabws + i * (size_t)glv_table_dwords<F2>(), q);
    AffP<F2> aff;
    jac_to_affine(aff, res);
    if (q == 0) g2_store_aff(out + i * GPBC_G2_BYTES, aff);
}

// ... and G2 calls of a few points: one point per OCTET of lanes, the two halves of every Fp2 product on two lanes (csrc/curve29_oct.hip.hpp)
constexpr size_t SMUL_OCT_MAX = 2048;
GPBC_KERNEL k_g2_scalar_mul_oct(const uint8_t *__restrict__ bases, int shared_base, const uint8_t *__restrict__ scalars, uint8_t *__restrict__ out, size_t n, int32_t *__restrict__ tabws) {
    const size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x, i = lane >> 3;
    const int q = (int)(lane & 7);
    if (i >= n) return;
    AffP<F2> b = g2_load_aff(bases + (shared_base ? 0 : i * GPBC_G2_BYTES));
    uint32_t k[8];
    load_scalar(k, scalars + i * GPBC_SCALAR_BYTES);
    JacP<F2> res;
    scalar_mul29_gls_oct(res, b, k, tabws + i * (size_t)glv_table_dwords<F2>(), q);
    AffP<F2> aff;
    jac_to_affine(aff, res);
    if (q == 0) g2_store_aff(out + i * GPBC_G2_BYTES, aff);
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

// ---- fixed-base tables (8-bit windows): entry ((b * 32 + w) * 255 + d - 1) = [d * 2^(8w)] base_b, affine, internal limb
// form in the 128-byte-aligned row layout of curve29.hip.hpp (tab_store / tab_load).  1 MB per G1 base, 2 MB per G2 base.
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
    scalar_mul29_best(r, base, k, tabws + t * (size_t)glv_table_dwords<F>());
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
// QUAD: one (chunk, sum) per quad of lanes, the additions five products deep instead of eleven long (csrc/curve29_quad.hip.hpp) — for
// calls with so few (chunk, sum) pairs that a lone lane's chain of additions is all the call consists of (one ScalarMultiplicationBase:
// 32 additions; one 256-term commitment)
template <class F, bool QUAD = false> __device__ __forceinline__ void fb_msm_lane(const int32_t *table, const uint8_t *base_inf, size_t nbase, const uint8_t *scalars,
                                                               size_t n_msm, size_t C, size_t n_chunks, uint8_t *partial) {
    const size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t t = QUAD ? lane >> 2 : lane;
    const int q = (int)(lane & 3);
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
                if constexpr (QUAD) jac_add_mixed_quad(acc, e, q); else jac_add_mixed(acc, acc, e);
            }
        }
    }
    AffP<F> a;
    jac_to_affine(a, acc);
    if (QUAD && q != 0) return;
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

constexpr size_t FB_QUAD_MAX = 16384;              // (chunk, sum) pairs per call up to which the quad form is taken
GPBC_KERNEL_G1 k_g1_fb_msm_quad(const int32_t *__restrict__ table, const uint8_t *__restrict__ base_inf, size_t nbase, const uint8_t *__restrict__ scalars,
                                size_t n_msm, size_t C, size_t n_chunks, uint8_t *__restrict__ partial) {
    fb_msm_lane<Fe, true>(table, base_inf, nbase, scalars, n_msm, C, n_chunks, partial);
}
GPBC_KERNEL k_g2_fb_msm_quad(const int32_t *__restrict__ table, const uint8_t *__restrict__ base_inf, size_t nbase, const uint8_t *__restrict__ scalars,
                             size_t n_msm, size_t C, size_t n_chunks, uint8_t *__restrict__ partial) {
    fb_msm_lane<F2, true>(table, base_inf, nbase, scalars, n_msm, C, n_chunks, partial);
}

extern "C" {

constexpr size_t FB_AUTO_MIN = 16384;
static int shared_base_mul_dev(bool g2, const void *d_base, const void *d_scalars, size_t n, void *d_out, hipStream_t st) {
    const size_t row_dwords = g2 ? (size_t)TabLayout<F2>::ENTRY_DWORDS : (size_t)TabLayout<Fe>::ENTRY_DWORDS;
    const size_t table_bytes = (size_t)FB_ENTRIES * row_dwords * sizeof(int32_t);
    Scratch tmp;
    if (tmp.open(st, 0, Scratch::padded(table_bytes) + 256) != GPBC_OK) return GPBC_ERR_WORKSPACE;
    int32_t *table = tmp.take<int32_t>(table_bytes);
    uint8_t *base_inf = tmp.take(256);
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
            profile_mark(g2 ? "k_g2_fb_build" : "k_g1_fb_build", st);
        }
    }
    if (rc == GPBC_OK) {
        if (g2) k_g2_fb_msm<<<grid_for(n), BLOCK, 0, st>>>(table, base_inf, 1, (const uint8_t *)d_scalars, n, 1, 1, (uint8_t *)d_out);
        else k_g1_fb_msm<<<grid_for(n), BLOCK, 0, st>>>(table, base_inf, 1, (const uint8_t *)d_scalars, n, 1, 1, (uint8_t *)d_out);
        rc = check_launch("k_fb_msm");
        profile_mark(g2 ? "k_g2_fb_msm" : "k_g1_fb_msm", st);
    }
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
        if (g2 && n <= SMUL_OCT_MAX) {
            k_g2_scalar_mul_oct<<<grid_for(8 * m), BLOCK, 0, st>>>(b, shared, k, o, m, tabws);
            TRY(check_launch("k_g2_scalar_mul_oct"));
            profile_mark("k_g2_scalar_mul_oct", st);
            continue;
        }
        if (n <= SMUL_QUAD_MAX) {
            if (g2) k_g2_scalar_mul_quad<<<grid_for(4 * m), BLOCK, 0, st>>>(b, shared, k, o, m, tabws);
            else k_g1_scalar_mul_quad<<<grid_for(4 * m), BLOCK, 0, st>>>(b, shared, k, o, m, tabws);
            TRY(check_launch(g2 ? "k_g2_scalar_mul_quad" : "k_g1_scalar_mul_quad"));
            profile_mark(g2 ? "k_g2_scalar_mul_quad" : "k_g1_scalar_mul_quad", st);
            continue;
        }
        if (g2) k_g2_scalar_mul<<<grid_for(m), BLOCK, 0, st>>>(b, shared, k, o, m, tabws);
        else k_g1_scalar_mul<<<grid_for(m), BLOCK, 0, st>>>(b, shared, k, o, m, tabws);
        TRY(check_launch(g2 ? "k_g2_scalar_mul" : "k_g1_scalar_mul"));
        profile_mark(g2 ? "k_g2_scalar_mul" : "k_g1_scalar_mul", st);
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
        profile_mark(g2 ? "k_g2_sum_level" : "k_g1_sum_level", (hipStream_t)stream);
        if (n_out == 1) break;
        in = out; ws += n_out * pt; n_in = n_out;
    }
    return GPBC_OK;
}
int gpbc_g1_sum_dev(const void *p, size_t n, void *o, void *w, size_t wb, void *s) { return sum_dev(false, p, n, o, w, wb, s); }
int gpbc_g2_sum_dev(const void *p, size_t n, void *o, void *w, size_t wb, void *s) { return sum_dev(true, p, n, o, w, wb, s); }

constexpr size_t SMUL_PIPE_CHUNK = 131072;
// Small calls (gpbc_common.hpp "Small host-pointer calls"): every waiting ScalarMultiplication of one group in one launch of the
// quad-of-lanes kernel on a call lane — bases and scalars read from the lane's pinned block, results written into it, the GLV
// tables in the lane's device block.  A call with one base for all its scalars gets that base repeated per scalar.
static int small_mul_run(bool G2, CallLane &lane, SmallCall *const *calls, size_t nc) {
    const size_t pt = G2 ? GPBC_G2_BYTES : GPBC_G1_BYTES;
    const size_t tab_bytes = sizeof(int32_t) * (G2 ? (size_t)glv_table_dwords<F2>() : (size_t)glv_table_dwords<Fe>());
    size_t N = 0;
    for (size_t c = 0; c < nc; c++) N += calls[c]->units;
    const size_t oB = 0, oS = Scratch::padded(N * pt), oO = oS + Scratch::padded(N * GPBC_SCALAR_BYTES), total = oO + Scratch::padded(N * pt);
    TRY(lane.reserve(total, N * tab_bytes));
    size_t n0 = 0;
    for (size_t c = 0; c < nc; c++) {
        const SmallCall &r = *calls[c];
        if (r.in_one[0]) for (size_t i = 0; i < r.units; i++) memcpy(lane.pin + oB + (n0 + i) * pt, r.in[0], pt);
        else memcpy(lane.pin + oB + n0 * pt, r.in[0], r.units * pt);
        memcpy(lane.pin + oS + n0 * GPBC_SCALAR_BYTES, r.in[1], r.units * GPBC_SCALAR_BYTES);
        n0 += r.units;
    }
    if (G2) k_g2_scalar_mul_oct<<<grid_for(8 * N), BLOCK, 0, lane.stream>>>(lane.d_pin + oB, 0, lane.d_pin + oS, lane.d_pin + oO, N, (int32_t *)lane.dev);   // N <= 2 048 = SMUL_OCT_MAX
    else k_g1_scalar_mul_quad<<<grid_for(4 * N), BLOCK, 0, lane.stream>>>(lane.d_pin + oB, 0, lane.d_pin + oS, lane.d_pin + oO, N, (int32_t *)lane.dev);
    TRY(check_launch(G2 ? "k_g2_scalar_mul_quad" : "k_g1_scalar_mul_quad"));
    profile_mark(G2 ? "k_g2_scalar_mul_quad" : "k_g1_scalar_mul_quad", lane.stream);
    HIP_TRY(hipStreamSynchronize(lane.stream));
    n0 = 0;
    for (size_t c = 0; c < nc; c++) { memcpy(calls[c]->out[0], lane.pin + oO + n0 * pt, calls[c]->units * pt); n0 += calls[c]->units; }
    return GPBC_OK;
}
static int small_g1_mul_run(CallLane &l, SmallCall *const *c, size_t n) { return small_mul_run(false, l, c, n); }
static int small_g2_mul_run(CallLane &l, SmallCall *const *c, size_t n) { return small_mul_run(true, l, c, n); }
static int scalar_mul_one(bool g2, const void *bases, size_t nbase, const void *scalars, size_t n, void *out) {
    TRY(bind_device());
    size_t pt = g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES;
    if (n <= SMALL_CALL_MAX_UNITS) {
        SmallCall c;
        c.in[0] = bases; c.in_one[0] = nbase == 1 && n != 1; c.in[1] = scalars; c.out[0] = out; c.units = n;
        return g2 ? small_call(CALL_G2_MUL, c, small_g2_mul_run) : small_call(CALL_G1_MUL, c, small_g1_mul_run);
    }
    if (nbase == n && n >= 2 * SMUL_PIPE_CHUNK) {
        // one base per scalar, large batch: transfers of neighbouring chunks overlap the kernels (pipelined_chunks)
        DevBuf dB, dS, dO;
        TRY(dB.alloc(n * pt)); TRY(dS.alloc(n * GPBC_SCALAR_BYTES)); TRY(dO.alloc(n * pt));
        int rc = pipelined_chunks(n, SMUL_PIPE_CHUNK,
            [&](size_t off, size_t m, hipStream_t st) {
                HIP_TRY(hipMemcpyAsync(dB.u8() + off * pt, (const uint8_t *)bases + off * pt, m * pt, hipMemcpyHostToDevice, st));
                HIP_TRY(hipMemcpyAsync(dS.u8() + off * GPBC_SCALAR_BYTES, (const uint8_t *)scalars + off * GPBC_SCALAR_BYTES, m * GPBC_SCALAR_BYTES, hipMemcpyHostToDevice, st));
                return (int)GPBC_OK;
            },
            [&](size_t off, size_t m, hipStream_t st) { return scalar_mul_dev(g2, dB.u8() + off * pt, m, dS.u8() + off * GPBC_SCALAR_BYTES, m, dO.u8() + off * pt, st); },
            [&](size_t off, size_t m, hipStream_t st) {
                HIP_TRY(hipMemcpyAsync((uint8_t *)out + off * pt, dO.u8() + off * pt, m * pt, hipMemcpyDeviceToHost, st));
                return (int)GPBC_OK;
            });
        if (rc != GPBC_OK) { (void)hipDeviceSynchronize(); return rc; }
        return GPBC_OK;
    }
    DevBuf dB, dS, dO;
    TRY(dB.upload(bases, nbase * pt)); TRY(dS.upload(scalars, n * GPBC_SCALAR_BYTES)); TRY(dO.alloc(n * pt));
    TRY(scalar_mul_dev(g2, dB.p, nbase, dS.p, n, dO.p, nullptr));
    TRY(sync_default());
    return dO.download(out, n * pt);
}
// host-pointer entries shard [0, n) over the bound devices (run_sharded, gpbc_core.hip)
constexpr size_t SMUL_SHARD_MIN = 4096;
static int scalar_mul_host(bool g2, const void *bases, size_t nbase, const void *scalars, size_t n, void *out) {
    if (!n) return GPBC_OK;
    if (!bases || !scalars || !out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    if (nbase != 1 && nbase != n) return fail(GPBC_ERR_INVALID_ARG, "nbase must be 1 or n");
    const size_t pt = g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES;
    return run_sharded(n, SMUL_SHARD_MIN, [=](size_t lo, size_t hi) {
        const bool shared = nbase == 1 && n != 1;
        return scalar_mul_one(g2, (const uint8_t *)bases + (shared ? 0 : lo * pt), shared ? 1 : hi - lo, (const uint8_t *)scalars + lo * GPBC_SCALAR_BYTES,
                              hi - lo, (uint8_t *)out + lo * pt);
    });
}
int gpbc_g1_scalar_mul_batch(const void *b, size_t nb, const void *s, size_t n, void *o) { return scalar_mul_host(false, b, nb, s, n, o); }
int gpbc_g2_scalar_mul_batch(const void *b, size_t nb, const void *s, size_t n, void *o) { return scalar_mul_host(true, b, nb, s, n, o); }

static int sum_one(bool g2, const void *pts, size_t n, void *out) {
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
// Partial sums of the shards of a host-pointer call: every shard leaves one point in the host array `parts` and the calling
// thread's device adds them up.  No collective here — the data of a host-pointer call is on the host anyway; RCCL carries the
// partial sums only where they are device-resident per rank (gpbc_g1/g2_scalar_mul_sum_dev).
// `partial(lo, hi, d_out)` leaves the shard's sum in device memory at d_out (stream 0).
static int sharded_point_sum(bool g2, size_t n, size_t min_units, const std::function<int(size_t, size_t, uint8_t *)> &partial, void *out) {
    const size_t pt = g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES;
    const int nd = device_count_initialised();
    std::vector<uint8_t> parts((size_t)(nd > 0 ? nd : 1) * pt, 0);   // rows of devices without a shard stay the point at infinity
    std::atomic<int> used{0};
    const int before = gpbc_get_device();
    TRY(run_sharded(n, min_units, [&](size_t lo, size_t hi) {
        TRY(bind_device());
        const int idx = gpbc_get_device();
        used.fetch_add(1);
        DevBuf dp;
        TRY(dp.alloc(pt));
        TRY(partial(lo, hi, dp.u8()));
        TRY(sync_default());
        return dp.download(parts.data() + (size_t)idx * pt, pt);
    }));
    if (used.load() == 1) { memcpy(out, parts.data() + (size_t)(before > 0 && before < nd ? before : 0) * pt, pt); return GPBC_OK; }
    return sum_one(g2, parts.data(), (size_t)nd, out);
}
static int sum_host(bool g2, const void *pts, size_t n, void *out) {
    if (!out || (n && !pts)) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    const size_t pt = g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES;
    if (device_count_initialised() <= 1 || n < 2 * 65536) return sum_one(g2, pts, n, out);
    return sharded_point_sum(g2, n, 65536, [=](size_t lo, size_t hi, uint8_t *d_out) {
        DevBuf dP, dW;
        TRY(dP.upload((const uint8_t *)pts + lo * pt, (hi - lo) * pt));
        const size_t wsb = gpbc_sum_workspace_bytes(hi - lo, g2);
        TRY(dW.alloc(wsb));
        TRY(sum_dev(g2, dP.p, hi - lo, d_out, dW.p, wsb, nullptr));
        return sync_default();
    }, out);
}
// sum_i [s_i] P_i, host pointers: scalar multiplications and the point-sum tree per shard, partial sums combined as above
static int scalar_mul_sum_host(bool g2, const void *bases, const void *scalars, size_t n, void *out) {
    if (!out || (n && (!bases || !scalars))) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    const size_t pt = g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES;
    if (!n) { memset(out, 0, pt); return GPBC_OK; }
    return sharded_point_sum(g2, n, SMUL_SHARD_MIN, [=](size_t lo, size_t hi, uint8_t *d_out) {
        const size_t m = hi - lo;
        DevBuf dB, dS, dM, dW;
        TRY(dB.upload((const uint8_t *)bases + lo * pt, m * pt)); TRY(dS.upload((const uint8_t *)scalars + lo * GPBC_SCALAR_BYTES, m * GPBC_SCALAR_BYTES));
        if (m >= MSM_MIN_TERMS) {                                  // bucket method instead of m independent multiplications
            const int rc = msm_dev(g2, dB.p, dS.p, m, d_out, nullptr);
            if (rc != MSM_SKEWED) { TRY(rc); return sync_default(); }
        }                                                          // (skewed scalars: term by term, below)
        TRY(dM.alloc(m * pt));
        TRY(scalar_mul_dev(g2, dB.p, m, dS.p, m, dM.p, nullptr));
        const size_t wsb = gpbc_sum_workspace_bytes(m, g2);
        TRY(dW.alloc(wsb));
        TRY(sum_dev(g2, dM.p, m, d_out, dW.p, wsb, nullptr));
        return sync_default();
    }, out);
}
int gpbc_g1_scalar_mul_sum(const void *b, const void *s, size_t n, void *o) { return scalar_mul_sum_host(false, b, s, n, o); }
int gpbc_g2_scalar_mul_sum(const void *b, const void *s, size_t n, void *o) { return scalar_mul_sum_host(true, b, s, n, o); }
// Device-resident form, one rank of a multi-process job: local sum, all-gather of one point per rank, sum of the partials.
static int scalar_mul_sum_dev(bool g2, const void *d_bases, const void *d_scalars, size_t n, void *d_out, void *stream) {
    if (!d_out || (n && (!d_bases || !d_scalars))) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    hipStream_t st = (hipStream_t)stream;
    const size_t pt = g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES;
    const int ranks = comm_ranks();
    bool bucket = n >= MSM_MIN_TERMS;                             // Pippenger instead of n independent multiplications + a sum tree
    const size_t wsb_all = gpbc_sum_workspace_bytes((size_t)(ranks > 1 ? ranks : 1), g2);
    Scratch tmp;                                                   // level 1: msm_dev and the shared-base path use level 0 underneath
    uint8_t *local = nullptr, *all = nullptr, *ws2 = nullptr;
    int rc = GPBC_OK;
    for (;;) {
        // one stream-ordered scratch block: products | local tree workspace | local sum | gathered sums | final tree workspace
        const size_t wsb_local = bucket ? 0 : gpbc_sum_workspace_bytes(n, g2);
        const size_t total = (bucket ? 0 : n * pt) + wsb_local + pt + (size_t)(ranks > 1 ? ranks : 1) * pt + wsb_all;
        TRY(tmp.open(st, 1, total));
        uint8_t *prod = tmp.base, *ws1 = prod + (bucket ? 0 : n * pt);
        local = ws1 + wsb_local; all = local + pt; ws2 = all + (size_t)(ranks > 1 ? ranks : 1) * pt;
        uint8_t *local_out = ranks > 1 ? local : (uint8_t *)d_out;
        if (bucket) {
            rc = msm_dev(g2, d_bases, d_scalars, n, local_out, st);
            if (rc == MSM_SKEWED) { bucket = false; continue; }    // skewed scalars: term by term
        } else {
            if (n) rc = scalar_mul_dev(g2, d_bases, n, d_scalars, n, prod, stream);
            if (rc == GPBC_OK) rc = sum_dev(g2, prod, n, local_out, ws1, wsb_local, stream);
        }
        break;
    }
    if (rc == GPBC_OK && ranks > 1) {
        rc = comm_allgather(local, pt, all, st);
        if (rc == GPBC_OK) rc = sum_dev(g2, all, (size_t)ranks, d_out, ws2, wsb_all, stream);
    }
    return rc;
}
int gpbc_g1_scalar_mul_sum_dev(const void *b, const void *s, size_t n, void *o, void *st) { return scalar_mul_sum_dev(false, b, s, n, o, st); }
int gpbc_g2_scalar_mul_sum_dev(const void *b, const void *s, size_t n, void *o, void *st) { return scalar_mul_sum_dev(true, b, s, n, o, st); }
int gpbc_g1_sum(const void *p, size_t n, void *o) { return sum_host(false, p, n, o); }
int gpbc_g2_sum(const void *p, size_t n, void *o) { return sum_host(true, p, n, o); }

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
    gpbc_fixed_base *h = new gpbc_fixed_base{current_device(), is_g2 ? 1 : 0, nbase, nullptr, nullptr};
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
            profile_mark(is_g2 ? "k_g2_fb_build" : "k_g1_fb_build", st);
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
    if (current_device() != h->device) return fail(GPBC_ERR_INVALID_ARG, "table was built on device %d", h->device);
    size_t C, n_chunks;
    fb_shape(h, n_msm, &C, &n_chunks);
    if (n_chunks > 1 && (!d_workspace || workspace_bytes < gpbc_fixed_base_msm_workspace_bytes(h, n_msm))) return fail(GPBC_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const size_t pt = h->is_g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES;
    uint8_t *partial = n_chunks > 1 ? (uint8_t *)d_workspace : (uint8_t *)d_out;
    const size_t lanes = n_msm * n_chunks;
    if (lanes <= FB_QUAD_MAX) {
        if (h->is_g2) k_g2_fb_msm_quad<<<grid_for(4 * lanes), BLOCK, 0, st>>>(h->table, h->base_inf, h->nbase, (const uint8_t *)d_scalars, n_msm, C, n_chunks, partial);
        else k_g1_fb_msm_quad<<<grid_for(4 * lanes), BLOCK, 0, st>>>(h->table, h->base_inf, h->nbase, (const uint8_t *)d_scalars, n_msm, C, n_chunks, partial);
    } else if (h->is_g2) k_g2_fb_msm<<<grid_for(lanes), BLOCK, 0, st>>>(h->table, h->base_inf, h->nbase, (const uint8_t *)d_scalars, n_msm, C, n_chunks, partial);
    else k_g1_fb_msm<<<grid_for(lanes), BLOCK, 0, st>>>(h->table, h->base_inf, h->nbase, (const uint8_t *)d_scalars, n_msm, C, n_chunks, partial);
    TRY(check_launch("k_fb_msm"));
    profile_mark(h->is_g2 ? "k_g2_fb_msm" : "k_g1_fb_msm", st);
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
        profile_mark(h->is_g2 ? "k_g2_sum_level" : "k_g1_sum_level", st);
        in = out; ws += c2 * n_msm * pt; c = c2;
    }
    return GPBC_OK;
}
// Small fixed-base sums (ScalarMultiplicationBase as the reference calls it: one scalar against the generator table,
// cpabe/bsw07/bsw07_cpabe.go:69, signature/bls01_signature/bls_signature.go:45) COMBINED per table: key = the handle, in[0] = the
// call's scalars (segs rows of nbase each), in[1] = the handle, units = scalars (the batch cap counts terms).
static int small_fixed_base_run(CallLane &lane, SmallCall *const *calls, size_t nc) {
    const gpbc_fixed_base *h = (const gpbc_fixed_base *)calls[0]->in[1];
    const size_t pt = h->is_g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES, row = h->nbase * GPBC_SCALAR_BYTES;
    size_t M = 0;
    for (size_t c = 0; c < nc; c++) M += calls[c]->segs;
    const size_t o_out = Scratch::padded(M * row), wsb = gpbc_fixed_base_msm_workspace_bytes(h, M);
    TRY(lane.reserve(o_out + Scratch::padded(M * pt), wsb));
    size_t m0 = 0;
    for (size_t c = 0; c < nc; c++) { memcpy(lane.pin + m0 * row, calls[c]->in[0], calls[c]->segs * row); m0 += calls[c]->segs; }
    TRY(gpbc_fixed_base_msm_dev(h, lane.d_pin, M, lane.d_pin + o_out, lane.dev, wsb, lane.stream));
    HIP_TRY(hipStreamSynchronize(lane.stream));
    m0 = 0;
    for (size_t c = 0; c < nc; c++) { memcpy(calls[c]->out[0], lane.pin + o_out + m0 * pt, calls[c]->segs * pt); m0 += calls[c]->segs; }
    return GPBC_OK;
}
int gpbc_fixed_base_msm(const gpbc_fixed_base *h, const void *scalars, size_t n_msm, void *out) {
    if (!h) return fail(GPBC_ERR_INVALID_ARG, "null table handle");
    if (!n_msm) return GPBC_OK;
    if (!scalars || !out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    const size_t pt = h->is_g2 ? GPBC_G2_BYTES : GPBC_G1_BYTES;
    if (n_msm * h->nbase <= SMALL_CALL_MAX_UNITS) {
        SmallCall c;
        c.in[0] = scalars; c.in[1] = h; c.out[0] = out; c.units = n_msm * h->nbase; c.segs = n_msm; c.key = &h; c.key_len = sizeof h;
        return small_call(CALL_FIXED_BASE, c, small_fixed_base_run);
    }
    if (n_msm * h->nbase <= LANE_CALL_MAX_UNITS)      // larger, still small: one launch on a call lane of its own, gpbc_common.hpp
        return with_call_lane([&](CallLane &l) {
            const size_t sb = n_msm * h->nbase * GPBC_SCALAR_BYTES, o_out = Scratch::padded(sb), wsb = gpbc_fixed_base_msm_workspace_bytes(h, n_msm);
            TRY(l.reserve(o_out + Scratch::padded(n_msm * pt), wsb));
            memcpy(l.pin, scalars, sb);
            TRY(gpbc_fixed_base_msm_dev(h, l.d_pin, n_msm, l.d_pin + o_out, l.dev, wsb, l.stream));
            HIP_TRY(hipStreamSynchronize(l.stream));
            memcpy(out, l.pin + o_out, n_msm * pt);
            return (int)GPBC_OK;
        });
    DevBuf dS, dO, dW;
    TRY(dS.upload(scalars, n_msm * h->nbase * GPBC_SCALAR_BYTES)); TRY(dO.alloc(n_msm * pt));
    const size_t wsb = gpbc_fixed_base_msm_workspace_bytes(h, n_msm);
    TRY(dW.alloc(wsb));
    TRY(gpbc_fixed_base_msm_dev(h, dS.p, n_msm, dO.p, dW.p, wsb, nullptr));
    TRY(sync_default());
    return dO.download(out, n_msm * pt);
}

}  // extern "C"
