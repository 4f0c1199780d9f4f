// libgpbc_bn254.so, unit 5 of 5: bucket (Pippenger) multi-scalar multiplication over variable bases (csrc/msm29.hip.hpp) —
// the engine behind gpbc_g1/g2_scalar_mul_sum(_dev) from 16 384 terms on.  gfx950 only.
//
// Device plan for n terms, c-bit windows (c = 16 from 2^17 terms on, else 12), W = ceil(256 / c), M = W * 2^c bucket keys:
//   k_msm_count      one lane per term: its W digits -> histogram of the keys (atomics in L2; digit 0 and infinite bases skipped)
//   k_scan_*         exclusive prefix sum of the histogram (three small kernels)
//   k_msm_scatter    one lane per term again: term index into its slot of every bucket it belongs to
//   k_msm_bases      one lane per term: the base in internal limb form (the Montgomery conversion once, not once per window)
//   k_size_hist / k_size_scatter   buckets ordered by size (largest first; block-wise counting sort over 256 size classes) so
//                    that the 64 lanes of a wave run chains of equal length
//   k_msm_buckets    one lane per bucket: mixed additions of its terms (gathered 80 / 160-byte rows) -> Jacobian row
//   k_msm_groups     one lane per (window, 8 digits): running-sum reduction + one multiplication by a 16-bit integer
//   k_msm_rows_sum   fan-in-8 sums of the group rows of a window, level by level (serial chains of 8 additions, not 64:
//                    these last steps have few lanes and are latency-bound)
//   k_msm_finish     one lane: Horner over the windows, affine result in gnark's layout
// The PARTIAL top window (c = 12: W = 22 and window 21 holds bits 252..255 only; uniform scalars below r put a third of the terms
// into each of its digits 1 and 2): its 2^c keys are  (term index mod S) * D + digit  with D = 2^(256 - c (W - 1)) digit values and
// S = 2^c / D sub-buckets per digit, so that its terms spread over the whole key range like any other window's; step 3 then weighs
// a key by its low bits (the digit) instead of the key itself.  Nothing else changes: the tree sum of the window adds the sub-buckets.
// All scratch is one stream-ordered allocation; the call is asynchronous on `stream`.
#include "gpbc_common.hpp"
#include "msm29.hip.hpp"
#include "curve29_quad.hip.hpp"

constexpr int SCAN_BLOCK = 256, SCAN_ITEMS = 4, SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

template <class F> __device__ __forceinline__ bool base_is_inf(const uint8_t *p) {
    if constexpr (sizeof(F) == sizeof(Fe)) return g1_bytes_inf(p); else return g2_bytes_inf(p);
}
template <class F> __device__ __forceinline__ AffP<F> base_load(const uint8_t *p) {
    if constexpr (sizeof(F) == sizeof(Fe)) return g1_load_aff(p); else return g2_load_aff(p);
}

template <class F, bool SCATTER>
__global__ void __launch_bounds__(BLOCK) k_msm_keys(const uint8_t *__restrict__ bases, const uint8_t *__restrict__ scalars, size_t n, int c, int W,
                                                   uint32_t *__restrict__ counts, const uint32_t *__restrict__ offsets, uint32_t *__restrict__ idx) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    constexpr size_t PT = sizeof(F) == sizeof(Fe) ? GPBC_G1_BYTES : GPBC_G2_BYTES;
    if (base_is_inf<F>(bases + i * PT)) return;
    uint32_t k[8];
    load_scalar(k, scalars + i * GPBC_SCALAR_BYTES);
    const int top_bits = 256 - c * (W - 1);                  // digit bits of the last window (= c when c divides 256)
    for (int w = 0; w < W; w++) {
        uint32_t d = msm_digit(k, w, c);
        if (!d) continue;
        if (w == W - 1 && top_bits < c) d |= ((uint32_t)i & ((1u << (c - top_bits)) - 1u)) << top_bits;     // sub-bucket (i mod S) above the digit
        const uint32_t key = ((uint32_t)w << c) | d;
        const uint32_t pos = atomicAdd(&counts[key], 1u);
        if (SCATTER) idx[offsets[key] + pos] = (uint32_t)i;
    }
}

// the longest bucket: one lane adds a whole bucket serially (k_msm_buckets), so its length is the depth of the bucket stage
// out[w] = the longest bucket of window w (a block of 256 keys lies inside one window: 2^c is a multiple of 256)
__global__ void __launch_bounds__(SCAN_BLOCK) k_msm_max_count(const uint32_t *__restrict__ counts, size_t M, int c, uint32_t *__restrict__ out) {
    __shared__ uint32_t sh;
    if (threadIdx.x == 0) sh = 0;
    __syncthreads();
    const size_t key = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    if (key < M && counts[key]) atomicMax(&sh, counts[key]);
    __syncthreads();
    if (threadIdx.x == 0 && sh) atomicMax(&out[((size_t)blockIdx.x * SCAN_BLOCK) >> c], sh);
}

// exclusive scan of m counters: per-tile scan + tile totals, scan of the totals (one block), add-back
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_tiles(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t *__restrict__ tile_sum, size_t m) {
    __shared__ uint32_t sh[SCAN_BLOCK];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], s = 0;
    for (int j = 0; j < SCAN_ITEMS; j++) { v[j] = base + j < m ? in[base + j] : 0; s += v[j]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
        uint32_t t = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = sh[threadIdx.x] - s;                      // exclusive prefix of this thread within the tile
    for (int j = 0; j < SCAN_ITEMS; j++) { if (base + j < m) out[base + j] = run; run += v[j]; }
    if (threadIdx.x == SCAN_BLOCK - 1) tile_sum[blockIdx.x] = sh[threadIdx.x];
}
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_tops(uint32_t *__restrict__ tile_sum, size_t n_tiles, uint32_t *__restrict__ total) {
    __shared__ uint32_t sh[SCAN_BLOCK];
    uint32_t carry = 0;
    for (size_t b = 0; b < n_tiles; b += SCAN_BLOCK) {
        const size_t i = b + threadIdx.x;
        const uint32_t v = i < n_tiles ? tile_sum[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
            uint32_t t = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n_tiles) tile_sum[i] = carry + sh[threadIdx.x] - v;
        const uint32_t block_total = sh[SCAN_BLOCK - 1];
        __syncthreads();
        carry += block_total;
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_add(uint32_t *__restrict__ out, const uint32_t *__restrict__ tile_sum, size_t m, const uint32_t *__restrict__ total) {
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    const uint32_t add = tile_sum[blockIdx.x];
    for (int j = 0; j < SCAN_ITEMS; j++) if (base + j < m) out[base + j] += add;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[m] = *total;   // sentinel: offsets[m] = number of (term, window) entries
}

// bases in internal limb form: x, y limbs + infinity flag, 20 (G1) / 40 (G2) dwords per point
template <class F> struct BaseRow { static constexpr int DWORDS = sizeof(F) == sizeof(Fe) ? 20 : 40; };
template <class F> __global__ void __launch_bounds__(BLOCK) k_msm_bases(const uint8_t *__restrict__ bases, size_t n, int32_t *__restrict__ rows) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    constexpr size_t PT = sizeof(F) == sizeof(Fe) ? GPBC_G1_BYTES : GPBC_G2_BYTES;
    constexpr int E = sizeof(F) / sizeof(Fe) * NL;
    AffP<F> a = base_load<F>(bases + i * PT);
    int32_t *r = rows + i * BaseRow<F>::DWORDS;
    limbs_store(r, a.x); limbs_store(r + E, a.y);
    r[2 * E] = a.inf ? 1 : 0;
    r[2 * E + 1] = 0;
}
template <class F> __device__ __forceinline__ AffP<F> base_row_load(const int32_t *r) {
    constexpr int E = sizeof(F) / sizeof(Fe) * NL;
    AffP<F> a;
    limbs_load(a.x, r); limbs_load(a.y, r + E);
    a.inf = r[2 * E] != 0;
    return a;
}

// Buckets ordered by size, largest first: a block-wise counting sort over 256 size classes (size 255 and more share one).
// k_size_hist: per block of 256 keys, how many fall into each class, stored class-major (hist[class * n_blocks + block]) so that
// ONE exclusive scan of that array yields every block's first slot for every class; k_size_scatter: each key takes the next slot
// of its (class, block) cell.  perm[slot] = key.
__device__ __forceinline__ uint32_t size_class(const uint32_t *offsets, size_t key, size_t M, int c) {
    if (key >= M || (key & (((size_t)1 << c) - 1)) == 0) return 255u;          // digit 0 / padding: empty, sorted last
    const uint32_t cnt = offsets[key + 1] - offsets[key];
    return 255u - (cnt > 255u ? 255u : cnt);
}
__global__ void __launch_bounds__(SCAN_BLOCK) k_size_hist(const uint32_t *__restrict__ offsets, size_t M, int c, uint32_t *__restrict__ hist, size_t n_blocks) {
    __shared__ uint32_t sh[256];
    sh[threadIdx.x] = 0;
    __syncthreads();
    const size_t key = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    if (key < M) atomicAdd(&sh[size_class(offsets, key, M, c)], 1u);
    __syncthreads();
    hist[(size_t)threadIdx.x * n_blocks + blockIdx.x] = sh[threadIdx.x];
}
__global__ void __launch_bounds__(SCAN_BLOCK) k_size_scatter(const uint32_t *__restrict__ offsets, size_t M, int c, const uint32_t *__restrict__ hist_scanned, size_t n_blocks,
                                                            uint32_t *__restrict__ perm) {
    __shared__ uint32_t sh[256];
    sh[threadIdx.x] = hist_scanned[(size_t)threadIdx.x * n_blocks + blockIdx.x];
    __syncthreads();
    const size_t key = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    if (key < M) perm[atomicAdd(&sh[size_class(offsets, key, M, c)], 1u)] = (uint32_t)key;
}

template <class F> __global__ void __launch_bounds__(BLOCK, sizeof(F) == sizeof(Fe) ? GPBC_WAVES_G1 : GPBC_WAVES_PER_SIMD)
k_msm_buckets(const int32_t *__restrict__ base_rows, const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ perm, int c, size_t M,
              int32_t *__restrict__ rows) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= M) return;
    const size_t key = perm[t];                                              // neighbouring lanes: buckets of (nearly) equal size
    JacP<F> acc;
    if ((key & (((size_t)1 << c) - 1)) == 0) jac_set_inf(acc);               // digit 0 has no bucket
    else msm_bucket_sum(acc, offsets[key], offsets[key + 1], [&](size_t j) { return base_row_load<F>(base_rows + (size_t)idx[j] * BaseRow<F>::DWORDS); });
    jac_row_store(rows + key * JacRow<F>::DWORDS, acc);
}
template <class F> __global__ void __launch_bounds__(BLOCK, sizeof(F) == sizeof(Fe) ? GPBC_WAVES_G1 : GPBC_WAVES_PER_SIMD)
k_msm_groups(const int32_t *__restrict__ buckets, int c, int W, size_t n_groups_total, int32_t *__restrict__ out) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n_groups_total) return;
    const size_t groups_per_window = ((size_t)1 << c) / MSM_GROUP;
    const size_t w = t / groups_per_window, g = t % groups_per_window;
    const int top_bits = 256 - c * (W - 1);
    // keys [first, first + 8) of the window; a key's weight is the key itself, or — in a partial top window — its low top_bits bits
    // (the digit; the bits above number the sub-bucket): weights lo..hi sit at keys key0 + lo .. key0 + hi
    const uint32_t first = (uint32_t)(g * MSM_GROUP);
    const uint32_t dmask = ((int)w == W - 1 && top_bits < c) ? (1u << top_bits) - 1u : ~0u;
    const uint32_t lo = first & dmask, hi = lo + MSM_GROUP - 1, key0 = first - lo;
    const int32_t *win = buckets + ((w << c) + key0) * JacRow<F>::DWORDS;
    JacP<F> r;
    msm_group_reduce(r, lo ? lo : 1u, hi, [&](uint32_t d) { JacP<F> b; jac_row_load(b, win + (size_t)d * JacRow<F>::DWORDS); return b; });
    jac_row_store(out + t * JacRow<F>::DWORDS, r);
}
// out[t] = sum_{j < fan} in[t * fan + j]
template <class F> __global__ void __launch_bounds__(BLOCK, sizeof(F) == sizeof(Fe) ? GPBC_WAVES_G1 : GPBC_WAVES_PER_SIMD)
k_msm_rows_sum(const int32_t *__restrict__ in, size_t n_out, size_t fan, int32_t *__restrict__ out) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n_out) return;
    JacP<F> acc, b, s;
    jac_set_inf(acc);
    for (size_t j = 0; j < fan; j++) {
        jac_row_load(b, in + (t * fan + j) * JacRow<F>::DWORDS);
        jac_add(s, acc, b);
        acc = s;
    }
    jac_row_store(out + t * JacRow<F>::DWORDS, acc);
}
// Horner over the windows is ONE chain of (W - 1) c doublings — 240 for sixteen 16-bit windows — and a lone lane walks it at ~14 k
// cycles per doubling: a quarter of a 2^20-term sum was this kernel.  The seven products of jac_dbl are three levels deep
// (X^2, Y^2, Y 2Z | X 4B, B 8B, (3A)^2 | E (S - x3)), so three lanes of a quad take one product each per level and pass the results
// round by DPP broadcasts (jac_dbl_quad, csrc/curve29_quad.hip.hpp): three product times per doubling instead of seven.
template <class F> __global__ void __launch_bounds__(BLOCK) k_msm_finish(const int32_t *__restrict__ window_sums, int c, int W, uint8_t *__restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x >= 4) return;            // one quad
    const int q = threadIdx.x;
    JacP<F> acc, s, t;
    jac_row_load(acc, window_sums + (size_t)(W - 1) * JacRow<F>::DWORDS);
    for (int w = W - 2; w >= 0; w--) {
        jac_row_load(s, window_sums + (size_t)w * JacRow<F>::DWORDS);
        for (int i = 0; i < c; i++) jac_dbl_quad(acc, q);
        jac_add(t, acc, s);                                     // every lane the same sum: nothing to exchange
        acc = t;
    }
    if (q != 0) return;
    AffP<F> r;
    jac_to_affine(r, acc);
    if constexpr (sizeof(F) == sizeof(Fe)) g1_store_aff(out, r); else g2_store_aff(out, r);
}

constexpr size_t MSM_FAN = 8;
static std::atomic<uint64_t> g_msm_bucket_runs{0}, g_msm_skewed{0};       // which way msm_run went (gpbc_msm_stats: tests assert the path)
struct MsmPlan { int c, W; size_t M, n_groups, n_size_blocks, bytes; size_t off_counts, off_offsets, off_tiles, off_total, off_idx, off_buckets, off_groups, off_tmp, off_bases, off_hist, off_hist_scanned, off_perm; };
static MsmPlan msm_plan(bool g2, size_t n) {
    MsmPlan p;
    p.c = n >= ((size_t)1 << 17) ? 16 : 12;
    p.W = (256 + p.c - 1) / p.c;
    static_assert(256 - 12 * 21 == 4 && 256 % 16 == 0, "c = 12 leaves a 4-bit top window (16 digit values >= MSM_GROUP: a group of step 3 stays inside one sub-bucket row), c = 16 none");
    p.M = (size_t)p.W << p.c;
    p.n_groups = p.M / MSM_GROUP;
    const size_t row = (g2 ? JacRow<F2>::DWORDS : JacRow<Fe>::DWORDS) * sizeof(int32_t);
    const size_t gpw = ((size_t)1 << p.c) / MSM_GROUP;                 // groups per window: 8192 (c = 16) or 512 (c = 12)
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o = 0;
    p.off_counts = o;  o += up(p.M * 4);
    p.off_offsets = o; o += up((p.M + 1) * 4);
    p.off_tiles = o;   o += up((((p.M > 256 * ((p.M + SCAN_BLOCK - 1) / SCAN_BLOCK) ? p.M : 256 * ((p.M + SCAN_BLOCK - 1) / SCAN_BLOCK)) + SCAN_TILE - 1) / SCAN_TILE) * 4);   // tile sums of the larger of the two scans
    p.off_total = o;   o += 256;
    p.off_idx = o;     o += up(n * (size_t)p.W * 4);
    p.off_buckets = o; o += up(p.M * row);
    p.off_groups = o;  o += up(p.n_groups * row);
    p.off_tmp = o;     o += up((size_t)p.W * (gpw / MSM_FAN) * row);   // the tree's ping-pong partner of the groups array
    p.n_size_blocks = (p.M + SCAN_BLOCK - 1) / SCAN_BLOCK;
    p.off_bases = o;   o += up(n * (g2 ? BaseRow<F2>::DWORDS : BaseRow<Fe>::DWORDS) * sizeof(int32_t));
    p.off_hist = o;    o += up(256 * p.n_size_blocks * 4);
    p.off_hist_scanned = o; o += up((256 * p.n_size_blocks + 1) * 4);
    p.off_perm = o;    o += up(p.M * 4);
    p.bytes = o;
    return p;
}

template <class F> static int msm_run(const uint8_t *d_bases, const uint8_t *d_scalars, size_t n, uint8_t *d_out, hipStream_t st) {
    constexpr bool G2 = sizeof(F) != sizeof(Fe);
    const MsmPlan p = msm_plan(G2, n);
    if (n * (size_t)p.W >= ((size_t)1 << 32)) return fail(GPBC_ERR_INVALID_ARG, "MSM of %zu terms exceeds the 32-bit index space of one call", n);
    Scratch scratch;
    TRY(scratch.open(st, 0, p.bytes));
    uint8_t *mem = scratch.base;
    uint32_t *counts = (uint32_t *)(mem + p.off_counts), *offsets = (uint32_t *)(mem + p.off_offsets), *tiles = (uint32_t *)(mem + p.off_tiles);
    uint32_t *total = (uint32_t *)(mem + p.off_total), *idx = (uint32_t *)(mem + p.off_idx);
    int32_t *buckets = (int32_t *)(mem + p.off_buckets), *groups = (int32_t *)(mem + p.off_groups), *tmp = (int32_t *)(mem + p.off_tmp);
    int32_t *base_rows = (int32_t *)(mem + p.off_bases);
    uint32_t *hist = (uint32_t *)(mem + p.off_hist), *hist_scanned = (uint32_t *)(mem + p.off_hist_scanned), *perm = (uint32_t *)(mem + p.off_perm);
    const size_t hist_len = 256 * p.n_size_blocks, hist_tiles = (hist_len + SCAN_TILE - 1) / SCAN_TILE;
    const size_t n_tiles = (p.M + SCAN_TILE - 1) / SCAN_TILE;
    const size_t gpw = ((size_t)1 << p.c) / MSM_GROUP;
    int rc = GPBC_OK;
    auto step = [&](const char *name) { if (rc == GPBC_OK) { rc = check_launch(name); profile_mark(name, st); } };
    if (hipMemsetAsync(counts, 0, p.M * 4, st) != hipSuccess) rc = fail(GPBC_ERR_HIP, "hipMemsetAsync failed");
    if (rc == GPBC_OK) { k_msm_keys<F, false><<<grid_for(n), BLOCK, 0, st>>>(d_bases, d_scalars, n, p.c, p.W, counts, nullptr, nullptr); step(G2 ? "k_msm_count_g2" : "k_msm_count_g1"); }
    // Skewed scalars (all rho equal, small integers, one repeated value): a window's terms land in ONE bucket, and the lane that owns
    // it would run n dependent additions — seconds.  The histogram says so before any point is touched: when the longest bucket is far
    // above the mean the caller takes the per-term path (n independent scalar multiplications + the sum tree: milliseconds).  Costs
    // one 4-byte read-back and a host wait for the counting kernel (~20 us of a multi-millisecond call).
    if (rc == GPBC_OK) {
        uint8_t *pin = nullptr;
        rc = pinned_staging(st, 4096, &pin);
        // one maximum per window (W <= 22 words in the 256-byte `total` slot), each judged against the mean of a full window; the
        // partial top window is spread over its sub-buckets (k_msm_keys), so the same bound holds for it
        static_assert(sizeof(uint32_t) * 32 <= 256, "per-window maxima live in the `total` slot");
        if (rc == GPBC_OK && hipMemsetAsync(total, 0, 4 * (size_t)p.W, st) != hipSuccess) rc = fail(GPBC_ERR_HIP, "hipMemsetAsync failed");
        if (rc == GPBC_OK) { k_msm_max_count<<<(unsigned)((p.M + SCAN_BLOCK - 1) / SCAN_BLOCK), SCAN_BLOCK, 0, st>>>(counts, p.M, p.c, total); step("k_msm_max_count"); }
        if (rc == GPBC_OK && (hipMemcpyAsync(pin, total, 4 * (size_t)p.W, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) rc = fail(GPBC_ERR_HIP, "read-back of the longest buckets failed");
        if (rc == GPBC_OK) {
            uint32_t longest = 0;
            for (int w = 0; w < p.W; w++) if (((const uint32_t *)pin)[w] > longest) longest = ((const uint32_t *)pin)[w];
            if (longest > MSM_MAX_BUCKET_BASE + 8 * (uint32_t)(n >> p.c)) { g_msm_skewed.fetch_add(1); return MSM_SKEWED; }
            g_msm_bucket_runs.fetch_add(1);
        }
    }
    if (rc == GPBC_OK) { k_scan_tiles<<<(unsigned)n_tiles, SCAN_BLOCK, 0, st>>>(counts, offsets, tiles, p.M); step("k_scan_tiles"); }
    if (rc == GPBC_OK) { k_scan_tops<<<1, SCAN_BLOCK, 0, st>>>(tiles, n_tiles, total); step("k_scan_tops"); }
    if (rc == GPBC_OK) { k_scan_add<<<(unsigned)n_tiles, SCAN_BLOCK, 0, st>>>(offsets, tiles, p.M, total); step("k_scan_add"); }
    if (rc == GPBC_OK && hipMemsetAsync(counts, 0, p.M * 4, st) != hipSuccess) rc = fail(GPBC_ERR_HIP, "hipMemsetAsync failed");
    if (rc == GPBC_OK) { k_msm_keys<F, true><<<grid_for(n), BLOCK, 0, st>>>(d_bases, d_scalars, n, p.c, p.W, counts, offsets, idx); step(G2 ? "k_msm_scatter_g2" : "k_msm_scatter_g1"); }
    if (rc == GPBC_OK) { k_msm_bases<F><<<grid_for(n), BLOCK, 0, st>>>(d_bases, n, base_rows); step(G2 ? "k_msm_bases_g2" : "k_msm_bases_g1"); }
    // buckets by size (the tile-sum scratch of the first scan is free again: hist_tiles <= n_tiles of the key scan for c = 16; a
    // separate region keeps it simple)
    if (rc == GPBC_OK) { k_size_hist<<<(unsigned)p.n_size_blocks, SCAN_BLOCK, 0, st>>>(offsets, p.M, p.c, hist, p.n_size_blocks); step("k_size_hist"); }
    if (rc == GPBC_OK) { k_scan_tiles<<<(unsigned)hist_tiles, SCAN_BLOCK, 0, st>>>(hist, hist_scanned, tiles, hist_len); step("k_scan_tiles"); }
    if (rc == GPBC_OK) { k_scan_tops<<<1, SCAN_BLOCK, 0, st>>>(tiles, hist_tiles, total); step("k_scan_tops"); }
    if (rc == GPBC_OK) { k_scan_add<<<(unsigned)hist_tiles, SCAN_BLOCK, 0, st>>>(hist_scanned, tiles, hist_len, total); step("k_scan_add"); }
    if (rc == GPBC_OK) { k_size_scatter<<<(unsigned)p.n_size_blocks, SCAN_BLOCK, 0, st>>>(offsets, p.M, p.c, hist_scanned, p.n_size_blocks, perm); step("k_size_scatter"); }
    if (rc == GPBC_OK) { k_msm_buckets<F><<<grid_for(p.M), BLOCK, 0, st>>>(base_rows, offsets, idx, perm, p.c, p.M, buckets); step(G2 ? "k_msm_buckets_g2" : "k_msm_buckets_g1"); }
    if (rc == GPBC_OK) { k_msm_groups<F><<<grid_for(p.n_groups), BLOCK, 0, st>>>(buckets, p.c, p.W, p.n_groups, groups); step(G2 ? "k_msm_groups_g2" : "k_msm_groups_g1"); }
    // tree over the groups of every window: rows are window-major, so segments of `fan` consecutive rows never cross a window
    int32_t *src = groups, *dst = tmp;
    for (size_t per_window = gpw; rc == GPBC_OK && per_window > 1;) {
        const size_t fan = per_window >= MSM_FAN ? MSM_FAN : per_window, next = per_window / fan;
        k_msm_rows_sum<F><<<grid_for((size_t)p.W * next), BLOCK, 0, st>>>(src, (size_t)p.W * next, fan, dst);
        step(G2 ? "k_msm_rows_sum_g2" : "k_msm_rows_sum_g1");
        int32_t *t = src; src = dst; dst = t;
        per_window = next;
    }
    if (rc == GPBC_OK) { k_msm_finish<F><<<1, BLOCK, 0, st>>>(src, p.c, p.W, d_out); step(G2 ? "k_msm_finish_g2" : "k_msm_finish_g1"); }
    return rc;
}

// sum_i [s_i] P_i over n >= MSM_MIN_TERMS terms in device memory, result (one affine point, gnark layout) at d_out; MSM_SKEWED
// (nothing written) when one bucket would be too long a chain — the caller then multiplies term by term
int msm_dev(bool g2, const void *d_bases, const void *d_scalars, size_t n, void *d_out, hipStream_t st) {
    TRY(bind_device());
    return g2 ? msm_run<F2>((const uint8_t *)d_bases, (const uint8_t *)d_scalars, n, (uint8_t *)d_out, st)
              : msm_run<Fe>((const uint8_t *)d_bases, (const uint8_t *)d_scalars, n, (uint8_t *)d_out, st);
}
extern "C" int gpbc_msm_stats(uint64_t *bucket_runs_out, uint64_t *skewed_fallbacks_out) {
    if (!bucket_runs_out || !skewed_fallbacks_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    *bucket_runs_out = g_msm_bucket_runs.load(); *skewed_fallbacks_out = g_msm_skewed.load();
    return GPBC_OK;
}
