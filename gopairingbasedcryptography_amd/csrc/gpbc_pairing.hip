// libgpbc_bn254.so, unit 2 of 4: Miller loop (two phases, single pairs / shared-squaring chunks / fixed Q), final
// exponentiation, segment products and the GT kernels, with their C-ABI entries (include/gpbc_bn254.h).  gfx950 only.
#include "gpbc_common.hpp"
#include "wide29.hip.hpp"
#include "pairing29.hip.hpp"
#include "pairing29_pair.hip.hpp"

static_assert(LINE_BYTES_PER_PAIR == (size_t)MILLER_LINES * LINE_WORDS * sizeof(int32_t), "lines workspace row size");

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

// The accumulator's line stream through LDS, loaded by the DMA path (global_load_lds_dword: HBM -> LDS without a destination
// register): line s + 1 is requested as soon as line s has been read out of the stage, so its HBM latency passes behind the sparse
// product and the next squaring instead of in front of the product (the compiler cannot hoist 54 register loads above six leaf
// calls; an LDS stage costs no registers).  Each lane of a pair requests half of the 54 words — the even lane words 0..26, the odd
// lane 27..53, one 256-byte LDS row per request — and both read all of them back.  (The register-load form it replaced measured
// 55.4 against 54.6 ms, profiles/r02_variant_line_dma.txt; it lives in the history before round 3.)
__shared__ int32_t g_line_stage[27][BLOCK];
__device__ __forceinline__ void line_request(const int32_t *__restrict__ buf, size_t stride, size_t lane, int step, bool odd) {
    const int32_t *b = buf + ((size_t)step * LINE_WORDS + (odd ? 27 : 0)) * stride + lane;
#pragma unroll
    for (int r = 0; r < 27; r++) __builtin_amdgcn_global_load_lds(b + (size_t)r * stride, &g_line_stage[r][0], 4, 0, 0);
}
__device__ __forceinline__ LineS line_from_stage() {
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0): the requested words have landed
    const unsigned e = threadIdx.x & ~1u;
    LineS l;
    Fe *fe[6] = {&l.c0.a0, &l.c0.a1, &l.c3.a0, &l.c3.a1, &l.c4.a0, &l.c4.a1};
#pragma unroll
    for (int w = 0; w < LINE_WORDS; w++) fe[w / NL]->v[w % NL] = g_line_stage[w < 27 ? w : w - 27][e + (w < 27 ? 0 : 1)];
    __builtin_amdgcn_s_waitcnt(0xC07F);                          // lgkmcnt(0): read out before the next request may overwrite the stage
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
// value, halves swapped by DPP — tower29_pair.hip.hpp), so a batch of n pairings is a grid of 2n lanes.
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
        line_request(lines, stride, i, 0, x.odd);
        h = miller_accumulate_pair(x, [&]() -> LineS {
            LineS l = line_from_stage();
            if (++step < MILLER_LINES) line_request(lines, stride, i, step, x.odd);
            return l;
        });
    }
    f6_store(f_out + i * GPBC_GT_BYTES + (x.odd ? 192 : 0), h);
}

// Small batches: both phases in ONE launch, running CONCURRENTLY.  A single pairing is a chain of ~3 M dependent instructions
// (0.7 M line phase, 1.0 M accumulator, 1.3 M final exponentiation) and a lone wave issues one every 6-8 cycles, so a call of a few
// pairings is pure latency; the accumulator only ever needs line s after it has consumed line s - 1, so it can start as soon as the
// first line exists.  Blocks [0, n_line_blocks) are producers (k_miller_lines' body: after every line a lane publishes its count
// with an agent-scope release store), the blocks behind them consumers (k_miller_accumulate's body: before line s a lane pair waits
// until the count of its pairing exceeds s, with acquire loads).  Producer blocks have the lower indices and the grid is far
// smaller than the chip (n <= PIPELINED_MAX_PAIRS: at most 256 + 512 of the 2048 resident waves; at 32 768 pairs the waiting consumers cost the producers more than the overlap gains, profiles/r02_pipelined_sizes.txt), so every producer is resident
// before any consumer waits.  The wait is bounded all the same: a lane pair that has spun PIPELINED_SPIN_LIMIT times computes its
// own lines (same values into the same slots) and goes on — every wave finishes whatever the dispatch order.
constexpr size_t PIPELINED_MAX_PAIRS = 16384;
constexpr uint32_t PIPELINED_SPIN_LIMIT = 1u << 22;
GPBC_KERNEL k_miller_pipelined(const uint8_t *__restrict__ P, const uint8_t *__restrict__ Q, int32_t *lines, uint32_t *progress,
                               uint8_t *__restrict__ f_out, size_t n, size_t stride, unsigned n_line_blocks, uint32_t spin_limit) {
    if (blockIdx.x < n_line_blocks) {
        size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
        if (i >= n) return;
        const uint8_t *p = P + i * GPBC_G1_BYTES, *q = Q + i * GPBC_G2_BYTES;
        if (g1_bytes_inf(p) || g2_bytes_inf(q)) return;      // the consumer checks for itself and does not wait
        G1A a{fe_load(p), fe_load(p + 32)};
        G2A b{f2_load(q), f2_load(q + 64)};
        int step = 0;
        miller_lines(a, b, [&](const LineS &l) {
            line_store(lines, stride, i, step++, l);
            __hip_atomic_store(progress + i, (uint32_t)step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        });
        return;
    }
    size_t lane = (size_t)(blockIdx.x - n_line_blocks) * BLOCK + threadIdx.x;
    size_t i = lane >> 1;
    if (i >= n) return;
    PairDpp x{(bool)(lane & 1)};
    const uint8_t *p = P + i * GPBC_G1_BYTES, *q = Q + i * GPBC_G2_BYTES;
    F6 h;
    if (g1_bytes_inf(p) || g2_bytes_inf(q)) h = f12p_one(x);
    else {
        int step = 0;
        bool own_lines = false;
        h = miller_accumulate_pair(x, [&]() -> LineS {
            if (!own_lines) {
                uint32_t spins = 0;
                while (__hip_atomic_load(progress + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) <= (uint32_t)step) {
                    __builtin_amdgcn_s_sleep(16);
                    if (++spins > spin_limit) { own_lines = true; break; }
                }
                if (own_lines) {                              // never seen in practice; both lanes of the pair write the same values
                    G1A a{fe_load(p), fe_load(p + 32)};
                    G2A b{f2_load(q), f2_load(q + 64)};
                    int s2 = 0;
                    miller_lines(a, b, [&](const LineS &l) { line_store(lines, stride, i, s2++, l); });
                    __threadfence();
                }
            }
            return line_load(lines, stride, i, step++);
        });
    }
    f6_store(f_out + i * GPBC_GT_BYTES + (x.odd ? 192 : 0), h);
}

// Multi-pairing form of the two phases (host entry gpbc_multi_pair / gpbc_pairing_check): the pairs of a segment are cut
// into chunks of at most MULTI_CHUNK pairs, one lane pair accumulates a whole chunk with SHARED squarings
// (miller_accumulate_multi), and the lines workspace is laid out by slot = i * n_chunks + c (pair i of chunk c) so that
// adjacent lane pairs read adjacent words whatever the chunk lengths are.
constexpr int MULTI_CHUNK = 8;
constexpr size_t MULTI_GROUP = 65536;
// Both kernels ECHO the chunk bounds they read from the device copy of the host's table (`seen_lines`, `seen`: two words per
// chunk) — k_segment_product folds the echo into "first pair, pairs consumed" per segment and the host entry compares that with
// ITS table before it reports anything (multi_pair_core: a product over fewer pairs than the caller passed must never come back as
// a result, least of all as "PairingCheck = true").
GPBC_KERNEL k_miller_lines_chunks(const uint8_t *__restrict__ P, const uint8_t *__restrict__ Q, int32_t *__restrict__ lines,
                                  const uint64_t *__restrict__ chunk_off, size_t n_chunks, size_t n_slots, uint64_t *__restrict__ seen_lines) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n_slots) return;
    const size_t i = t / n_chunks, c = t % n_chunks;
    const uint64_t c_lo = chunk_off[c], c_hi = chunk_off[c + 1];
    if (i == 0) { seen_lines[2 * c] = c_lo; seen_lines[2 * c + 1] = c_hi; }
    const uint64_t pair = c_lo + i;
    if (pair >= c_hi) return;                                  // slot beyond this chunk's length
    const uint8_t *p = P + pair * GPBC_G1_BYTES, *q = Q + pair * GPBC_G2_BYTES;
    if (g1_bytes_inf(p) || g2_bytes_inf(q)) return;
    G1A a{fe_load(p), fe_load(p + 32)};
    G2A b{f2_load(q), f2_load(q + 64)};
    int step = 0;
    miller_lines(a, b, [&](const LineS &l) { line_store(lines, n_slots, t, step++, l); });
}
GPBC_KERNEL k_miller_accumulate_chunks(const uint8_t *__restrict__ P, const uint8_t *__restrict__ Q, const int32_t *__restrict__ lines,
                                       const uint64_t *__restrict__ chunk_off, uint8_t *__restrict__ f_out, size_t n_chunks, size_t n_slots,
                                       const uint64_t *__restrict__ seen_lines, uint64_t *__restrict__ seen) {
    size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t c = lane >> 1;
    if (c >= n_chunks) return;
    PairDpp x{(bool)(lane & 1)};
    const uint64_t lo = chunk_off[c], hi = chunk_off[c + 1];
    if (!x.odd) {                                              // the bounds THIS kernel used; poisoned if the line phase used others
        const bool same = seen_lines[2 * c] == lo && seen_lines[2 * c + 1] == hi && hi >= lo && hi - lo <= (uint64_t)MULTI_CHUNK;
        seen[2 * c] = same ? lo : ~0ull;
        seen[2 * c + 1] = same ? hi : 0;
    }
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
// are computed once (k_q_lines: 88 x 54 int32 per Q_i, laid out [line][word][i]) and scaled to c0 = 1 (k_q_lines_scale), the
// P's become (x/y, 1/y) in internal form once (k_g1_line_point), and the accumulator kernel evaluates a line at its own P (two
// Fp x Fp2 products, one per lane of the pair) right before the sparse multiplication.  Lane pairs are numbered chunk-major (t = c * k + j): the 32 lane pairs of a wave then work on
// the same Q_i at the same time, so their line loads are one broadcast transaction.
// A chunk here may be much longer than MULTI_CHUNK: the lines live once per Q_i, not once per (pair, slot), so the only cost of a
// long chunk is fewer lane pairs — and every pair beyond the first of a chunk saves its 64 squarings.
constexpr int FIXED_Q_CHUNK = 64;
GPBC_KERNEL k_q_lines(const uint8_t *__restrict__ Q, int32_t *__restrict__ qlines, size_t m) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= m) return;
    const uint8_t *q = Q + i * GPBC_G2_BYTES;
    if (g2_bytes_inf(q)) return;
    G2A b{f2_load(q), f2_load(q + 64)};
    int step = 0;
    miller_lines_raw(b, [&](const LineE &l) { line_store(qlines, m, i, step++, LineS{l.r0, l.r1, l.r2}); });
}
// Scaling of the table to c0 = 1: line (r0, r1, r2) -> (r1 / r0, r2 / r0), one lane per (Q_i, line).  The factor 1 / r0 lies in
// Fp2, a proper subfield of Fp12: the final exponentiation removes it, so the GT values are unchanged bit for bit while every
// sparse product in the accumulator loses its three F2 products by c0.  (r0 = 0 only for points outside the order-r subgroup —
// 2-torsion, or a chord through T = +-Q; the inverse of zero is zero here and the line degenerates to one: no fault, and bn254.Pair
// promises nothing for such inputs either.)  Compact layout [line][36 words][i].
constexpr int LINE34_WORDS = 4 * NL;
GPBC_KERNEL k_q_lines_scale(const int32_t *__restrict__ qlines, int32_t *__restrict__ q34, size_t m) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= m * MILLER_LINES) return;
    const size_t i = t % m;
    const int li = (int)(t / m);
    LineS r = line_load(qlines, m, i, li);
    F2 inv = f2_inv(f2_norm(r.c0));                              // the raw coefficients come un-normalised from the point formulas
    F2 c3 = f2_mul(f2_norm(r.c3), inv), c4 = f2_mul(f2_norm(r.c4), inv);
    int32_t *o = q34 + (size_t)li * LINE34_WORDS * m + i;
    const Fe *fe[4] = {&c3.a0, &c3.a1, &c4.a0, &c4.a1};
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
        for (int w = 0; w < NL; w++) o[(size_t)(e * NL + w) * m] = fe[e]->v[w];
}
// The evaluation point of such a line: (x / y, 1 / y) in internal limbs (the division by yP is the Fp factor that makes c0 = 1).
// 20 int32 per point: x/y (9), 1/y (9), infinity flag, pad.  One lane converts LINE_POINT_GROUP consecutive points with ONE Fp
// inversion (fe_batch_inverse); a zero y — infinity, or a point outside the group — does not spoil its neighbours.
constexpr int LINE_POINT_GROUP = 8;
GPBC_KERNEL_G1 k_g1_line_point(const uint8_t *__restrict__ P, int32_t *__restrict__ out, size_t n) {
    const size_t base = ((size_t)blockIdx.x * BLOCK + threadIdx.x) * LINE_POINT_GROUP;
    if (base >= n) return;
    const int cnt = n - base < (size_t)LINE_POINT_GROUP ? (int)(n - base) : LINE_POINT_GROUP;
    fe_batch_inverse<LINE_POINT_GROUP>(cnt, [&](int j) { return fe_load(P + (base + j) * GPBC_G1_BYTES + 32); }, [&](int j, const Fe &yinv) {
        const uint8_t *pb = P + (base + j) * GPBC_G1_BYTES;
        Fe xoy = fe_mul(fe_load(pb), yinv);
        int32_t *o = out + (base + j) * 20;
#pragma unroll
        for (int w = 0; w < NL; w++) { o[w] = xoy.v[w]; o[NL + w] = yinv.v[w]; }
        o[18] = g1_bytes_inf(pb) ? 1 : 0;
        o[19] = 0;
    });
}
GPBC_KERNEL k_miller_accumulate_fixed_q(const int32_t *__restrict__ Pint, const uint8_t *__restrict__ Q, const int32_t *__restrict__ q34,
                                        uint8_t *__restrict__ f_out, size_t m, size_t k, size_t L, size_t n_c) {
    size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t t = lane >> 1;
    if (t >= n_c * k) return;
    PairDpp x{(bool)(lane & 1)};
    const size_t c = t / k, j = t % k;
    const size_t lo = c * L, hi = (c + 1) * L < m ? (c + 1) * L : m;
    int vi[FIXED_Q_CHUNK], n = 0;
    for (size_t i = lo; i < hi; i++)
        if (!Pint[(j * m + i) * 20 + 18] && !g2_bytes_inf(Q + i * GPBC_G2_BYTES)) vi[n++] = (int)i;
    F6 h;
    if (n == 0) h = f12p_one(x);
    else h = miller_accumulate_multi_34(x, n, [&](int p, int li) -> Line34 {
        const size_t i = (size_t)vi[p];
        // the line at P: c3 = (r1 / r0) (xP / yP), c4 = (r2 / r0) / yP.  Both lanes of the pair need both; each loads ONE coefficient
        // of the table row (the same address for all even / all odd lanes of the wave) and ONE coordinate, computes its Fp x Fp2
        // product (even lane c3, odd lane c4), and they swap the products.  (The LDS stage of k_miller_accumulate was tried here as
        // well and is 1 % SLOWER: these loads hit L2 — the table row is shared by the whole grid — and have little latency to hide.)
        const int32_t *lb = q34 + ((size_t)li * LINE34_WORDS + (x.odd ? 2 * NL : 0)) * m + i;
        const int32_t *pp = Pint + (j * m + i) * 20 + (x.odd ? NL : 0);                 // odd: 1 / yP, even: xP / yP
        F2 coef;
        Fe pc;
#pragma unroll
        for (int w = 0; w < NL; w++) { coef.a0.v[w] = lb[(size_t)w * m]; coef.a1.v[w] = lb[(size_t)(NL + w) * m]; pc.v[w] = pp[w]; }
        const F2 mine = f2_mul_fe(coef, pc), other = x.swap(mine);
        return Line34{f2_sel(x.odd, other, mine), f2_sel(x.odd, mine, other)};
    });
    f6_store(f_out + (j * n_c + c) * GPBC_GT_BYTES + (x.odd ? 192 : 0), h);     // segment-major: chunks of a segment are adjacent
}

// ---- the latency form (csrc/wide29.hip.hpp): ONE pairing per workgroup, Fp12 values as F2 slots in LDS (96 bytes each), the lanes
// working on the F2 products inside the pairing.
typedef int32_t w128 __attribute__((ext_vector_type(4)));
// one slot = 96 bytes: a0 in three 128-bit words (nine limbs + padding), a1 in the next three — either half can be read or written alone
__device__ __forceinline__ Fe wide_half_load(const w128 *mem, int slot, int h) {
    const w128 t0 = mem[slot * 6 + 3 * h], t1 = mem[slot * 6 + 3 * h + 1], t2 = mem[slot * 6 + 3 * h + 2];
    Fe r;
    r.v[0] = t0.x; r.v[1] = t0.y; r.v[2] = t0.z; r.v[3] = t0.w; r.v[4] = t1.x; r.v[5] = t1.y; r.v[6] = t1.z; r.v[7] = t1.w; r.v[8] = t2.x;
    return r;
}
__device__ __forceinline__ void wide_half_store(w128 *mem, int slot, int h, const Fe &v) {
    mem[slot * 6 + 3 * h] = w128{v.v[0], v.v[1], v.v[2], v.v[3]};
    mem[slot * 6 + 3 * h + 1] = w128{v.v[4], v.v[5], v.v[6], v.v[7]};
    mem[slot * 6 + 3 * h + 2] = w128{v.v[8], 0, 0, 0};
}
__device__ __forceinline__ F2 wide_slot_load(const w128 *mem, int slot) { return F2{wide_half_load(mem, slot, 0), wide_half_load(mem, slot, 1)}; }
__device__ __forceinline__ void wide_slot_store(w128 *mem, int slot, const F2 &v) { wide_half_store(mem, slot, 0, v.a0); wide_half_store(mem, slot, 1, v.a1); }
// The phases of ONE WAVE: its LDS instructions execute in program order, so a phase's stores are visible to the next phase's loads
// of any lane of the same wave; the fence / wave barrier only keeps the compiler from moving LDS accesses across the phase boundary.
// A workgroup of the exponentiation / product kernels may bring a HELPER wave (128 threads instead of 64): it sits out every phase
// except the product phase of wide_mul (run2), where the 72 halves of the 36 F2 products go to 72 lanes of the two waves instead of
// 36 whole products to one — a lone wave issues one MAD per ~9 cycles, so that phase is halved.  Calls of more than 1 024 elements
// come without it (two waves per element would need a second round of the chip).
struct WideLds {
    w128 *mem;
    int lane;                                                // lane number inside the wave
    int helper = 0, n_waves = 1;                             // helper: this wave only takes part in run2 phases
    __device__ __forceinline__ int waves() const { return n_waves; }
    // limb-parallel phases (wide29.hip.hpp: wide_cyclo_out_limbs): lane = component * NL + limb
    static constexpr bool LIMB_PARALLEL = true;
    __device__ __forceinline__ int comp() const { return lane / NL; }
    __device__ __forceinline__ int limb() const { return lane % NL; }
    __device__ __forceinline__ int32_t p_limb() const {      // this lane's limb of p
        constexpr int32_t P[NL] = F29_P;
        const int i = limb();
        int32_t r = P[0];
#pragma unroll
        for (int j = 1; j < NL; j++) r = i == j ? P[j] : r;
        return r;
    }
    __device__ __forceinline__ int32_t ldw(int slot, int h, int i) const { return reinterpret_cast<const int32_t *>(mem)[(slot * 6 + 3 * h) * 4 + i]; }
    __device__ __forceinline__ void stw(int slot, int h, int i, int32_t v) const { reinterpret_cast<int32_t *>(mem)[(slot * 6 + 3 * h) * 4 + i] = v; }
    __device__ __forceinline__ static int32_t below(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true); }   // the value of lane - 1 (wave_shr:1)
    __device__ __forceinline__ int32_t from_top(int32_t v) const { return __builtin_amdgcn_ds_bpermute((comp() * NL + NL - 1) * 4, v); }   // ... of the lane that holds the component's top limb
    __device__ __forceinline__ F2 ld(int slot) const { return wide_slot_load(mem, slot); }
    __device__ __forceinline__ void st(int slot, const F2 &v) const { wide_slot_store(mem, slot, v); }
    __device__ __forceinline__ Fe ldh(int slot, int h) const { return wide_half_load(mem, slot, h); }
    __device__ __forceinline__ void sth(int slot, int h, const Fe &v) const { wide_half_store(mem, slot, h, v); }
    template <class B> __device__ __forceinline__ void run(int n, B &&body) const {
        if (!helper && lane < n) body(lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // a phase of BOTH waves (n <= 128 lanes; only where waves() == 2): workgroup barriers on both sides — every wave of the workgroup
    // walks the same (uniform) control flow, so both reach them
    // one-limb-per-lane operations of the linear phases (wide29.hip.hpp "linear phases"): lane = component * NL + limb
    struct LimbOps {
        using V = int32_t;
        const WideLds &m;
        int i;
        int32_t low, carry_in, p_i;
        __device__ __forceinline__ explicit LimbOps(const WideLds &w) : m(w), i(w.limb()), low(w.limb() == NL - 1 ? -1 : LMASK), carry_in(w.limb() == 0 ? 0 : -1), p_i(w.p_limb()) {}
        __device__ __forceinline__ V ld(int slot, int h) const { return m.ldw(slot, h, i); }
        __device__ __forceinline__ void st(int slot, int h, V v) const { m.stw(slot, h, i, v); }
        __device__ __forceinline__ static V add(V a, V b) { return a + b; }
        __device__ __forceinline__ static V sub(V a, V b) { return a - b; }
        __device__ __forceinline__ static V dbl(V a) { return a + a; }
        __device__ __forceinline__ static V neg(V a) { return -a; }
        __device__ __forceinline__ static V sel(bool c, V a, V b) { return c ? a : b; }
        __device__ __forceinline__ V norm(V v) const { return (v & low) + ((below(v) >> LB) & carry_in); }                  // fe_norm
        __device__ __forceinline__ V halve(V v) const {                                                                    // fe_halve
            const int32_t odd = -(__builtin_amdgcn_ds_bpermute(m.comp() * NL * 4, v) & 1);                                   // parity of the value = parity of limb 0
            const int32_t t = v + (p_i & odd);
            const int32_t up = __builtin_amdgcn_update_dpp(0, t, 0x130, 0xF, 0xF, true);                                     // the limb above (wave_shl:1)
            return (t >> 1) + (i == NL - 1 ? 0 : (up & 1) << (LB - 1));
        }
    };
    template <class B> __device__ __forceinline__ void limbs(int n_comp, B &&body) const {
        run(n_comp * NL, [&](int) { LimbOps o(*this); body(o, comp()); });
    }
    __device__ __forceinline__ void sync_waves() const { if (n_waves == 2) __syncthreads(); }    // before control flow that depends on LDS contents
    template <class B> __device__ __forceinline__ void run2(int n, B &&body) const {
        __syncthreads();
        const int l = helper * 64 + lane;
        if (l < n) body(l);
        __syncthreads();
    }
};
#define GPBC_KERNEL_WIDE __global__ void __launch_bounds__(128, GPBC_WAVES_PER_SIMD)
constexpr size_t WIDE_HELPER_MAX = 1024;                     // elements per call up to which the helper wave comes along
static inline unsigned wide_threads(size_t n) { return n <= WIDE_HELPER_MAX ? 128u : 64u; }
// Two waves per pairing: wave 0 walks the G2 point and leaves each of the 88 lines in an LDS ring (21 KB), wave 1 runs the Fp12
// accumulator and takes line j as soon as the ring's counter says it is there — the two chains (~235 k and ~180 k instructions) run
// side by side on two SIMDs instead of one after the other.  Both waves belong to one workgroup, so both are resident: the consumer's
// wait always ends.
constexpr int WIDE_MILLER_THREADS = 128;
// q_mod > 0: pair i takes Q[i % q_mod] (one G2 list shared by every segment: the small calls of the fixed-Q entry)
__global__ void __launch_bounds__(WIDE_MILLER_THREADS, 2) k_miller_wide(const uint8_t *__restrict__ P, const uint8_t *__restrict__ Q, uint8_t *__restrict__ f_out, size_t n, size_t q_mod) {
    __shared__ w128 w_mem[W_SLOTS * 6];
    __shared__ w128 w_ring[MILLER_LINES * 3 * 6];
    __shared__ int w_lines_ready;
    const size_t i = blockIdx.x;
    if (i >= n) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const WideLds m{w_mem, lane};
    const uint8_t *p = P + i * GPBC_G1_BYTES, *q = Q + (q_mod ? i % q_mod : i) * GPBC_G2_BYTES;
    if (g1_bytes_inf(p) || g2_bytes_inf(q)) {                // the same for the whole workgroup
        if (wave == 1 && lane < 6) f2_store(f_out + i * GPBC_GT_BYTES + 64 * lane, f2_sel(lane == 0, f2_one(), f2_zero()));
        return;
    }
    if (threadIdx.x == 0) w_lines_ready = 0;
    __syncthreads();
    if (wave == 0) {
        const G1A a{fe_load(p), fe_load(p + 32)};
        const G2A b{f2_load(q), f2_load(q + 64)};
        wide_miller_lines(m, a, b, [&](int j) {
            m.run(3, [&](int t) { wide_slot_store(w_ring, 3 * j + t, m.ld(W_L0 + t)); });
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(&w_lines_ready, j + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        });
    } else {
        wide_miller_accumulate(m, wv(0), [&](int j) {
            while (__hip_atomic_load(&w_lines_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= j) __builtin_amdgcn_s_sleep(2);
            m.run(3, [&](int t) { m.st(W_CL + t, wide_slot_load(w_ring, 3 * j + t)); });
        });
        m.run(6, [&](int k) { f2_store(f_out + i * GPBC_GT_BYTES + 64 * k, m.ld(k)); });
    }
}
// The fixed-Q form's line table (k_q_lines) by the latency form's G2 walk, one Q_i per wavefront (lists of up to the latency limit: a key's 513 points on 513 lone
// lanes took longer than everything else in a call of a few ciphertexts).  wide_miller_lines evaluates its lines at P; at P = (1, 1)
// they are the raw coefficients — up to the factor in Fp2 that k_q_lines_scale divides out anyway.
GPBC_KERNEL k_q_lines_wide(const uint8_t *__restrict__ Q, int32_t *__restrict__ qlines, size_t m_pts) {
    __shared__ w128 w_mem[W_SLOTS * 6];
    const size_t i = blockIdx.x;
    if (i >= m_pts) return;
    const uint8_t *q = Q + i * GPBC_G2_BYTES;
    if (g2_bytes_inf(q)) return;
    const WideLds m{w_mem, (int)threadIdx.x};
    const G1A a{fe_one(), fe_one()};
    const G2A b{f2_load(q), f2_load(q + 64)};
    wide_miller_lines(m, a, b, [&](int j) {
        m.run(3, [&](int t) {
            const F2 c = m.ld(W_L0 + t);
            int32_t *o = qlines + (size_t)j * LINE_WORDS * m_pts + i;
#pragma unroll
            for (int w = 0; w < NL; w++) { o[(size_t)((2 * t) * NL + w) * m_pts] = c.a0.v[w]; o[(size_t)((2 * t + 1) * NL + w) * m_pts] = c.a1.v[w]; }
        });
    });
}
GPBC_KERNEL_WIDE k_final_exp_wide(const uint8_t *f_in, uint8_t *gt_out, size_t n) {
    __shared__ w128 w_mem[W_SLOTS * 6];
    const size_t i = blockIdx.x;
    if (i >= n) return;
    const WideLds m{w_mem, (int)(threadIdx.x & 63), (int)(threadIdx.x >> 6), (int)(blockDim.x >> 6)};
    m.run(6, [&](int k) { m.st(k, f2_load(f_in + i * GPBC_GT_BYTES + 64 * k)); });
    wide_final_exp(m);
    m.run(6, [&](int k) { f2_store(gt_out + i * GPBC_GT_BYTES + 64 * k, m.ld(k)); });
}

// Long segments of a latency call (one BSW07 ciphertext is 513 pairs, a BB04 identity 257): `fold` wavefronts per segment each
// multiply one residue class of the segment's Miller values — lo + s, lo + s + fold, ... — and leave the product in place at lo + s,
// so that the wavefront of k_segment_final_exp_wide multiplies `fold` values instead of the whole run.  Same clamping of the table.
GPBC_KERNEL_WIDE k_segment_fold_wide(uint8_t *f, const uint64_t *__restrict__ seg_off, size_t uniform_len, size_t k, size_t n_vals, unsigned fold) {
    __shared__ w128 w_mem[W_SLOTS * 6];
    const size_t j = blockIdx.x / fold;
    const uint64_t s = blockIdx.x % fold;
    if (j >= k) return;
    uint64_t lo = seg_off ? seg_off[j] : j * uniform_len, hi = seg_off ? seg_off[j + 1] : (j + 1) * uniform_len;   // no table: equal runs
    if (hi > n_vals) hi = n_vals;
    if (lo > hi) lo = hi;
    if (hi - lo <= s + fold) return;                        // this class holds at most one value: it stays where it is (the same for the whole workgroup)
    const WideLds m{w_mem, (int)(threadIdx.x & 63), (int)(threadIdx.x >> 6), (int)(blockDim.x >> 6)};
    m.run(6, [&](int c) { m.st(c, f2_load(f + (lo + s) * GPBC_GT_BYTES + 64 * c)); });
    for (uint64_t i = lo + s + fold; i < hi; i += fold) {
        m.run(6, [&](int c) { m.st(wv(1) + c, f2_load(f + i * GPBC_GT_BYTES + 64 * c)); });
        wide_mul(m, wv(0), wv(0), wv(1));
    }
    m.run(6, [&](int c) { f2_store(f + (lo + s) * GPBC_GT_BYTES + 64 * c, m.ld(c)); });
}

// The product of a segment's Miller values and its final exponentiation in one launch, one segment per wavefront: what
// k_segment_product (one lane, serial Fp12 products) + k_final_exp_wide do for the small calls of the latency path.  Same clamping
// of the table and the same echo (first value, values consumed) as k_segment_product.
// `fold` > 0: k_segment_fold_wide ran first with that many residue classes, so the segment's product is the product of its first
// `fold` values.
// `ok_out` (may be null): 1 where the segment's value is GT's one — PairingCheck's answer, taken from the canonical words the lanes
// are about to store (a later k_gt_is_one would have to read them back, and in a small call they live in host memory).
GPBC_KERNEL_WIDE k_segment_final_exp_wide(const uint8_t *__restrict__ f, const uint64_t *__restrict__ seg_off, size_t uniform_len, uint8_t *__restrict__ out, size_t k, size_t n_vals,
                                     uint64_t *__restrict__ echo, unsigned fold, uint8_t *__restrict__ ok_out) {
    __shared__ w128 w_mem[W_SLOTS * 6];
    const size_t j = blockIdx.x;
    if (j >= k) return;
    const WideLds m{w_mem, (int)(threadIdx.x & 63), (int)(threadIdx.x >> 6), (int)(blockDim.x >> 6)};
    uint64_t lo = seg_off ? seg_off[j] : j * uniform_len, hi = seg_off ? seg_off[j + 1] : (j + 1) * uniform_len;
    if (hi > n_vals) hi = n_vals;
    if (lo > hi) lo = hi;
    const uint64_t end = (fold && hi - lo > fold) ? lo + fold : hi;
    if (lo == hi) m.run(6, [&](int c) { m.st(c, f2_sel(c == 0, f2_one(), f2_zero())); });
    else m.run(6, [&](int c) { m.st(c, f2_load(f + lo * GPBC_GT_BYTES + 64 * c)); });
    for (uint64_t i = lo + 1; i < end; i++) {
        m.run(6, [&](int c) { m.st(wv(1) + c, f2_load(f + i * GPBC_GT_BYTES + 64 * c)); });
        wide_mul(m, wv(0), wv(0), wv(1));
    }
    wide_final_exp(m);
    uint32_t diff = 0;
    if (threadIdx.x < 6) {
        const F2 v = m.ld((int)threadIdx.x);
        uint32_t w[16];
        fe_to_words(w, v.a0); fe_to_words(w + 8, v.a1);
        uint32_t *o = reinterpret_cast<uint32_t *>(out + j * GPBC_GT_BYTES + 64 * threadIdx.x);
        constexpr uint64_t ONE[4] = BN254_FP_ONE;
#pragma unroll
        for (int t = 0; t < 16; t++) { o[t] = w[t]; diff |= w[t] ^ ((threadIdx.x == 0 && t < 8) ? (uint32_t)(ONE[t >> 1] >> ((t & 1) * 32)) : 0u); }
    }
    if (ok_out) {
        const bool one = __builtin_amdgcn_ballot_w64(diff != 0) == 0;       // lanes 6..63 carry diff = 0
        if (threadIdx.x == 0) ok_out[j] = one ? 1 : 0;
    }
    if (echo && threadIdx.x == 0) { echo[2 * j] = lo; echo[2 * j + 1] = hi - lo; }
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
// the number of values so that a malformed table can never read outside the Miller-value buffer).
// `echo` (two words per segment, may be null): what this thread actually consumed — the first pair index and the number of pairs.
// With `seen` null the values are single Miller values and the echo is the table entry as used; with `seen` the values are chunk
// products and the echo is folded from the bounds k_miller_accumulate_chunks recorded per chunk (count ~0 if the chunks of the
// segment were not contiguous).  The host-table entries compare the echo with the table they copied (multi_pair_core).
// Long segments (one bn254.Pair of a few hundred pairs: a BSW07 or BB04 call) would leave that thread a chain of hundreds of Fp12
// products at a lone lane's pace — 21 ms for 513 values, 5 s for 10^5.  k_segment_fold runs first, in passes: `fold` lane pairs per
// segment each multiply one residue class of the segment's first `limit` values (0 = all of them) — lo + s, lo + s + fold, ... — in
// place at lo + s, so that the next pass, and at the end this kernel, see at most `fold` values (segment_fold_passes).  Segments come
// from the table (clamped as below) or, with seg_off null, are the runs of `uniform_len` values of the fixed-Q form.
GPBC_KERNEL k_segment_fold(uint8_t *f, const uint64_t *__restrict__ seg_off, size_t uniform_len, size_t k, size_t n_vals, unsigned fold, uint64_t limit) {
    const size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x, t = lane >> 1;
    const size_t j = t / fold;
    const uint64_t s = t % fold;
    if (j >= k) return;
    PairDpp x{(bool)(lane & 1)};
    uint64_t lo = seg_off ? seg_off[j] : j * uniform_len, hi = seg_off ? seg_off[j + 1] : (j + 1) * uniform_len;
    if (hi > n_vals) hi = n_vals;
    if (lo > hi) lo = hi;
    if (limit && hi - lo > limit) hi = lo + limit;
    if (hi - lo <= s + fold) return;                          // at most one value in this class (both lanes of the pair agree)
    const size_t half = x.odd ? 192 : 0;
    F6 acc = f6_load(f + (lo + s) * GPBC_GT_BYTES + half);
    for (uint64_t i = lo + s + fold; i < hi; i += fold) {
        const F6 v = f6_load(f + i * GPBC_GT_BYTES + half);
        f12p_mul_to(x, acc, acc, v);
    }
    f6_store(f + (lo + s) * GPBC_GT_BYTES + half, acc);
}
GPBC_KERNEL k_segment_product(const uint8_t *__restrict__ f, const uint64_t *__restrict__ seg_off, uint8_t *__restrict__ out, size_t k, size_t n_vals,
                              const uint64_t *__restrict__ seen, uint64_t *__restrict__ echo, uint64_t limit) {
    size_t j = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= k) return;
    F12 acc = f12_one(), t;
    uint64_t lo = seg_off[j], hi = seg_off[j + 1];
    if (hi > n_vals) hi = n_vals;
    if (lo > hi) lo = hi;
    uint64_t first = lo, count = hi - lo;
    if (seen) {
        first = lo < hi ? seen[2 * lo] : 0;
        uint64_t expect = first;
        bool contiguous = true;
        for (uint64_t i = lo; i < hi; i++) { contiguous = contiguous && seen[2 * i] == expect && seen[2 * i + 1] >= expect; expect = seen[2 * i + 1]; }
        count = contiguous ? expect - first : ~0ull;
    }
    const uint64_t end = (limit && hi - lo > limit) ? lo + limit : hi;        // k_segment_fold left the product in the first `limit` values
    for (uint64_t i = lo; i < end; i++) {
        f12_load(t, f + i * GPBC_GT_BYTES);
        acc = f12_mul(acc, t);
    }
    f12_store(out + j * GPBC_GT_BYTES, acc);
    if (echo) { echo[2 * j] = first; echo[2 * j + 1] = count; }
}

// the same product over equal runs of n_c values per output (fixed-Q multi-pairing: no table needed)
GPBC_KERNEL k_chunk_product(const uint8_t *__restrict__ f, uint8_t *__restrict__ out, size_t k, size_t n_c, uint64_t limit) {
    size_t j = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= k) return;
    F12 acc = f12_one(), t;
    const size_t end = (limit && n_c > limit) ? j * n_c + limit : (j + 1) * n_c;
    for (size_t i = j * n_c; i < end; i++) {
        f12_load(t, f + i * GPBC_GT_BYTES);
        acc = f12_mul(acc, t);
    }
    f12_store(out + j * GPBC_GT_BYTES, acc);
}

// validation of a device-resident segment table (gpbc_check_segments_dev): bit 0 first entry not zero, bit 1 not monotone,
// bit 2 last entry is not the number of pairs
__global__ void __launch_bounds__(BLOCK) k_check_segments(const uint64_t *__restrict__ seg_off, size_t k, size_t n_pairs, int *__restrict__ flag) {
    size_t j = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j > k) return;
    int bad = 0;
    if (j == 0 && seg_off[0] != 0) bad |= 1;
    if (j < k && seg_off[j + 1] < seg_off[j]) bad |= 2;
    if (j == k && seg_off[k] != (uint64_t)n_pairs) bad |= 4;
    if (bad) atomicOr(flag, bad);
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

// GT.Exp: left-to-right square-and-multiply on a 256-bit plain exponent (k = 0 -> one)
GPBC_KERNEL k_gt_exp(const uint8_t *__restrict__ x, const uint8_t *__restrict__ kk, uint8_t *__restrict__ out, size_t n, int32_t *__restrict__ tabws) {
    size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t i = lane >> 1;                                   // one Fp12 per lane pair
    if (i >= n) return;
    PairDpp px{(bool)(lane & 1)};
    size_t off = i * GPBC_GT_BYTES + (px.odd ? 192 : 0);
    uint32_t k[8];
    load_scalar(k, kk + i * GPBC_SCALAR_BYTES);
    f6_store(out + off, f12p_exp256(px, f6_load(x + off), k, tabws + lane * (size_t)GT_EXP_TAB_DWORDS));
}

// the same for the calls of the latency path: one element per wavefront (wide_exp256)
GPBC_KERNEL_WIDE k_gt_exp_wide(const uint8_t *__restrict__ x, const uint8_t *__restrict__ kk, uint8_t *__restrict__ out, size_t n) {
    __shared__ w128 w_mem[W_SLOTS * 6];
    const size_t i = blockIdx.x;
    if (i >= n) return;
    const WideLds m{w_mem, (int)(threadIdx.x & 63), (int)(threadIdx.x >> 6), (int)(blockDim.x >> 6)};
    uint32_t k[8];
#pragma unroll
    for (int w = 0; w < 8; w++) k[w] = __builtin_amdgcn_readfirstlane(reinterpret_cast<const uint32_t *>(kk + i * GPBC_SCALAR_BYTES)[w]);
    m.run(6, [&](int c) { m.st(wv(1) + c, f2_load(x + i * GPBC_GT_BYTES + 64 * c)); });
    wide_exp256(m, k);
    m.run(6, [&](int c) { f2_store(out + i * GPBC_GT_BYTES + 64 * c, m.ld(c)); });
}

// op 0: a*b   1: a*b^-1   2: a^-1
GPBC_KERNEL k_gt_binary(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint8_t *__restrict__ out, size_t n, int op) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    F12 x, y, z;
    f12_load(x, a + i * GPBC_GT_BYTES);
    if (op == 2) z = f12_inv_gt(x, PairDpp::all);
    else {
        f12_load(y, b + i * GPBC_GT_BYTES);
        if (op == 1) y = f12_inv_gt(y, PairDpp::all);
        z = f12_mul(x, y);
    }
    f12_store(out + i * GPBC_GT_BYTES, z);
}

extern "C" {

// test / measurement knob: 0 = always the two-kernel form, 2 = pipelined with a spin limit of zero (every consumer that finds its
// line missing computes its own: the fallback path, which a healthy run never takes)
static std::atomic<int> g_pipelined{1};
int gpbc_set_pipelined_miller(int on) { g_pipelined.store(on == 2 ? 2 : on ? 1 : 0); return GPBC_OK; }
// Calls of up to this many pairings take the latency form (one pairing per wavefront): up to one wave per SIMD pair of the chip they
// all run side by side at a lone wave's speed, so the whole call costs one pairing's chain (~2 ms instead of ~6).  0 switches it off.
constexpr size_t WIDE_DEFAULT_MAX_PAIRS = 2048;
static std::atomic<size_t> g_wide_max{WIDE_DEFAULT_MAX_PAIRS};
int gpbc_set_latency_path(long max_pairs) {
    // (the wavefront forms launch one workgroup per pairing or per segment chunk: the limit keeps every such grid far inside 2^31)
    if (max_pairs < 0 || max_pairs > 65536) return fail(GPBC_ERR_INVALID_ARG, "max_pairs must be 0 (off) .. 65536");
    g_wide_max.store((size_t)max_pairs);
    return GPBC_OK;
}
int gpbc_miller_loop_dev(const void *dP, const void *dQ, size_t n, void *d_f_out, void *stream) {
    if (!n) return GPBC_OK;
    if (!dP || !dQ || !d_f_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    hipStream_t st = (hipStream_t)stream;
    size_t chunk = n < MILLER_CHUNK ? n : MILLER_CHUNK;
    if (n <= g_wide_max.load()) {
        k_miller_wide<<<(unsigned)n, WIDE_MILLER_THREADS, 0, st>>>((const uint8_t *)dP, (const uint8_t *)dQ, (uint8_t *)d_f_out, n, 0);
        TRY(check_launch("k_miller_wide"));
        profile_mark("k_miller_wide", st);
        return GPBC_OK;
    }
    if (n <= PIPELINED_MAX_PAIRS && g_pipelined.load()) {
        Scratch flags;                                            // level 2: callers may hold levels 0 and 1 on this stream
        TRY(flags.open(st, 2, n * sizeof(uint32_t)));
        uint32_t *progress = flags.take<uint32_t>(n * sizeof(uint32_t));
        std::lock_guard<std::mutex> seq(g_ws_seq_mu);
        int32_t *lines = nullptr;
        TRY(lines_workspace(st, n, &lines));
        HIP_TRY(hipMemsetAsync(progress, 0, n * sizeof(uint32_t), st));
        const unsigned n_line_blocks = grid_for(n);
        k_miller_pipelined<<<n_line_blocks + grid_for(2 * n), BLOCK, 0, st>>>((const uint8_t *)dP, (const uint8_t *)dQ, lines, progress, (uint8_t *)d_f_out, n, n, n_line_blocks,
                                                                                       g_pipelined.load() == 2 ? 0u : PIPELINED_SPIN_LIMIT);
        TRY(check_launch("k_miller_pipelined"));
        profile_mark("k_miller_pipelined", st);
        return GPBC_OK;
    }
    std::lock_guard<std::mutex> seq(g_ws_seq_mu);
    int32_t *lines = nullptr;
    TRY(lines_workspace(st, chunk, &lines));
    for (size_t off = 0; off < n; off += chunk) {
        size_t m = n - off < chunk ? n - off : chunk;
        const uint8_t *p = (const uint8_t *)dP + off * GPBC_G1_BYTES, *q = (const uint8_t *)dQ + off * GPBC_G2_BYTES;
        k_miller_lines<<<grid_for(m), BLOCK, 0, st>>>(p, q, lines, m, chunk);
        TRY(check_launch("k_miller_lines"));
        profile_mark("k_miller_lines", st);
        k_miller_accumulate<<<grid_for(2 * m), BLOCK, 0, st>>>(p, q, lines, (uint8_t *)d_f_out + off * GPBC_GT_BYTES, m, chunk);
        TRY(check_launch("k_miller_accumulate"));
        profile_mark("k_miller_accumulate", st);
    }
    return GPBC_OK;
}
int gpbc_final_exp_dev(const void *d_f, size_t n, void *d_gt_out, void *stream) {
    if (!n) return GPBC_OK;
    if (!d_f || !d_gt_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    // (two rounds of the chip still beat the lane-pair kernel's 2.6 ms floor: 3 072 / 4 096 values 1.38 / 1.75 ms; 6 144: 2.54 — even.
    // The Miller loop's switch-over stays at one round: its wavefront form holds 768 pairings at a time.)
    if (n <= 2 * g_wide_max.load()) {
        k_final_exp_wide<<<(unsigned)n, wide_threads(n), 0, (hipStream_t)stream>>>((const uint8_t *)d_f, (uint8_t *)d_gt_out, n);
        profile_mark("k_final_exp_wide", (hipStream_t)stream);
        return check_launch("k_final_exp_wide");
    }
    k_final_exp<<<grid_for(2 * n), BLOCK, 0, (hipStream_t)stream>>>((const uint8_t *)d_f, (uint8_t *)d_gt_out, n);
    profile_mark("k_final_exp", (hipStream_t)stream);
    return check_launch("k_final_exp");
}
int gpbc_pair_batch_dev(const void *dP, const void *dQ, size_t n, void *d_gt_out, void *stream) {
    if (!n) return fail(GPBC_ERR_INVALID_ARG, "invalid inputs sizes");
    TRY(gpbc_miller_loop_dev(dP, dQ, n, d_gt_out, stream));        // f staged in the output buffer
    return gpbc_final_exp_dev(d_gt_out, n, d_gt_out, stream);      // each lane rewrites its own 384 B
}
size_t gpbc_multi_pair_workspace_bytes(size_t n_pairs, size_t k) { (void)k; return n_pairs * GPBC_GT_BYTES; }
// The passes of k_segment_fold that leave at most 8 values per segment for the one-thread product (`limit_out`; 0 = no pass ran).
// Fold widths 4096 / 512 / 64 / 8, starting at the smallest one whose classes hold about 8 values for an AVERAGE segment — a table in
// device memory is not known to the host, and a skewed one only makes its long segments' chains longer, never wrong.
static int segment_fold_passes(uint8_t *vals, const uint64_t *d_seg_off, size_t uniform_len, size_t k, size_t n_vals, hipStream_t st, uint64_t *limit_out) {
    *limit_out = 0;
    const size_t avg = n_vals / k;
    if (avg <= 8) return GPBC_OK;
    unsigned fold = 8;
    while (fold < 4096 && (size_t)fold * 8 < avg) fold *= 8;
    uint64_t limit = 0;
    for (;; fold /= 8) {
        k_segment_fold<<<grid_for(2 * k * fold), BLOCK, 0, st>>>(vals, d_seg_off, uniform_len, k, n_vals, fold, limit);
        TRY(check_launch("k_segment_fold"));
        profile_mark("k_segment_fold", st);
        limit = fold;
        if (fold == 8) break;
    }
    *limit_out = 8;
    return GPBC_OK;
}
// latency path, after the Miller loop: the values of each segment (table, or equal runs of `uniform_len`) multiplied and exponentiated
static int segments_wide(uint8_t *vals, const uint64_t *d_seg_off, size_t uniform_len, size_t k, size_t n_vals, uint8_t *d_gt_out, uint64_t *d_echo, hipStream_t st, uint8_t *d_ok = nullptr) {
    const unsigned fold = n_vals >= 128 * k ? 16u : n_vals >= 32 * k ? 8u : 0u;
    if (fold) {
        k_segment_fold_wide<<<(unsigned)(k * fold), wide_threads(k * fold), 0, st>>>(vals, d_seg_off, uniform_len, k, n_vals, fold);
        TRY(check_launch("k_segment_fold_wide"));
        profile_mark("k_segment_fold_wide", st);
    }
    k_segment_final_exp_wide<<<(unsigned)k, wide_threads(k), 0, st>>>(vals, d_seg_off, uniform_len, d_gt_out, k, n_vals, d_echo, fold, d_ok);
    TRY(check_launch("k_segment_final_exp_wide"));
    profile_mark("k_segment_final_exp_wide", st);
    return GPBC_OK;
}
static int multi_pair_dev_echo(const void *dP, const void *dQ, const uint64_t *d_seg_off, size_t n_pairs, size_t k,
                               void *d_gt_out, void *d_workspace, size_t workspace_bytes, uint64_t *d_echo, void *stream) {
    if (!k) return fail(GPBC_ERR_INVALID_ARG, "invalid inputs sizes");
    if (!d_seg_off || !d_gt_out || (n_pairs && (!dP || !dQ || !d_workspace))) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    if (workspace_bytes < gpbc_multi_pair_workspace_bytes(n_pairs, k)) return fail(GPBC_ERR_WORKSPACE, "workspace too small");
    TRY(gpbc_miller_loop_dev(dP, dQ, n_pairs, d_workspace, stream));
    if (k <= g_wide_max.load() && n_pairs <= g_wide_max.load()) {              // the latency path: product and exponentiation per wavefront
        // segments of 32 values and more on average: fold them over 8 or 16 wavefronts first (a device-resident table is not
        // known here, so the average decides; a skewed table only makes the choice slower or faster, never wrong)
        return segments_wide((uint8_t *)d_workspace, d_seg_off, 0, k, n_pairs, (uint8_t *)d_gt_out, d_echo, (hipStream_t)stream);
    }
    uint64_t limit = 0;
    TRY(segment_fold_passes((uint8_t *)d_workspace, d_seg_off, 0, k, n_pairs, (hipStream_t)stream, &limit));
    k_segment_product<<<grid_for(k), BLOCK, 0, (hipStream_t)stream>>>((const uint8_t *)d_workspace, d_seg_off, (uint8_t *)d_gt_out, k, n_pairs, nullptr, d_echo, limit);
    TRY(check_launch("k_segment_product"));
    profile_mark("k_segment_product", (hipStream_t)stream);
    return gpbc_final_exp_dev(d_gt_out, k, d_gt_out, stream);
}
int gpbc_multi_pair_dev(const void *dP, const void *dQ, const uint64_t *d_seg_off, size_t n_pairs, size_t k,
                        void *d_gt_out, void *d_workspace, size_t workspace_bytes, void *stream) {
    return multi_pair_dev_echo(dP, dQ, d_seg_off, n_pairs, k, d_gt_out, d_workspace, workspace_bytes, nullptr, stream);
}
int gpbc_check_segments_dev(const uint64_t *d_seg_off, size_t n_pairs, size_t k, void *stream) {
    if (!k) return fail(GPBC_ERR_INVALID_ARG, "invalid inputs sizes");
    if (!d_seg_off) return fail(GPBC_ERR_INVALID_ARG, "null segment table");
    TRY(bind_device());
    hipStream_t st = (hipStream_t)stream;
    DevBuf dFlag;
    TRY(dFlag.alloc(sizeof(int)));
    HIP_TRY(hipMemsetAsync(dFlag.p, 0, sizeof(int), st));
    k_check_segments<<<grid_for(k + 1), BLOCK, 0, st>>>(d_seg_off, k, n_pairs, (int *)dFlag.p);
    TRY(check_launch("k_check_segments"));
    int flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, dFlag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (flag & 1) return fail(GPBC_ERR_INVALID_ARG, "seg_off[0] must be 0");
    if (flag & 2) return fail(GPBC_ERR_INVALID_ARG, "segment table not monotone");
    if (flag & 4) return fail(GPBC_ERR_INVALID_ARG, "seg_off[k] must equal the number of pairs");
    return GPBC_OK;
}
int gpbc_gt_exp_batch_dev(const void *d_x, const void *d_k, size_t n, void *d_out, void *stream) {
    if (!n) return GPBC_OK;
    if (!d_x || !d_k || !d_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    hipStream_t st = (hipStream_t)stream;
    // one round of the chip holds 2 048 wavefronts (1.1 ms per round when full), the lane-pair kernel needs 3.6 ms whatever the size
    // below 65 536: measured 2 048 / 4 096 / 8 192 elements 1.09 / 2.45 / 4.23 ms against 3.6 — two rounds is where the wavefront form stops winning
    if (n <= 2 * g_wide_max.load()) {
        k_gt_exp_wide<<<(unsigned)n, wide_threads(n), 0, st>>>((const uint8_t *)d_x, (const uint8_t *)d_k, (uint8_t *)d_out, n);
        TRY(check_launch("k_gt_exp_wide"));
        profile_mark("k_gt_exp_wide", st);
        return GPBC_OK;
    }
    constexpr size_t CHUNK = 131072;                       // elements per launch: 1 GB of window tables (4 KB per lane)
    const size_t chunk = n < CHUNK ? n : CHUNK;
    std::lock_guard<std::mutex> seq(g_ws_seq_mu);
    int32_t *tabws = nullptr;
    TRY(stream_workspace(st, 2 * chunk * GT_EXP_TAB_DWORDS * sizeof(int32_t), &tabws));
    for (size_t off = 0; off < n; off += chunk) {
        const size_t m = n - off < chunk ? n - off : chunk;
        k_gt_exp<<<grid_for(2 * m), BLOCK, 0, st>>>((const uint8_t *)d_x + off * GPBC_GT_BYTES, (const uint8_t *)d_k + off * GPBC_SCALAR_BYTES,
                                                    (uint8_t *)d_out + off * GPBC_GT_BYTES, m, tabws);
        TRY(check_launch("k_gt_exp"));
        profile_mark("k_gt_exp", st);
    }
    return GPBC_OK;
}
static int gt_binary_dev(int op, const void *a, const void *b, size_t n, void *out, void *stream) {
    if (!n) return GPBC_OK;
    if (!a || (op != 2 && !b) || !out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    k_gt_binary<<<grid_for(n), BLOCK, 0, (hipStream_t)stream>>>((const uint8_t *)a, (const uint8_t *)b, (uint8_t *)out, n, op);
    profile_mark("k_gt_binary", (hipStream_t)stream);
    return check_launch("k_gt_binary");
}
int gpbc_gt_mul_batch_dev(const void *a, const void *b, size_t n, void *o, void *s) { return gt_binary_dev(0, a, b, n, o, s); }
int gpbc_gt_div_batch_dev(const void *a, const void *b, size_t n, void *o, void *s) { return gt_binary_dev(1, a, b, n, o, s); }
int gpbc_gt_inverse_batch_dev(const void *a, size_t n, void *o, void *s) { return gt_binary_dev(2, a, nullptr, n, o, s); }

// Host-pointer entries: [0, n) is split over the bound devices (run_sharded, gpbc_core.hip); each shard uploads, computes and
// downloads on its own device from its own host thread.
constexpr size_t SHARD_MIN_UNITS = 4096;
static int miller_loop_one(const void *P, const void *Q, size_t n, void *f_out) {
    TRY(bind_device());
    DevBuf dP, dQ, dF;
    TRY(dP.upload(P, n * GPBC_G1_BYTES)); TRY(dQ.upload(Q, n * GPBC_G2_BYTES)); TRY(dF.alloc(n * GPBC_GT_BYTES));
    TRY(gpbc_miller_loop_dev(dP.p, dQ.p, n, dF.p, nullptr));
    TRY(sync_default());
    return dF.download(f_out, n * GPBC_GT_BYTES);
}
int gpbc_miller_loop(const void *P, const void *Q, size_t n, void *f_out) {
    if (!n) return GPBC_OK;
    if (!P || !Q || !f_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    return run_sharded(n, SHARD_MIN_UNITS, [=](size_t lo, size_t hi) {
        return miller_loop_one((const uint8_t *)P + lo * GPBC_G1_BYTES, (const uint8_t *)Q + lo * GPBC_G2_BYTES, hi - lo, (uint8_t *)f_out + lo * GPBC_GT_BYTES);
    });
}
static int final_exp_one(const void *f, size_t n, void *gt_out) {
    TRY(bind_device());
    DevBuf dF;
    TRY(dF.upload(f, n * GPBC_GT_BYTES));
    TRY(gpbc_final_exp_dev(dF.p, n, dF.p, nullptr));
    TRY(sync_default());
    return dF.download(gt_out, n * GPBC_GT_BYTES);
}
int gpbc_final_exp(const void *f, size_t n, void *gt_out) {
    if (!n) return GPBC_OK;
    if (!f || !gt_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    return run_sharded(n, SHARD_MIN_UNITS, [=](size_t lo, size_t hi) {
        return final_exp_one((const uint8_t *)f + lo * GPBC_GT_BYTES, hi - lo, (uint8_t *)gt_out + lo * GPBC_GT_BYTES);
    });
}
// ---- small calls (gpbc_common.hpp "Small host-pointer calls"): every waiting Pair / PairingCheck / multi-pairing of the latency
// path in ONE launch pair on a call lane.  The batch is a multi-pairing over the concatenated pairs with the concatenated segment
// table (a pair_batch call contributes one segment per pair); k_miller_wide reads P and Q straight from the lane's pinned block,
// k_segment_final_exp_wide writes the GT values, the PairingCheck flags and the echo of the table straight into it.
static std::atomic<int> g_fault_table{0};
static int verify_echo(const uint64_t *echo, const uint64_t *seg_off, size_t k);
static int small_pairs_run(CallLane &lane, SmallCall *const *calls, size_t nc) {
    size_t N = 0, K = 0;
    for (size_t c = 0; c < nc; c++) { N += calls[c]->units; K += calls[c]->segs; }
    const size_t oP = 0, oQ = oP + Scratch::padded(N * GPBC_G1_BYTES), oSeg = oQ + Scratch::padded(N * GPBC_G2_BYTES), oEcho = oSeg + Scratch::padded((K + 1) * sizeof(uint64_t)),
                 oGt = oEcho + Scratch::padded(2 * K * sizeof(uint64_t)), oOk = oGt + Scratch::padded(K * GPBC_GT_BYTES), total = oOk + Scratch::padded(K);
    TRY(lane.reserve(total, N * GPBC_GT_BYTES));
    std::vector<uint64_t> table(K + 1);                          // what the host says; the device answers with what it consumed
    size_t n0 = 0, k0 = 0;
    for (size_t c = 0; c < nc; c++) {
        const SmallCall &r = *calls[c];
        memcpy(lane.pin + oP + n0 * GPBC_G1_BYTES, r.in[0], r.units * GPBC_G1_BYTES);
        memcpy(lane.pin + oQ + n0 * GPBC_G2_BYTES, r.in[1], r.units * GPBC_G2_BYTES);
        for (size_t j = 0; j < r.segs; j++) table[k0 + j] = n0 + (r.seg ? r.seg[j] : j);
        n0 += r.units; k0 += r.segs;
    }
    table[K] = N;
    uint64_t *pin_seg = (uint64_t *)(lane.pin + oSeg), *h_echo = (uint64_t *)(lane.pin + oEcho);
    memcpy(pin_seg, table.data(), (K + 1) * sizeof(uint64_t));
    if (g_fault_table.exchange(0)) pin_seg[K] = pin_seg[K - 1];  // test knob: the device sees a last segment that is empty
    memset(h_echo, 0xff, 2 * K * sizeof(uint64_t));
    k_miller_wide<<<(unsigned)N, WIDE_MILLER_THREADS, 0, lane.stream>>>(lane.d_pin + oP, lane.d_pin + oQ, lane.dev, N, 0);
    TRY(check_launch("k_miller_wide"));
    profile_mark("k_miller_wide", lane.stream);
    TRY(segments_wide(lane.dev, (const uint64_t *)(lane.d_pin + oSeg), 0, K, N, lane.d_pin + oGt, (uint64_t *)(lane.d_pin + oEcho), lane.stream, lane.d_pin + oOk));
    HIP_TRY(hipStreamSynchronize(lane.stream));
    k0 = 0;
    for (size_t c = 0; c < nc; c++) {
        SmallCall &r = *calls[c];
        // fail closed, call by call: a result computed from a table that was not the caller's does not leave the library
        const int rc = verify_echo(h_echo + 2 * k0, table.data() + k0, r.segs);
        if (rc != GPBC_OK) {
            r.rc = rc; snprintf(r.err, sizeof r.err, "%s", g_err);
            if (r.out[0]) memset(r.out[0], 0, r.segs * GPBC_GT_BYTES);
            if (r.out[1]) memset(r.out[1], 0, r.segs);
        } else {
            if (r.out[0]) memcpy(r.out[0], lane.pin + oGt + k0 * GPBC_GT_BYTES, r.segs * GPBC_GT_BYTES);
            if (r.out[1]) memcpy(r.out[1], lane.pin + oOk + k0, r.segs);
        }
        k0 += r.segs;
    }
    return GPBC_OK;
}
// GT.Exp / Mul / Div / Inverse one call at a time (access/tree/access_tree_node.go:114,123,156-157): elementwise batches on a lane
static int small_gt_run(int OP, CallLane &lane, SmallCall *const *calls, size_t nc) {     // OP 0 mul, 1 div, 2 inverse, 3 exp
    size_t N = 0;
    for (size_t c = 0; c < nc; c++) N += calls[c]->units;
    const size_t b_unit = OP == 3 ? GPBC_SCALAR_BYTES : OP == 2 ? 0 : GPBC_GT_BYTES;
    const size_t oA = 0, oB = Scratch::padded(N * GPBC_GT_BYTES), oO = oB + Scratch::padded(N * b_unit), total = oO + Scratch::padded(N * GPBC_GT_BYTES);
    TRY(lane.reserve(total, OP == 3 ? 0 : 2 * N * GPBC_GT_BYTES));
    size_t n0 = 0;
    for (size_t c = 0; c < nc; c++) {
        const SmallCall &r = *calls[c];
        memcpy(lane.pin + oA + n0 * GPBC_GT_BYTES, r.in[0], r.units * GPBC_GT_BYTES);
        if (b_unit) memcpy(lane.pin + oB + n0 * b_unit, r.in[1], r.units * b_unit);
        n0 += r.units;
    }
    if (OP == 3) {
        k_gt_exp_wide<<<(unsigned)N, wide_threads(N), 0, lane.stream>>>(lane.d_pin + oA, lane.d_pin + oB, lane.d_pin + oO, N);
        TRY(check_launch("k_gt_exp_wide"));
        profile_mark("k_gt_exp_wide", lane.stream);
    } else {
        // one lane per element walks its 384 + 384 bytes many times: operands into device memory first
        HIP_TRY(hipMemcpyAsync(lane.dev, lane.pin + oA, N * GPBC_GT_BYTES, hipMemcpyHostToDevice, lane.stream));
        if (b_unit) HIP_TRY(hipMemcpyAsync(lane.dev + N * GPBC_GT_BYTES, lane.pin + oB, N * GPBC_GT_BYTES, hipMemcpyHostToDevice, lane.stream));
        k_gt_binary<<<grid_for(N), BLOCK, 0, lane.stream>>>(lane.dev, lane.dev + N * GPBC_GT_BYTES, lane.d_pin + oO, N, OP);
        TRY(check_launch("k_gt_binary"));
        profile_mark("k_gt_binary", lane.stream);
    }
    HIP_TRY(hipStreamSynchronize(lane.stream));
    n0 = 0;
    for (size_t c = 0; c < nc; c++) { memcpy(calls[c]->out[0], lane.pin + oO + n0 * GPBC_GT_BYTES, calls[c]->units * GPBC_GT_BYTES); n0 += calls[c]->units; }
    return GPBC_OK;
}
static int small_gt_mul_run(CallLane &l, SmallCall *const *c, size_t n) { return small_gt_run(0, l, c, n); }
static int small_gt_div_run(CallLane &l, SmallCall *const *c, size_t n) { return small_gt_run(1, l, c, n); }
static int small_gt_inv_run(CallLane &l, SmallCall *const *c, size_t n) { return small_gt_run(2, l, c, n); }
static int small_gt_exp_run(CallLane &l, SmallCall *const *c, size_t n) { return small_gt_run(3, l, c, n); }
static bool small_call_ok(size_t units) { const size_t lim = g_wide_max.load(); return units && units <= lim && units <= SMALL_CALL_MAX_UNITS; }

constexpr size_t PIPE_CHUNK = 131072;          // pairs per pipelined chunk: 2.5 GB of lines per stream
static int pair_batch_one(const void *P, const void *Q, size_t n, void *gt_out) {
    TRY(bind_device());
    if (n >= 2 * PIPE_CHUNK) {
        // upload / kernels / download of neighbouring chunks overlap on the slot's two streams (pipelined_chunks)
        DevBuf dP, dQ, dG;
        TRY(dP.alloc(n * GPBC_G1_BYTES)); TRY(dQ.alloc(n * GPBC_G2_BYTES)); TRY(dG.alloc(n * GPBC_GT_BYTES));
        int rc = pipelined_chunks(n, PIPE_CHUNK,
            [&](size_t off, size_t m, hipStream_t st) {
                HIP_TRY(hipMemcpyAsync(dP.u8() + off * GPBC_G1_BYTES, (const uint8_t *)P + off * GPBC_G1_BYTES, m * GPBC_G1_BYTES, hipMemcpyHostToDevice, st));
                HIP_TRY(hipMemcpyAsync(dQ.u8() + off * GPBC_G2_BYTES, (const uint8_t *)Q + off * GPBC_G2_BYTES, m * GPBC_G2_BYTES, hipMemcpyHostToDevice, st));
                return (int)GPBC_OK;
            },
            [&](size_t off, size_t m, hipStream_t st) {
                return gpbc_pair_batch_dev(dP.u8() + off * GPBC_G1_BYTES, dQ.u8() + off * GPBC_G2_BYTES, m, dG.u8() + off * GPBC_GT_BYTES, st);
            },
            [&](size_t off, size_t m, hipStream_t st) {
                HIP_TRY(hipMemcpyAsync((uint8_t *)gt_out + off * GPBC_GT_BYTES, dG.u8() + off * GPBC_GT_BYTES, m * GPBC_GT_BYTES, hipMemcpyDeviceToHost, st));
                return (int)GPBC_OK;
            });
        if (rc != GPBC_OK) { (void)hipDeviceSynchronize(); return rc; }     // nothing may still use the buffers when they are freed
        return GPBC_OK;
    }
    if (small_call_ok(n)) {
        // a latency call (bn254.Pair as the reference makes it): through the device's call lanes, combined with whatever other
        // threads are asking for at the same moment
        SmallCall c;
        c.in[0] = P; c.in[1] = Q; c.out[0] = gt_out; c.units = n; c.segs = n;
        return small_call(CALL_PAIRS, c, small_pairs_run);
    }
    DevBuf dP, dQ, dG;
    TRY(dP.upload(P, n * GPBC_G1_BYTES)); TRY(dQ.upload(Q, n * GPBC_G2_BYTES)); TRY(dG.alloc(n * GPBC_GT_BYTES));
    TRY(gpbc_pair_batch_dev(dP.p, dQ.p, n, dG.p, nullptr));
    TRY(sync_default());
    return dG.download(gt_out, n * GPBC_GT_BYTES);
}
int gpbc_pair_batch(const void *P, const void *Q, size_t n, void *gt_out) {
    if (!n) return fail(GPBC_ERR_INVALID_ARG, "invalid inputs sizes");
    if (!P || !Q || !gt_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    return run_sharded(n, SHARD_MIN_UNITS, [=](size_t lo, size_t hi) {
        return pair_batch_one((const uint8_t *)P + lo * GPBC_G1_BYTES, (const uint8_t *)Q + lo * GPBC_G2_BYTES, hi - lo, (uint8_t *)gt_out + lo * GPBC_GT_BYTES);
    });
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
    if (pairs_per_chunk < 0 || pairs_per_chunk > FIXED_Q_CHUNK) return fail(GPBC_ERR_INVALID_ARG, "chunk length must be 0 (automatic) .. %d", FIXED_Q_CHUNK);
    g_multi_chunk.store(pairs_per_chunk);
    return GPBC_OK;
}
// The product of a segment is only as good as the table the kernels read.  Round 2 recorded a verifier that answered "true" for a
// forged signature because k_segment_product had seen an EMPTY segment where the host's table said two pairs
// (profiles/r02_pool_bisect.txt; DESIGN.md §4 names the unordered operation).  The invariant since: (1) host tables travel through
// library-owned PINNED staging memory (pinned_staging: one buffer per device and stream, held under the scratch lock until the
// call has synchronised), so the copy is a stream-ordered DMA from memory nobody else writes; (2) the kernels echo what they
// consumed and the call FAILS (GPBC_ERR_INTERNAL, outputs zeroed) unless that equals the host's table segment by segment.
// test knob: the NEXT host-table multi-pairing sends the device a table whose last segment is empty while the host keeps the real
// one — what a stale or unordered table copy looks like to the kernels; the call must then fail (tests/cpp/test_bls_flow.cpp)
int gpbc_debug_stale_table_once(void) {
    // a fault-injection knob has no business in a production process: it answers only where the environment asks for it
    const char *e = getenv("GPBC_TEST_KNOBS");
    if (!e || e[0] != '1') return fail(GPBC_ERR_INVALID_ARG, "test knobs are off (set GPBC_TEST_KNOBS=1 in the environment of a test process)");
    g_fault_table.store(1);
    return GPBC_OK;
}
static int verify_echo(const uint64_t *echo, const uint64_t *seg_off, size_t k) {
    for (size_t j = 0; j < k; j++) {
        const uint64_t want = seg_off[j + 1] - seg_off[j];
        if (echo[2 * j + 1] != want || (want && echo[2 * j] != seg_off[j]))
            return fail(GPBC_ERR_INTERNAL, "segment %zu: the device consumed %llu pairs from %llu, the host table says %llu from %llu — result withheld", j,
                        (unsigned long long)echo[2 * j + 1], (unsigned long long)echo[2 * j], (unsigned long long)want, (unsigned long long)seg_off[j]);
    }
    return GPBC_OK;
}
static int multi_pair_core(const uint8_t *dP, const uint8_t *dQ, const uint64_t *seg_off, size_t k, size_t n_pairs, uint8_t *dG, uint8_t *dOk, hipStream_t st) {
    // Chunk length L: as long as possible (more shared squarings) while ~131072 lane pairs stay in flight, and at most
    // MULTI_CHUNK; with at least 65536 segments that all fit a chunk (AFP25: 3 pairs, BLS checks: 2) a chunk is a whole
    // segment — the chip is full with one lane pair per segment.  Chunks are launched in groups of MULTI_GROUP = 65536
    // (131072 lanes = 2048 waves: exactly one full round of two waves per SIMD on 256 CUs — a lane pair here runs for tens
    // of milliseconds, so a partially filled second round would cost as much as a full one); the slot grid of a group,
    // L x 65536 lines rows (twice that for an odd L, see the loop), is at most 17.4 GB.
    uint64_t max_len = 0;
    for (size_t j = 0; j < k; j++) if (seg_off[j + 1] - seg_off[j] > max_len) max_len = seg_off[j + 1] - seg_off[j];
    uint64_t L = (n_pairs + 131071) / 131072;
    if (max_len <= (uint64_t)MULTI_CHUNK) L = k >= MULTI_GROUP ? max_len : 1;      // short segments: whole or not at all
    if (g_multi_chunk.load() > 0) L = (uint64_t)g_multi_chunk.load();                 // (capped at MULTI_CHUNK below)
    const size_t echo_bytes = 2 * k * sizeof(uint64_t);
    Scratch tmp;                     // held until the stream has been synchronised: the pinned staging below belongs to this call until then
    uint64_t *h_echo = nullptr, *dEcho = nullptr;
    int rc = GPBC_OK;
    if (L <= 1 && g_multi_chunk.load() <= 0) {
        // nothing to share (few pairs, or single-pair segments): one Miller loop per lane pair and one product per segment
        const size_t wsb = gpbc_multi_pair_workspace_bytes(n_pairs, k), seg_bytes = (k + 1) * sizeof(uint64_t);
        TRY(tmp.open(st, 0, Scratch::padded(seg_bytes) + Scratch::padded(echo_bytes) + Scratch::padded(wsb)));
        uint64_t *dSeg = tmp.take<uint64_t>(seg_bytes);
        dEcho = tmp.take<uint64_t>(echo_bytes);
        uint8_t *dW = tmp.take(wsb);
        uint8_t *pin = nullptr;
        TRY(pinned_staging(st, Scratch::padded(seg_bytes) + echo_bytes, &pin));
        memcpy(pin, seg_off, seg_bytes);
        if (g_fault_table.exchange(0)) ((uint64_t *)pin)[k] = ((uint64_t *)pin)[k - 1];
        h_echo = (uint64_t *)(pin + Scratch::padded(seg_bytes));
        HIP_TRY(hipMemcpyAsync(dSeg, pin, seg_bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(dEcho, 0xff, echo_bytes, st));
        TRY(multi_pair_dev_echo(dP, dQ, dSeg, n_pairs, k, dG, dW, wsb, dEcho, st));
    } else {
        if (L < 1) L = 1;
        if (L > (uint64_t)MULTI_CHUNK) L = MULTI_CHUNK;
        std::vector<uint64_t> chunk_off(1, 0), seg_chunk(1, 0);
        for (size_t j = 0; j < k; j++) {
            for (uint64_t a = seg_off[j]; a < seg_off[j + 1]; a += L)
                chunk_off.push_back(a + L < seg_off[j + 1] ? a + L : seg_off[j + 1]);
            seg_chunk.push_back(chunk_off.size() - 1);
        }
        const size_t n_chunks = chunk_off.size() - 1;
        const size_t co_bytes = chunk_off.size() * sizeof(uint64_t), sc_bytes = seg_chunk.size() * sizeof(uint64_t), seen_bytes = 2 * (n_chunks + 1) * sizeof(uint64_t);
        TRY(tmp.open(st, 0, Scratch::padded(co_bytes) + Scratch::padded(sc_bytes) + 2 * Scratch::padded(seen_bytes) + Scratch::padded(echo_bytes) + Scratch::padded(n_chunks * GPBC_GT_BYTES)));
        uint64_t *dChunkOff = tmp.take<uint64_t>(co_bytes), *dSegChunk = tmp.take<uint64_t>(sc_bytes);
        uint64_t *dSeenLines = tmp.take<uint64_t>(seen_bytes), *dSeen = tmp.take<uint64_t>(seen_bytes);
        dEcho = tmp.take<uint64_t>(echo_bytes);
        uint8_t *dPart = tmp.take(n_chunks * GPBC_GT_BYTES);
        uint8_t *pin = nullptr;
        TRY(pinned_staging(st, Scratch::padded(co_bytes) + Scratch::padded(sc_bytes) + echo_bytes, &pin));
        memcpy(pin, chunk_off.data(), co_bytes);
        memcpy(pin + Scratch::padded(co_bytes), seg_chunk.data(), sc_bytes);
        if (g_fault_table.exchange(0)) ((uint64_t *)(pin + Scratch::padded(co_bytes)))[k] = ((uint64_t *)(pin + Scratch::padded(co_bytes)))[k - 1];
        h_echo = (uint64_t *)(pin + Scratch::padded(co_bytes) + Scratch::padded(sc_bytes));
        HIP_TRY(hipMemcpyAsync(dChunkOff, pin, co_bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(dSegChunk, pin + Scratch::padded(co_bytes), sc_bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(dSeenLines, 0xff, seen_bytes, st));
        HIP_TRY(hipMemsetAsync(dEcho, 0xff, echo_bytes, st));
        std::lock_guard<std::mutex> seq(g_ws_seq_mu);
        // the line phase runs one lane per pair: with an odd chunk length a group of 65 536 chunks fills an odd number of half
        // rounds of the chip (3 pairs: 1.5 rounds, i.e. two) — two groups at a time make it whole (3 rounds for 131 072 chunks)
        const size_t group = MULTI_GROUP * ((L & 1) && L > 1 ? 2 : 1);
        for (size_t cb = 0; cb < n_chunks; cb += group) {
            const size_t g = n_chunks - cb < group ? n_chunks - cb : group;
            size_t longest = 0;
            for (size_t c = cb; c < cb + g; c++) { size_t len = (size_t)(chunk_off[c + 1] - chunk_off[c]); if (len > longest) longest = len; }
            const size_t n_slots = longest * g;
            int32_t *lines = nullptr;
            TRY(lines_workspace(st, n_slots, &lines));
            const uint64_t *co = dChunkOff + cb;
            k_miller_lines_chunks<<<grid_for(n_slots), BLOCK, 0, st>>>(dP, dQ, lines, co, g, n_slots, dSeenLines + 2 * cb);
            TRY(check_launch("k_miller_lines_chunks"));
            profile_mark("k_miller_lines_chunks", st);
            k_miller_accumulate_chunks<<<grid_for(2 * g), BLOCK, 0, st>>>(dP, dQ, lines, co, dPart + cb * GPBC_GT_BYTES, g, n_slots, dSeenLines + 2 * cb, dSeen + 2 * cb);
            TRY(check_launch("k_miller_accumulate_chunks"));
            profile_mark("k_miller_accumulate_chunks", st);
        }
        uint64_t limit = 0;
        TRY(segment_fold_passes(dPart, dSegChunk, 0, k, n_chunks, st, &limit));
        k_segment_product<<<grid_for(k), BLOCK, 0, st>>>(dPart, dSegChunk, dG, k, n_chunks, dSeen, dEcho, limit);
        TRY(check_launch("k_segment_product (segments)"));
        profile_mark("k_segment_product", st);
        TRY(gpbc_final_exp_dev(dG, k, dG, st));
    }
    if (dOk) {
        k_gt_is_one<<<grid_for(k), BLOCK, 0, st>>>(dG, dOk, k);
        TRY(check_launch("k_gt_is_one"));
        profile_mark("k_gt_is_one", st);
    }
    HIP_TRY(hipMemcpyAsync(h_echo, dEcho, echo_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    rc = verify_echo(h_echo, seg_off, k);
    if (rc != GPBC_OK) {                                    // fail closed: nothing computed from a table that was not the caller's leaves this call
        (void)hipMemsetAsync(dG, 0, k * GPBC_GT_BYTES, st);
        if (dOk) (void)hipMemsetAsync(dOk, 0, k, st);
        (void)hipStreamSynchronize(st);
    }
    return rc;
}
// out[j] = Pair(P[j*m .. (j+1)*m), Q[0 .. m)), j < k.  Asynchronous on the stream (its temporaries live in the stream's scratch).
int gpbc_multi_pair_fixed_q_dev(const void *dP, const void *dQ, size_t m, size_t k, void *d_gt_out, void *stream) {
    if (!k || !m) return fail(GPBC_ERR_INVALID_ARG, "invalid inputs sizes");
    if (!dP || !dQ || !d_gt_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    hipStream_t st = (hipStream_t)stream;
    if (m * k <= g_wide_max.load() && g_multi_chunk.load() <= 0) {
        // a few ciphertexts' worth of pairs (one BSW07 decrypt is 513): the line table of the list would be computed by m lanes at a
        // lone lane's pace and cost more than it saves — every pair gets its own wavefront instead, the list indexed modulo m
        Scratch tmp;
        TRY(tmp.open(st, 0, Scratch::padded(m * k * GPBC_GT_BYTES)));
        uint8_t *vals = tmp.take(m * k * GPBC_GT_BYTES);
        k_miller_wide<<<(unsigned)(m * k), WIDE_MILLER_THREADS, 0, st>>>((const uint8_t *)dP, (const uint8_t *)dQ, vals, m * k, m);
        TRY(check_launch("k_miller_wide"));
        profile_mark("k_miller_wide", st);
        return segments_wide(vals, nullptr, m, k, m * k, (uint8_t *)d_gt_out, nullptr, st);
    }
    // Chunks of the Q list per lane pair: n_c equal chunks of L = ceil(m / n_c) <= FIXED_Q_CHUNK pairs.  One lane pair costs about
    // L line steps + 0.85 (its 64 squarings, in units of one pair's 88 line steps) and the chip runs 65536 lane pairs at a time, so
    // the estimate to minimise is  ceil(n_c k / 65536) * (L + 0.85): long chunks share squarings, but a last partly filled round of
    // long chunks costs as much as a full one.
    size_t n_c = 0, L = 0;
    if (g_multi_chunk.load() > 0) {
        L = (size_t)g_multi_chunk.load() < m ? (size_t)g_multi_chunk.load() : m;
        n_c = (m + L - 1) / L;
    } else {
        double best = 0;
        for (size_t c = (m + FIXED_Q_CHUNK - 1) / FIXED_Q_CHUNK; c <= m; c++) {
            const size_t len = (m + c - 1) / c, rounds = (c * k + 65535) / 65536;
            const double cost = (double)rounds * ((double)len + 0.85);
            if (!n_c || cost < best) { best = cost; n_c = c; L = len; }
            if (len == 1) break;
        }
    }
    {
        const size_t pint_bytes = m * k * 20 * sizeof(int32_t), part_bytes = n_c * k * GPBC_GT_BYTES;
        const size_t q34_bytes = m * (size_t)MILLER_LINES * LINE34_WORDS * sizeof(int32_t);
        Scratch tmp;
        TRY(tmp.open(st, 0, Scratch::padded(m * LINE_BYTES_PER_PAIR) + Scratch::padded(q34_bytes) + Scratch::padded(pint_bytes) + Scratch::padded(part_bytes)));
        int32_t *dLines = tmp.take<int32_t>(m * LINE_BYTES_PER_PAIR), *dQ34 = tmp.take<int32_t>(q34_bytes), *dPint = tmp.take<int32_t>(pint_bytes);
        uint8_t *dPart = tmp.take(part_bytes);
        HIP_TRY(hipMemsetAsync(dLines, 0, m * LINE_BYTES_PER_PAIR, st));           // rows of points at infinity are never written, but are scaled
        if (m <= g_wide_max.load() && g_wide_max.load() > 0) k_q_lines_wide<<<(unsigned)m, BLOCK, 0, st>>>((const uint8_t *)dQ, dLines, m);
        else k_q_lines<<<grid_for(m), BLOCK, 0, st>>>((const uint8_t *)dQ, dLines, m);
        TRY(check_launch("k_q_lines"));
        profile_mark("k_q_lines", st);
        k_q_lines_scale<<<grid_for(m * MILLER_LINES), BLOCK, 0, st>>>(dLines, dQ34, m);
        TRY(check_launch("k_q_lines_scale"));
        profile_mark("k_q_lines_scale", st);
        k_g1_line_point<<<grid_for((m * k + LINE_POINT_GROUP - 1) / LINE_POINT_GROUP), BLOCK, 0, st>>>((const uint8_t *)dP, dPint, m * k);
        TRY(check_launch("k_g1_line_point"));
        profile_mark("k_g1_line_point", st);
        k_miller_accumulate_fixed_q<<<grid_for(2 * n_c * k), BLOCK, 0, st>>>(dPint, (const uint8_t *)dQ, dQ34, dPart, m, k, L, n_c);
        TRY(check_launch("k_miller_accumulate_fixed_q"));
        profile_mark("k_miller_accumulate_fixed_q", st);
        uint64_t limit = 0;
        TRY(segment_fold_passes(dPart, nullptr, n_c, k, n_c * k, st, &limit));
        k_chunk_product<<<grid_for(k), BLOCK, 0, st>>>(dPart, (uint8_t *)d_gt_out, k, n_c, limit);      // ciphertext j owns chunk values [j n_c, (j+1) n_c)
        TRY(check_launch("k_chunk_product"));
        profile_mark("k_chunk_product", st);
    }
    return gpbc_final_exp_dev(d_gt_out, k, d_gt_out, st);
}
static int multi_pair_fixed_q_one(const void *P, const void *Q, size_t m, size_t k, void *gt_out) {
    TRY(bind_device());
    DevBuf dP, dQ, dG;
    TRY(dP.upload(P, m * k * GPBC_G1_BYTES)); TRY(dQ.upload(Q, m * GPBC_G2_BYTES)); TRY(dG.alloc(k * GPBC_GT_BYTES));
    TRY(gpbc_multi_pair_fixed_q_dev(dP.p, dQ.p, m, k, dG.p, nullptr));
    return dG.download(gt_out, k * GPBC_GT_BYTES);
}
int gpbc_multi_pair_fixed_q(const void *P, const void *Q, size_t m, size_t k, void *gt_out) {
    if (!k || !m) return fail(GPBC_ERR_INVALID_ARG, "invalid inputs sizes");
    if (!P || !Q || !gt_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    // segments are the independent units; every device computes the lines of the shared Q list for itself
    const size_t min_seg = (SHARD_MIN_UNITS + m - 1) / m;
    return run_sharded(k, min_seg, [=](size_t lo, size_t hi) {
        return multi_pair_fixed_q_one((const uint8_t *)P + lo * m * GPBC_G1_BYTES, Q, m, hi - lo, (uint8_t *)gt_out + lo * GPBC_GT_BYTES);
    });
}
int gpbc_multi_pair_hostseg_dev(const void *dP, const void *dQ, const uint64_t *seg_off, size_t k, void *d_gt_out, void *stream) {
    if (!k) return fail(GPBC_ERR_INVALID_ARG, "invalid inputs sizes");
    size_t n_pairs = 0;
    TRY(check_segments(seg_off, k, &n_pairs));
    if ((n_pairs && (!dP || !dQ)) || !d_gt_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    return multi_pair_core((const uint8_t *)dP, (const uint8_t *)dQ, seg_off, k, n_pairs, (uint8_t *)d_gt_out, nullptr, (hipStream_t)stream);
}
static int multi_pair_host_one(const void *P, const void *Q, const uint64_t *seg_off, size_t k, size_t n_pairs, void *gt_out, uint8_t *ok_out) {
    TRY(bind_device());
    if (small_call_ok(n_pairs) && small_call_ok(k) && g_multi_chunk.load() <= 0) {
        // a latency call (Pair / PairingCheck as the reference makes them): through the call lanes (small_pairs_run)
        SmallCall c;
        c.in[0] = P; c.in[1] = Q; c.out[0] = gt_out; c.out[1] = ok_out; c.units = n_pairs; c.seg = seg_off; c.segs = k;
        return small_call(CALL_PAIRS, c, small_pairs_run);
    }
    DevBuf dP, dQ, dG, dOk;
    TRY(dP.upload(P, n_pairs * GPBC_G1_BYTES)); TRY(dQ.upload(Q, n_pairs * GPBC_G2_BYTES));
    TRY(dG.alloc(k * GPBC_GT_BYTES));
    if (ok_out) TRY(dOk.alloc(k));
    TRY(multi_pair_core(dP.u8(), dQ.u8(), seg_off, k, n_pairs, dG.u8(), ok_out ? dOk.u8() : nullptr, nullptr));
    if (gt_out) TRY(dG.download(gt_out, k * GPBC_GT_BYTES));
    if (ok_out) TRY(dOk.download(ok_out, k));
    return GPBC_OK;
}
static int multi_pair_host(const void *P, const void *Q, const uint64_t *seg_off, size_t k, void *gt_out, uint8_t *ok_out) {
    if (!k) return fail(GPBC_ERR_INVALID_ARG, "invalid inputs sizes");
    size_t n_pairs = 0;
    TRY(check_segments(seg_off, k, &n_pairs));
    if ((n_pairs && (!P || !Q)) || (!gt_out && !ok_out)) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    // segments are the independent units: a shard is a run of whole segments with its table rebased to zero
    const size_t avg = n_pairs / k ? n_pairs / k : 1;
    return run_sharded(k, (SHARD_MIN_UNITS + avg - 1) / avg, [=](size_t lo, size_t hi) {
        if (lo == 0 && hi == k) return multi_pair_host_one(P, Q, seg_off, k, n_pairs, gt_out, ok_out);
        std::vector<uint64_t> sub(hi - lo + 1);
        const uint64_t base = seg_off[lo];
        for (size_t j = lo; j <= hi; j++) sub[j - lo] = seg_off[j] - base;
        return multi_pair_host_one((const uint8_t *)P + base * GPBC_G1_BYTES, (const uint8_t *)Q + base * GPBC_G2_BYTES, sub.data(), hi - lo,
                                   (size_t)sub.back(), gt_out ? (uint8_t *)gt_out + lo * GPBC_GT_BYTES : nullptr, ok_out ? ok_out + lo : nullptr);
    });
}
int gpbc_multi_pair(const void *P, const void *Q, const uint64_t *seg_off, size_t k, void *gt_out) {
    if (!gt_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    return multi_pair_host(P, Q, seg_off, k, gt_out, nullptr);
}
int gpbc_pairing_check(const void *P, const void *Q, const uint64_t *seg_off, size_t k, uint8_t *ok_out) {
    if (!ok_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    return multi_pair_host(P, Q, seg_off, k, nullptr, ok_out);
}
static int gt_exp_one(const void *x, const void *k, size_t n, void *out) {
    TRY(bind_device());
    if (small_call_ok(n)) {
        SmallCall c;
        c.in[0] = x; c.in[1] = k; c.out[0] = out; c.units = n;
        return small_call(CALL_GT_EXP, c, small_gt_exp_run);
    }
    DevBuf dX, dK, dO;
    TRY(dX.upload(x, n * GPBC_GT_BYTES)); TRY(dK.upload(k, n * GPBC_SCALAR_BYTES)); TRY(dO.alloc(n * GPBC_GT_BYTES));
    TRY(gpbc_gt_exp_batch_dev(dX.p, dK.p, n, dO.p, nullptr));
    TRY(sync_default());
    return dO.download(out, n * GPBC_GT_BYTES);
}
int gpbc_gt_exp_batch(const void *x, const void *k, size_t n, void *out) {
    if (!n) return GPBC_OK;
    if (!x || !k || !out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    return run_sharded(n, SHARD_MIN_UNITS, [=](size_t lo, size_t hi) {
        return gt_exp_one((const uint8_t *)x + lo * GPBC_GT_BYTES, (const uint8_t *)k + lo * GPBC_SCALAR_BYTES, hi - lo, (uint8_t *)out + lo * GPBC_GT_BYTES);
    });
}
static int gt_binary_one(int op, const void *a, const void *b, size_t n, void *out) {
    TRY(bind_device());
    if (small_call_ok(n)) {
        SmallCall c;
        c.in[0] = a; c.in[1] = b; c.out[0] = out; c.units = n;
        return op == 0 ? small_call(CALL_GT_MUL, c, small_gt_mul_run) : op == 1 ? small_call(CALL_GT_DIV, c, small_gt_div_run) : small_call(CALL_GT_INV, c, small_gt_inv_run);
    }
    DevBuf dA, dB, dO;
    TRY(dA.upload(a, n * GPBC_GT_BYTES));
    if (op != 2) TRY(dB.upload(b, n * GPBC_GT_BYTES));
    TRY(dO.alloc(n * GPBC_GT_BYTES));
    TRY(gt_binary_dev(op, dA.p, dB.p, n, dO.p, nullptr));
    TRY(sync_default());
    return dO.download(out, n * GPBC_GT_BYTES);
}
static int gt_binary_host(int op, const void *a, const void *b, size_t n, void *out) {
    if (!n) return GPBC_OK;
    if (!a || (op != 2 && !b) || !out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    return run_sharded(n, 4 * SHARD_MIN_UNITS, [=](size_t lo, size_t hi) {
        return gt_binary_one(op, (const uint8_t *)a + lo * GPBC_GT_BYTES, b ? (const uint8_t *)b + lo * GPBC_GT_BYTES : nullptr, hi - lo, (uint8_t *)out + lo * GPBC_GT_BYTES);
    });
}
int gpbc_gt_mul_batch(const void *a, const void *b, size_t n, void *o) { return gt_binary_host(0, a, b, n, o); }
int gpbc_gt_div_batch(const void *a, const void *b, size_t n, void *o) { return gt_binary_host(1, a, b, n, o); }
int gpbc_gt_inverse_batch(const void *a, size_t n, void *o) { return gt_binary_host(2, a, nullptr, n, o); }

}  // extern "C"
