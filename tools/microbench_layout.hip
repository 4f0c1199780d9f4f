// north_star layout against the built layout, one number.
//
// north_star: "one (P,Q) pair per wavefront, Fp limbs ... staged in LDS, Montgomery mul/reduce using wavefront shuffles".
// Built:      one field element per LANE (nine signed 29-bit limbs in nine registers), one Fp12 value per lane pair.
//
// This program measures the core of the question — a lazy Fp2 product (the leaf every tower operation is made of):
//   per-lane      f2_mul_leaf of csrc/tower29.hip.hpp: 64 products per wave-call, 486 v_mad_i64_i32 + 127 other VALU instructions
//   cooperative   the same product with the LIMBS SPREAD OVER NINE LANES (seven products per wave, lane 63 idle): lane j holds
//                 limb j of a0, a1, b0, b1; operand limbs travel by ds_bpermute_b32 (the LDS crossbar, the "wavefront shuffle"
//                 of gfx950 for arbitrary lane patterns), column sums live one per lane, the Montgomery reduction walks the
//                 nine low columns in order and hands the carry to the next lane.
// Both compute (a0 b0 - a1 b1, a0 b1 + a1 b0) / 2^261 mod p on the same operands and the outputs are compared (after
// normalisation to canonical bytes), so the cooperative version is a working multiplier, not a sketch.
//
// build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/microbench_layout.hip -o tools/microbench_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../gopairingbasedcryptography_amd/csrc/tower29.hip.hpp"
using namespace gpbc;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ int32_t bperm(int src_lane, int32_t v) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ int64_t bperm64(int src_lane, int64_t v) {
    uint32_t lo = (uint32_t)bperm(src_lane, (int32_t)(uint32_t)v), hi = (uint32_t)bperm(src_lane, (int32_t)(v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// One lazy Fp2 product per group of nine lanes.  In: this lane's limb of a0, a1, b0, b1.  Out: this lane's limb of the real
// and imaginary part (limbs 0..7 in [0, 2^29 + small), limb 8 signed: the class the per-lane leaf returns).
// prot[t] = p_((j - t) mod 9): the modulus limb this lane multiplies m_t by (constant per lane, loaded once per kernel).
__device__ __forceinline__ void coop_f2_mul(int gb, int j, int32_t a0, int32_t a1, int32_t b0, int32_t b1, const int32_t (&prot)[NL], int32_t &r0, int32_t &r1) {
    int64_t lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;                  // columns j and j + 9 of the real / imaginary part
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int32_t x0 = bperm(gb + i, a0), x1 = bperm(gb + i, a1);          // limb i of a, to every lane of the group
        int idx = j - i; if (idx < 0) idx += NL;                                // (j - i) mod 9: the one b limb this lane pairs with a_i
        const int32_t y0 = bperm(gb + idx, b0), y1 = bperm(gb + idx, b1);
        const int32_t ml = (i <= j) ? -1 : 0, mh = ~ml;                         // i + idx == j -> column j, else column j + 9
        const int32_t y0l = y0 & ml, y1l = y1 & ml, y0h = y0 & mh, y1h = y1 & mh;
        lo0 += (int64_t)x0 * y0l - (int64_t)x1 * y1l;  hi0 += (int64_t)x0 * y0h - (int64_t)x1 * y1h;
        lo1 += (int64_t)x0 * y1l + (int64_t)x1 * y0l;  hi1 += (int64_t)x0 * y1h + (int64_t)x1 * y0h;
    }
    // Montgomery reduction: for t = 0..8 lane t owns the lowest live column; m_t goes to every lane, m_t * p lands on
    // columns t .. t + 8, and lane t's cleared column hands its carry to the next column (lane t + 1, or column 9 = lane 0)
#pragma unroll
    for (int t = 0; t < NL; t++) {
        const int32_t m0 = bperm(gb + t, (int32_t)(((uint32_t)lo0 * (uint32_t)F29_PINV) & (uint32_t)LMASK));
        const int32_t m1 = bperm(gb + t, (int32_t)(((uint32_t)lo1 * (uint32_t)F29_PINV) & (uint32_t)LMASK));
        const int32_t pl = (j >= t) ? prot[t] : 0, ph = (j >= t) ? 0 : prot[t];
        lo0 += (int64_t)m0 * pl;  hi0 += (int64_t)m0 * ph;
        lo1 += (int64_t)m1 * pl;  hi1 += (int64_t)m1 * ph;
        // carry of column t (its low 29 bits are zero now) into column t + 1
        const int from = (j == 0) ? gb + NL - 1 : gb + j - 1;
        const int64_t c0 = bperm64(from, lo0 >> LB), c1 = bperm64(from, lo1 >> LB);
        if (t < NL - 1) { if (j == t + 1) { lo0 += c0; lo1 += c1; } }
        else if (j == 0) { hi0 += c0; hi1 += c1; }
    }
    // result limb j = column j + 9 with the carries of the columns below: two carry-save passes (a column holds up to 2^63)
    const int from = (j == 0) ? gb : gb + j - 1;
    int64_t c0 = bperm64(from, hi0 >> LB), c1 = bperm64(from, hi1 >> LB);
    if (j == 0) { c0 = 0; c1 = 0; }
    int64_t t0 = (j == NL - 1 ? hi0 : (hi0 & LMASK)) + c0, t1 = (j == NL - 1 ? hi1 : (hi1 & LMASK)) + c1;
    int32_t d0 = bperm(from, (int32_t)(t0 >> LB)), d1 = bperm(from, (int32_t)(t1 >> LB));
    if (j == 0) { d0 = 0; d1 = 0; }
    r0 = (j == NL - 1 ? (int32_t)t0 : (int32_t)(t0 & LMASK)) + d0;
    r1 = (j == NL - 1 ? (int32_t)t1 : (int32_t)(t1 & LMASK)) + d1;
}

// limbs[(elem * 4 + which) * 9 + limb]: internal-form operands a0, a1, b0, b1 of n Fp2 products
__global__ void __launch_bounds__(64, 2) k_coop(const int32_t *__restrict__ limbs, int32_t *__restrict__ out, int n_groups, int iters) {
    const int lane = threadIdx.x, g = lane / NL, j = lane % NL, gb = g * NL;
    const size_t elem = ((size_t)blockIdx.x * 7 + g) % (size_t)n_groups;
    constexpr int32_t PL[NL] = F29_P;
    int32_t prot[NL];
#pragma unroll
    for (int t = 0; t < NL; t++) { int idx = j - t; if (idx < 0) idx += NL; int32_t v = 0;
#pragma unroll
        for (int q = 0; q < NL; q++) v = (idx == q) ? PL[q] : v;
        prot[t] = v; }
    const bool live = lane < 63;
    const int32_t *e = limbs + elem * 36;
    int32_t a0 = live ? e[j] : 0, a1 = live ? e[9 + j] : 0, b0 = live ? e[18 + j] : 0, b1 = live ? e[27 + j] : 0;
    int32_t r0 = 0, r1 = 0;
    for (int it = 0; it < iters; it++) {
        coop_f2_mul(gb, j, a0, a1, b0, b1, prot, r0, r1);
        a0 = r0; a1 = r1;                                     // chain: a <- a * b
    }
    if (live) { out[(((size_t)blockIdx.x * 7 + g) * 2) * 9 + j] = r0; out[(((size_t)blockIdx.x * 7 + g) * 2 + 1) * 9 + j] = r1; }
}
__global__ void __launch_bounds__(64, 2) k_lane(const int32_t *__restrict__ limbs, int32_t *__restrict__ out, int n_groups, int iters, int per_block) {
    const size_t slot = (size_t)blockIdx.x * per_block + threadIdx.x;
    if ((int)threadIdx.x >= per_block) return;
    const int32_t *e = limbs + (slot % (size_t)n_groups) * 36;
    F2 a, b;
#pragma unroll
    for (int i = 0; i < NL; i++) { a.a0.v[i] = e[i]; a.a1.v[i] = e[9 + i]; b.a0.v[i] = e[18 + i]; b.a1.v[i] = e[27 + i]; }
    for (int it = 0; it < iters; it++) a = f2_mul(a, b);
#pragma unroll
    for (int i = 0; i < NL; i++) { out[(slot * 2) * 9 + i] = a.a0.v[i]; out[(slot * 2 + 1) * 9 + i] = a.a1.v[i]; }
}
// canonical 32-byte form of internal limbs (value mod p, independent of the lazy representation)
__global__ void k_canon(const int32_t *__restrict__ limbs, uint8_t *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    Fe x;
#pragma unroll
    for (int q = 0; q < NL; q++) x.v[q] = limbs[i * 9 + q];
    fe_store(out + 32 * i, fe_reduce(fe_norm(x)));
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, waves = 2, grid = ncu * 4 * waves;
    const int n_groups = 4096;
    static int32_t h[4096 * 36];
    srand(7);
    for (int i = 0; i < n_groups * 36; i++) h[i] = (i % 9 == 8) ? (rand() & 0xfffff) : (rand() & LMASK);   // normalised limbs, value < 2^253
    int32_t *din, *dc, *dl; uint8_t *bc, *bl;
    CHECK(hipMalloc(&din, sizeof h)); CHECK(hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice));
    const size_t n_coop = (size_t)grid * 7, n_lane = (size_t)grid * 7;      // compare the same 7 elements per block
    CHECK(hipMalloc(&dc, n_coop * 18 * 4)); CHECK(hipMalloc(&dl, (size_t)grid * 64 * 18 * 4));
    CHECK(hipMalloc(&bc, n_coop * 64)); CHECK(hipMalloc(&bl, n_lane * 64));
    // ---- equality on the first 7 elements of every block, 5 chained products
    k_coop<<<grid, 64>>>(din, dc, n_groups, 5);
    k_lane<<<grid, 64>>>(din, dl, n_groups, 5, 7);
    CHECK(hipDeviceSynchronize());
    k_canon<<<(unsigned)((n_coop * 2 + 63) / 64), 64>>>(dc, bc, n_coop * 2);
    k_canon<<<(unsigned)((n_lane * 2 + 63) / 64), 64>>>(dl, bl, n_lane * 2);
    CHECK(hipDeviceSynchronize());
    static uint8_t hc[1 << 20], hl[1 << 20];
    size_t cmp = n_coop * 64 < sizeof hc ? n_coop * 64 : sizeof hc;
    CHECK(hipMemcpy(hc, bc, cmp, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hl, bl, cmp, hipMemcpyDeviceToHost));
    if (memcmp(hc, hl, cmp) != 0) { printf("MISMATCH: the cooperative product differs from the per-lane leaf\n"); return 1; }
    printf("cooperative and per-lane Fp2 products agree on %zu products (5 chained each)\n", cmp / 64);
    // ---- timing: products per second per GPU
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto time_it = [&](auto launch) { float best = 1e30f; for (int r = 0; r < 3; r++) { CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } return best; };
    const int iters = 2000;
    float tc = time_it([&] { k_coop<<<grid, 64>>>(din, dc, n_groups, iters); });
    float tl = time_it([&] { k_lane<<<grid, 64>>>(din, dl, n_groups, iters, 64); });
    double pc = (double)grid * 7 * iters / (tc * 1e-3), pl = (double)grid * 64 * iters / (tl * 1e-3);
    double cyc_c = tc * 1e-3 * 2.4e9 / iters / waves, cyc_l = tl * 1e-3 * 2.4e9 / iters / waves;
    printf("cooperative (limbs across 9 lanes, ds_bpermute)  %8.3f ms  %9.1f SIMD cycles / wave-call  7 products / wave  -> %8.2f G Fp2 products/s\n", tc, cyc_c, pc / 1e9);
    printf("per lane (f2_mul_leaf, 9 limbs in registers)     %8.3f ms  %9.1f SIMD cycles / wave-call 64 products / wave  -> %8.2f G Fp2 products/s\n", tl, cyc_l, pl / 1e9);
    printf("per-lane / cooperative throughput ratio: %.1fx\n", pl / pc);
    return 0;
}
