// PROTOTYPE (measurement only, not in the library): the cyclotomic squaring of the final exponentiation with TEN 27-BIT LIMBS
// (Montgomery radix 2^270) and the sums formed in the 64-bit column domain — the candidate VERDICT r03 item 1 names ("ten 26-bit limbs
// for this kernel only … dot-product leaves with one reduction per output coefficient").  One Fp12 value per lane pair as in
// csrc/tower29_pair.hip.hpp; the three Fp4 pairs of the Granger-Scott squaring are kept lane-local:
//   even lane: (C0.b0, C1.b1) whole, of (C1.b0, C0.b2): C0.b2^2 and the real half of C1.b0 C0.b2
//   odd  lane: (C0.b1, C1.b2) whole, of (C1.b0, C0.b2): C1.b0^2 and the imaginary half of the product
// so per lane: U = xi y^2 + x^2 accumulated in ONE column set per component (operands pre-scaled by 9: 27 + 3.2 bits fit int32, the
// columns stay below 2^62), V = 2 x y, one F2 square, one half product: 7 reductions instead of 9, no normalisation before a product,
// xi by shift-and-add (no re-splitting), one normalisation per output.  It is checked against f12p_cyclo_sqr (9 x 29) on the same
// inputs through canonical bytes, and timed beside f12p_cyclo_sqr_alt at two waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -I gopairingbasedcryptography_amd/csrc tools/microbench_sqr27.hip -o tools/microbench_sqr27
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "tower29_pair.hip.hpp"
using namespace gpbc;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int N7 = 10, B7 = 27;
constexpr int32_t M7 = (1 << B7) - 1;
struct G { int32_t v[N7]; };                   // value = sum v[i] 2^(27 i), Montgomery radix 2^270
struct G2 { G a0, a1; };
struct G6 { G2 b0, b1, b2; };
__device__ __forceinline__ constexpr int32_t p7(int i) { constexpr int32_t P[N7] = {8191303, 68256475, 120206576, 88650808, 98138134, 84083376, 101978477, 26018125, 72250081, 1548}; return P[i]; }
constexpr uint32_t PINV7 = 0x4866389u;
#define D __device__ __forceinline__
D G g_add(const G &a, const G &b) { G r; for (int i = 0; i < N7; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
D G g_sub(const G &a, const G &b) { G r; for (int i = 0; i < N7; i++) r.v[i] = a.v[i] - b.v[i]; return r; }
D G g_neg(const G &a) { G r; for (int i = 0; i < N7; i++) r.v[i] = -a.v[i]; return r; }
D G g_scale(const G &a, int32_t k) { G r; for (int i = 0; i < N7; i++) r.v[i] = a.v[i] * k; return r; }   // small k: shift-and-add
D G g_sel(bool c, const G &a, const G &b) { G r; for (int i = 0; i < N7; i++) r.v[i] = c ? a.v[i] : b.v[i]; return r; }
D G g_norm(const G &a) {
    G r;
    r.v[0] = a.v[0] & M7;
    for (int i = 1; i < N7 - 1; i++) r.v[i] = (a.v[i] & M7) + (a.v[i - 1] >> B7);
    r.v[N7 - 1] = a.v[N7 - 1] + (a.v[N7 - 2] >> B7);
    return r;
}
D G2 g2_sel(bool c, const G2 &a, const G2 &b) { return G2{g_sel(c, a.a0, b.a0), g_sel(c, a.a1, b.a1)}; }
D G g_swap(const G &a) { G r; for (int i = 0; i < N7; i++) r.v[i] = __builtin_amdgcn_mov_dpp(a.v[i], 0xB1, 0xF, 0xF, true); return r; }
D G2 g2_swap(const G2 &a) { return G2{g_swap(a.a0), g_swap(a.a1)}; }

// ---- column domain
struct Col { int64_t c[2 * N7 - 1]; };
D void col_zero(Col &w) { for (int k = 0; k < 2 * N7 - 1; k++) w.c[k] = 0; }
D void col_mul(Col &w, const G &a, const G &b) {                    // += a b   (100 MADs)
    for (int i = 0; i < N7; i++) for (int j = 0; j < N7; j++) w.c[i + j] += (int64_t)a.v[i] * (int64_t)b.v[j];
}
D void col_sqr(Col &w, const G &a, const G &s) {                    // += a s with s = k a for a constant k: the symmetric half (55 MADs)
    for (int i = 0; i < N7; i++) for (int j = i; j < N7; j++) w.c[i + j] += (int64_t)(i == j ? a.v[i] : 2 * a.v[i]) * (int64_t)s.v[j];
}
D G col_reduce(const Col &w) {                                      // Montgomery, ten 27-bit digits (100 MADs)
    int32_t m[N7];
    G r;
    int64_t acc = 0;
    for (int k = 0; k < 2 * N7 - 1; k++) {
        acc += w.c[k];
        for (int i = 0; i < N7; i++) { const int j = k - i; if (j < 1 || j >= N7) continue; acc += (int64_t)m[i] * (int64_t)p7(j); }   // digits m_i with i < k only (j >= 1)
        if (k < N7) { m[k] = (int32_t)(((uint32_t)acc * PINV7) & (uint32_t)M7); acc += (int64_t)m[k] * (int64_t)p7(0); }
        else r.v[k - N7] = (int32_t)(acc & M7);
        acc >>= B7;
    }
    r.v[N7 - 1] = (int32_t)acc;
    return r;
}
D G g_mul(const G &a, const G &b) { Col w; col_zero(w); col_mul(w, a, b); return col_reduce(w); }
// value reduction: subtract k p with k from the top limb (p's top limb 1548 is small: go by the top TWO limbs)
D G g_reduce(const G &a) {
    const float top = (float)a.v[N7 - 1] * 134217728.0f + (float)a.v[N7 - 2];
    int32_t k = (int32_t)rintf(top * (1.0f / (1548.0f * 134217728.0f + 72250081.0f)));
    k = k > 15 ? 15 : k < -15 ? -15 : k;                            // k p_i must fit 32 bits; larger values take two passes (g_to_words)
    G r;
    for (int i = 0; i < N7; i++) r.v[i] = a.v[i] - k * p7(i);       // |k| < 2^4 here: k p_i fits 32 bits
    return g_norm(r);
}

// ---- the squaring (lane pair; h = this lane's half)
D G6 sqr27(bool odd, const G6 &h) {
    const G2 pb1 = g2_swap(h.b1), pother = g2_swap(g2_sel(odd, h.b0, h.b2));     // even receives C1.b1, C1.b0; odd receives C0.b1, C0.b2
    const G2 x4 = g2_sel(odd, pb1, h.b0), y4 = g2_sel(odd, h.b2, pb1);           // own Fp4 pair (x, y)
    // U = xi y^2 + x^2:   re = 9 (y0^2 - y1^2) - 2 y0 y1 + x0^2 - x1^2      im = 9 * 2 y0 y1 + (y0^2 - y1^2) + 2 x0 x1
    Col w;
    G2 U, V, Z;
    {
        const G y0_9 = g_scale(y4.a0, 9), y1_9n = g_scale(y4.a1, -9);
        col_zero(w);
        col_sqr(w, y4.a0, y0_9); col_sqr(w, y4.a1, y1_9n); col_mul(w, g_scale(y4.a0, -2), y4.a1); col_sqr(w, x4.a0, x4.a0); col_sqr(w, x4.a1, g_neg(x4.a1));
        U.a0 = col_reduce(w);
        col_zero(w);
        // (18 y0 would leave int32: 9 y0 times 2 y1)
        col_mul(w, y0_9, g_scale(y4.a1, 2)); col_sqr(w, y4.a0, y4.a0); col_sqr(w, y4.a1, g_neg(y4.a1)); col_mul(w, g_scale(x4.a0, 2), x4.a1);
        U.a1 = col_reduce(w);
    }
    {   // V = 2 x y
        const G x0_2 = g_scale(x4.a0, 2), x1_2 = g_scale(x4.a1, 2);
        col_zero(w); col_mul(w, x0_2, y4.a0); col_mul(w, g_neg(x1_2), y4.a1); V.a0 = col_reduce(w);
        col_zero(w); col_mul(w, x0_2, y4.a1); col_mul(w, x1_2, y4.a0); V.a1 = col_reduce(w);
    }
    {   // Z = z^2 with z = C0.b2 (even) | C1.b0 (odd)
        const G2 z = g2_sel(odd, h.b0, h.b2);
        col_zero(w); col_sqr(w, z.a0, z.a0); col_sqr(w, z.a1, g_neg(z.a1)); Z.a0 = col_reduce(w);
        col_zero(w); col_mul(w, g_scale(z.a0, 2), z.a1); Z.a1 = col_reduce(w);
    }
    // half of C1.b0 * C0.b2: even the real part (x = partner's C1.b0, y = own C0.b2), odd the imaginary part (x = own, y = partner's)
    G mh;
    {
        const G A = g_sel(odd, h.b0.a0, pother.a0), Bq = g_sel(odd, pother.a1, h.b2.a0), Cq = g_sel(odd, h.b0.a1, g_neg(pother.a1)), Dq = g_sel(odd, pother.a0, h.b2.a1);
        col_zero(w); col_mul(w, A, Bq); col_mul(w, Cq, Dq); mh = col_reduce(w);
    }
    // exchange: even sends V (C1'.b1 needs it) and its half product; odd sends U (C0'.b2) and Z = C1.b0^2
    const G2 s1 = g2_swap(g2_sel(odd, U, V)), s2 = g2_swap(g2_sel(odd, Z, G2{mh, mh}));
    const G2 q = g2_sel(odd, V, Z);                                   // what this lane multiplies by xi: odd V (C1'.b0), even Z = t2
    const G2 xq{g_norm(g_sub(g_scale(q.a0, 9), q.a1)), g_norm(g_add(g_scale(q.a1, 9), q.a0))};                  // (normalised: it is tripled below)
    // T: even (U, xi Z + t3, U_C)   odd (xi V, V_A, 2 (m_re + m_im i))
    const G2 T0 = g2_sel(odd, xq, U);
    const G2 T1 = g2_sel(odd, s1, G2{g_add(xq.a0, s2.a0), g_add(xq.a1, s2.a1)});
    const G2 T2 = g2_sel(odd, G2{g_scale(s2.a0, 2), g_scale(mh, 2)}, s1);
    const int32_t sg = odd ? 2 : -2;
    auto out = [&](const G2 &T, const G2 &x) { return G2{g_norm(g_add(g_scale(T.a0, 3), g_scale(x.a0, sg))), g_norm(g_add(g_scale(T.a1, 3), g_scale(x.a1, sg)))}; };
    return G6{out(T0, h.b0), out(T1, h.b1), out(T2, h.b2)};
}
D G6 g6_reduce(const G6 &h) { return G6{G2{g_reduce(h.b0.a0), g_reduce(h.b0.a1)}, G2{g_reduce(h.b1.a0), g_reduce(h.b1.a1)}, G2{g_reduce(h.b2.a0), g_reduce(h.b2.a1)}}; }

// ---- conversions through gnark's canonical words (x 2^256 mod p, eight 32-bit words)
D G g_from_words(const uint32_t w[8]) {
    G x;
    for (int i = 0; i < N7; i++) {
        const int bit = B7 * i, wi = bit >> 5, sh = bit & 31;
        uint64_t two = wi < 8 ? (uint64_t)w[wi] : 0;
        if (wi + 1 < 8) two |= (uint64_t)w[wi + 1] << 32;
        x.v[i] = (int32_t)((two >> sh) & (uint64_t)M7);
    }
    const G cin{{132570834, 76169053, 84900857, 13839673, 39193585, 16616571, 26299750, 30077364, 22913795, 594}};   // 2^284 mod p
    return g_mul(x, cin);                                            // (x 2^256) 2^284 / 2^270 = x 2^270
}
D void g_to_words(uint32_t w[8], const G &a) {
    const G cout{{93261213, 61370808, 70055757, 93616867, 46180238, 116454028, 26978523, 4127099, 41402778, 449}};   // 2^256 mod p
    G x = g_mul(g_reduce(g_reduce(g_reduce(a))), cout);              // x 2^256, value in (-eps p, (1 + eps) p)
    int32_t t[N7];
    int64_t c = 0;
    for (int i = 0; i < N7 - 1; i++) { int64_t s = (int64_t)x.v[i] + 2 * (int64_t)p7(i) + c; t[i] = (int32_t)(s & M7); c = s >> B7; }
    t[N7 - 1] = (int32_t)((int64_t)x.v[N7 - 1] + 2 * (int64_t)p7(N7 - 1) + c);
    for (int rep = 0; rep < 4; rep++) {
        int32_t d[N7], b = 0;
        for (int i = 0; i < N7 - 1; i++) { int32_t s = t[i] - p7(i) + b; d[i] = s & M7; b = s >> B7; }
        d[N7 - 1] = t[N7 - 1] - p7(N7 - 1) + b;
        const bool ge = d[N7 - 1] >= 0;
        for (int i = 0; i < N7; i++) t[i] = ge ? d[i] : t[i];
    }
    uint64_t acc = 0;
    int have = 0, wi = 0;
    for (int i = 0; i < N7; i++) { acc |= (uint64_t)(uint32_t)t[i] << have; have += B7; while (have >= 32 && wi < 8) { w[wi++] = (uint32_t)acc; acc >>= 32; have -= 32; } }
    if (wi < 8) w[wi] = (uint32_t)acc;
}
D G2 g2_from_f2(const F2 &x) { uint32_t w[8]; G2 r; fe_to_words(w, x.a0); r.a0 = g_from_words(w); fe_to_words(w, x.a1); r.a1 = g_from_words(w); return r; }
D G6 g6_from_f6(const F6 &x) { return G6{g2_from_f2(x.b0), g2_from_f2(x.b1), g2_from_f2(x.b2)}; }
D F2 ld2(const uint8_t *p) { return F2{fe_load(p), fe_load(p + 32)}; }

// correctness: one squaring each way, results as canonical words
__global__ void __launch_bounds__(64, 2) check(const uint8_t *in, uint32_t *out29, uint32_t *out27) {
    const size_t i = threadIdx.x;
    const uint8_t *base = in + 32 * ((i * 12) & 1023);
    const F6 h{ld2(base), ld2(base + 64), ld2(base + 128)};
    PairDpp x{(bool)(threadIdx.x & 1)};
    const F6 r = f12p_cyclo_sqr<true>(x, h);
    const G6 s = sqr27(x.odd, g6_from_f6(h));
    const Fe *fr[6] = {&r.b0.a0, &r.b0.a1, &r.b1.a0, &r.b1.a1, &r.b2.a0, &r.b2.a1};
    const G *gs[6] = {&s.b0.a0, &s.b0.a1, &s.b1.a0, &s.b1.a1, &s.b2.a0, &s.b2.a1};
    for (int e = 0; e < 6; e++) { fe_to_words(out29 + (i * 6 + e) * 8, *fr[e]); g_to_words(out27 + (i * 6 + e) * 8, *gs[e]); }
}
template <int OP> __global__ void __launch_bounds__(64, 2) bench(const uint8_t *in, uint32_t *out, int iters) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    const uint8_t *base = in + 32 * ((i * 12) & 1023);
    F6 h{ld2(base), ld2(base + 64), ld2(base + 128)};
    PairDpp x{(bool)(threadIdx.x & 1)};
    uint32_t acc = 0;
    if (OP == 0) {
        h = f6_norm(h);
        for (int it = 0; it < iters; it++) h = f12p_cyclo_sqr_alt(x, h);
        acc = (uint32_t)(h.b0.a0.v[0] + h.b1.a1.v[3] + h.b2.a0.v[8]);
    } else {
        G6 g = g6_from_f6(h);
        for (int it = 0; it < iters; it++) { g = sqr27(x.odd, g); if ((it & 3) == 3) g = g6_reduce(g); }       // one value reduction per run of four
        acc = (uint32_t)(g.b0.a0.v[0] + g.b1.a1.v[3] + g.b2.a0.v[9]);
    }
    if (acc == 0x12345678u) out[i & 63] = acc;
}
template <int OP> void run(const char *name, const uint8_t *din, uint32_t *dout, int ncu, int iters) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int waves = 2, grid = ncu * 4 * waves;
    bench<OP><<<grid, 64>>>(din, dout, 4); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CHECK(hipEventRecord(e0)); bench<OP><<<grid, 64>>>(din, dout, iters); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-52s %8.3f ms %9.1f cycles per wave-call (2 waves/SIMD, 2.4 GHz nominal)\n", name, best, best * 1e-3 * 2.4e9 / iters / waves);
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    uint8_t *din; uint32_t *d29, *d27;
    CHECK(hipMalloc(&din, 32 * 1024 + 4096)); CHECK(hipMalloc(&d29, 64 * 6 * 32)); CHECK(hipMalloc(&d27, 64 * 6 * 32));
    static uint8_t h[32 * 1024 + 4096]; srand(1);
    for (size_t i = 0; i < sizeof h; i++) h[i] = (i % 32 == 31) ? (rand() & 0x1f) : (rand() & 0xff);
    CHECK(hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice));
    check<<<1, 64>>>(din, d29, d27); CHECK(hipDeviceSynchronize());
    static uint32_t a[64 * 6 * 8], b[64 * 6 * 8];
    CHECK(hipMemcpy(a, d29, sizeof a, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(b, d27, sizeof b, hipMemcpyDeviceToHost));
    int bad = 0, by[12] = {0};
    for (int i = 0; i < 64 * 6; i++) if (memcmp(a + 8 * i, b + 8 * i, 32) != 0) { bad++; by[((i / 6) & 1) * 6 + i % 6]++; }
    printf("mismatches by (lane parity, coefficient half): "); for (int e = 0; e < 12; e++) printf("%d ", by[e]); printf("\n");
    printf("27-bit squaring against f12p_cyclo_sqr (9 x 29), 32 values x 12 coefficients as canonical bytes: %d mismatches\n", bad);
    run<0>("f12p_cyclo_sqr_alt (9 x 29, shipped)", din, d29, prop.multiProcessorCount, 400);
    run<1>("sqr27 (10 x 27, column-domain sums, prototype)", din, d29, prop.multiProcessorCount, 400);
    return bad ? 1 : 0;
}
