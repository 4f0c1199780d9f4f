// Micro-benchmark of the lane-pair Fp12 primitives (csrc/tower29_pair.hip.hpp) on gfx950: SIMD cycles per wave-call of
// f12p_cyclo_sqr / f12p_mul / f12p_sqr / f12p_mul_034 and of the F2 leaves at 2 waves per SIMD, beside the cycles their
// MAD instructions alone would take (4.5 cycles per wave-instruction, profiles/r01_microbench_valu.txt).
// build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -I gopairingbasedcryptography_amd/csrc tools/microbench_pair.hip -o tools/microbench_pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "tower29_pair.hip.hpp"      // build with -I <csrc dir>: the tree's or a variant copy's
using namespace gpbc;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ F2 ld2(const uint8_t *p) { return F2{fe_load(p), fe_load(p + 32)}; }

template <int OP>
__global__ void __launch_bounds__(64, 2) bench(const uint8_t *in, uint8_t *out, int iters) {
    size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    const uint8_t *base = in + 32 * ((i * 12) & 1023);
    F6 h{ld2(base), ld2(base + 64), ld2(base + 128)};
    F6 g{ld2(base + 192), ld2(base + 256), ld2(base + 320)};
    PairDpp x{(bool)(threadIdx.x & 1)};
    for (int it = 0; it < iters; it++) {
        if (OP == 0) h = f12p_cyclo_sqr<true>(x, h);
        else if (OP == 1) h = f12p_mul(x, h, g);
        else if (OP == 2) h = f6_reduce(f12p_sqr(x, h));
        else if (OP == 3) h = f12p_mul_034(x, h, g.b0, g.b1, g.b2);
        else if (OP == 4) { h.b0 = f2_mul(h.b0, g.b0); h.b1 = f2_mul(h.b1, g.b1); h.b2 = f2_mul(h.b2, g.b2); }
        else if (OP == 5) { h.b0 = f2_sqr(h.b0); h.b1 = f2_sqr(h.b1); h.b2 = f2_sqr(h.b2); }
        else if (OP == 6) { h = f6_reduce(f6_norm(f6_add(h, g))); }
        else if (OP == 7) { h = f6_norm(f6_add(h, x.swap(h))); }
        // single-lane forms (tower29.hip.hpp): one whole Fp12 value per lane, 64 values per wave-call instead of 32
        else if (OP == 8) { F12 z = f12_cyclo_sqr(F12{h, g}); h = z.c0; g = z.c1; }
        else if (OP == 9) { F12 z = f12_mul(F12{h, g}, F12{g, h}); h = z.c0; g = z.c1; }
        else if (OP == 10) { F12 z = f12_sqr(F12{h, g}); h = f6_reduce(z.c0); g = f6_reduce(z.c1); }
        else if (OP == 11) h = f12p_cyclo_sqr_alt(x, h);          // the alternating-sign form used inside runs (x^u)
    }
    uint8_t *o = out + 384 * (i & 63);
    if (OP >= 8) h = f6_norm(f6_add(h, g));
    fe_store(o, fe_reduce(fe_norm(fe_add(fe_add(h.b0.a0, h.b1.a1), fe_add(h.b2.a0, fe_add(h.b0.a1, fe_add(h.b1.a0, h.b2.a1)))))));
}
template <int OP> void run(const char *name, const uint8_t *din, uint8_t *dout, int ncu, int waves, int iters, double mads) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int grid = ncu * 4 * waves;
    bench<OP><<<grid, 64>>>(din, dout, 4); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CHECK(hipEventRecord(e0)); bench<OP><<<grid, 64>>>(din, dout, iters); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double cyc = best * 1e-3 * 2.4e9 / iters / waves;                    // SIMD cycles per call per wave
    printf("%-34s waves/SIMD=%d %8.3f ms %9.1f cycles/wave-call   MAD-only %8.1f   MAD share %5.1f %%\n", name, waves, best, cyc, mads * 4.5,
           100.0 * mads * 4.5 / cyc);
    fflush(stdout);
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    uint8_t *din, *dout; CHECK(hipMalloc(&din, 32 * 1024 + 4096)); CHECK(hipMalloc(&dout, 384 * 64));
    static uint8_t h[32 * 1024 + 4096]; srand(1);
    for (size_t i = 0; i < sizeof h; i++) h[i] = (i % 32 == 31) ? (rand() & 0x1f) : (rand() & 0xff);
    CHECK(hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice));
    for (int w : {2, 1}) {
        run<0>("f12p_cyclo_sqr (4.5 F2 sqr)", din, dout, ncu, w, 400, 4.5 * 324);
        run<11>("f12p_cyclo_sqr_alt (run form)", din, dout, ncu, w, 400, 4.5 * 324);
        run<1>("f12p_mul (9 F2 mul)", din, dout, ncu, w, 200, 9 * 486);
        run<2>("f12p_sqr (6 F2 mul) + reduce", din, dout, ncu, w, 200, 6 * 486);
        run<3>("f12p_mul_034 (8 F2 mul)", din, dout, ncu, w, 200, 8 * 486);
        run<4>("3 x f2_mul leaf", din, dout, ncu, w, 400, 3 * 486);
        run<5>("3 x f2_sqr leaf", din, dout, ncu, w, 400, 3 * 324);
        run<6>("f6 add+norm+reduce", din, dout, ncu, w, 2000, 0);
        run<7>("f6 swap+add+norm", din, dout, ncu, w, 2000, 0);
        run<8>("1-lane f12_cyclo_sqr (9 F2 sqr)", din, dout, ncu, w, 200, 9 * 324);
        run<9>("1-lane f12_mul (18 F2 mul)", din, dout, ncu, w, 100, 18 * 486);
        run<10>("1-lane f12_sqr (12 F2 mul)+reduce", din, dout, ncu, w, 100, 12 * 486);
    }
    return 0;
}
