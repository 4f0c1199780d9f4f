// Micro-benchmark of the field-layer leaves (csrc/fe29.hip.hpp) on gfx950: cycles per fe_mul / fe_mul2 / fe_norm /
// fe_reduce call at 1, 2 and 4 waves per SIMD.  build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/microbench_fe.hip -o tools/microbench_fe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../gopairingbasedcryptography_amd/csrc/tower29.hip.hpp"
using namespace gpbc;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int OP>
__global__ void __launch_bounds__(64) bench(const uint8_t *in, uint8_t *out, int iters) {
    size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    Fe a = fe_load(in + 32 * (i & 1023)), b = fe_load(in + 32 * ((i + 7) & 1023));
    F2 x{a, b}, y{b, a};
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { a = fe_mul(a, b); b = fe_mul(b, a); }
        else if (OP == 1) { a = fe_mul2(a, b, b, a); b = fe_mul2(b, a, a, b); }
        else if (OP == 2) { a = fe_norm(fe_add(a, b)); b = fe_norm(fe_sub(b, a)); }
        else if (OP == 3) { a = fe_reduce(fe_norm(fe_add(a, b))); b = fe_reduce(fe_norm(fe_sub(b, a))); }
        else if (OP == 4) { x = f2_mul(x, y); y = f2_mul(y, x); }
        else if (OP == 5) { x = f2_sqr(x); y = f2_sqr(f2_norm(f2_add(y, x))); }
    }
    if (OP >= 4) { a = fe_add(x.a0, y.a1); b = fe_add(x.a1, y.a0); }
    fe_store(out + 32 * (i & 1023), fe_norm(fe_add(a, b)));
}
template <int OP> void run(const char *name, const uint8_t *din, uint8_t *dout, int ncu, int waves, double calls_per_iter) {
    int iters = 2000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int grid = ncu * 4 * waves;
    bench<OP><<<grid, 64>>>(din, dout, 10); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CHECK(hipEventRecord(e0)); bench<OP><<<grid, 64>>>(din, dout, iters); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double calls = (double)iters * calls_per_iter;                       // per wave
    double cyc = best * 1e-3 * 2.4e9 / calls / waves;                    // SIMD cycles per call per wave-slot
    printf("%-28s waves/SIMD=%d  %8.3f ms  %8.1f SIMD-cycles per wave-call (nominal 2.4 GHz)  %.1f Gcalls/s chip\n", name, waves, best,
           cyc, (double)grid * 64 * calls / (best * 1e-3) * 1e-9);
    fflush(stdout);
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    uint8_t *din, *dout; CHECK(hipMalloc(&din, 32 * 1024)); CHECK(hipMalloc(&dout, 32 * 1024));
    uint8_t h[32 * 1024]; srand(1); for (int i = 0; i < 32 * 1024; i++) h[i] = (i % 32 == 31) ? (rand() & 0x1f) : (rand() & 0xff);
    CHECK(hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice));
    for (int w : {1, 2, 4}) {
        run<0>("fe_mul (81+81 MAD)", din, dout, ncu, w, 2);
        run<1>("fe_mul2 (162+81 MAD)", din, dout, ncu, w, 2);
        run<2>("fe_add + fe_norm", din, dout, ncu, w, 2);
        run<3>("fe_add + fe_norm + fe_reduce", din, dout, ncu, w, 2);
        run<4>("f2_mul (2 fe_mul2)", din, dout, ncu, w, 2);
        run<5>("f2_sqr (2 fe_mul + norms)", din, dout, ncu, w, 2);
    }
    return 0;
}
