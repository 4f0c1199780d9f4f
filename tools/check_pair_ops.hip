// Device-side cross-check of the lane-pair Fp12 primitives (csrc/tower29_pair.hip.hpp) against the single-lane tower
// (csrc/tower29.hip.hpp) computed by the same lanes: catches what the host interval harness cannot — code generation on the device
// (DPP folding, LDS slots): round 3 found v_subrev_u32_dpp computing dpp(src1) - src0 on gfx950 this way (tools/subdpp_probe.hip,
// gopairingbasedcryptography_amd/_build.py: _check_isa).  Prints one line per primitive; exit code = number of failing primitives.  build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -I <csrc> tools/check_pair_ops.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "tower29_pair.hip.hpp"
using namespace gpbc;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ F2 ld2(const uint8_t *p) { return F2{fe_load(p), fe_load(p + 32)}; }
__device__ __forceinline__ F6 ld6(const uint8_t *p) { return F6{ld2(p), ld2(p + 64), ld2(p + 128)}; }
__device__ bool same6(const F6 &a, const F6 &b) {
    uint32_t wa[8], wb[8];
    const Fe *fa[6] = {&a.b0.a0, &a.b0.a1, &a.b1.a0, &a.b1.a1, &a.b2.a0, &a.b2.a1}, *fb[6] = {&b.b0.a0, &b.b0.a1, &b.b1.a0, &b.b1.a1, &b.b2.a0, &b.b2.a1};
    bool ok = true;
    for (int e = 0; e < 6; e++) { fe_to_words(wa, fe_reduce(fe_norm(*fa[e]))); fe_to_words(wb, fe_reduce(fe_norm(*fb[e]))); for (int i = 0; i < 8; i++) ok = ok && wa[i] == wb[i]; }
    return ok;
}
template <int OP> __global__ void __launch_bounds__(64, 2) k(const uint8_t *in, int *bad) {
    const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x, i = lane >> 1;
    PairDpp x{(bool)(lane & 1)};
    const uint8_t *base = in + 768 * (i & 63);
    F12 a{ld6(base), ld6(base + 192)}, b{ld6(base + 384), ld6(base + 576)};
    F6 ha = x.odd ? a.c1 : a.c0, hb = x.odd ? b.c1 : b.c0, got, want;
    if (OP == 0) { got = f12p_mul(x, ha, hb); F12 z = f12_mul(a, b); want = x.odd ? z.c1 : z.c0; }
    if (OP == 1) { got = f12p_sqr(x, ha); F12 z = f12_sqr(a); want = x.odd ? z.c1 : z.c0; }
    if (OP == 2) { got = f12p_mul_034(x, ha, b.c0.b0, b.c0.b1, b.c0.b2); F12 z = f12_mul_034(a, b.c0.b0, b.c0.b1, b.c0.b2); want = x.odd ? z.c1 : z.c0; }
    if (OP == 3) { got = f12p_cyclo_sqr<true>(x, ha); F12 z = f12_cyclo_sqr(a); want = x.odd ? z.c1 : z.c0; }      // the Granger-Scott FORMULA on any input
    if (OP == 4) { got = f12p_cyclo_sqr_alt(x, f6_norm(ha)); F12 z = f12_cyclo_sqr(a); want = x.odd ? f6_neg(z.c1) : z.c0; }   // conjugate of the square
    if (OP == 5) { bool fl = false; got = f12p_cyclo_sqr_run(x, ha, 2, fl); F12 z = f12_cyclo_sqr(f12_cyclo_sqr(a)); want = x.odd ? z.c1 : z.c0; if (fl) got = f6_neg(got); }
    if (OP == 6) { got = f12p_mul_34(x, ha, b.c0.b1, b.c0.b2); F12 z = f12_mul_034(a, f2_one(), b.c0.b1, b.c0.b2); want = x.odd ? z.c1 : z.c0; }
    if (!same6(got, want)) atomicAdd(bad, 1);
}
static int g_failed = 0;
template <int OP> void run(const char *name, const uint8_t *din, int *dbad) {
    CHECK(hipMemset(dbad, 0, 4));
    k<OP><<<64, 64>>>(din, dbad);
    int bad = -1;
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost));
    printf("%-40s %s (%d of 4096 lanes differ)\n", name, bad ? "MISMATCH" : "ok", bad);
    if (bad) g_failed++;
}
int main() {
    static uint8_t h[768 * 64];
    srand(7);
    for (size_t i = 0; i < sizeof h; i++) h[i] = (i % 32 == 31) ? (rand() & 0x1f) : (rand() & 0xff);
    uint8_t *din; int *dbad;
    CHECK(hipMalloc(&din, sizeof h)); CHECK(hipMalloc(&dbad, 4));
    CHECK(hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice));
    run<0>("f12p_mul", din, dbad);
    run<1>("f12p_sqr", din, dbad);
    run<2>("f12p_mul_034", din, dbad);
    run<3>("f12p_cyclo_sqr", din, dbad);
    run<4>("f12p_cyclo_sqr_alt", din, dbad);
    run<5>("f12p_cyclo_sqr_run(2)", din, dbad);
    run<6>("f12p_mul_34", din, dbad);
    return g_failed;
}
