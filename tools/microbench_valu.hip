// Dependency-controlled VALU issue-rate micro-benchmark for gfx950 (MI355X).
// Defines the denominator of the VALU roofline used by bench.py: the peak rate of
// v_mad_u64_u32 (one 32x32+64 MAC per lane), next to the alternatives a big-integer multiplier could
// be built from (v_mul_lo/hi_u32, 24-bit multiplies, v_fma_f64) and the carry-chain adds.
//
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench_valu.hip -o tools/microbench_valu
// run:   tools/microbench_valu            (prints one line per instruction / occupancy / ILP)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned long long u64;
typedef unsigned int u32;

enum Op { MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, MAD_U32_U24, MUL_U32_U24, MUL_HI_U32_U24, FMA_F64, FMA_F32,
          ADD_U32, ADD_CO_PAIR, ADD3_U32, MAD_U64_THEN_ADD, LSHL_ADD_U64, N_OPS };
static const char *OP_NAME[N_OPS] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_u32_u24",
                                     "v_mul_hi_u32_u24", "v_fma_f64", "v_fma_f32", "v_add_u32", "v_add_co+v_addc_co (pair)",
                                     "v_add3_u32", "v_mad_u64_u32+v_add_u32 (pair)", "v_lshl_add_u64"};
// instructions counted per "op" (pairs count 2)
static const int OP_INSTR[N_OPS] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 2, 1};

template <int OP>
__device__ __forceinline__ void step(u64 &acc, u32 &x, u32 a, u32 b, double &d, double da, double db, float &f) {
    if constexpr (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
    else if constexpr (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == MUL_U32_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == MUL_HI_U32_U24) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == FMA_F64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d) : "v"(da), "v"(db));
    else if constexpr (OP == FMA_F32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f) : "v"(a), "v"(b));
    else if constexpr (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == ADD_CO_PAIR) {
        u32 lo = (u32)acc, hi = (u32)(acc >> 32);
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
        acc = ((u64)hi << 32) | lo;
    } else if constexpr (OP == ADD3_U32) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == MAD_U64_THEN_ADD) {
        asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %1, %2" : "+v"(acc), "+v"(x) : "v"(a), "v"(b) : "vcc");
    } else if constexpr (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc) : "v"(d));
}

template <int OP, int CHAINS>
__global__ void __launch_bounds__(256) bench_kernel(u64 *out, int iters, u32 a, u32 b) {
    u64 acc[CHAINS]; u32 x[CHAINS]; double d[CHAINS]; float f[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { acc[c] = threadIdx.x + c; x[c] = threadIdx.x * 7 + c; d[c] = 1.0 + c; f[c] = 1.0f + c; }
    double da = 1.0000001 + a * 1e-9, db = 1e-9 * b;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) step<OP>(acc[c], x[c], a, b, d[c], da, db, f[c]);
        }
    }
    u64 r = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) r += acc[c] + x[c] + (u64)d[c] + (u64)f[c];
    if (r == 0x123456789abcdefULL) out[0] = r;   // keep everything live
}

template <int OP, int CHAINS>
static void run(int waves_per_simd, u64 *dout, int ncu, double clk_ghz) {
    const int iters = 4096;
    dim3 block(256), grid(ncu * waves_per_simd);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    bench_kernel<OP, CHAINS><<<grid, block>>>(dout, 16, 3, 5);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        bench_kernel<OP, CHAINS><<<grid, block>>>(dout, iters, 3, 5);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double ops = (double)grid.x * 256 * iters * 8 * CHAINS;           // per-lane "ops"
    double rate = ops / (best * 1e-3);                                  // lane-ops per second
    double per_clk_cu = rate / (ncu * clk_ghz * 1e9);                   // lane-ops / clk / CU
    printf("%-32s waves/SIMD=%d chains=%d  %8.3f ms  %8.3f Tlane-op/s  %7.2f lane-op/clk/CU  (cyc per wave-instr per SIMD: %.2f)\n",
           OP_NAME[OP], waves_per_simd, CHAINS, best, rate * 1e-12, per_clk_cu,
           64.0 * 4.0 * OP_INSTR[OP] / (per_clk_cu * OP_INSTR[OP]));
    fflush(stdout);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

template <int OP>
static void sweep(u64 *dout, int ncu, double clk) {
    run<OP, 1>(1, dout, ncu, clk);
    run<OP, 4>(1, dout, ncu, clk);
    run<OP, 8>(1, dout, ncu, clk);
    run<OP, 1>(2, dout, ncu, clk);
    run<OP, 4>(2, dout, ncu, clk);
    run<OP, 8>(2, dout, ncu, clk);
    run<OP, 4>(4, dout, ncu, clk);
    run<OP, 4>(8, dout, ncu, clk);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e-6;
    printf("device %s  arch %s  CUs %d  clock %.3f GHz  (rates below use this nominal clock)\n", prop.name, prop.gcnArchName, ncu, clk);
    u64 *dout; CHECK(hipMalloc(&dout, 8));
    sweep<MAD_U64_U32>(dout, ncu, clk);
    sweep<MUL_LO_U32>(dout, ncu, clk);
    sweep<MUL_HI_U32>(dout, ncu, clk);
    sweep<MAD_U32_U24>(dout, ncu, clk);
    sweep<MUL_U32_U24>(dout, ncu, clk);
    sweep<MUL_HI_U32_U24>(dout, ncu, clk);
    sweep<FMA_F64>(dout, ncu, clk);
    sweep<FMA_F32>(dout, ncu, clk);
    sweep<ADD_U32>(dout, ncu, clk);
    sweep<ADD_CO_PAIR>(dout, ncu, clk);
    sweep<ADD3_U32>(dout, ncu, clk);
    sweep<MAD_U64_THEN_ADD>(dout, ncu, clk);
    sweep<LSHL_ADD_U64>(dout, ncu, clk);
    CHECK(hipFree(dout));
    return 0;
}
