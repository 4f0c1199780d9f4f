// Second VALU issue-rate micro-benchmark for gfx950: the "cheap" instructions the limb glue is made of (adds, masks,
// shifts, selects, DPP moves) in their different encodings (VOP1/VOP2 32-bit, VOP2 + 32-bit literal, VOP3 64-bit, DPP),
// at 1 / 2 / 4 waves per SIMD with 8 independent chains, and mixed 1:1 with v_mad_i64_i32.
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench_valu2.hip -o tools/microbench_valu2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned long long u64;
typedef unsigned int u32;

enum Op { ADD_VOP2, ADD_LIT, AND_LIT, AND_SGPR, AND_VGPR, ASHR_INL, LSHL_INL, SUB_VOP2, MOV_VOP1, MOV_DPP, ADD_DPP, CND_E32, CND_E64,
          BFE_I32, ALIGNBIT, ADD3, LSHL_ADD, AND_OR, MAD_I64, MAD_PLUS_ADD, MAD_PLUS_ANDLIT, MAD_PLUS_ADD3, NORM_TRIPLE, NORM_TRIPLE_SGPR, CND_E32_VCCSET, CND_E64_VCC, CND_E32_AFTER_CMP, BFI_B32, N_OPS };
static const char *NAME[N_OPS] = {"v_add_u32 (VOP2)", "v_add_u32 + literal", "v_and_b32 + literal", "v_and_b32 sgpr mask", "v_and_b32 vgpr mask",
    "v_ashrrev_i32 inline 29", "v_lshlrev_b32 inline", "v_sub_u32 (VOP2)", "v_mov_b32 (VOP1)", "v_mov_b32_dpp quad_perm", "v_add_u32_dpp quad_perm",
    "v_cndmask_b32_e32 (vcc)", "v_cndmask_b32_e64 (sgpr pair)", "v_bfe_i32 (VOP3)", "v_alignbit_b32 (VOP3)", "v_add3_u32 (VOP3)",
    "v_lshl_add_u32 (VOP3)", "v_and_or_b32 (VOP3)", "v_mad_i64_i32", "mad + v_add_u32", "mad + v_and literal", "mad + v_add3",
    "ashr + and-literal + add", "ashr + and-sgpr + add", "v_cndmask_b32_e32, vcc set by s_mov", "v_cndmask_b32_e64 with vcc", "v_cmp + 8 x v_cndmask_b32_e32",
    "v_bfi_b32 (VOP3)"};
static const int INSTR[N_OPS] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 3, 3, 1, 1, 1, 1};

template <int OP>
__device__ __forceinline__ void step(u64 &acc, u32 &x, u32 &y, u32 a, u32 b, u32 smask) {
    if constexpr (OP == ADD_VOP2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == ADD_LIT) asm volatile("v_add_u32 %0, 0x12345678, %0" : "+v"(x));
    else if constexpr (OP == AND_LIT) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(x));
    else if constexpr (OP == AND_SGPR) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "s"(smask));
    else if constexpr (OP == AND_VGPR) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "v"(a));
    else if constexpr (OP == ASHR_INL) asm volatile("v_ashrrev_i32 %0, 29, %0" : "+v"(x));
    else if constexpr (OP == LSHL_INL) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x));
    else if constexpr (OP == SUB_VOP2) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == MOV_VOP1) asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %0" : "+v"(x), "+v"(y));
    else if constexpr (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x));
    else if constexpr (OP == ADD_DPP) asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x) : "v"(a));
    else if constexpr (OP == CND_E32) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : );
    else if constexpr (OP == CND_E64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "s"((u64)smask));
    else if constexpr (OP == BFE_I32) asm volatile("v_bfe_i32 %0, %0, 0, 29" : "+v"(x));
    else if constexpr (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(x) : "v"(a));
    else if constexpr (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == MAD_I64) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
    else if constexpr (OP == MAD_PLUS_ADD) asm volatile("v_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %1, %2" : "+v"(acc), "+v"(x) : "v"(a), "v"(b) : "vcc");
    else if constexpr (OP == MAD_PLUS_ANDLIT) asm volatile("v_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_and_b32 %1, 0x1fffffff, %1" : "+v"(acc), "+v"(x) : "v"(a), "v"(b) : "vcc");
    else if constexpr (OP == MAD_PLUS_ADD3) asm volatile("v_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_add3_u32 %1, %1, %2, %3" : "+v"(acc), "+v"(x) : "v"(a), "v"(b) : "vcc");
    else if constexpr (OP == NORM_TRIPLE) asm volatile("v_ashrrev_i32 %1, 29, %0\n\tv_and_b32 %0, 0x1fffffff, %0\n\tv_add_u32 %0, %0, %1" : "+v"(x), "+v"(y));
    else if constexpr (OP == CND_E32_VCCSET) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : );
    else if constexpr (OP == CND_E64_VCC) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : );
    else if constexpr (OP == CND_E32_AFTER_CMP) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : );
    else if constexpr (OP == BFI_B32) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == NORM_TRIPLE_SGPR) asm volatile("v_ashrrev_i32 %1, 29, %0\n\tv_and_b32 %0, %2, %0\n\tv_add_u32 %0, %0, %1" : "+v"(x), "+v"(y) : "s"(smask));
}

template <int OP, int CHAINS>
__global__ void __launch_bounds__(256) bench_kernel(u64 *out, int iters, u32 a, u32 b, u32 smask) {
    u64 acc[CHAINS]; u32 x[CHAINS], y[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { acc[c] = threadIdx.x + c; x[c] = threadIdx.x * 7 + c; y[c] = c; }
    if constexpr (OP == CND_E32_VCCSET || OP == CND_E64_VCC) asm volatile("s_mov_b64 vcc, 0x55" ::: "vcc");
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if constexpr (OP == CND_E32_AFTER_CMP) asm volatile("v_cmp_gt_u32 vcc, %0, %1" :: "v"(x[0]), "v"(a) : "vcc");
#pragma unroll
            for (int c = 0; c < CHAINS; c++) step<OP>(acc[c], x[c], y[c], a, b, smask);
        }
    }
    u64 r = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) r += acc[c] + x[c] + y[c];
    if (r == 0x123456789abcdefULL) out[0] = r;
}

template <int OP>
static void run(int waves_per_simd, u64 *dout, int ncu) {
    const int iters = 2048, CH = 8;
    dim3 block(256), grid(ncu * waves_per_simd);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    bench_kernel<OP, CH><<<grid, block>>>(dout, 16, 3, 5, 0x1fffffff); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0)); bench_kernel<OP, CH><<<grid, block>>>(dout, iters, 3, 5, 0x1fffffff); CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    int mult = (OP == MOV_VOP1) ? 2 : INSTR[OP];
    double instrs = (double)iters * 8 * CH * mult * waves_per_simd;       // wave-instructions per SIMD
    printf("%-32s waves/SIMD=%d  %8.3f ms   %.2f cycles per wave-instruction per SIMD\n", NAME[OP], waves_per_simd, best, best * 1e-3 * 2.4e9 / instrs);
    fflush(stdout);
}
// the same instruction in fewer independent chains per wave (1 = every MAD waits for the one before it)
template <int OP, int CH>
static void run_chains(int waves_per_simd, u64 *dout, int ncu) {
    const int iters = 2048;
    dim3 block(256), grid(ncu * waves_per_simd);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    bench_kernel<OP, CH><<<grid, block>>>(dout, 16, 3, 5, 0x1fffffff); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0)); bench_kernel<OP, CH><<<grid, block>>>(dout, iters, 3, 5, 0x1fffffff); CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double instrs = (double)iters * 8 * CH * INSTR[OP] * waves_per_simd;
    printf("%-22s %d chain(s) per wave  waves/SIMD=%d  %8.3f ms   %.2f cycles per wave-instruction per SIMD\n", NAME[OP], CH, waves_per_simd, best, best * 1e-3 * 2.4e9 / instrs);
    fflush(stdout);
}
template <int OP> static void sweep(u64 *dout, int ncu) { run<OP>(1, dout, ncu); run<OP>(2, dout, ncu); run<OP>(4, dout, ncu); }
template <int OP> static void all(u64 *dout, int ncu) { sweep<OP>(dout, ncu); if constexpr (OP + 1 < N_OPS) all<OP + 1>(dout, ncu); }

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  arch %s  CUs %d  (cycles at the nominal 2.4 GHz)\n", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    u64 *dout; CHECK(hipMalloc(&dout, 8));
    all<0>(dout, prop.multiProcessorCount);
    for (int w : {1, 2, 4}) { run_chains<MAD_I64, 1>(w, dout, prop.multiProcessorCount); run_chains<MAD_I64, 2>(w, dout, prop.multiProcessorCount); run_chains<MAD_I64, 4>(w, dout, prop.multiProcessorCount); }
    for (int w : {1, 2, 4}) { run_chains<MAD_PLUS_ADD, 1>(w, dout, prop.multiProcessorCount); run_chains<MAD_PLUS_ADD, 2>(w, dout, prop.multiProcessorCount); }
    return 0;
}
