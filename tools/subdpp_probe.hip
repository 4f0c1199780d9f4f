// Does a subtraction with a DPP-swapped operand compute what the ISA manual says on gfx950?  y - swap(x) and swap(x) - y as the compiler folds them
// (v_subrev_u32_dpp / v_sub_u32_dpp) and as hand-written instructions, against the host's values.  build: hipcc -O3 --offload-arch=gfx950 tools/subdpp_probe.hip -o tools/subdpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ int sw(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }
__global__ void k(const int *a, const int *b, int *out) {
    int x = a[threadIdx.x], y = b[threadIdx.x];
    out[threadIdx.x] = y - sw(x);              // compiler: v_subrev_u32_dpp
    out[64 + threadIdx.x] = sw(x) - y;         // compiler: v_sub_u32_dpp
    int r;
    asm volatile("v_subrev_u32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(x), "v"(y));
    out[128 + threadIdx.x] = r;
    asm volatile("v_sub_u32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(x), "v"(y));
    out[192 + threadIdx.x] = r;
}
int main() {
    int ha[64], hb[64], ho[256], *da, *db, *dout;
    for (int i = 0; i < 64; i++) { ha[i] = 1000 + i; hb[i] = 5 * i; }
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dout, 1024);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dout);
    hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
    int bad[4] = {0, 0, 0, 0};
    for (int i = 0; i < 64; i++) {
        int p = i ^ 1;
        if (ho[i] != hb[i] - ha[p]) bad[0]++;
        if (ho[64 + i] != ha[p] - hb[i]) bad[1]++;
        if (ho[128 + i] != hb[i] - ha[p]) bad[2]++;
        if (ho[192 + i] != ha[p] - hb[i]) bad[3]++;
    }
    printf("lane 0: y - sw(x) = %d (want %d), sw(x) - y = %d (want %d), asm subrev = %d, asm sub = %d\n", ho[0], hb[0] - ha[1], ho[64], ha[1] - hb[0], ho[128], ho[192]);
    printf("compiler y - sw(x): %d wrong; compiler sw(x) - y: %d wrong; asm v_subrev_u32_dpp as S1 - dpp(S0): %d wrong; asm v_sub_u32_dpp as dpp(S0) - S1: %d wrong\n", bad[0], bad[1], bad[2], bad[3]);
    return 0;
}
