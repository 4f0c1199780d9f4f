// Diagnostic for the round-2 silent accept (profiles/r02_pool_bisect.txt): which operation of
//     block = hipMallocAsync(st); hipMemcpyAsync(block, small host table, H2D, st); kernel<<<st>>>(block); hipFreeAsync(block, st); sync
// is not ordered the way the multi-pairing assumed?  Every variant repeats the call shape of multi_pair_host_one (plain
// hipMalloc / hipMemcpy / hipFree of the point buffers around it) eight times with a different 16-byte table each time; the kernel
// echoes the table it read and the host compares.  One process, one run; prints one line per variant.
// build: hipcc -O3 --offload-arch=gfx950 tools/pool_order_probe.hip -o tools/pool_order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_echo(const uint64_t *tab, uint64_t *out) {
    if (threadIdx.x < 2) out[threadIdx.x] = tab[threadIdx.x];
}
__global__ void k_touch(uint8_t *p, size_t n) {            // stands in for the Miller kernels: reads and writes the point buffers
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (uint8_t)(p[i] + 1);
}

enum Alloc { POOL, PLAIN_REUSED, PLAIN_FRESH };
enum Src { PAGEABLE_STACK, PAGEABLE_HEAP, PINNED };
enum Copy { ASYNC, SYNC, ASYNC_THEN_STREAM_SYNC };

static int run(const char *name, Alloc alloc, Src src, Copy copy, bool own_stream, bool release_threshold) {
    hipStream_t st = nullptr;
    if (own_stream) CHECK(hipStreamCreate(&st));
    if (release_threshold) {
        hipMemPool_t pool; CHECK(hipDeviceGetDefaultMemPool(&pool, 0));
        uint64_t thr = ~0ull; CHECK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
    }
    uint64_t *reused = nullptr, *echo = nullptr, *pinned = nullptr, *heap = (uint64_t *)malloc(16);
    CHECK(hipMalloc(&reused, 512)); CHECK(hipMalloc(&echo, 16)); CHECK(hipHostMalloc(&pinned, 16));
    int bad = 0; char detail[256] = "";
    for (int it = 0; it < 8; it++) {
        uint64_t stack_tab[2];
        uint64_t *host = src == PAGEABLE_STACK ? stack_tab : src == PAGEABLE_HEAP ? heap : pinned;
        host[0] = 0; host[1] = (uint64_t)(it + 2);
        uint8_t *dP, *dQ; CHECK(hipMalloc(&dP, 128)); CHECK(hipMalloc(&dQ, 256));
        uint8_t hp[256] = {0}; CHECK(hipMemcpy(dP, hp, 128, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dQ, hp, 256, hipMemcpyHostToDevice));
        uint64_t *tab = nullptr;
        if (alloc == POOL) CHECK(hipMallocAsync((void **)&tab, 512, st));
        else if (alloc == PLAIN_FRESH) CHECK(hipMalloc(&tab, 512));
        else tab = reused;
        if (copy == SYNC) CHECK(hipMemcpy(tab, host, 16, hipMemcpyHostToDevice));
        else CHECK(hipMemcpyAsync(tab, host, 16, hipMemcpyHostToDevice, st));
        if (copy == ASYNC_THEN_STREAM_SYNC) CHECK(hipStreamSynchronize(st));
        k_touch<<<1, 256, 0, st>>>(dQ, 256);
        k_echo<<<1, 64, 0, st>>>(tab, echo);
        k_touch<<<1, 128, 0, st>>>(dP, 128);
        if (alloc == POOL) CHECK(hipFreeAsync(tab, st));
        CHECK(hipStreamSynchronize(st));
        uint64_t got[2] = {~0ull, ~0ull};
        CHECK(hipMemcpy(got, echo, 16, hipMemcpyDeviceToHost));
        if (alloc == PLAIN_FRESH) CHECK(hipFree(tab));
        CHECK(hipFree(dP)); CHECK(hipFree(dQ));
        if (got[0] != 0 || got[1] != (uint64_t)(it + 2)) {
            if (!bad) snprintf(detail, sizeof detail, " first at call %d: kernel read {%llu, %llu}, host table {0, %d}", it + 1, (unsigned long long)got[0], (unsigned long long)got[1], it + 2);
            bad++;
        }
    }
    printf("%-86s %s%s\n", name, bad ? "STALE" : "ok", detail);
    fflush(stdout);
    CHECK(hipFree(reused)); CHECK(hipFree(echo)); CHECK(hipHostFree(pinned)); free(heap);
    if (own_stream) CHECK(hipStreamDestroy(st));
    return bad;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int rt = 0; CHECK(hipRuntimeGetVersion(&rt));
    printf("device %s, HIP runtime %d\n", prop.gcnArchName, rt);
    run("plain reused block, pageable stack table, hipMemcpyAsync, null stream (the library today)", PLAIN_REUSED, PAGEABLE_STACK, ASYNC, false, false);
    run("plain fresh block,  pageable stack table, hipMemcpyAsync, null stream", PLAIN_FRESH, PAGEABLE_STACK, ASYNC, false, false);
    run("pool block, pageable stack table, hipMemcpyAsync, null stream (round 2's failing form)", POOL, PAGEABLE_STACK, ASYNC, false, false);
    run("pool block, pageable heap table,  hipMemcpyAsync, null stream", POOL, PAGEABLE_HEAP, ASYNC, false, false);
    run("pool block, pinned table,         hipMemcpyAsync, null stream", POOL, PINNED, ASYNC, false, false);
    run("pool block, pageable stack table, hipMemcpy (synchronous), null stream", POOL, PAGEABLE_STACK, SYNC, false, false);
    run("pool block, pageable stack table, hipMemcpyAsync + hipStreamSynchronize, null stream", POOL, PAGEABLE_STACK, ASYNC_THEN_STREAM_SYNC, false, false);
    run("pool block, pageable stack table, hipMemcpyAsync, created stream", POOL, PAGEABLE_STACK, ASYNC, true, false);
    run("pool block, pinned table,         hipMemcpyAsync, created stream", POOL, PINNED, ASYNC, true, false);
    run("pool block, pageable stack table, hipMemcpyAsync, null stream, release threshold raised", POOL, PAGEABLE_STACK, ASYNC, false, true);
    return 0;
}
