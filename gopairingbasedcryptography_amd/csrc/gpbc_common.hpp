// Shared by the translation units of libgpbc_bn254.so (gpbc_core.hip, gpbc_pairing.hip, gpbc_curve.hip, gpbc_wire.hip):
// launch-bound macros, small device-side load / store helpers, and the host-side plumbing every entry point uses
// (thread-local error text, device binding, RAII device buffers, the per-stream internal workspace).  The units are compiled
// separately (each carries its own copy of the device code it needs; no relocatable device code) and linked into one
// library.
#ifndef GPBC_COMMON_HPP
#define GPBC_COMMON_HPP
#include <hip/hip_runtime.h>
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/gpbc_bn254.h"
#include "curve29.hip.hpp"

using namespace gpbc;

// One batch element per lane.  Inputs/outputs are gnark structs (Montgomery R = 2^256, canonical); each kernel
// converts to the internal 9 x 29-bit signed-limb form on load and back to canonical bytes on store.
constexpr int BLOCK = 64;
static_assert(BLOCK == 64, "one wave per workgroup: the F6 parking slot (tower29_pair.hip.hpp) is indexed by threadIdx.x < 64; the F2 leaf's argument slots allow 128");
#define GPBC_WAVES_PER_SIMD 2
#define GPBC_KERNEL __global__ void __launch_bounds__(BLOCK, GPBC_WAVES_PER_SIMD)
// G1 arithmetic is light enough on registers for three waves per SIMD (168 VGPRs): measured +12 % over two, while the Fp2 /
// Fp12 kernels lose 15-75 % to the extra spills (profiles/r01_microbench_valu2.txt explains the gain: a wave issues
// at most one VALU instruction per ~4.5 cycles, so the 2.4-cycle VOP2 glue only gets cheaper with more waves).
#define GPBC_WAVES_G1 3
#define GPBC_KERNEL_G1 __global__ void __launch_bounds__(BLOCK, GPBC_WAVES_G1)

__device__ __forceinline__ bool g1_bytes_inf(const uint8_t *p) { return bytes_all_zero(p, 16); }
__device__ __forceinline__ bool g2_bytes_inf(const uint8_t *p) { return bytes_all_zero(p, 32); }
__device__ __forceinline__ void load_scalar(uint32_t k[8], const uint8_t *p) {
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = q[i];
}
__device__ __forceinline__ AffP<Fe> g1_load_aff(const uint8_t *p) { return AffP<Fe>{fe_load(p), fe_load(p + 32), g1_bytes_inf(p)}; }
__device__ __forceinline__ AffP<F2> g2_load_aff(const uint8_t *p) { return AffP<F2>{f2_load(p), f2_load(p + 64), g2_bytes_inf(p)}; }
__device__ __forceinline__ void g1_store_aff(uint8_t *p, const AffP<Fe> &r) { fe_store(p, r.x); fe_store(p + 32, r.y); }
__device__ __forceinline__ void g2_store_aff(uint8_t *p, const AffP<F2> &r) { f2_store(p, r.x); f2_store(p + 64, r.y); }

// ---- host side
extern thread_local char g_err[512];
int fail(int code, const char *fmt, ...);
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(GPBC_ERR_HIP, "%s failed: %s", #x, hipGetErrorString(e_)); } while (0)
#define TRY(x) do { int rc_ = (x); if (rc_ != GPBC_OK) return rc_; } while (0)
// Devices: gpbc_init_devices() binds the process to a LIST of HIP devices.  Every host thread has a current device
// (gpbc_set_device, thread-local, default = the first of the list): *_dev entries and the un-sharded host entries run
// there.  bind_device() makes it the calling thread's HIP device; current_device() is its HIP ordinal (-1 before init).
int bind_device();
int current_device();
int device_count_initialised();
// Host-pointer batch entries over n independent units: with more than one device initialised the range [0, n) is cut by
// shard_range (contiguous, sizes differ by at most one) and each part runs on its own device from its own host thread
// (body(lo, hi) is called with that thread's current device set); one device, a batch below 2 * min_units or a call made
// from inside a shard run body(0, n) on the calling thread.  Returns the first failing status (its message becomes the
// caller's gpbc_last_error()).
int run_sharded(size_t n, size_t min_units, const std::function<int(size_t, size_t)> &body);
// Host-pointer calls over many units on ONE device: chunks alternate on the slot's two compute streams while a helper thread
// drains the results on a third.  The caller's thread runs upload(c) — a blocking copy from pageable memory, during which the
// GPU works on earlier chunks — then enqueue(c) and records an event; the helper waits for that event on the download stream
// and copies chunk c back (a blocking copy again, but nobody queues behind it).  The compute streams never wait for a
// download, so the call takes about upload(first) + kernels + download(last); it returns with everything downloaded and all
// three streams idle.  current_slot / set_slot let the helper thread adopt the caller's device slot.
int pipe_streams(hipStream_t out[3]);
int current_slot();
int set_slot(int index);
template <class Up, class Run, class Down> int pipelined_chunks(size_t n, size_t chunk, Up upload, Run enqueue, Down download) {
    hipStream_t st[3];
    TRY(pipe_streams(st));
    const size_t n_chunks = (n + chunk - 1) / chunk;
    std::vector<hipEvent_t> ev(n_chunks, nullptr);
    for (auto &e : ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::mutex mu;
    std::condition_variable cv;
    size_t recorded = 0;
    bool abort_all = false;
    int rc_down = GPBC_OK;
    char err_down[512] = "";
    const int slot = current_slot();
    std::thread drain([&]() {
        if (set_slot(slot) != GPBC_OK || bind_device() != GPBC_OK) { rc_down = GPBC_ERR_HIP; snprintf(err_down, sizeof err_down, "%s", g_err); return; }
        for (size_t c = 0; c < n_chunks; c++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return recorded > c || abort_all; });
                if (recorded <= c) return;
            }
            const size_t off = c * chunk, m = n - off < chunk ? n - off : chunk;
            int rc = hipStreamWaitEvent(st[2], ev[c], 0) == hipSuccess ? download(off, m, st[2]) : fail(GPBC_ERR_HIP, "hipStreamWaitEvent failed");
            if (rc == GPBC_OK && hipStreamSynchronize(st[2]) != hipSuccess) rc = fail(GPBC_ERR_HIP, "hipStreamSynchronize (download stream) failed");
            if (rc != GPBC_OK) { rc_down = rc; snprintf(err_down, sizeof err_down, "%s", g_err); return; }
        }
    });
    int rc = GPBC_OK;
    for (size_t c = 0; c < n_chunks && rc == GPBC_OK; c++) {
        const size_t off = c * chunk, m = n - off < chunk ? n - off : chunk;
        rc = upload(off, m, st[c & 1]);
        if (rc == GPBC_OK) rc = enqueue(off, m, st[c & 1]);
        if (rc == GPBC_OK && hipEventRecord(ev[c], st[c & 1]) != hipSuccess) rc = fail(GPBC_ERR_HIP, "hipEventRecord failed");
        if (rc == GPBC_OK) { std::lock_guard<std::mutex> lk(mu); recorded = c + 1; }
        cv.notify_all();
    }
    if (rc != GPBC_OK) { std::lock_guard<std::mutex> lk(mu); abort_all = true; }
    cv.notify_all();
    drain.join();
    (void)hipStreamSynchronize(st[0]); (void)hipStreamSynchronize(st[1]); (void)hipStreamSynchronize(st[2]);
    for (auto &e : ev) if (e) (void)hipEventDestroy(e);
    if (rc != GPBC_OK) return rc;
    if (rc_down != GPBC_OK) return fail(rc_down, "%s", err_down);
    return GPBC_OK;
}
// Bucket (Pippenger) multi-scalar multiplication over variable bases (gpbc_msm.hip): sum_i [s_i] P_i, n terms in device memory,
// one affine point out; asynchronous on `st`.  Used by the scalar_mul_sum entries from MSM_MIN_TERMS terms on.
constexpr size_t MSM_MIN_TERMS = 16384;
constexpr int MSM_SKEWED = 1;                     // msm_dev: the scalars are too unevenly spread for buckets (not an error; nothing was written)
constexpr uint32_t MSM_MAX_BUCKET_BASE = 256;     // longest bucket accepted: this + 8 x the mean bucket length
int msm_dev(bool g2, const void *d_bases, const void *d_scalars, size_t n, void *d_out, hipStream_t st);
// RCCL communicator of the current device (gpbc_core.hip): number of ranks (0 = none), this device's rank, all-gather
int comm_ranks();
int comm_rank();
int comm_allgather(const void *d_send, size_t bytes, void *d_recv, hipStream_t st);
static inline unsigned grid_for(size_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }
int check_launch(const char *what);
void profile_mark(const char *kernel_name, hipStream_t st);   // bench.py's per-kernel timing (gpbc_profile_begin / _end); a no-op otherwise
int sync_default();

// RAII device buffer for the host-pointer entry points
// Device blocks of the host-pointer entries.  hipMalloc / hipFree per call cost 0.1-0.2 ms on a good day and, now and then, 10 ms and
// more (profiles/r03_size_sweep.txt: a 1.6 ms decode call measured at 12-15 ms) — so freed blocks go to a small per-device cache
// (gpbc_core.hip: size classes, at most 1 GiB kept per device, gpbc_release_workspaces() empties it) and come back from there.
// A block returns to the cache only after the device has drained, which is what hipFree's implicit synchronisation did.
int dev_block_alloc(size_t bytes, void **p, size_t *cap, int *dev);
void dev_block_free(void *p, size_t cap, int dev);
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int dev = -1;                    // the device slot the block lives on (the thread's current device at alloc())
    ~DevBuf() { if (p) dev_block_free(p, cap, dev); }
    int alloc(size_t bytes) { return dev_block_alloc(bytes ? bytes : 1, &p, &cap, &dev); }
    int upload(const void *src, size_t bytes) {
        TRY(alloc(bytes));
        if (bytes) HIP_TRY(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        return GPBC_OK;
    }
    int download(void *dst, size_t bytes) const {
        if (bytes) HIP_TRY(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost));
        return GPBC_OK;
    }
    uint8_t *u8() const { return static_cast<uint8_t *>(p); }
};

// Internal workspace: the Miller lines (88 x 54 int32 per pairing) and the per-point GLV tables of the scalar
// multiplications (2 KB / 4 KB per point).  One grow-only buffer per (device, stream) — calls on one stream are ordered and
// may share it, calls on different streams get different buffers — and batches are processed in chunks so the footprint
// stays bounded.
constexpr size_t MILLER_CHUNK = 262144;                               // pairings per chunk: 4.98 GB of lines
constexpr size_t SMUL_CHUNK = 262144;                                 // points per chunk: 0.5 GB (G1) / 1 GB (G2) of tables
constexpr size_t LINE_BYTES_PER_PAIR = (size_t)88 * 54 * sizeof(int32_t);
// Held while a call enqueues the kernels that share the stream's workspace (lines then accumulate; table build and loop
// in one kernel): two host threads launching on the same stream must not interleave such sequences.  Enqueueing is
// asynchronous, so the lock is held for microseconds.
std::mutex &ws_seq_mutex();                                             // of the calling thread's current device
#define g_ws_seq_mu (ws_seq_mutex())
int stream_workspace(hipStream_t stream, size_t bytes, int32_t **out);
// Per-call temporaries of the device-pointer entries (chunk tables and values of the multi-pairings, the fixed-Q line table, the
// MSM's keys / buckets / rows, the fixed-base window table): a second family of grow-only buffers per (device, stream) and
// nesting level — level 0 for a routine that calls no other scratch user, level 1 for its caller (scalar_mul_sum_dev around
// msm_dev).  A repeated call pays no hipMalloc / hipFree, which for the gigabytes of a BASELINE-size BSW07 decrypt were tens of
// milliseconds per call.  (hipMallocAsync pools were measured instead and dropped: a small host-to-device table copy into a
// block the pool had just recycled was not seen by the next kernel on the null stream — profiles/r02_pool_bisect.txt.)
// open() takes the device's scratch lock, which the object holds until it goes out of scope; like the workspace lock it only
// has to cover the ENQUEUE of the kernels that use the buffer, later calls on the same stream are ordered behind them.
// Lock order: scratch before workspace.
std::recursive_mutex &scratch_mutex();                                  // of the calling thread's current device
int stream_scratch(hipStream_t stream, int level, size_t bytes, void **out);
struct Scratch {
    std::unique_lock<std::recursive_mutex> lock;
    uint8_t *base = nullptr;
    size_t off = 0, cap = 0;
    static size_t padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    int open(hipStream_t stream, int level, size_t bytes) {
        lock = std::unique_lock<std::recursive_mutex>(scratch_mutex());
        void *p = nullptr;
        TRY(stream_scratch(stream, level, bytes ? bytes : 256, &p));
        base = (uint8_t *)p; off = 0; cap = bytes;
        return GPBC_OK;
    }
    template <class T = uint8_t> T *take(size_t bytes) {            // the next 256-byte aligned piece (sizes summed with padded())
        T *p = (T *)(base + off);
        off += padded(bytes);
        return p;
    }
};
// Library-owned PINNED host memory, one grow-only buffer per (device, stream): small host tables (segment / chunk offsets) are
// copied here first and travel to the device by a stream-ordered DMA from memory that only this library writes, and the kernels'
// echo of what they consumed comes back the same way.  The caller must hold the scratch lock of the stream from the moment it
// writes the buffer until it has synchronised the stream (multi_pair_core does).
int pinned_staging(hipStream_t stream, size_t bytes, uint8_t **out);
static inline int lines_workspace(hipStream_t stream, size_t pairs, int32_t **out) { return stream_workspace(stream, pairs * LINE_BYTES_PER_PAIR, out); }
void free_workspaces();

// ---- Small host-pointer calls: the reference's call shape.  Every reference call site is ONE bn254.Pair of one pair, one
// PairingCheck of two, one ScalarMultiplication (access/tree/access_tree_node.go:106-123, signature/bls01_signature/bls_signature.go:45,81)
// and a cgo replacement is called like that from many OS threads at once.  A lone such call is a dependent chain
// on a handful of wavefronts; T of them side by side cost the same GPU time as one, so calls that arrive while the device is
// busy are COMBINED: a caller enqueues its request and, if one of the device's call lanes is free, becomes the leader — it takes
// every request of its kind that is waiting (its own among them), copies the inputs into the lane's pinned block, launches ONE
// batch on the lane's stream, waits for that stream alone and hands each caller its slice.  Callers that arrive while all lanes
// are busy sleep until a leader has served them or a lane is free.  No timer, no helper thread: a lone call runs at once on the
// calling thread (nothing to wait for), a crowd rides in batches as large as the arrival rate makes them.
// A lane = a non-blocking stream + a pinned host block that is mapped into the device (kernels read the inputs from it and
// write the results into it: no copy engine on the path of a 192-byte call) + a device block for intermediates.  Nothing here
// touches the null stream, the device-wide scratch lock or hipDeviceSynchronize.
constexpr int CALL_LANES = 4;
constexpr size_t SMALL_CALL_MAX_UNITS = 2048;       // per batch (and per call): one round of wavefronts on the chip
enum SmallKind { CALL_PAIRS = 0, CALL_G1_MUL, CALL_G2_MUL, CALL_GT_EXP, CALL_GT_MUL, CALL_GT_DIV, CALL_GT_INV, CALL_HASH_G1, CALL_HASH_G2, CALL_FIXED_BASE, CALL_KINDS };
struct CallLane {
    int device = -1;                 // index into the bound device list
    hipStream_t stream = nullptr;
    uint8_t *pin = nullptr, *d_pin = nullptr;       // the pinned block: host address / the device's address of the same bytes
    size_t pin_bytes = 0;
    uint8_t *dev = nullptr;
    size_t dev_bytes = 0;
    bool busy = false;
    int reserve(size_t pin_need, size_t dev_need);  // grow-only; called by the batch runner before it writes anything
};
struct SmallCall {
    const void *in[3] = {nullptr, nullptr, nullptr};   // caller's input columns (meaning per kind)
    bool in_one[3] = {false, false, false};            // column holds ONE element for all units (a shared base)
    void *out[2] = {nullptr, nullptr};
    size_t units = 0;                                  // pairs / points / GT elements
    const uint64_t *seg = nullptr;                     // CALL_PAIRS: the call's own segment table (segs + 1 entries); null = one pair per segment
    size_t segs = 0;                                   //   (CALL_HASH_*: seg = the call's message offsets, units + 1 entries)
    // Calls of one kind share a launch only if their keys are equal byte for byte: the domain-separation tag of a hash (a kernel
    // argument), the table handle of a fixed-base sum.  No key: every call of the kind combines.
    const void *key = nullptr;
    size_t key_len = 0;
    int rc = GPBC_OK;
    char err[512] = "";
    bool taken = false, done = false;
    std::condition_variable cv;
};
using SmallBatchFn = int (*)(CallLane &lane, SmallCall *const *calls, size_t n_calls);
// Runs `c` through the combiner of the calling thread's current device; returns the call's status with gpbc_last_error() set.
int small_call(SmallKind kind, SmallCall &c, SmallBatchFn run);
// The same lanes for small calls that are NOT combined (wire formats, hash to curve, fixed-base sums: one call = one launch): the caller
// takes a free lane (waits for one), stages through its pinned block, launches on its stream, waits for that stream alone — off the null
// stream and without the device-wide synchronisation a DevBuf costs when it is freed, so such a call neither waits for the batches of
// other threads nor makes them wait.  body runs on the calling thread with the lane's buffers reserved by itself (lane.reserve).
int with_call_lane(const std::function<int(CallLane &)> &body);
constexpr size_t LANE_CALL_MAX_UNITS = 16384;       // elements per such call (the quad-of-lanes kernels' range); larger calls keep the bulk path
void free_call_lanes();

#endif
