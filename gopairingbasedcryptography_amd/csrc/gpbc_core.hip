// libgpbc_bn254.so, unit 1 of 4: process-wide state and lifetime entries of the C ABI (include/gpbc_bn254.h): the list of
// bound devices with a thread-local current device, host-side sharding of batch entries over the devices, the RCCL
// communicator(s) and the all-gather entries, the per-(device, stream) internal workspace, and the field-level test
// entry.  gfx950 only.
#include "gpbc_common.hpp"
#include <cstdlib>
#include <deque>
#include <dlfcn.h>
#include <exception>
#include <rccl/rccl.h>      // types and enums only: the library is opened with dlopen when a communicator is first asked for
#include <string>
#include <thread>

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

// ---- devices.  Slots are written under g_dev_mu by init / shutdown and read lock-free afterwards (g_ndev is published
// last); a slot's mutex and communicator live as long as the process.
constexpr int MAX_DEVICES = 64;
struct DeviceCtx {
    int hip = -1;                 // HIP ordinal
    std::mutex ws_seq;            // held while a call enqueues kernels that share this device's stream workspace
    std::recursive_mutex scratch; // held while a call enqueues kernels that use this device's stream scratch (gpbc_common.hpp)
    ncclComm_t comm = nullptr;    // RCCL communicator of this device (rank = comm_rank of comm_ranks), or null
    hipStream_t pipe[3] = {nullptr, nullptr, nullptr};   // large host-pointer calls: two compute streams + one download stream
    std::mutex pipe_mu;
    // small host-pointer calls (gpbc_common.hpp): the lanes and one queue of waiting requests per kind, all under calls_mu
    std::mutex calls_mu;
    CallLane lanes[CALL_LANES];
    std::deque<SmallCall *> waiting[CALL_KINDS];
    int next_kind = 0;                                   // where the search for the next leader starts (round robin over the kinds)
    std::condition_variable lane_cv;                     // with_call_lane waiters
    int lane_waiters = 0;
};
static DeviceCtx g_ctx[MAX_DEVICES];
static std::atomic<int> g_ndev{0};
static std::mutex g_dev_mu;
static std::atomic<int> g_host_sharding{1};
static thread_local int tl_index = -1;        // current device of this thread (index into g_ctx); -1 = the first
static thread_local bool tl_in_shard = false;

static inline int cur_index() { return tl_index < 0 ? 0 : tl_index; }
int device_count_initialised() { return g_ndev.load(); }
int current_device() {
    int n = g_ndev.load();
    if (n <= 0) return -1;
    int i = cur_index();
    return g_ctx[i < n ? i : 0].hip;
}
int bind_device() {
    int d = current_device();
    if (d < 0) return fail(GPBC_ERR_NO_DEVICE, "gpbc_init() has not bound a HIP device (no CPU fallback exists)");
    HIP_TRY(hipSetDevice(d));
    return GPBC_OK;
}
std::mutex &ws_seq_mutex() {
    int n = g_ndev.load(), i = cur_index();
    return g_ctx[(n > 0 && i < n) ? i : 0].ws_seq;
}
std::recursive_mutex &scratch_mutex() {
    int n = g_ndev.load(), i = cur_index();
    return g_ctx[(n > 0 && i < n) ? i : 0].scratch;
}
// Three non-blocking streams per bound device slot, created on first use: a large host-pointer call cuts its batch into
// chunks, alternates them on the first two and drains results on the third (gpbc_common.hpp: pipelined_chunks).
int current_slot() { return cur_index(); }
int set_slot(int index) { return gpbc_set_device(index); }
int pipe_streams(hipStream_t out[3]) {
    const int n = g_ndev.load(), i = cur_index();
    if (n <= 0 || i >= n) return fail(GPBC_ERR_NO_DEVICE, "gpbc_init() has not bound a HIP device");
    DeviceCtx &c = g_ctx[i];
    std::lock_guard<std::mutex> lk(c.pipe_mu);
    for (int k = 0; k < 3; k++)
        if (!c.pipe[k]) HIP_TRY(hipStreamCreateWithFlags(&c.pipe[k], hipStreamNonBlocking));
    out[0] = c.pipe[0]; out[1] = c.pipe[1]; out[2] = c.pipe[2];
    return GPBC_OK;
}
static void free_pipe_streams() {
    for (int i = 0; i < MAX_DEVICES; i++)
        for (int k = 0; k < 3; k++)
            if (g_ctx[i].pipe[k]) {
                if (g_ctx[i].hip >= 0) (void)hipSetDevice(g_ctx[i].hip);
                (void)hipStreamSynchronize(g_ctx[i].pipe[k]);
                (void)hipStreamDestroy(g_ctx[i].pipe[k]);
                g_ctx[i].pipe[k] = nullptr;
            }
}
int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GPBC_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return GPBC_OK;
}
int sync_default() { HIP_TRY(hipStreamSynchronize(nullptr)); return GPBC_OK; }

static inline void shard_range(size_t n, size_t part, size_t parts, size_t *lo, size_t *hi) {
    size_t base = n / parts, extra = n % parts;
    *lo = part * base + (part < extra ? part : extra);
    *hi = *lo + base + (part < extra ? 1 : 0);
}
int run_sharded(size_t n, size_t min_units, const std::function<int(size_t, size_t)> &body) {
    const int nd = g_ndev.load();
    if (min_units < 1) min_units = 1;
    if (nd <= 1 || tl_in_shard || !g_host_sharding.load() || n < 2 * min_units) return body(0, n);
    size_t parts = n / min_units;
    if (parts > (size_t)nd) parts = (size_t)nd;
    std::vector<int> rc(parts, GPBC_OK);
    std::vector<std::string> msg(parts);
    std::vector<std::thread> th;
    th.reserve(parts);
    for (size_t d = 0; d < parts; d++)
        th.emplace_back([&, d]() {
            tl_index = (int)d;
            tl_in_shard = true;
            size_t lo, hi;
            shard_range(n, d, parts, &lo, &hi);
            rc[d] = body(lo, hi);
            if (rc[d] != GPBC_OK) msg[d] = g_err;
        });
    for (auto &t : th) t.join();
    for (size_t d = 0; d < parts; d++)
        if (rc[d] != GPBC_OK) return fail(rc[d], "device %d (shard %zu of %zu): %s", g_ctx[d].hip, d, parts, msg[d].c_str());
    return GPBC_OK;
}

// ---- small host-pointer calls: lanes, queues, leader election (the scheme is described in gpbc_common.hpp)
int CallLane::reserve(size_t pin_need, size_t dev_need) {
    if (pin_need > pin_bytes) {
        size_t want = pin_bytes ? pin_bytes : (size_t)256 << 10;
        while (want < pin_need) want <<= 1;
        if (pin) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipHostFree(pin)); pin = d_pin = nullptr; pin_bytes = 0; }
        void *h = nullptr, *d = nullptr;
        // mapped into the device AND coherent, said explicitly: kernels write results here and the host reads them right after the stream
        // synchronisation (the default follows an environment variable)
        HIP_TRY(hipHostMalloc(&h, want, hipHostMallocMapped | hipHostMallocCoherent));
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipHostFree(h); return fail(GPBC_ERR_HIP, "hipHostGetDevicePointer failed for a call lane's pinned block"); }
        pin = (uint8_t *)h; d_pin = (uint8_t *)d; pin_bytes = want;
    }
    if (dev_need > dev_bytes) {
        size_t want = dev_bytes ? dev_bytes : (size_t)1 << 20;
        while (want < dev_need) want <<= 1;
        if (dev) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(dev)); dev = nullptr; dev_bytes = 0; }
        void *d = nullptr;
        HIP_TRY(hipMalloc(&d, want));
        dev = (uint8_t *)d; dev_bytes = want;
    }
    return GPBC_OK;
}
int small_call(SmallKind kind, SmallCall &c, SmallBatchFn run) {
    const int n = g_ndev.load(), di = cur_index();
    if (n <= 0 || di >= n) return fail(GPBC_ERR_NO_DEVICE, "gpbc_init() has not bound a HIP device (no CPU fallback exists)");
    if (!c.units || c.units > SMALL_CALL_MAX_UNITS) return fail(GPBC_ERR_INTERNAL, "small_call: %zu units", c.units);
    DeviceCtx &x = g_ctx[di];
    std::unique_lock<std::mutex> lk(x.calls_mu);
    x.waiting[kind].push_back(&c);
    std::vector<SmallCall *> batch;
    while (!c.done) {
        CallLane *lane = nullptr;
        if (!c.taken)
            for (auto &l : x.lanes) if (!l.busy) { lane = &l; break; }
        if (!lane) { c.cv.wait(lk); continue; }           // a leader is serving this call, or every lane is busy: woken when served / when a lane is free
        // leader: everything of this kind that is waiting, oldest first, up to one round of the chip
        auto &q = x.waiting[kind];
        batch.clear();
        size_t units = 0;
        // (the oldest request sets the batch's key; requests with another key stay queued, in order, for the next leader)
        const SmallCall *head = q.front();
        for (auto it = q.begin(); it != q.end();) {
            SmallCall *r = *it;
            const bool same = r->key_len == head->key_len && (r->key_len == 0 || memcmp(r->key, head->key, r->key_len) == 0);
            if (!same || (!batch.empty() && units + r->units > SMALL_CALL_MAX_UNITS)) { ++it; continue; }
            units += r->units;
            r->taken = true;
            batch.push_back(r);
            it = q.erase(it);
        }
        lane->busy = true;
        lane->device = di;
        lk.unlock();
        int rc = GPBC_OK;
        if (!lane->stream && hipStreamCreateWithFlags(&lane->stream, hipStreamNonBlocking) != hipSuccess) rc = fail(GPBC_ERR_HIP, "hipStreamCreateWithFlags failed for a call lane");
        if (rc == GPBC_OK) {
            try { rc = run(*lane, batch.data(), batch.size()); }
            catch (const std::exception &e) { rc = fail(GPBC_ERR_INTERNAL, "small-call batch: %s", e.what()); }      // (allocation failure of a host table: no exception may cross the C ABI)
        }
        if (rc != GPBC_OK && lane->stream) (void)hipStreamSynchronize(lane->stream);      // nothing of a failed batch may still be running when the lane is reused
        lk.lock();
        lane->busy = false;
        for (SmallCall *b : batch) {
            if (rc != GPBC_OK && b->rc == GPBC_OK) { b->rc = rc; snprintf(b->err, sizeof b->err, "%s", g_err); }
            b->done = true;
            if (b != &c) b->cv.notify_one();
        }
        // lanes are free: the oldest waiting request of each kind may lead next (one per free lane, kinds in turn)
        int free_lanes = 0;
        for (auto &l : x.lanes) free_lanes += l.busy ? 0 : 1;
        for (int t = 0; t < CALL_KINDS && free_lanes > 0; t++) {
            const int k2 = (x.next_kind + t) % CALL_KINDS;
            if (!x.waiting[k2].empty()) { x.waiting[k2].front()->cv.notify_one(); free_lanes--; }
        }
        x.next_kind = (x.next_kind + 1) % CALL_KINDS;
        if (x.lane_waiters) x.lane_cv.notify_one();
    }
    if (c.rc != GPBC_OK) return fail(c.rc, "%s", c.err);
    return GPBC_OK;
}
int with_call_lane(const std::function<int(CallLane &)> &body) {
    const int n = g_ndev.load(), di = cur_index();
    if (n <= 0 || di >= n) return fail(GPBC_ERR_NO_DEVICE, "gpbc_init() has not bound a HIP device (no CPU fallback exists)");
    TRY(bind_device());
    DeviceCtx &x = g_ctx[di];
    CallLane *lane = nullptr;
    {
        std::unique_lock<std::mutex> lk(x.calls_mu);
        for (;;) {
            for (auto &l : x.lanes) if (!l.busy) { lane = &l; break; }
            if (lane) break;
            x.lane_waiters++;
            x.lane_cv.wait(lk);
            x.lane_waiters--;
        }
        lane->busy = true;
        lane->device = di;
    }
    int rc = GPBC_OK;
    if (!lane->stream && hipStreamCreateWithFlags(&lane->stream, hipStreamNonBlocking) != hipSuccess) rc = fail(GPBC_ERR_HIP, "hipStreamCreateWithFlags failed for a call lane");
    if (rc == GPBC_OK) {
        try { rc = body(*lane); }
        catch (const std::exception &e) { rc = fail(GPBC_ERR_INTERNAL, "small call: %s", e.what()); }
    }
    if (lane->stream) (void)hipStreamSynchronize(lane->stream);      // (a no-op after a successful body, which has synchronised; after a failure nothing may still run)
    {
        std::lock_guard<std::mutex> lk(x.calls_mu);
        lane->busy = false;
        // the lane is free: a waiting combined request leads next, or another uncombined call takes it
        bool woke = false;
        for (int t = 0; t < CALL_KINDS && !woke; t++) {
            const int k2 = (x.next_kind + t) % CALL_KINDS;
            if (!x.waiting[k2].empty()) { x.waiting[k2].front()->cv.notify_one(); woke = true; }
        }
        if (x.lane_waiters) x.lane_cv.notify_one();
    }
    return rc;
}
void free_call_lanes() {
    int keep = -1;
    (void)hipGetDevice(&keep);
    for (int i = 0; i < MAX_DEVICES; i++) {
        std::lock_guard<std::mutex> lk(g_ctx[i].calls_mu);
        for (auto &l : g_ctx[i].lanes) {
            if (!l.stream && !l.pin && !l.dev) continue;
            if (g_ctx[i].hip >= 0) (void)hipSetDevice(g_ctx[i].hip);
            if (l.stream) { (void)hipStreamSynchronize(l.stream); (void)hipStreamDestroy(l.stream); }
            if (l.pin) (void)hipHostFree(l.pin);
            if (l.dev) (void)hipFree(l.dev);
            l = CallLane();
        }
    }
    if (keep >= 0) (void)hipSetDevice(keep);
}

// ---- internal workspace, one grow-only buffer per (bound device slot, stream)
struct StreamWs { int device; hipStream_t stream; int kind; void *ptr; size_t bytes; };      // device = index into g_ctx; kind 0 = workspace, 1 + level = scratch, KIND_PINNED = pinned host staging
constexpr int KIND_PINNED = 100;
static hipError_t ws_alloc(int kind, void **p, size_t bytes) { return kind == KIND_PINNED ? hipHostMalloc(p, bytes, hipHostMallocDefault) : hipMalloc(p, bytes); }
static hipError_t ws_free(int kind, void *p) { return kind == KIND_PINNED ? hipHostFree(p) : hipFree(p); }
static std::mutex g_ws_mu;
static std::vector<StreamWs> g_ws;
void free_device_blocks();
static int stream_buffer(int kind, hipStream_t stream, size_t bytes, void **out) {
    int dev = cur_index();
    std::lock_guard<std::mutex> lk(g_ws_mu);
    for (auto &w : g_ws)
        if (w.device == dev && w.stream == stream && w.kind == kind) {
            if (w.bytes < bytes) {
                HIP_TRY(hipStreamSynchronize(stream));
                HIP_TRY(ws_free(kind, w.ptr));
                w.ptr = nullptr; w.bytes = 0;
                if (ws_alloc(kind, &w.ptr, bytes) != hipSuccess) { w.ptr = nullptr; free_device_blocks(); HIP_TRY(ws_alloc(kind, &w.ptr, bytes)); }
                w.bytes = bytes;
            }
            *out = w.ptr;
            return GPBC_OK;
        }
    void *ptr = nullptr;
    if (ws_alloc(kind, &ptr, bytes) != hipSuccess) { free_device_blocks(); HIP_TRY(ws_alloc(kind, &ptr, bytes)); }   // the block cache may hold what is missing
    g_ws.push_back(StreamWs{dev, stream, kind, ptr, bytes});
    *out = ptr;
    return GPBC_OK;
}
int stream_workspace(hipStream_t stream, size_t bytes, int32_t **out) { return stream_buffer(0, stream, bytes, (void **)out); }
int stream_scratch(hipStream_t stream, int level, size_t bytes, void **out) { return stream_buffer(1 + level, stream, bytes, out); }
int pinned_staging(hipStream_t stream, size_t bytes, uint8_t **out) { return stream_buffer(KIND_PINNED, stream, bytes < 4096 ? 4096 : bytes, (void **)out); }
// ---- cache of the host-pointer entries' device blocks (DevBuf).  Classes: powers of two from 4 KiB to 64 MiB, multiples of 64 MiB
// above; a cached block serves a request of its class only, so at most twice the asked-for bytes are held per block.
struct DevBlock { int device; void *ptr; size_t cap; };
static std::mutex g_blk_mu;
static std::vector<DevBlock> g_blk;
static size_t g_blk_bytes[MAX_DEVICES] = {};
constexpr size_t BLK_KEEP_PER_DEVICE = (size_t)1 << 30, BLK_BIG = (size_t)64 << 20;
void free_device_blocks();
static size_t blk_class(size_t bytes) {
    if (bytes > BLK_BIG) return (bytes + BLK_BIG - 1) / BLK_BIG * BLK_BIG;
    size_t c = 4096;
    while (c < bytes) c <<= 1;
    return c;
}
int dev_block_alloc(size_t bytes, void **p, size_t *cap, int *dev_out) {
    const size_t c = blk_class(bytes);
    const int dev = cur_index();
    *dev_out = dev;
    {
        std::lock_guard<std::mutex> lk(g_blk_mu);
        for (size_t i = 0; i < g_blk.size(); i++)
            if (g_blk[i].device == dev && g_blk[i].cap == c) {
                *p = g_blk[i].ptr; *cap = c;
                g_blk_bytes[dev] -= c;
                g_blk[i] = g_blk.back(); g_blk.pop_back();
                return GPBC_OK;
            }
    }
    hipError_t e = hipMalloc(p, c);
    if (e != hipSuccess) {                                   // memory held by the cache may be what is missing: empty it and try once more
        free_device_blocks();
        e = hipMalloc(p, c);
    }
    if (e != hipSuccess) { *p = nullptr; *cap = 0; return fail(GPBC_ERR_HIP, "hipMalloc(%zu) failed: %s", c, hipGetErrorString(e)); }
    *cap = c;
    return GPBC_OK;
}
void dev_block_free(void *p, size_t cap, int dev) {
    // `dev` = the device slot the block was allocated on (DevBuf remembers it: the thread may have switched since).  Nothing in
    // flight THERE may still use the block — what hipFree guarantees — so that device is drained, whichever one is current.
    int keep = -1;
    (void)hipGetDevice(&keep);
    const int owner = (dev >= 0 && dev < MAX_DEVICES) ? g_ctx[dev].hip : -1;
    if (owner >= 0 && owner != keep) (void)hipSetDevice(owner);
    (void)hipDeviceSynchronize();
    struct Restore { int keep, owner; ~Restore() { if (owner >= 0 && keep >= 0 && owner != keep) (void)hipSetDevice(keep); } } restore{keep, owner};
    {
        std::lock_guard<std::mutex> lk(g_blk_mu);
        if (dev >= 0 && dev < MAX_DEVICES && g_blk_bytes[dev] + cap <= BLK_KEEP_PER_DEVICE) {
            g_blk.push_back(DevBlock{dev, p, cap});
            g_blk_bytes[dev] += cap;
            return;
        }
    }
    (void)hipFree(p);
}
void free_device_blocks() {
    std::lock_guard<std::mutex> lk(g_blk_mu);
    int keep = -1;
    (void)hipGetDevice(&keep);
    for (auto &b : g_blk) { if (g_ctx[b.device].hip >= 0) (void)hipSetDevice(g_ctx[b.device].hip); (void)hipFree(b.ptr); }
    g_blk.clear();
    for (auto &x : g_blk_bytes) x = 0;
    if (keep >= 0) (void)hipSetDevice(keep);
}
void free_workspaces() {
    free_device_blocks();
    free_call_lanes();
    std::lock_guard<std::mutex> lk(g_ws_mu);
    for (auto &w : g_ws) if (w.ptr) { if (g_ctx[w.device].hip >= 0) (void)hipSetDevice(g_ctx[w.device].hip); (void)ws_free(w.kind, w.ptr); }
    g_ws.clear();
}

// ---- RCCL, opened on demand.  A process that already carries an RCCL (PyTorch-ROCm ships one and torch.distributed uses
// it) must not get a second copy, so an already loaded library is taken first.
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;
static std::mutex g_rccl_mu;
static int g_comm_ranks = 0;          // 0: no communicator
static int g_comm_mode = 0;           // 1: one communicator per bound device in this process, 2: one rank of a multi-process job
static int g_comm_rank0 = 0;          // mode 2: this process's rank
static int rccl_load() {
    if (g_rccl.handle) return GPBC_OK;
    void *h = nullptr;
    // GPBC_RCCL_LIBRARY names one library file to use instead (a particular RCCL build; tests/stub_rccl's test double in bench.py's
    // one-GPU rehearsal, where PyTorch's real RCCL is already in the process and would refuse two ranks on one device)
    const char *override_path = getenv("GPBC_RCCL_LIBRARY");
    if (override_path && *override_path) {
        h = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
        if (!h) return fail(GPBC_ERR_COMM, "GPBC_RCCL_LIBRARY=%s: %s", override_path, dlerror());
    }
    if (!h) for (const char *name : {"librccl.so", "librccl.so.1"}) if ((h = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h) for (const char *name : {"librccl.so.1", "librccl.so"}) if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return fail(GPBC_ERR_COMM, "RCCL not found (dlopen librccl.so.1): %s", dlerror());
    Rccl r;
    r.handle = h;
#define GPBC_SYM(field, sym) *(void **)(&r.field) = dlsym(h, sym); if (!r.field) return fail(GPBC_ERR_COMM, "RCCL symbol %s missing", sym)
    GPBC_SYM(GetUniqueId, "ncclGetUniqueId");
    GPBC_SYM(CommInitRank, "ncclCommInitRank");
    GPBC_SYM(CommInitAll, "ncclCommInitAll");
    GPBC_SYM(CommDestroy, "ncclCommDestroy");
    GPBC_SYM(AllGather, "ncclAllGather");
    GPBC_SYM(GroupStart, "ncclGroupStart");
    GPBC_SYM(GroupEnd, "ncclGroupEnd");
    GPBC_SYM(GetErrorString, "ncclGetErrorString");
#undef GPBC_SYM
    g_rccl = r;
    return GPBC_OK;
}
#define NCCL_TRY(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return fail(GPBC_ERR_COMM, "%s failed: %s", #x, g_rccl.GetErrorString(r_)); } while (0)
static void comm_destroy_locked() {
    const int n = g_ndev.load();
    for (int i = 0; i < MAX_DEVICES; i++)
        if (g_ctx[i].comm) {
            if (i < n) (void)hipSetDevice(g_ctx[i].hip);
            if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g_ctx[i].comm);
            g_ctx[i].comm = nullptr;
        }
    g_comm_ranks = 0; g_comm_mode = 0; g_comm_rank0 = 0;
}
int comm_ranks() { return g_comm_ranks; }
int comm_rank() { return g_comm_mode == 2 ? g_comm_rank0 : cur_index(); }
// all-gather on the communicator of the calling thread's current device
int comm_allgather(const void *d_send, size_t bytes, void *d_recv, hipStream_t st) {
    if (!g_comm_ranks) return fail(GPBC_ERR_COMM, "no communicator: call gpbc_comm_init_all() or gpbc_comm_init_rank() first");
    const int n = g_ndev.load(), i = cur_index();
    ncclComm_t c = (i < n) ? g_ctx[i].comm : nullptr;
    if (!c) return fail(GPBC_ERR_COMM, "the current device (index %d) has no communicator", i);
    NCCL_TRY(g_rccl.AllGather(d_send, d_recv, bytes, ncclUint8, c, st));
    return GPBC_OK;
}

// ---- per-kernel timing for bench.py (HIP events on the launch stream).  Entries call profile_mark(name, stream) right
// after a launch; with profiling on, that records an event, and the time between consecutive marks of a stream is the
// duration of the kernel that ended at the later one (the stream is kept busy, so there are no gaps to speak of).
struct ProfMark { char name[32]; hipEvent_t ev; };
static std::mutex g_prof_mu;
static std::vector<ProfMark> g_prof;
static std::atomic<int> g_prof_on{0};
void profile_mark(const char *name, hipStream_t st) {
    if (!g_prof_on.load()) return;
    ProfMark m;
    snprintf(m.name, sizeof m.name, "%s", name);
    if (hipEventCreate(&m.ev) != hipSuccess) return;
    if (hipEventRecord(m.ev, st) != hipSuccess) { (void)hipEventDestroy(m.ev); return; }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(m);
}

__global__ void __launch_bounds__(BLOCK) k_fp_mul(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint8_t *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    fe_store(out + i * 32, fe_mul(fe_load(a + i * 32), fe_load(b + i * 32)));
}

extern "C" {

int gpbc_abi_version(void) { return 6; }   // 6: hash-to-curve entries, release_workspaces, pipelined-Miller knob; fixed-Q _dev entry asynchronous
const char *gpbc_last_error(void) { return g_err; }

int gpbc_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(GPBC_ERR_NO_DEVICE, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return n;
}

int gpbc_init_devices(const int *devices, int n_devices) {
    if (!devices || n_devices < 1) return fail(GPBC_ERR_INVALID_ARG, "gpbc_init_devices needs at least one device");
    if (n_devices > MAX_DEVICES) return fail(GPBC_ERR_INVALID_ARG, "at most %d devices", MAX_DEVICES);
    int n = gpbc_device_count();
    if (n <= 0) return fail(GPBC_ERR_NO_DEVICE, "no HIP device visible (this engine has no CPU fallback)");
    for (int i = 0; i < n_devices; i++) {
        if (devices[i] < 0 || devices[i] >= n) return fail(GPBC_ERR_INVALID_ARG, "device %d out of range [0,%d)", devices[i], n);
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, devices[i]));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(GPBC_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", devices[i], prop.gcnArchName);
    }
    std::lock_guard<std::mutex> lk(g_dev_mu);
    bool same = g_ndev.load() == n_devices;
    for (int i = 0; same && i < n_devices; i++) same = g_ctx[i].hip == devices[i];
    if (same) return GPBC_OK;                                   // idempotent
    if (g_ndev.load() > 0) {                                    // a different list: drop what belongs to the old one
        std::lock_guard<std::mutex> ck(g_rccl_mu);
        comm_destroy_locked();
        free_pipe_streams();
        free_workspaces();
        g_ndev.store(0);
    }
    for (int i = 0; i < n_devices; i++) g_ctx[i].hip = devices[i];
    g_ndev.store(n_devices);
    HIP_TRY(hipSetDevice(devices[0]));
    return GPBC_OK;
}
int gpbc_init(int device) { return gpbc_init_devices(&device, 1); }
int gpbc_num_devices(void) { return g_ndev.load(); }
int gpbc_device_at(int index) {
    if (index < 0 || index >= g_ndev.load()) return fail(GPBC_ERR_INVALID_ARG, "device index %d out of range [0,%d)", index, g_ndev.load());
    return g_ctx[index].hip;
}
int gpbc_set_device(int index) {
    if (index < 0 || index >= g_ndev.load()) return fail(GPBC_ERR_INVALID_ARG, "device index %d out of range [0,%d)", index, g_ndev.load());
    tl_index = index;
    return GPBC_OK;
}
int gpbc_get_device(void) { return g_ndev.load() > 0 ? cur_index() : fail(GPBC_ERR_NO_DEVICE, "gpbc_init() has not bound a HIP device"); }
int gpbc_set_host_sharding(int on) { g_host_sharding.store(on ? 1 : 0); return GPBC_OK; }

int gpbc_release_workspaces(void) {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    for (int i = 0; i < g_ndev.load(); i++) {
        HIP_TRY(hipSetDevice(g_ctx[i].hip));
        HIP_TRY(hipDeviceSynchronize());
    }
    free_workspaces();
    if (g_ndev.load() > 0) HIP_TRY(hipSetDevice(g_ctx[cur_index() < g_ndev.load() ? cur_index() : 0].hip));
    return GPBC_OK;
}
int gpbc_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    {
        std::lock_guard<std::mutex> ck(g_rccl_mu);
        comm_destroy_locked();
    }
    free_pipe_streams();
    free_workspaces();
    g_ndev.store(0);
    for (int i = 0; i < MAX_DEVICES; i++) g_ctx[i].hip = -1;
    return GPBC_OK;
}

// ---- RCCL communicators and the all-gather (the only collective of this path)
int gpbc_comm_init_all(void) {
    const int n = g_ndev.load();
    if (n <= 0) return fail(GPBC_ERR_NO_DEVICE, "gpbc_init_devices() first");
    std::lock_guard<std::mutex> ck(g_rccl_mu);
    if (g_comm_mode == 1 && g_comm_ranks == n) return GPBC_OK;
    TRY(rccl_load());
    comm_destroy_locked();
    ncclComm_t comms[MAX_DEVICES];
    int devs[MAX_DEVICES];
    for (int i = 0; i < n; i++) devs[i] = g_ctx[i].hip;
    NCCL_TRY(g_rccl.CommInitAll(comms, n, devs));
    for (int i = 0; i < n; i++) g_ctx[i].comm = comms[i];
    g_comm_ranks = n; g_comm_mode = 1;
    return GPBC_OK;
}
int gpbc_comm_get_unique_id(void *id_out) {
    if (!id_out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    std::lock_guard<std::mutex> ck(g_rccl_mu);
    TRY(rccl_load());
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, GPBC_COMM_ID_BYTES);
    return GPBC_OK;
}
int gpbc_comm_init_rank(const void *id_in, int n_ranks, int rank) {
    if (!id_in) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(GPBC_ERR_INVALID_ARG, "rank %d of %d", rank, n_ranks);
    TRY(bind_device());
    std::lock_guard<std::mutex> ck(g_rccl_mu);
    TRY(rccl_load());
    comm_destroy_locked();
    ncclUniqueId id;
    memcpy(&id, id_in, GPBC_COMM_ID_BYTES);
    ncclComm_t c = nullptr;
    NCCL_TRY(g_rccl.CommInitRank(&c, n_ranks, id, rank));
    g_ctx[cur_index()].comm = c;
    g_comm_ranks = n_ranks; g_comm_mode = 2; g_comm_rank0 = rank;
    return GPBC_OK;
}
int gpbc_comm_ranks(void) { return g_comm_ranks; }
int gpbc_comm_rank(void) { return g_comm_ranks ? comm_rank() : fail(GPBC_ERR_COMM, "no communicator"); }
int gpbc_comm_destroy(void) {
    std::lock_guard<std::mutex> ck(g_rccl_mu);
    comm_destroy_locked();
    return GPBC_OK;
}
int gpbc_allgather_dev(const void *d_send, size_t bytes_per_rank, void *d_recv, void *stream) {
    if (!bytes_per_rank) return GPBC_OK;
    if (!d_send || !d_recv) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    return comm_allgather(d_send, bytes_per_rank, d_recv, (hipStream_t)stream);
}
int gpbc_allgather_all_dev(const void *const *d_send, size_t bytes_per_rank, void *const *d_recv, void *const *streams) {
    if (!bytes_per_rank) return GPBC_OK;
    if (!d_send || !d_recv) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    if (g_comm_mode != 1) return fail(GPBC_ERR_COMM, "gpbc_allgather_all_dev needs gpbc_comm_init_all()");
    const int n = g_comm_ranks;
    for (int i = 0; i < n; i++) if (!d_send[i] || !d_recv[i]) return fail(GPBC_ERR_INVALID_ARG, "null pointer for device index %d", i);
    NCCL_TRY(g_rccl.GroupStart());
    for (int i = 0; i < n; i++) {
        ncclResult_t r = g_rccl.AllGather(d_send[i], d_recv[i], bytes_per_rank, ncclUint8, g_ctx[i].comm, streams ? (hipStream_t)streams[i] : nullptr);
        if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return fail(GPBC_ERR_COMM, "ncclAllGather (device index %d) failed: %s", i, g_rccl.GetErrorString(r)); }
    }
    NCCL_TRY(g_rccl.GroupEnd());
    return GPBC_OK;
}

int gpbc_profile_begin(void *stream) {
    TRY(bind_device());
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        for (auto &m : g_prof) (void)hipEventDestroy(m.ev);
        g_prof.clear();
    }
    g_prof_on.store(1);
    profile_mark("(begin)", (hipStream_t)stream);
    return GPBC_OK;
}
int gpbc_profile_end(char *names_out, double *total_ms_out, int *launches_out, int max_kernels, int *n_kernels_out) {
    if (!names_out || !total_ms_out || !launches_out || !n_kernels_out || max_kernels < 1) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    g_prof_on.store(0);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    int rc = GPBC_OK;
    if (!g_prof.empty() && hipEventSynchronize(g_prof.back().ev) != hipSuccess) rc = fail(GPBC_ERR_HIP, "hipEventSynchronize failed");
    for (size_t i = 1; rc == GPBC_OK && i < g_prof.size(); i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, g_prof[i - 1].ev, g_prof[i].ev) != hipSuccess) { rc = fail(GPBC_ERR_HIP, "hipEventElapsedTime failed"); break; }
        int k = 0;
        while (k < n && strncmp(names_out + 32 * k, g_prof[i].name, 32) != 0) k++;
        if (k == n) {
            if (n == max_kernels) continue;
            memset(names_out + 32 * k, 0, 32);
            snprintf(names_out + 32 * k, 32, "%s", g_prof[i].name);
            total_ms_out[k] = 0; launches_out[k] = 0;
            n++;
        }
        total_ms_out[k] += ms; launches_out[k]++;
    }
    for (auto &m : g_prof) (void)hipEventDestroy(m.ev);
    g_prof.clear();
    *n_kernels_out = n;
    return rc;
}

// the dependency-free MAD rate and the shader clock, same process, same moment (tools/microbench_valu.hip is the long form)
__global__ void __launch_bounds__(256) k_valu_probe(uint64_t *out, int iters, uint32_t a, uint32_t b) {
    uint64_t acc[4];
#pragma unroll
    for (int c = 0; c < 4; c++) acc[c] = threadIdx.x + c;
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < 4; c++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint64_t r = acc[0] + acc[1] + acc[2] + acc[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (r == 0x123456789abcdefULL) out[2] = r;                  // keeps the chains alive
}
int gpbc_valu_probe(double *out4) {
    if (!out4) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, current_device()));
    const int ncu = prop.multiProcessorCount, waves_per_simd = 8, iters = 4096;
    DevBuf d;
    TRY(d.alloc(3 * sizeof(uint64_t)));
    HIP_TRY(hipMemset(d.p, 0, 3 * sizeof(uint64_t)));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    k_valu_probe<<<ncu * waves_per_simd, 256>>>((uint64_t *)d.p, 16, 3u, 5u);        // warm-up (clocks, code object)
    float best = 1e30f;
    uint64_t h[3] = {0, 0, 0}, hb[3] = {0, 0, 0};
    int rc = check_launch("k_valu_probe");
    for (int rep = 0; rep < 3 && rc == GPBC_OK; rep++) {
        float ms = 0;
        if (hipEventRecord(e0, nullptr) != hipSuccess) rc = fail(GPBC_ERR_HIP, "hipEventRecord failed");
        k_valu_probe<<<ncu * waves_per_simd, 256>>>((uint64_t *)d.p, iters, 3u, 5u);
        if (rc == GPBC_OK) rc = check_launch("k_valu_probe");
        if (rc == GPBC_OK && (hipEventRecord(e1, nullptr) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)) rc = fail(GPBC_ERR_HIP, "timing of the probe failed");
        if (rc == GPBC_OK) rc = d.download(h, sizeof h);
        if (rc == GPBC_OK && ms < best) { best = ms; memcpy(hb, h, sizeof h); }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    TRY(rc);
    const double lane_ops = (double)ncu * waves_per_simd * 256.0 * iters * 8 * 4;
    const double clock_hz = hb[1] ? (double)hb[0] / ((double)hb[1] / 100e6) : 0.0;
    const double wave_instr_per_simd = (double)waves_per_simd * iters * 8 * 4;      // each SIMD runs waves_per_simd waves of the 4 per block
    out4[0] = lane_ops / (best * 1e-3);
    out4[1] = clock_hz;
    out4[2] = clock_hz ? (best * 1e-3 * clock_hz) / wave_instr_per_simd : 0.0;
    out4[3] = best;
    return GPBC_OK;
}

int gpbc_fp_mul_batch(const void *a, const void *b, size_t n, void *out) {
    if (!n) return GPBC_OK;
    if (!a || !b || !out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    DevBuf dA, dB, dO;
    TRY(dA.upload(a, n * 32)); TRY(dB.upload(b, n * 32)); TRY(dO.alloc(n * 32));
    k_fp_mul<<<grid_for(n), BLOCK>>>(dA.u8(), dB.u8(), dO.u8(), n);
    TRY(check_launch("k_fp_mul"));
    TRY(sync_default());
    return dO.download(out, n * 32);
}

}  // extern "C"
