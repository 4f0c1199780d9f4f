// TEST DOUBLE for librccl.so.1 — test infrastructure only, never shipped and never on the product's library path.
//
// Why: the N > 1 collective code of libgpbc_bn254.so (csrc/gpbc_core.hip: ncclCommInitAll over several slots, the grouped
// ncclAllGather of gpbc_allgather_all_dev, the per-thread ncclAllGather inside gpbc_g1/g2_scalar_mul_sum_dev) needs more than one
// rank, real RCCL refuses a device list that names one GPU twice, and the builder's boxes have one GPU.  This library implements the
// eight symbols the product binds (gpbc_core.hip: rccl_load) for ranks that live in ONE process and may share one device: an
// all-gather is a rendezvous of the ranks' calls on the host followed by device-to-device copies — stronger ordering than RCCL
// gives (every stream involved is synchronised), same data movement: rank q's `count` bytes land at offset q * count of every
// rank's receive buffer.  tests/test_gpu_parity.py builds it into tests/stub_rccl/librccl.so.1 and runs tests/cpp/test_multi_device.cpp
// with LD_LIBRARY_PATH pointing here.  Rates measured through it mean nothing.
//
// build: g++ -O1 -shared -fPIC -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/stub_rccl/rccl_stub.cpp -L/opt/rocm/lib -lamdhip64 -o tests/stub_rccl/librccl.so.1
#include <hip/hip_runtime_api.h>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct ncclComm;
typedef struct ncclComm *ncclComm_t;
}

namespace {
struct Op { const void *send; void *recv; size_t bytes; hipStream_t stream; bool posted; };
struct Group {                     // the ranks of one communicator clique
    int n, refs;
    std::vector<int> dev;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Op> ops;
    int count = 0;
    unsigned long generation = 0;
    ncclResult_t last = ncclSuccess;
    explicit Group(int n_) : n(n_), refs(n_), dev(n_, 0), ops(n_, Op{nullptr, nullptr, 0, nullptr, false}) {}
};
}  // namespace
struct ncclComm { Group *group; int rank; };

namespace {
struct Ticket { Group *g; unsigned long gen; };
thread_local int tl_depth = 0;
thread_local std::vector<Ticket> tl_tickets;
std::mutex g_id_mu;
int g_next_id = 1;

// all ranks have posted: synchronise every rank's stream (the send buffers are then complete), copy, synchronise
ncclResult_t run_round(Group &g) {
    int before = 0;
    if (hipGetDevice(&before) != hipSuccess) return ncclUnhandledCudaError;
    ncclResult_t rc = ncclSuccess;
    for (int q = 0; q < g.n && rc == ncclSuccess; q++)
        if (hipSetDevice(g.dev[q]) != hipSuccess || hipStreamSynchronize(g.ops[q].stream) != hipSuccess) rc = ncclUnhandledCudaError;
    for (int q = 1; q < g.n && rc == ncclSuccess; q++)
        if (g.ops[q].bytes != g.ops[0].bytes) rc = ncclInvalidArgument;
    for (int r = 0; r < g.n && rc == ncclSuccess; r++) {
        if (hipSetDevice(g.dev[r]) != hipSuccess) { rc = ncclUnhandledCudaError; break; }
        for (int q = 0; q < g.n && rc == ncclSuccess; q++)
            if (g.ops[q].bytes && hipMemcpy((char *)g.ops[r].recv + (size_t)q * g.ops[q].bytes, g.ops[q].send, g.ops[q].bytes, hipMemcpyDefault) != hipSuccess) rc = ncclUnhandledCudaError;
    }
    (void)hipSetDevice(before);
    return rc;
}
// a rank's call arrives; the last one to arrive runs the round.  Returns the generation to wait for.
Ticket post(ncclComm_t c, const void *send, void *recv, size_t bytes, hipStream_t st) {
    Group &g = *c->group;
    std::unique_lock<std::mutex> lk(g.mu);
    g.cv.wait(lk, [&] { return !g.ops[c->rank].posted; });            // this rank's slot of the previous round has been consumed
    g.ops[c->rank] = Op{send, recv, bytes, st, true};
    const unsigned long gen = g.generation;
    if (++g.count == g.n) {
        g.last = run_round(g);
        for (auto &o : g.ops) o.posted = false;
        g.count = 0;
        g.generation++;
        g.cv.notify_all();
    }
    return Ticket{&g, gen};
}
ncclResult_t wait_done(const Ticket &t) {
    std::unique_lock<std::mutex> lk(t.g->mu);
    t.g->cv.wait(lk, [&] { return t.g->generation > t.gen; });
    return t.g->last;
}
}  // namespace

extern "C" {
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error (rccl test double)" : "error in the rccl test double"; }
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    std::lock_guard<std::mutex> lk(g_id_mu);
    snprintf(id->internal, sizeof id->internal, "gpbc-rccl-test-double-%d", g_next_id++);
    return ncclSuccess;
}
// one process per rank needs a cross-process rendezvous, which this double does not have: one rank only
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId, int rank) {
    if (!comm || nranks != 1 || rank != 0) return ncclInvalidUsage;
    Group *g = new Group(1);
    if (hipGetDevice(&g->dev[0]) != hipSuccess) { delete g; return ncclUnhandledCudaError; }
    *comm = new ncclComm{g, 0};
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist) {      // a device may appear more than once: that is the point
    if (!comms || ndev < 1) return ncclInvalidArgument;
    Group *g = new Group(ndev);
    for (int i = 0; i < ndev; i++) { g->dev[i] = devlist ? devlist[i] : i; comms[i] = new ncclComm{g, i}; }
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    Group *g = c->group;
    bool last;
    { std::lock_guard<std::mutex> lk(g->mu); last = --g->refs == 0; }
    delete c;
    if (last) delete g;
    return ncclSuccess;
}
ncclResult_t ncclGroupStart() { tl_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (tl_depth <= 0) return ncclInvalidUsage;
    if (--tl_depth) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    for (const Ticket &t : tl_tickets) { ncclResult_t r = wait_done(t); if (r != ncclSuccess) rc = r; }
    tl_tickets.clear();
    return rc;
}
ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream) {
    if (!comm || (sendcount && (!sendbuff || !recvbuff)) || (dt != ncclUint8 && dt != ncclInt8)) return ncclInvalidArgument;
    const Ticket t = post(comm, sendbuff, recvbuff, sendcount, stream);
    if (tl_depth) { tl_tickets.push_back(t); return ncclSuccess; }                 // grouped: GroupEnd waits
    return wait_done(t);
}
}
