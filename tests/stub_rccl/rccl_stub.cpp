// TEST DOUBLE for librccl.so.1 — test infrastructure only, never shipped and never on the product's library path.
//
// Why: the N > 1 collective code of libgpbc_bn254.so (csrc/gpbc_core.hip: ncclCommInitAll over several slots, the grouped
// ncclAllGather of gpbc_allgather_all_dev, the per-thread ncclAllGather inside gpbc_g1/g2_scalar_mul_sum_dev) needs more than one
// rank, real RCCL refuses a device list that names one GPU twice, and the builder's boxes have one GPU.  This library implements the
// eight symbols the product binds (gpbc_core.hip: rccl_load) for ranks that live in ONE process and may share one device: an
// all-gather is a rendezvous of the ranks' calls on the host followed by device-to-device copies — stronger ordering than RCCL
// gives (every stream involved is synchronised), same data movement: rank q's `count` bytes land at offset q * count of every
// rank's receive buffer.  Ranks in DIFFERENT processes (ncclCommInitRank with nranks > 1: what bench.py --gpus N does) meet in POSIX
// shared memory named after the unique id: every rank copies its rows to its own shared file, a counter barrier, every rank copies
// all files into its receive buffer, a second barrier.  tests/test_gpu_parity.py builds it into tests/stub_rccl/librccl.so.1 and runs tests/cpp/test_multi_device.cpp
// with LD_LIBRARY_PATH pointing here.  Rates measured through it mean nothing.
//
// build: g++ -O1 -shared -fPIC -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/stub_rccl/rccl_stub.cpp -L/opt/rocm/lib -lamdhip64 -o tests/stub_rccl/librccl.so.1
#include <hip/hip_runtime_api.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
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
// ranks of other processes: control block in shared memory + one data file per rank
struct Shared { std::atomic<unsigned long> arrive, leave; std::atomic<unsigned long> bytes[64]; };
struct Remote {
    std::string base;              // "/<unique id>"
    int nranks, rank;
    Shared *ctl = nullptr;
    unsigned long round = 0;
};
struct ncclComm { Group *group; int rank; Remote *remote; };

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
    snprintf(id->internal, sizeof id->internal, "gpbc-rccl-test-double-%d-%d", (int)getpid(), g_next_id++);
    return ncclSuccess;
}
static void *map_file(const std::string &name, size_t bytes, bool create) {
    int fd = shm_open(name.c_str(), create ? (O_CREAT | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) return nullptr;
    struct stat st;
    if (fstat(fd, &st) != 0 || ((size_t)st.st_size < bytes && ftruncate(fd, (off_t)bytes) != 0)) { close(fd); return nullptr; }
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    return p == MAP_FAILED ? nullptr : p;
}
// a rank that never arrives (its process died) must not hang the others: two minutes, then the collective fails
static bool spin_until(std::atomic<unsigned long> &c, unsigned long target) {
    const auto t0 = std::chrono::steady_clock::now();
    while (c.load(std::memory_order_acquire) < target) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;
        usleep(50);
    }
    return true;
}
// one rank of a multi-process job: all-gather through shared memory
static ncclResult_t remote_allgather(Remote &r, const void *send, void *recv, size_t bytes, hipStream_t stream) {
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    const unsigned long done = (unsigned long)r.nranks * (r.round + 1);
    if (bytes) {
        void *mine = map_file(r.base + "_r" + std::to_string(r.rank), bytes, true);
        if (!mine) return ncclSystemError;
        if (hipMemcpy(mine, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) { munmap(mine, bytes); return ncclUnhandledCudaError; }
        munmap(mine, bytes);
    }
    r.ctl->bytes[r.rank].store(bytes, std::memory_order_release);
    r.ctl->arrive.fetch_add(1, std::memory_order_acq_rel);
    if (!spin_until(r.ctl->arrive, done)) return ncclSystemError;
    ncclResult_t rc = ncclSuccess;
    for (int q = 0; q < r.nranks && rc == ncclSuccess; q++) {
        if (r.ctl->bytes[q].load(std::memory_order_acquire) != bytes) { rc = ncclInvalidArgument; break; }
        if (!bytes) continue;
        void *theirs = map_file(r.base + "_r" + std::to_string(q), bytes, false);
        if (!theirs) { rc = ncclSystemError; break; }
        if (hipMemcpy((char *)recv + (size_t)q * bytes, theirs, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
        munmap(theirs, bytes);
    }
    r.ctl->leave.fetch_add(1, std::memory_order_acq_rel);       // nobody overwrites its file before everybody has read it
    if (!spin_until(r.ctl->leave, done) && rc == ncclSuccess) rc = ncclSystemError;
    r.round++;
    return rc;
}
// one process per rank: ranks of other processes are met through shared memory named after the id
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (nranks == 1) {
        Group *g = new Group(1);
        if (hipGetDevice(&g->dev[0]) != hipSuccess) { delete g; return ncclUnhandledCudaError; }
        *comm = new ncclComm{g, 0, nullptr};
        return ncclSuccess;
    }
    id.internal[sizeof id.internal - 1] = 0;
    Remote *r = new Remote;
    r->base = std::string("/") + id.internal;
    r->nranks = nranks; r->rank = rank;
    r->ctl = (Shared *)map_file(r->base + "_ctl", sizeof(Shared), true);      // a fresh shared file is all zeros: counters start at 0
    if (!r->ctl) { delete r; return ncclSystemError; }
    *comm = new ncclComm{nullptr, rank, r};
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
    if (c->remote) {
        Remote *r = c->remote;
        shm_unlink((r->base + "_r" + std::to_string(r->rank)).c_str());
        if (r->rank == 0) shm_unlink((r->base + "_ctl").c_str());      // (the mapping stays valid for ranks still holding it)
        munmap(r->ctl, sizeof(Shared));
        delete r;
        delete c;
        return ncclSuccess;
    }
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
    if (comm->remote) return remote_allgather(*comm->remote, sendbuff, recvbuff, sendcount, stream);
    const Ticket t = post(comm, sendbuff, recvbuff, sendcount, stream);
    if (tl_depth) { tl_tickets.push_back(t); return ncclSuccess; }                 // grouped: GroupEnd waits
    return wait_done(t);
}
}
