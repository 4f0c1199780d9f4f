// ONE host process drives every visible MI355X through the C ABI (include/gpbc_bn254.h; SURVEY.md §8b "init(devices)",
// §8e "one host thread + one HIP stream per device") — what a Go host calling the cgo shim does:
//   1. the host-pointer batch entries shard their index range over the bound devices and must return the bits a single
//      device returns (pairings, ragged multi-pairings, PairingCheck, fixed-Q multi-pairing, G1 / G2 scalar multiplication,
//      GT.Exp / Mul, wire formats, point sums and the aggregate-verify sums  sum [rho_i] P_i);
//   2. one thread per device with device-resident buffers and its own stream (gpbc_set_device + *_dev entries);
//   3. the RCCL all-gather of GT rows and of partial sums over the in-process communicator (gpbc_comm_init_all).
// With one visible GPU the device list is {0, 0}: two slots share the GPU, so the sharding code (offsets, rebased segment
// tables, partial sums) still runs; RCCL is then exercised with one rank.
// build: g++ -std=c++17 -pthread -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/cpp/test_multi_device.cpp
//        -L gopairingbasedcryptography_amd -lgpbc_bn254 -L /opt/rocm/lib -lamdhip64
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <string>
#include <vector>
#include "gpbc_bn254.hpp"

using namespace bn254;
#define EXPECT(c) do { if (!(c)) { printf("FAIL line %d: %s  (%s)\n", __LINE__, #c, gpbc_last_error()); return 1; } } while (0)
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL line %d: %s: %s\n", __LINE__, #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class T> static bool same(const std::vector<T> &a, const std::vector<T> &b) {
    return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0;
}

int main() {
    setenv("GPBC_TEST_KNOBS", "1", 1);          // this is a test process: the fault-injection knob answers only with this set
    const int visible = gpbc_device_count();
    EXPECT(visible >= 1);
    std::vector<int> devs;
    for (int i = 0; i < visible; i++) devs.push_back(i);
    const bool shared_gpu = visible == 1;
    // one GPU: listed twice — or GPBC_TEST_SLOTS times (the rank arithmetic of an 8-GPU node rehearsed on one device: 8 slots, 8 host
    // threads, 8 communicator ranks over the test double; nothing here is a measurement)
    const int want_slots = getenv("GPBC_TEST_SLOTS") ? atoi(getenv("GPBC_TEST_SLOTS")) : 2;
    if (shared_gpu) while ((int)devs.size() < (want_slots < 2 ? 2 : want_slots)) devs.push_back(0);
    Init(devs);
    const int nd = NumDevices();
    EXPECT(nd == (int)devs.size());
    for (int i = 0; i < nd; i++) EXPECT(gpbc_device_at(i) == devs[i]);
    EXPECT(gpbc_set_device(nd) < 0 && gpbc_set_device(-1) < 0);          // out of range: error, current device unchanged
    EXPECT(gpbc_get_device() == 0);
    printf("%d visible GPU(s), %d bound slot(s)%s\n", visible, nd, shared_gpu ? " (one GPU listed twice)" : "");

    G1Affine g1; G2Affine g2; Generators(g1, g2);
    const size_t N = nd <= 4 ? 20000 : (size_t)nd * 4096 + 1234;        // > 2 x 4096, and >= 4096 per slot: every batch entry shards over ALL slots, ragged
    std::vector<Scalar> a(N), b(N), c(N);
    for (size_t i = 0; i < N; i++) { a[i] = Scalar(0x9E3779B97F4A7C15ull * (i + 1)); b[i] = Scalar(0xC2B2AE3D27D4EB4Full * (i + 3)); c[i] = Scalar(0x165667B19E3779F9ull * (i + 5)); }
    for (size_t i = 0; i < N; i++) for (int j = 8; j < 31; j++) c[i].le[j] = (uint8_t)(c[i].le[j - 8] * 31 + j);      // full-width scalars
    a[17] = Scalar(0);                                                   // a point at infinity inside a shard

    // ---- 1. sharded host entries against the single-device results
    check(gpbc_set_host_sharding(0));
    std::vector<G1Affine> P1 = G1ScalarMultiplicationBatch({g1}, a), Pc1 = G1ScalarMultiplicationBatch(P1, c);
    std::vector<G2Affine> Q1 = G2ScalarMultiplicationBatch({g2}, b), Qc1 = G2ScalarMultiplicationBatch(Q1, c);
    std::vector<GT> E1 = PairBatch(P1, Q1);
    check(gpbc_set_host_sharding(1));
    std::vector<G1Affine> P = G1ScalarMultiplicationBatch({g1}, a), Pc = G1ScalarMultiplicationBatch(P, c);
    std::vector<G2Affine> Q = G2ScalarMultiplicationBatch({g2}, b), Qc = G2ScalarMultiplicationBatch(Q, c);
    std::vector<GT> E = PairBatch(P, Q);
    EXPECT(same(P, P1) && same(Q, Q1) && same(Pc, Pc1) && same(Qc, Qc1) && same(E, E1));
    EXPECT(P[17].IsInfinity());

    // ragged multi-pairing segments (lengths 0..12) + PairingCheck over the same table
    std::vector<uint64_t> seg(1, 0);
    while (seg.back() + 13 <= N) seg.push_back(seg.back() + (seg.size() * 7) % 13);
    const size_t k = seg.size() - 1, npairs = (size_t)seg.back();
    std::vector<GT> M(k), M1(k);
    std::vector<uint8_t> ok(k), ok1(k);
    check(gpbc_set_host_sharding(0));
    check(gpbc_multi_pair(P.data(), Q.data(), seg.data(), k, M1.data()));
    check(gpbc_pairing_check(P.data(), Q.data(), seg.data(), k, ok1.data()));
    check(gpbc_set_host_sharding(1));
    check(gpbc_multi_pair(P.data(), Q.data(), seg.data(), k, M.data()));
    check(gpbc_pairing_check(P.data(), Q.data(), seg.data(), k, ok.data()));
    EXPECT(same(M, M1) && same(ok, ok1) && npairs > 8192);
    // fixed-Q: 600 segments x 16 pairs against one list
    {
        const size_t m = 16, ks = 600;
        std::vector<G2Affine> Qm(Q.begin(), Q.begin() + m);
        std::vector<G1Affine> Pm(P.begin(), P.begin() + m * ks);
        check(gpbc_set_host_sharding(0));
        std::vector<GT> F1 = PairFixedQ(Pm, Qm);
        check(gpbc_set_host_sharding(1));
        std::vector<GT> F = PairFixedQ(Pm, Qm);
        EXPECT(same(F, F1));
    }
    // GT.Exp / GT.Mul, wire formats
    {
        std::vector<GT> X(N), X1(N), Y(N), Y1(N);
        check(gpbc_set_host_sharding(0));
        check(gpbc_gt_exp_batch(E.data(), c.data(), N, X1.data()));
        check(gpbc_gt_mul_batch(E.data(), X1.data(), N, Y1.data()));
        check(gpbc_set_host_sharding(1));
        check(gpbc_gt_exp_batch(E.data(), c.data(), N, X.data()));
        check(gpbc_gt_mul_batch(E.data(), X.data(), N, Y.data()));
        EXPECT(same(X, X1) && same(Y, Y1));
        std::vector<uint8_t> w(N * 64), w1(N * 64), okw(N);
        std::vector<G2Affine> back(N);
        check(gpbc_g2_marshal_batch(Q.data(), N, 1, w.data()));
        check(gpbc_set_host_sharding(0));
        check(gpbc_g2_marshal_batch(Q.data(), N, 1, w1.data()));
        check(gpbc_set_host_sharding(1));
        check(gpbc_g2_unmarshal_batch(w.data(), 64, N, back.data(), okw.data()));
        EXPECT(w == w1 && same(back, Q));
        for (size_t i = 0; i < N; i++) EXPECT(okw[i] == 1);
    }
    {   // hash to curve from host pointers: the shards rebase the offset table of their run of messages
        const size_t nm = 40000;
        std::string data;
        std::vector<uint64_t> off(nm + 1, 0);
        for (size_t i = 0; i < nm; i++) { data.append(i % 71, (char)('a' + i % 26)); data += std::to_string(i); off[i + 1] = data.size(); }
        const std::string dst = "Hash Bytes To Element In G1";
        std::vector<G1Affine> H(nm), H1(nm);
        std::vector<uint8_t> U(nm * 128), U1(nm * 128);
        check(gpbc_hash_to_g1(data.data(), off.data(), nm, dst.data(), dst.size(), H.data()));
        check(gpbc_hash_to_field(data.data(), off.data(), nm, dst.data(), dst.size(), 4, U.data()));
        check(gpbc_set_host_sharding(0));
        check(gpbc_hash_to_g1(data.data(), off.data(), nm, dst.data(), dst.size(), H1.data()));
        check(gpbc_hash_to_field(data.data(), off.data(), nm, dst.data(), dst.size(), 4, U1.data()));
        check(gpbc_set_host_sharding(1));
        EXPECT(same(H, H1) && U == U1 && !H[0].IsInfinity() && !H[nm - 1].Equal(H[nm - 2]));
        EXPECT(H[12345].Equal(HashToG1(data.substr(off[12345], off[12346] - off[12345]), dst)));
    }
    // point sums and the aggregate-verify sums: sum_i [c_i] P_i must equal the sum of the products, on any number of devices
    G1Affine A, A1, As; G2Affine B, B1, Bs;
    check(gpbc_g1_scalar_mul_sum(P.data(), c.data(), N, &A));
    check(gpbc_g2_scalar_mul_sum(Q.data(), c.data(), N, &B));
    check(gpbc_set_host_sharding(0));
    check(gpbc_g1_scalar_mul_sum(P.data(), c.data(), N, &A1));
    check(gpbc_g2_scalar_mul_sum(Q.data(), c.data(), N, &B1));
    check(gpbc_g1_sum(Pc.data(), N, &As));
    check(gpbc_g2_sum(Qc.data(), N, &Bs));
    check(gpbc_set_host_sharding(1));
    EXPECT(A.Equal(A1) && A.Equal(As) && B.Equal(B1) && B.Equal(Bs) && !A.IsInfinity());
    {   // a sum long enough for gpbc_g1_sum itself to shard (>= 2 x 65536 points): the batch repeated 7 times = 7 x the sum
        std::vector<G1Affine> rep;
        for (int r = 0; r < 7; r++) rep.insert(rep.end(), Pc.begin(), Pc.end());
        G1Affine S7, want7;
        check(gpbc_g1_sum(rep.data(), rep.size(), &S7));
        want7.ScalarMultiplication(As, Scalar(7));
        EXPECT(S7.Equal(want7));
    }
    printf("sharded host entries identical to one device: %zu pairings, %zu ragged segments, scalar mults, GT, wire, sums\n", N, k);
    {   // a batch large enough for every shard to run PIPELINED (>= 2 x 131072 units per device: chunks on two streams, results
        // drained by a helper thread) against the same batch in slices small enough for the plain upload -> compute -> download
        const size_t NB = (size_t)(nd < 4 ? nd : 4) * 270000 + 17;     // (with more slots only the first ones get a pipelined share: the point is made with four)
        std::vector<Scalar> ka(NB), kb(NB);
        for (size_t i = 0; i < NB; i++) { ka[i] = Scalar(0x9E3779B97F4A7C15ull * (i + 11)); kb[i] = Scalar(0xC2B2AE3D27D4EB4Full * (i + 13)); }
        std::vector<G1Affine> PB = G1ScalarMultiplicationBatch({g1}, ka);
        std::vector<G2Affine> QB = G2ScalarMultiplicationBatch({g2}, kb);
        std::vector<GT> big = PairBatch(PB, QB), ref(NB);
        std::vector<G1Affine> PK = G1ScalarMultiplicationBatch(PB, kb), PKref(NB);          // one base per scalar: pipelined as well
        check(gpbc_set_host_sharding(0));
        for (size_t off = 0; off < NB; off += 100000) {
            const size_t m = NB - off < 100000 ? NB - off : 100000;
            check(gpbc_pair_batch(&PB[off], &QB[off], m, &ref[off]));
            check(gpbc_g1_scalar_mul_batch(&PB[off], m, &kb[off], m, &PKref[off]));
        }
        check(gpbc_set_host_sharding(1));
        EXPECT(same(big, ref) && same(PK, PKref));
        printf("pipelined + sharded host calls identical to plain ones: %zu pairings and G1 scalar multiplications\n", NB);
    }

    // ---- 2. one host thread per device slot, device-resident buffers, own stream
    {
        std::vector<int> bad(nd, 0);
        std::vector<std::thread> th;
        for (int d = 0; d < nd; d++)
            th.emplace_back([&, d]() {
                auto fail_here = [&](int line) { printf("thread %d failed at line %d: %s\n", d, line, gpbc_last_error()); bad[d] = 1; };
                if (gpbc_set_device(d) < 0) return fail_here(__LINE__);
                if (hipSetDevice(gpbc_device_at(d)) != hipSuccess) return fail_here(__LINE__);
                size_t lo = N * d / nd, hi = N * (d + 1) / nd, m = hi - lo;
                hipStream_t st;
                void *dP, *dQ, *dE;
                if (hipStreamCreate(&st) != hipSuccess) return fail_here(__LINE__);
                if (hipMalloc(&dP, m * 64) != hipSuccess || hipMalloc(&dQ, m * 128) != hipSuccess || hipMalloc(&dE, m * 384) != hipSuccess) return fail_here(__LINE__);
                hipMemcpyAsync(dP, &P[lo], m * 64, hipMemcpyHostToDevice, st);
                hipMemcpyAsync(dQ, &Q[lo], m * 128, hipMemcpyHostToDevice, st);
                if (gpbc_pair_batch_dev(dP, dQ, m, dE, st) < 0) return fail_here(__LINE__);
                std::vector<GT> out(m);
                hipMemcpyAsync(out.data(), dE, m * 384, hipMemcpyDeviceToHost, st);
                if (hipStreamSynchronize(st) != hipSuccess) return fail_here(__LINE__);
                if (std::memcmp(out.data(), &E[lo], m * 384) != 0) return fail_here(__LINE__);
                hipFree(dP); hipFree(dQ); hipFree(dE); hipStreamDestroy(st);
            });
        for (auto &t : th) t.join();
        for (int d = 0; d < nd; d++) EXPECT(bad[d] == 0);
        printf("one thread per device slot with *_dev entries: OK\n");
    }

    // ---- 3. RCCL all-gather inside the library.  Real RCCL refuses a GPU listed twice, so on a one-GPU box the list shrinks to {0}
    // (one rank) — unless GPBC_TEST_STUB_RCCL is set and tests/stub_rccl/librccl.so.1 (a host-rendezvous + device-copy test double)
    // is on the library path: then the {0, 0} list stays and everything below runs with TWO ranks.
    const bool stub = getenv("GPBC_TEST_STUB_RCCL") != nullptr;
    if (shared_gpu && !stub) Init(std::vector<int>{0});
    const int nr = NumDevices();
    if (gpbc_comm_init_all() < 0) { printf("FAIL: gpbc_comm_init_all: %s\n", gpbc_last_error()); return 1; }
    EXPECT(gpbc_comm_ranks() == nr);
    if (stub) EXPECT(nr >= 2);
    {
        const size_t rows = 1024, bytes = rows * 384;                    // GT rows per rank (BASELINE config 5 shape: the AFP25 masks)
        std::vector<void *> send(nr), recv(nr);
        for (int d = 0; d < nr; d++) {
            HIP_OK(hipSetDevice(gpbc_device_at(d)));
            HIP_OK(hipMalloc(&send[d], bytes)); HIP_OK(hipMalloc(&recv[d], bytes * nr));
            HIP_OK(hipMemset(recv[d], 0xEE, bytes * nr));
            HIP_OK(hipMemcpy(send[d], &E[d * rows], bytes, hipMemcpyHostToDevice));
        }
        check(gpbc_allgather_all_dev(send.data(), bytes, recv.data(), nullptr));
        for (int d = 0; d < nr; d++) {
            std::vector<GT> got(rows * nr);
            HIP_OK(hipSetDevice(gpbc_device_at(d)));
            HIP_OK(hipDeviceSynchronize());
            HIP_OK(hipMemcpy(got.data(), recv[d], bytes * nr, hipMemcpyDeviceToHost));
            EXPECT(std::memcmp(got.data(), E.data(), bytes * nr) == 0);      // rank order and offsets: rank q's rows at q * bytes, on every rank
            hipFree(send[d]); hipFree(recv[d]);
        }
        // the aggregate-verify sums through the host entry (sharded over the slots, partial sums combined on the host)
        G1Affine Ar; G2Affine Br;
        check(gpbc_g1_scalar_mul_sum(P.data(), c.data(), N, &Ar));
        check(gpbc_g2_scalar_mul_sum(Q.data(), c.data(), N, &Br));
        EXPECT(Ar.Equal(A) && Br.Equal(B));
        // ... and through the device-resident composite, one host thread per rank with its own shard and stream: local bucket
        // sum, all-gather of one point per rank INSIDE the library (per-thread ncclAllGather on the rank's communicator), sum of the
        // gathered partials — every rank must end with the global sum
        {
            std::vector<std::thread> th;
            std::vector<int> bad(nr, 0);
            std::vector<G1Affine> A_rank(nr);
            std::vector<G2Affine> B_rank(nr);
            for (int d = 0; d < nr; d++)
                th.emplace_back([&, d]() {
                    auto fail_here = [&](int line) { bad[d] = line; };
                    if (gpbc_set_device(d) < 0 || hipSetDevice(gpbc_device_at(d)) != hipSuccess) return fail_here(__LINE__);
                    const size_t lo = N * d / nr, hi = N * (d + 1) / nr, m = hi - lo;
                    hipStream_t st;
                    if (hipStreamCreate(&st) != hipSuccess) return fail_here(__LINE__);
                    void *dB1, *dB2, *dS, *dO1, *dO2;
                    if (hipMalloc(&dB1, m * 64) != hipSuccess || hipMalloc(&dB2, m * 128) != hipSuccess || hipMalloc(&dS, m * 32) != hipSuccess ||
                        hipMalloc(&dO1, 64) != hipSuccess || hipMalloc(&dO2, 128) != hipSuccess) return fail_here(__LINE__);
                    hipMemcpyAsync(dB1, &P[lo], m * 64, hipMemcpyHostToDevice, st);
                    hipMemcpyAsync(dB2, &Q[lo], m * 128, hipMemcpyHostToDevice, st);
                    hipMemcpyAsync(dS, &c[lo], m * 32, hipMemcpyHostToDevice, st);
                    if (gpbc_g1_scalar_mul_sum_dev(dB1, dS, m, dO1, st) < 0) return fail_here(__LINE__);
                    if (gpbc_g2_scalar_mul_sum_dev(dB2, dS, m, dO2, st) < 0) return fail_here(__LINE__);
                    hipMemcpyAsync(&A_rank[d], dO1, 64, hipMemcpyDeviceToHost, st);
                    hipMemcpyAsync(&B_rank[d], dO2, 128, hipMemcpyDeviceToHost, st);
                    if (hipStreamSynchronize(st) != hipSuccess) return fail_here(__LINE__);
                    hipFree(dB1); hipFree(dB2); hipFree(dS); hipFree(dO1); hipFree(dO2); hipStreamDestroy(st);
                });
            for (auto &t : th) t.join();
            for (int d = 0; d < nr; d++) { if (bad[d]) printf("rank %d failed at line %d: %s\n", d, bad[d], gpbc_last_error()); EXPECT(bad[d] == 0); }
            for (int d = 0; d < nr; d++) EXPECT(A_rank[d].Equal(A) && B_rank[d].Equal(B));
            printf("scalar_mul_sum_dev on %d rank(s), partial sums all-gathered inside the library: every rank holds the global sums\n", nr);
        }
        // BLS aggregate verification with the verifier's sums taken from the collective path: accepts, and rejects a forged signature
        {
            const size_t n = 10000;
            G2Affine H; H.ScalarMultiplicationBase(Scalar(0xABCDEF12345ull));
            std::vector<Scalar> x(a.begin() + 100, a.begin() + 100 + n), rho(c.begin(), c.begin() + n);
            for (auto &r : rho) for (int j = 16; j < 32; j++) r.le[j] = 0;
            std::vector<G1Affine> pk = G1ScalarMultiplicationBatch({g1}, x);
            std::vector<G2Affine> sg = G2ScalarMultiplicationBatch({H}, x);
            for (int forged = 0; forged < 2; forged++) {
                if (forged) sg[n / 3] = sg[n / 3 + 1];
                std::vector<G1Affine> Ag(nr); std::vector<G2Affine> Bg(nr);
                std::vector<std::thread> th; std::vector<int> bad(nr, 0);
                for (int d = 0; d < nr; d++)
                    th.emplace_back([&, d]() {
                        if (gpbc_set_device(d) < 0 || hipSetDevice(gpbc_device_at(d)) != hipSuccess) { bad[d] = __LINE__; return; }
                        const size_t lo = n * d / nr, hi = n * (d + 1) / nr, m = hi - lo;
                        void *dB1, *dB2, *dS, *dO1, *dO2;
                        if (hipMalloc(&dB1, m * 64) != hipSuccess || hipMalloc(&dB2, m * 128) != hipSuccess || hipMalloc(&dS, m * 32) != hipSuccess ||
                            hipMalloc(&dO1, 64) != hipSuccess || hipMalloc(&dO2, 128) != hipSuccess) { bad[d] = __LINE__; return; }
                        hipMemcpy(dB1, &pk[lo], m * 64, hipMemcpyHostToDevice); hipMemcpy(dB2, &sg[lo], m * 128, hipMemcpyHostToDevice); hipMemcpy(dS, &rho[lo], m * 32, hipMemcpyHostToDevice);
                        if (gpbc_g1_scalar_mul_sum_dev(dB1, dS, m, dO1, nullptr) < 0 || gpbc_g2_scalar_mul_sum_dev(dB2, dS, m, dO2, nullptr) < 0) { bad[d] = __LINE__; return; }
                        if (hipDeviceSynchronize() != hipSuccess) { bad[d] = __LINE__; return; }
                        hipMemcpy(&Ag[d], dO1, 64, hipMemcpyDeviceToHost); hipMemcpy(&Bg[d], dO2, 128, hipMemcpyDeviceToHost);
                        hipFree(dB1); hipFree(dB2); hipFree(dS); hipFree(dO1); hipFree(dO2);
                    });
                for (auto &t : th) t.join();
                for (int d = 0; d < nr; d++) { if (bad[d]) printf("rank %d failed at line %d: %s\n", d, bad[d], gpbc_last_error()); EXPECT(bad[d] == 0); }
                for (int d = 0; d < nr; d++) {                            // every rank can run the 2-pairing check on its own copy of the sums
                    G2Affine nB; nB.Neg(Bg[d]);
                    EXPECT(PairingCheck({Ag[d], g1}, {H, nB}) == (forged == 0));
                }
            }
            printf("BLS aggregate verify over %d rank(s) with all-gathered partial sums: accepts, rejects a forged signature\n", nr);
        }
    }
    check(gpbc_comm_destroy());
    EXPECT(gpbc_comm_ranks() == 0);
    printf("RCCL all-gather over %d rank(s) inside the library%s: OK\n", nr, stub ? " (rccl TEST DOUBLE: tests/stub_rccl)" : "");

    // BLS aggregate verification (BASELINE config 3) through the boundary: x_i secret keys, pk_i = [x_i]g1, sigma_i = [x_i]H,
    // check e(sum rho_i pk_i, H) * e(g1, -sum rho_i sigma_i) == 1, then forge one signature.
    {
        const size_t n = 10000;
        G2Affine H; H.ScalarMultiplicationBase(Scalar(0xABCDEF12345ull));
        std::vector<Scalar> x(a.begin() + 100, a.begin() + 100 + n), rho(c.begin(), c.begin() + n);
        for (auto &r : rho) for (int j = 16; j < 32; j++) r.le[j] = 0;                 // 128-bit verifier scalars
        std::vector<G1Affine> pk = G1ScalarMultiplicationBatch({g1}, x);
        std::vector<G2Affine> sig = G2ScalarMultiplicationBatch({H}, x);
        G1Affine Ag; G2Affine Bg, nB;
        check(gpbc_g1_scalar_mul_sum(pk.data(), rho.data(), n, &Ag));
        check(gpbc_g2_scalar_mul_sum(sig.data(), rho.data(), n, &Bg));
        nB.Neg(Bg);
        EXPECT(PairingCheck({Ag, g1}, {H, nB}));
        const G2Affine sig0_mid = sig[n / 2];
        sig[n / 2] = sig[n / 2 + 1];
        check(gpbc_g2_scalar_mul_sum(sig.data(), rho.data(), n, &Bg));
        nB.Neg(Bg);
        EXPECT(!PairingCheck({Ag, g1}, {H, nB}));
        printf("BLS aggregate verify of %zu signatures: accepts, rejects a forged one\n", n);
        // the same two checks as ONE call with the points in device memory, the table on the host and a created stream
        // (gpbc_multi_pair_hostseg_dev), then the fail-closed self-check on that entry: a stale device table fails the call and
        // leaves zeroed outputs
        {
            G2Affine Bok, nBok;
            sig[n / 2] = sig0_mid;
            check(gpbc_g2_scalar_mul_sum(sig.data(), rho.data(), n, &Bok));
            nBok.Neg(Bok);
            const G1Affine Ps[4] = {Ag, g1, Ag, g1};
            const G2Affine Qs[4] = {H, nBok, H, nB};
            const uint64_t seg[3] = {0, 2, 4};
            hipStream_t st; HIP_OK(hipStreamCreate(&st));
            void *dP, *dQ, *dG;
            HIP_OK(hipMalloc(&dP, sizeof Ps)); HIP_OK(hipMalloc(&dQ, sizeof Qs)); HIP_OK(hipMalloc(&dG, 2 * sizeof(GT)));
            HIP_OK(hipMemcpy(dP, Ps, sizeof Ps, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(dQ, Qs, sizeof Qs, hipMemcpyHostToDevice));
            GT out[2], one, e = Pair({g1}, {g2});
            one.Div(e, e);
            for (int rep = 0; rep < 3; rep++) {
                check(gpbc_multi_pair_hostseg_dev(dP, dQ, seg, 2, dG, st));
                HIP_OK(hipMemcpy(out, dG, sizeof out, hipMemcpyDeviceToHost));
                EXPECT(out[0].Equal(one) && !out[1].Equal(one));
            }
            check(gpbc_debug_stale_table_once());
            EXPECT(gpbc_multi_pair_hostseg_dev(dP, dQ, seg, 2, dG, st) == GPBC_ERR_INTERNAL);
            HIP_OK(hipMemcpy(out, dG, sizeof out, hipMemcpyDeviceToHost));
            GT zero; std::memset(&zero, 0, sizeof zero);
            EXPECT(out[0].Equal(zero) && out[1].Equal(zero));
            check(gpbc_multi_pair_hostseg_dev(dP, dQ, seg, 2, dG, st));
            HIP_OK(hipMemcpy(out, dG, sizeof out, hipMemcpyDeviceToHost));
            EXPECT(out[0].Equal(one) && !out[1].Equal(one));
            hipFree(dP); hipFree(dQ); hipFree(dG); hipStreamDestroy(st);
            printf("hostseg_dev on a created stream: accepts, rejects, fails closed on a stale table\n");
        }
    }
    check(gpbc_shutdown());
    printf("multi-device OK\n");
    return 0;
}
