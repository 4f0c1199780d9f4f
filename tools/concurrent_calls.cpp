// Calls per second of the reference's call shape through the C ABI: T OS threads, each looping ONE-element calls —
// bn254.Pair of one pair (access/tree/access_tree_node.go:106-123), PairingCheck of two pairs
// (signature/bls01_signature/bls_signature.go:81), one G1 ScalarMultiplication (bls_signature.go:45) — for a fixed time, every
// result compared with the bytes a batched call returned for the same inputs.  What a cgo shim's goroutines would see
// ("may be called concurrently from many OS threads").  Prints one JSON object.
//
//   g++ -O2 -std=c++17 -pthread -Iinclude tools/concurrent_calls.cpp -Lgopairingbasedcryptography_amd -lgpbc_bn254 \
//       -Wl,-rpath,'$ORIGIN' -o gopairingbasedcryptography_amd/gpbc_concurrent_calls          (_build.py does this)
//   gpbc_concurrent_calls [--device D] [--seconds S] [--threads 1,8,64]
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "gpbc_bn254.h"

static void die(const char *what) { fprintf(stderr, "concurrent_calls: %s: %s\n", what, gpbc_last_error()); exit(1); }
#define OK(x) do { if ((x) < 0) die(#x); } while (0)

struct Pool {
    static constexpr int M = 256;                    // distinct inputs; thread u's call j takes input (u * 37 + j) % M
    std::vector<uint8_t> P, Q, negP, k, gt, sP;
};

struct Point { double calls_per_s, mean_ms; long calls, mismatches; };

template <class Body> static Point run_point(int T, double seconds, Body body) {
    std::atomic<bool> go{false}, stop{false};
    std::vector<long> calls(T, 0), bad(T, 0);
    std::vector<std::thread> th;
    for (int u = 0; u < T; u++)
        th.emplace_back([&, u] {
            while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
            for (long j = 0; !stop.load(std::memory_order_relaxed); j++) {
                if (!body(u, j)) bad[u]++;
                calls[u]++;
            }
        });
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop.store(true);
    for (auto &t : th) t.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    long c = 0, b = 0;
    for (int u = 0; u < T; u++) { c += calls[u]; b += bad[u]; }
    return Point{c / dt, c ? 1e3 * dt * T / c : 0.0, c, b};
}

int main(int argc, char **argv) {
    int device = 0;
    double seconds = 1.0;
    std::vector<int> threads = {1, 8, 64};
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (a == "--seconds" && i + 1 < argc) seconds = atof(argv[++i]);
        else if (a == "--threads" && i + 1 < argc) {
            threads.clear();
            for (char *tok = strtok(argv[++i], ","); tok; tok = strtok(nullptr, ",")) threads.push_back(atoi(tok));
        } else { fprintf(stderr, "usage: %s [--device D] [--seconds S] [--threads 1,8,64]\n", argv[0]); return 2; }
    }
    OK(gpbc_init(device));
    // g1 = (1, 2), g2 = the twist generator, gnark's Montgomery limbs (include/gpbc_bn254.hpp Generators)
    const uint64_t G1[8] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL,
                            0xa6ba871b8b1e1b3aULL, 0x14f1d651eb8e167bULL, 0xccdd46def0f28c58ULL, 0x1c14ef83340fbe5eULL};
    const uint64_t G2[16] = {0x8e83b5d102bc2026ULL, 0xdceb1935497b0172ULL, 0xfbb8264797811adfULL, 0x19573841af96503bULL,
                             0xafb4737da84c6140ULL, 0x6043dd5a5802d8c4ULL, 0x09e950fc52a02f86ULL, 0x14fef0833aea7b6bULL,
                             0x619dfa9d886be9f6ULL, 0xfe7fd297f59e9b78ULL, 0xff9e1a62231b7dfeULL, 0x28fd7eebae9e4206ULL,
                             0x64095b56c71856eeULL, 0xdc57f922327d3cbbULL, 0x55f935be33351076ULL, 0x0da4a0e693fd6482ULL};
    Pool p;
    const int M = Pool::M;
    p.P.resize(M * GPBC_G1_BYTES); p.Q.resize(M * GPBC_G2_BYTES); p.negP.resize(M * GPBC_G1_BYTES);
    p.k.assign(M * GPBC_SCALAR_BYTES, 0); p.gt.resize(M * GPBC_GT_BYTES); p.sP.resize(M * GPBC_G1_BYTES);
    std::vector<uint8_t> a(M * GPBC_SCALAR_BYTES, 0), b(M * GPBC_SCALAR_BYTES, 0), minus(M * GPBC_SCALAR_BYTES, 0);
    // r - 1 (little-endian): [r-1]P = -P
    const uint64_t RM1[4] = {0x43e1f593f0000000ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    uint64_t s = 0x424E323534ULL;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int i = 0; i < M; i++) {
        for (int w = 0; w < 3; w++) {                 // 192-bit scalars: below r
            uint64_t x = next(), y = next(), z = next();
            memcpy(&a[i * 32 + 8 * w], &x, 8); memcpy(&b[i * 32 + 8 * w], &y, 8); memcpy(&p.k[i * 32 + 8 * w], &z, 8);
        }
        memcpy(&minus[i * 32], RM1, 32);
    }
    OK(gpbc_g1_scalar_mul_batch(G1, 1, a.data(), M, p.P.data()));
    OK(gpbc_g2_scalar_mul_batch(G2, 1, b.data(), M, p.Q.data()));
    OK(gpbc_g1_scalar_mul_batch(p.P.data(), M, minus.data(), M, p.negP.data()));
    OK(gpbc_pair_batch(p.P.data(), p.Q.data(), M, p.gt.data()));                       // what every single call must reproduce
    OK(gpbc_g1_scalar_mul_batch(p.P.data(), M, p.k.data(), M, p.sP.data()));

    auto pair1 = [&](int u, long j) {
        const int i = (int)((u * 37 + j) % M);
        uint8_t out[GPBC_GT_BYTES];
        if (gpbc_pair_batch(&p.P[i * GPBC_G1_BYTES], &p.Q[i * GPBC_G2_BYTES], 1, out) < 0) return false;
        return memcmp(out, &p.gt[i * GPBC_GT_BYTES], GPBC_GT_BYTES) == 0;
    };
    auto check2 = [&](int u, long j) {                  // e(P, Q) e(-P, Q) = 1; every fourth call is a forgery (P twice) and must say 0
        const int i = (int)((u * 37 + j) % M);
        const bool forged = (j & 3) == 3;
        uint8_t Ps[2 * GPBC_G1_BYTES], Qs[2 * GPBC_G2_BYTES], ok = 9;
        memcpy(Ps, &p.P[i * GPBC_G1_BYTES], GPBC_G1_BYTES);
        memcpy(Ps + GPBC_G1_BYTES, forged ? &p.P[i * GPBC_G1_BYTES] : &p.negP[i * GPBC_G1_BYTES], GPBC_G1_BYTES);
        memcpy(Qs, &p.Q[i * GPBC_G2_BYTES], GPBC_G2_BYTES); memcpy(Qs + GPBC_G2_BYTES, &p.Q[i * GPBC_G2_BYTES], GPBC_G2_BYTES);
        const uint64_t seg[2] = {0, 2};
        if (gpbc_pairing_check(Ps, Qs, seg, 1, &ok) < 0) return false;
        return ok == (forged ? 0 : 1);
    };
    auto g1mul1 = [&](int u, long j) {
        const int i = (int)((u * 37 + j) % M);
        uint8_t out[GPBC_G1_BYTES];
        if (gpbc_g1_scalar_mul_batch(&p.P[i * GPBC_G1_BYTES], 1, &p.k[i * GPBC_SCALAR_BYTES], 1, out) < 0) return false;
        return memcmp(out, &p.sP[i * GPBC_G1_BYTES], GPBC_G1_BYTES) == 0;
    };
    // HashToG2 of one 32-byte message (every BLS Sign / Verify starts with it) and ScalarMultiplicationBase over the generator table
    const char DST[] = "Hash Bytes To Element In G2";
    std::vector<uint8_t> msgs(M * 32), hashed(M * GPBC_G2_BYTES), based(M * GPBC_G1_BYTES);
    std::vector<uint64_t> offs(M + 1);
    for (int i = 0; i < M; i++) { for (int w = 0; w < 4; w++) { uint64_t x = next(); memcpy(&msgs[i * 32 + 8 * w], &x, 8); } offs[i] = 32 * (uint64_t)i; }
    offs[M] = 32 * (uint64_t)M;
    OK(gpbc_hash_to_g2(msgs.data(), offs.data(), M, DST, sizeof DST - 1, hashed.data()));
    gpbc_fixed_base *gen = nullptr;
    OK(gpbc_g1_fixed_base_create(G1, 1, &gen));
    OK(gpbc_fixed_base_msm(gen, p.k.data(), M, based.data()));
    auto hash1 = [&](int u, long j) {
        const int i = (int)((u * 37 + j) % M);
        const uint64_t off[2] = {0, 32};
        uint8_t out[GPBC_G2_BYTES];
        if (gpbc_hash_to_g2(&msgs[i * 32], off, 1, DST, sizeof DST - 1, out) < 0) return false;
        return memcmp(out, &hashed[i * GPBC_G2_BYTES], GPBC_G2_BYTES) == 0;
    };
    auto base1 = [&](int u, long j) {
        const int i = (int)((u * 37 + j) % M);
        uint8_t out[GPBC_G1_BYTES];
        if (gpbc_fixed_base_msm(gen, &p.k[i * GPBC_SCALAR_BYTES], 1, out) < 0) return false;
        return memcmp(out, &based[i * GPBC_G1_BYTES], GPBC_G1_BYTES) == 0;
    };
    for (int w = 0; w < 8; w++) { pair1(0, w); check2(0, w); g1mul1(0, w); hash1(0, w); base1(0, w); }            // lanes, streams, code objects

    printf("{\"what\": \"T OS threads looping one-element host-pointer calls for %.2f s per point; every result compared with a batched call's bytes\", \"seconds_per_point\": %.3f", seconds, seconds);
    long total_bad = 0;
    const char *names[5] = {"pair_batch_1", "pairing_check_2_pairs", "g1_scalar_mul_1", "hash_to_g2_1", "g1_scalar_mul_base_1"};
    for (int op = 0; op < 5; op++) {
        printf(", \"%s\": {", names[op]);
        for (size_t t = 0; t < threads.size(); t++) {
            const int T = threads[t];
            Point r = op == 0 ? run_point(T, seconds, pair1) : op == 1 ? run_point(T, seconds, check2) : op == 2 ? run_point(T, seconds, g1mul1)
                    : op == 3 ? run_point(T, seconds, hash1) : run_point(T, seconds, base1);
            total_bad += r.mismatches;
            printf("%s\"%d\": {\"calls_per_s\": %.1f, \"mean_ms_per_call\": %.4f, \"calls\": %ld, \"mismatches\": %ld}", t ? ", " : "", T, r.calls_per_s, r.mean_ms, r.calls, r.mismatches);
        }
        printf("}");
    }
    printf(", \"mismatches\": %ld}\n", total_bad);
    gpbc_fixed_base_destroy(gen);
    gpbc_shutdown();
    return total_bad ? 1 : 0;
}
