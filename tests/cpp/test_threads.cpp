// The C ABI promises re-entrancy (include/gpbc_bn254.h; SURVEY §8b: "a cgo replacement may be called concurrently from
// many OS threads").  Several host threads call the host-pointer entries at the same time — they share the default
// stream and therefore the internal per-stream workspace (Miller lines, GLV tables) — and every thread must get the
// results a lone caller gets.
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "gpbc_bn254.hpp"

using namespace bn254;

int main() {
    Init(0);
    G1Affine g1; G2Affine g2; Generators(g1, g2);
    const int T = 6, N = 96;
    // per-thread inputs: P_i = [a_i] g1, Q_i = [b_i] g2 with thread-specific scalars
    std::vector<std::vector<G1Affine>> P(T);
    std::vector<std::vector<G2Affine>> Q(T);
    std::vector<std::vector<GT>> want(T), got(T);
    std::vector<std::vector<G1Affine>> wantP(T), gotP(T);
    std::vector<std::vector<Scalar>> ks(T);
    for (int t = 0; t < T; t++) {
        std::vector<Scalar> a(N), b(N);
        for (int i = 0; i < N; i++) { a[i] = Scalar(1000003ull * (t + 1) + 7919ull * i + 1); b[i] = Scalar(998244353ull * (t + 2) + 104729ull * i + 3); }
        P[t] = G1ScalarMultiplicationBatch({g1}, a);
        Q[t] = G2ScalarMultiplicationBatch({g2}, b);
        ks[t] = b;
        want[t] = PairBatch(P[t], Q[t]);                       // sequential reference
        wantP[t] = G1ScalarMultiplicationBatch(P[t], b);
    }
    std::vector<std::thread> th;
    std::vector<int> bad(T, 0);
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            for (int rep = 0; rep < 4; rep++) {
                got[t] = PairBatch(P[t], Q[t]);
                gotP[t] = G1ScalarMultiplicationBatch(P[t], ks[t]);
                for (int i = 0; i < N; i++) {
                    if (!got[t][i].Equal(want[t][i])) bad[t]++;
                    if (!gotP[t][i].Equal(wantP[t][i])) bad[t]++;
                }
            }
        });
    for (auto &x : th) x.join();
    int total = 0;
    for (int t = 0; t < T; t++) total += bad[t];
    if (total) { printf("FAIL: %d mismatches under concurrency\n", total); return 1; }
    printf("threads OK\n");
    return 0;
}
