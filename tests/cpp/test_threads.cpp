// The C ABI promises re-entrancy (include/gpbc_bn254.h; SURVEY §8b: "a cgo replacement may be called concurrently from
// many OS threads").  Several host threads call the host-pointer entries at the same time — they share the default
// stream and therefore the internal per-stream workspace (Miller lines, GLV tables) — and every thread must get the
// results a lone caller gets.
#include <cstdio>
#include <cstring>
#include <string>
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

    // The reference's call shape (access/tree/access_tree_node.go:106-123, signature/bls01_signature/bls_signature.go:45,81): 64 OS
    // threads, each making ONE-element calls in a loop — Pair of one pair, PairingCheck of two, one ScalarMultiplication in G1 and
    // G2, one GT.Exp / Mul / Div.  The library combines whatever is waiting into one launch per kind (csrc/gpbc_common.hpp "Small
    // host-pointer calls"); every caller must get the bytes of its own call, i.e. what the sequential calls above returned.
    const int T2 = 64, REPS = 6;
    std::vector<std::thread> th2;
    std::vector<int> bad2(T2, 0);
    // sequential references for the derived values
    std::vector<GT> wantMul(T), wantDiv(T), wantExp(T);
    std::vector<G2Affine> wantQ(T);
    for (int t = 0; t < T; t++) {
        wantMul[t].Mul(want[t][0], want[t][1]);
        wantDiv[t].Div(want[t][0], want[t][1]);
        wantExp[t].Exp(want[t][2], ks[t][3]);
        wantQ[t] = G2ScalarMultiplicationBatch({Q[t][4]}, {ks[t][5]})[0];
    }
    // ... and the calls that are not combined but take a call lane each: ScalarMultiplicationBase (generator tables), HashToG2, Marshal /
    // Unmarshal of one point — sequential references first
    std::vector<G1Affine> wantBase(T);
    std::vector<G2Affine> wantHash(T);
    std::vector<std::vector<uint8_t>> wantWire(T);
    for (int t = 0; t < T; t++) {
        wantBase[t].ScalarMultiplicationBase(ks[t][6]);
        wantHash[t] = HashToG2("message of thread class " + std::to_string(t), "Hash String To Element In G2");
        wantWire[t] = Q[t][7].Marshal();
    }
    for (int u = 0; u < T2; u++)
        th2.emplace_back([&, u] {
            const int t = u % T;
            for (int rep = 0; rep < REPS; rep++) {
                const int i = (u * 7 + rep * 13) % N;
                G1Affine bs; bs.ScalarMultiplicationBase(ks[t][6]);
                if (!bs.Equal(wantBase[t])) bad2[u]++;
                if (!HashToG2("message of thread class " + std::to_string(t), "Hash String To Element In G2").Equal(wantHash[t])) bad2[u]++;
                const std::vector<uint8_t> wire = Q[t][7].Marshal();
                G2Affine back; back.Unmarshal(wire);
                if (wire != wantWire[t] || !back.Equal(Q[t][7])) bad2[u]++;
                GT e = Pair({P[t][i]}, {Q[t][i]});
                if (!e.Equal(want[t][i])) bad2[u]++;
                G1Affine neg; neg.Neg(P[t][i]);
                if (!PairingCheck({P[t][i], neg}, {Q[t][i], Q[t][i]})) bad2[u]++;          // e(P,Q) e(-P,Q) = 1
                if (PairingCheck({P[t][i], P[t][i]}, {Q[t][i], Q[t][i]})) bad2[u]++;         // e(P,Q)^2 != 1
                G1Affine r; r.ScalarMultiplication(P[t][i], ks[t][i]);
                if (!r.Equal(wantP[t][i])) bad2[u]++;
                G2Affine q; q.ScalarMultiplication(Q[t][4], ks[t][5]);
                if (!q.Equal(wantQ[t])) bad2[u]++;
                GT m; m.Mul(want[t][0], want[t][1]);
                if (!m.Equal(wantMul[t])) bad2[u]++;
                GT d; d.Div(want[t][0], want[t][1]);
                if (!d.Equal(wantDiv[t])) bad2[u]++;
                GT x; x.Exp(want[t][2], ks[t][3]);
                if (!x.Equal(wantExp[t])) bad2[u]++;
            }
        });
    for (auto &x : th2) x.join();
    int total2 = 0;
    for (int u = 0; u < T2; u++) total2 += bad2[u];
    if (total2) { printf("FAIL: %d mismatches among 64 threads of single calls\n", total2); return 1; }
    printf("threads OK\n");
    return 0;
}
