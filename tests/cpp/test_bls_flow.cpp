// Mirrors the reference's BLS tests (signature/bls01_signature/bls_signature_test.go:8-72) on the C++ host mirror
// of the gnark surface (include/gpbc_bn254.hpp): key generation, sign, verify through PairingCheck on the GPU,
// plus the wrong-message / wrong-key negatives.  The message point is the real hash, bn254.HashToG2(msg, dst) with the
// reference's DST (hash/hash_to.go:204-210), computed on the device; bls_signature.go:58-89.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "gpbc_bn254.hpp"

using namespace bn254;

static Scalar scalar_from(const std::string &s, uint64_t salt) {
    Scalar k;
    uint64_t h = 1469598103934665603ULL ^ salt;
    for (int i = 0; i < 31; i++) {            // 248-bit value < r
        for (unsigned char c : s) { h ^= c; h *= 1099511628211ULL; }
        h ^= (uint64_t)i; h *= 1099511628211ULL;
        k.le[i] = (uint8_t)(h >> 32);
    }
    return k;
}
struct KeyPair { G1Affine pk; Scalar sk; };
static KeyPair KeyGenerate(uint64_t seed) {                      // bls_signature.go:38-56
    KeyPair kp;
    kp.sk = scalar_from("secret-key", seed);
    kp.pk.ScalarMultiplicationBase(kp.sk);
    return kp;
}
static G2Affine HashStandIn(const std::string &m) { return HashToG2(m, "Hash String To Element In G2"); }
static G2Affine Sign(const Scalar &sk, const std::string &m) {   // bls_signature.go:58-69
    G2Affine hm = HashStandIn(m), sig;
    sig.ScalarMultiplication(hm, sk);
    return sig;
}
static bool Verify(const G1Affine &pk, const std::string &m, const G2Affine &sigma) {   // bls_signature.go:71-89
    G1Affine g1; G2Affine g2; Generators(g1, g2);
    G2Affine hm = HashStandIn(m), inv;
    inv.Neg(sigma);
    return PairingCheck({pk, g1}, {hm, inv});
}
#define EXPECT(c) do { if (!(c)) { printf("FAIL line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main() {
    setenv("GPBC_TEST_KNOBS", "1", 1);          // this is a test process: the fault-injection knob answers only with this set
    Init(0);
    KeyPair a = KeyGenerate(1), b = KeyGenerate(2);
    G2Affine sig = Sign(a.sk, "hello pairing");
    EXPECT(Verify(a.pk, "hello pairing", sig));                 // TestBLSFlow
    EXPECT(!Verify(a.pk, "hello pairinG", sig));                // wrong message
    EXPECT(!Verify(b.pk, "hello pairing", sig));                // wrong key
    // bilinearity through the mirror: e([2]g1, g2) == e(g1, g2)^2 == e(g1,[2]g2)
    G1Affine g1, g1x2; G2Affine g2, g2x2; Generators(g1, g2);
    g1x2.ScalarMultiplication(g1, Scalar(2)); g2x2.ScalarMultiplication(g2, Scalar(2));
    GT e = Pair({g1}, {g2}), e2 = Pair({g1x2}, {g2}), e2b = Pair({g1}, {g2x2}), sq;
    sq.Mul(e, e);
    EXPECT(e2.Equal(sq) && e2b.Equal(sq));
    GT ex; ex.Exp(e, Scalar(2)); EXPECT(ex.Equal(sq));
    GT q; q.Div(sq, e); EXPECT(q.Equal(e));
    // the singles of the mirror the reference calls one at a time: GT.Inverse (access/tree/access_tree_node.go:157), generator
    // multiplications through the lazily built fixed-base tables against the variable-base kernel, HashToG1 with an EMPTY tag
    {
        GT inv, prod, one; inv.Inverse(e); prod.Mul(e, inv); one.Div(e, e);
        EXPECT(prod.Equal(one));
        for (uint64_t k : {0ull, 1ull, 5ull, 0xFFFFFFFFFFFFFFFFull}) {
            G1Affine b1, v1; b1.ScalarMultiplicationBase(Scalar(k)); v1.ScalarMultiplication(g1, Scalar(k));
            G2Affine b2, v2; b2.ScalarMultiplicationBase(Scalar(k)); v2.ScalarMultiplication(g2, Scalar(k));
            EXPECT(b1.Equal(v1) && b2.Equal(v2));
        }
        G1Affine h0 = HashToG1("abc", ""), h1 = HashToG1("abc", "d");
        EXPECT(!h0.IsInfinity() && !h0.Equal(h1) && h0.Equal(HashToG1("abc", "")));
    }
    bool threw = false;
    try { Pair({g1, g1}, {g2}); } catch (const std::invalid_argument &) { threw = true; }
    EXPECT(threw);                                               // "invalid inputs sizes"
    // wire formats: Marshal / Bytes / Unmarshal round trips (reference serialization/serialization_curve.go:5-33)
    {
        G1Affine back; G2Affine back2; GT backt;
        EXPECT(back.Unmarshal(a.pk.Marshal()) == 64 && back.Equal(a.pk));
        auto c = a.pk.Bytes();
        EXPECT(back.Unmarshal(std::vector<uint8_t>(c.begin(), c.end())) == 32 && back.Equal(a.pk));
        EXPECT(back2.Unmarshal(sig.Marshal()) == 128 && back2.Equal(sig));
        auto c2 = sig.Bytes();
        EXPECT(back2.Unmarshal(std::vector<uint8_t>(c2.begin(), c2.end())) == 64 && back2.Equal(sig));
        backt.Unmarshal(e.Marshal());
        EXPECT(backt.Equal(e));
        auto g1c = g1.Bytes();                                   // (1, 2): smaller-Y flag, x = 1
        EXPECT(g1c[0] == 0x80 && g1c[31] == 1);
        std::vector<uint8_t> bad(64, 0); bad[31] = 1; bad[63] = 3;   // (1, 3) is not on the curve
        bool rejected = false;
        try { back.Unmarshal(bad); } catch (const std::invalid_argument &) { rejected = true; }
        EXPECT(rejected);
    }
    // fixed-Q multi-pairing and fixed-base tables through the mirror
    {
        std::vector<G1Affine> Ps = {a.pk, g1, b.pk, g1};               // two segments of two points against {hm, -sigma}
        G2Affine hm = HashStandIn("hello pairing"), inv; inv.Neg(sig);
        std::vector<GT> fq = PairFixedQ(Ps, {hm, inv});
        GT one; one.Div(e, e);
        EXPECT(fq.size() == 2 && fq[0].Equal(one) && !fq[1].Equal(one));   // a's signature verifies, b's key does not
        EXPECT(fq[0].Equal(Pair({a.pk, g1}, {hm, inv})) && fq[1].Equal(Pair({b.pk, g1}, {hm, inv})));
        G1FixedBase fb({g1, a.pk});
        std::vector<G1Affine> s = fb.Msm({Scalar(5), Scalar(7), Scalar(0), Scalar(1)});
        G1Affine t5, t7, want;
        t5.ScalarMultiplication(g1, Scalar(5)); t7.ScalarMultiplication(a.pk, Scalar(7));
        check(gpbc_g1_sum(std::vector<G1Affine>{t5, t7}.data(), 2, &want));
        EXPECT(s.size() == 2 && s[0].Equal(want) && s[1].Equal(a.pk));
    }
    // forged-signature reject through every host-table multi-pairing path (general = one Miller loop per pair, pipelined or two
    // kernels; shared-squaring chunks), repeated so that every call finds the previous call's tables in the recycled scratch and
    // pinned staging buffers; then the fail-closed self-check: a device table that is not the caller's must fail the call
    // (profiles/r02_pool_bisect.txt is the failure this guards against: "true" for a forged signature from an empty product)
    {
        G2Affine hm = HashStandIn("hello pairing"), inv, forged_inv; inv.Neg(sig);
        G2Affine forged = Sign(b.sk, "hello pairing"); forged_inv.Neg(forged);       // b's signature offered under a's key
        struct Mode { const char *name; int chunk, pipelined; long latency; };
        const Mode modes[] = {{"latency form (one pairing per wavefront)", 0, 1, 2048}, {"throughput kernels, pipelined Miller loop", 0, 1, 0},
                              {"two-kernel Miller loop", 0, 0, 0}, {"pipelined without waiting", 0, 2, 0}, {"chunks of 2", 2, 1, 0}, {"chunks of 1", 1, 1, 2048}};
        for (const Mode &m : modes) {
            check(gpbc_set_multi_pair_chunk(m.chunk)); check(gpbc_set_pipelined_miller(m.pipelined)); check(gpbc_set_latency_path(m.latency));
            for (int rep = 0; rep < 3; rep++) {
                EXPECT(Verify(a.pk, "hello pairing", sig));
                EXPECT(!PairingCheck({a.pk, g1}, {hm, forged_inv}));
                // valid, forged and empty segment in one call (an empty product IS one); then the same pairs in the other order
                const G1Affine Ps[4] = {a.pk, g1, a.pk, g1};
                const G2Affine Qv[4] = {hm, inv, hm, forged_inv}, Qw[4] = {hm, forged_inv, hm, inv};
                const uint64_t seg[4] = {0, 2, 4, 4};
                uint8_t ok[3] = {9, 9, 9};
                check(gpbc_pairing_check(Ps, Qv, seg, 3, ok));
                EXPECT(ok[0] == 1 && ok[1] == 0 && ok[2] == 1);
                check(gpbc_pairing_check(Ps, Qw, seg, 3, ok));
                EXPECT(ok[0] == 0 && ok[1] == 1 && ok[2] == 1);
            }
            const G1Affine Ps[2] = {a.pk, g1};
            const G2Affine Qf[2] = {hm, forged_inv};
            const uint64_t seg[2] = {0, 2};
            uint8_t ok[1] = {9};
            GT out;
            check(gpbc_debug_stale_table_once());
            EXPECT(gpbc_pairing_check(Ps, Qf, seg, 1, ok) == GPBC_ERR_INTERNAL && ok[0] != 1);      // the device saw an empty segment: refused, not "true"
            check(gpbc_debug_stale_table_once());
            EXPECT(gpbc_multi_pair(Ps, Qf, seg, 1, &out) == GPBC_ERR_INTERNAL);
            EXPECT(!PairingCheck({a.pk, g1}, {hm, forged_inv}) && Verify(a.pk, "hello pairing", sig));   // and the next calls are sound
            printf("  multi-pairing path '%s': accepts, rejects, fails closed on a stale table\n", m.name);
        }
        check(gpbc_set_multi_pair_chunk(0)); check(gpbc_set_pipelined_miller(1)); check(gpbc_set_latency_path(2048));
    }
    printf("BLS flow OK\n");
    return 0;
}
