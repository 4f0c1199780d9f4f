// Host build of the DEVICE arithmetic (csrc/fe29.hip.hpp, tower29.hip.hpp, curve29.hip.hpp, pairing29.hip.hpp) with -DGPBC_BOUNDS:
// every field element carries data-independent magnitude bounds and every product asserts that its int64 column
// accumulators cannot overflow (abort() on violation).  This is a verification harness for tests/ only — it is
// never loaded by the product path (which has no CPU fallback).
//
// build: g++ -O2 -std=c++17 -DGPBC_BOUNDS -shared -fPIC -o tools/libgpbc_bounds.so tools/bounds_check.cpp
#ifndef GPBC_BOUNDS
#define GPBC_BOUNDS
#endif
#include <cstring>
#include <tuple>
#include "../gopairingbasedcryptography_amd/csrc/curve29.hip.hpp"
#include "../gopairingbasedcryptography_amd/csrc/pairing29.hip.hpp"
#include "../gopairingbasedcryptography_amd/csrc/pairing29_pair.hip.hpp"
#include "../gopairingbasedcryptography_amd/csrc/wide29.hip.hpp"
#include "../gopairingbasedcryptography_amd/csrc/wire29.hip.hpp"
#include "../gopairingbasedcryptography_amd/csrc/h2c29.hip.hpp"
#include "../gopairingbasedcryptography_amd/csrc/msm29.hip.hpp"
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

using namespace gpbc;

// Host stand-in for the DPP lane swap: the two lanes of a pair run as two threads; swap() is a rendezvous.
struct PairRendezvous {
    std::mutex mu; std::condition_variable cv;
    const void *slot[2] = {nullptr, nullptr}; int arrived = 0, left = 0; long gen = 0;
    template <class T> T exchange(int me, const T &v) {
        std::unique_lock<std::mutex> lk(mu);
        while (left != 0) cv.wait(lk);                 // previous exchange fully drained
        slot[me] = &v;
        long g = gen;
        if (++arrived == 2) { gen++; left = 2; cv.notify_all(); }
        else while (gen == g) cv.wait(lk);
        T out = *static_cast<const T *>(slot[1 - me]);
        if (--left == 0) { arrived = 0; cv.notify_all(); }
        else while (left != 0) cv.wait(lk);            // keep my value alive until the partner has copied it
        return out;
    }
};
struct PairHost {
    bool odd; PairRendezvous *rv;
    Fe swap(const Fe &a) const { return rv->exchange<Fe>(odd, a); }
    F2 swap(const F2 &a) const { return rv->exchange<F2>(odd, a); }
    F6 swap(const F6 &a) const { return rv->exchange<F6>(odd, a); }
    Fe sgn(const Fe &a) const { return odd ? a : fe_neg(a); }                       // device: one multiplication by the lane's +-1
    Fe add_swap(const Fe &a, const Fe &b) const { return fe_add(a, swap(b)); }       // device: v_add_u32_dpp
    static bool all(bool c) { return c; }                                            // device: the whole wavefront must agree
};
static std::mutex g_stats_mu;
static BoundStats g_stats_total;
static void stats_flush() { std::lock_guard<std::mutex> lk(g_stats_mu); bound_stats_merge(g_stats_total, bound_stats()); }

template <class F, class LoadA, class StoreA> static void smul_batch(const uint8_t *B, const uint8_t *K, size_t n, uint8_t *out, size_t pt, LoadA ld, StoreA st) {
    constexpr int KK = 4;                                  // same grouping as the kernels: one shared inversion per 4 points
    size_t T = (n + KK - 1) / KK;
    for (size_t t = 0; t < T; t++) {
        JacP<F> res[KK];
        for (int j = 0; j < KK; j++) {
            size_t i = t + (size_t)j * T;
            if (i >= n) { jac_set_inf(res[j]); continue; }
            uint32_t k[8]; memcpy(k, K + 32 * i, 32);
            alignas(16) int32_t tab[glv_table_dwords<F>()];
            scalar_mul29_best(res[j], ld(B + pt * i), k, tab);
        }
        AffP<F> aff[KK];
        jac_to_affine_batch<F, KK>(aff, res);
        for (int j = 0; j < KK; j++) { size_t i = t + (size_t)j * T; if (i < n) st(out + pt * i, aff[j]); }
    }
}

// Bucket MSM exactly as csrc/gpbc_msm.hip runs it — the same per-lane pieces of msm29.hip.hpp in the same order (bucket sums by
// mixed additions, group reduction with the small multiplication, fan-in sums, Horner over the windows) — with the points kept as
// objects so that their tracked intervals flow from step to step.  out = sum_i [K_i] B_i (affine, gnark layout).
template <class F, class LoadA, class StoreA> static void msm_host(const uint8_t *B, const uint8_t *K, size_t n, int c, uint8_t *out, size_t pt, LoadA ld, StoreA st) {
    const int W = (256 + c - 1) / c;
    const size_t nb = (size_t)1 << c;
    std::vector<std::vector<size_t>> members((size_t)W * nb);
    for (size_t i = 0; i < n; i++) {
        if (bytes_all_zero(B + pt * i, (int)(pt / 4))) continue;
        uint32_t k[8]; memcpy(k, K + 32 * i, 32);
        for (int w = 0; w < W; w++) { uint32_t d = msm_digit(k, w, c); if (d) members[(size_t)w * nb + d].push_back(i); }
    }
    std::vector<JacP<F>> buckets((size_t)W * nb);
    for (size_t key = 0; key < buckets.size(); key++) {
        const auto &m = members[key];
        msm_bucket_sum(buckets[key], 0, m.size(), [&](size_t j) { return ld(B + pt * m[j]); });
    }
    const size_t gpw = nb >= (size_t)MSM_GROUP ? nb / MSM_GROUP : 1, gsz = nb >= (size_t)MSM_GROUP ? MSM_GROUP : nb;
    std::vector<JacP<F>> wsum(W);
    for (int w = 0; w < W; w++) {
        JacP<F> acc, s; jac_set_inf(acc);
        for (size_t g = 0; g < gpw; g++) {
            JacP<F> r;
            uint32_t lo = (uint32_t)(g * gsz), hi = lo + (uint32_t)gsz - 1;
            msm_group_reduce(r, lo ? lo : 1u, hi, [&](uint32_t d) { return buckets[(size_t)w * nb + d]; });
            jac_add(s, acc, r); acc = s;
        }
        wsum[w] = acc;
    }
    JacP<F> acc = wsum[W - 1];
    for (int w = W - 2; w >= 0; w--) msm_horner_step(acc, wsum[w], c);
    AffP<F> a; jac_to_affine(a, acc);
    st(out, a);
}
// host memory policy of the latency ("wide") form, csrc/wide29.hip.hpp: the slots are interval-carrying values, the lanes of a
// phase run one after the other and their stores land when the phase ends (what the barrier does on the device)
struct WideHost {
    static constexpr bool LIMB_PARALLEL = false;          // the device spreads some linear phases over one limb per lane; the host runs their Fe-level form
    std::vector<F2> s = std::vector<F2>(W_SLOTS, f2_zero());
    std::vector<std::pair<int, F2>> pending;
    std::vector<std::tuple<int, int, Fe>> pending_half;
    F2 ld(int i) const { return s[i]; }
    Fe ldh(int i, int h) const { return h ? s[i].a1 : s[i].a0; }
    void st(int i, const F2 &v) { pending.emplace_back(i, v); }
    void sth(int i, int h, const Fe &v) { pending_half.emplace_back(i, h, v); }
    int waves() const { return 2; }                        // the form with the helper wave (half products in wide_mul); the one-wave form multiplies whole F2 products: f2_mul, covered by the throughput kernels' runs
    template <class B> void run2(int n, B &&body) { run(n, body); }
    void sync_waves() const {}
    template <class B> void limbs(int n_comp, B &&body) { LimbOpsFe<WideHost> o{*this}; run(n_comp, [&](int c) { body(o, c); }); }
    template <class B> void run(int n, B &&body) {
        for (int l = 0; l < n; l++) body(l);
        for (auto &p : pending) s[p.first] = p.second;
        for (auto &p : pending_half) (std::get<1>(p) ? s[std::get<0>(p)].a1 : s[std::get<0>(p)].a0) = std::get<2>(p);
        pending.clear(); pending_half.clear();
    }
};
extern "C" {

// one pairing per "wavefront": Miller loop and final exponentiation of the latency form (k_miller_wide / k_final_exp_wide)
void hc_pair_wide(const uint8_t *P, const uint8_t *Q, size_t n, uint8_t *out, int do_final_exp) {
    for (size_t i = 0; i < n; i++) {
        const uint8_t *p = P + 64 * i, *q = Q + 128 * i;
        WideHost m;
        if (bytes_all_zero(p, 16) || bytes_all_zero(q, 32)) { m.s[0] = f2_one(); }
        else {
            G1A a{fe_load(p), fe_load(p + 32)};
            G2A b{f2_load(q), f2_load(q + 64)};
            std::vector<F2> ring;                          // the device passes the 88 lines through an LDS ring, wave to wave
            wide_miller_lines(m, a, b, [&](int) { m.run(3, [&](int t) { ring.push_back(m.ld(W_L0 + t)); }); });
            wide_miller_accumulate(m, wv(0), [&](int j) { m.run(3, [&](int t) { m.st(W_CL + t, ring[3 * j + t]); }); });
        }
        if (do_final_exp) wide_final_exp(m);
        for (int k = 0; k < 6; k++) f2_store(out + 384 * i + 64 * k, m.s[k]);
        stats_flush();
    }
}
// GT.Exp of the latency form (k_gt_exp_wide): x any Fp12 element, k a 256-bit plain exponent
void hc_gt_exp_wide(const uint8_t *x, const uint8_t *k, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        WideHost m;
        uint32_t kw[8];
        memcpy(kw, k + 32 * i, 32);
        for (int c = 0; c < 6; c++) m.s[wv(1) + c] = f2_load(x + 384 * i + 64 * c);
        wide_exp256(m, kw);
        for (int c = 0; c < 6; c++) f2_store(out + 384 * i + 64 * c, m.s[c]);
        stats_flush();
    }
}
void hc_pair(const uint8_t *P, const uint8_t *Q, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        const uint8_t *p = P + 64 * i, *q = Q + 128 * i;
        F12 f;
        if (bytes_all_zero(p, 16) || bytes_all_zero(q, 32)) f = f12_one();
        else {
            G1A a{fe_load(p), fe_load(p + 32)};
            G2A b{f2_load(q), f2_load(q + 64)};
            f = final_exp29(miller_loop29(a, b));
        }
        f12_store(out + 384 * i, f);
    }
}
void hc_miller(const uint8_t *P, const uint8_t *Q, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        G1A a{fe_load(P + 64 * i), fe_load(P + 64 * i + 32)};
        G2A b{f2_load(Q + 128 * i), f2_load(Q + 128 * i + 64)};
        f12_store(out + 384 * i, miller_loop29(a, b));
    }
}
void hc_final_exp(const uint8_t *F, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) { F12 f; f12_load(f, F + 384 * i); f12_store(out + 384 * i, final_exp29(f)); }
}
void hc_g1_mul(const uint8_t *B, const uint8_t *K, size_t n, uint8_t *out) {
    smul_batch<Fe>(B, K, n, out, 64,
                   [](const uint8_t *p) { return AffP<Fe>{fe_load(p), fe_load(p + 32), bytes_all_zero(p, 16)}; },
                   [](uint8_t *p, const AffP<Fe> &r) { fe_store(p, r.x); fe_store(p + 32, r.y); });
}
void hc_g2_mul(const uint8_t *B, const uint8_t *K, size_t n, uint8_t *out) {
    smul_batch<F2>(B, K, n, out, 128,
                   [](const uint8_t *p) { return AffP<F2>{f2_load(p), f2_load(p + 64), bytes_all_zero(p, 32)}; },
                   [](uint8_t *p, const AffP<F2> &r) { f2_store(p, r.x); f2_store(p + 64, r.y); });
}
// GLV split of n 256-bit scalars: out rows = [k1 (5 x u32), k2 (5 x u32), neg1, neg2] as 12 u32
void hc_glv_split(const uint8_t *K, size_t n, uint32_t *out) {
    for (size_t i = 0; i < n; i++) {
        uint32_t k[8]; memcpy(k, K + 32 * i, 32);
        GlvSplit s; glv_split(s, k);
        memcpy(out + 12 * i, s.k1, 20); memcpy(out + 12 * i + 5, s.k2, 20);
        out[12 * i + 10] = s.neg1; out[12 * i + 11] = s.neg2;
    }
}
// G2 scalar multiplication by the two-dimensional GLV loop (the G2 kernels use the four-dimensional GLS loop; this keeps the
// GLV instantiation for F2 under test: the fixed-base table builder may use either)
void hc_g2_mul_glv(const uint8_t *B, const uint8_t *K, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        uint32_t k[8]; memcpy(k, K + 32 * i, 32);
        alignas(16) int32_t tab[glv_table_dwords<F2>()];
        AffP<F2> b{f2_load(B + 128 * i), f2_load(B + 128 * i + 64), bytes_all_zero(B + 128 * i, 32)}, r;
        JacP<F2> j;
        scalar_mul29_jac<F2>(j, b, k, tab);
        jac_to_affine(r, j);
        f2_store(out + 128 * i, r.x); f2_store(out + 128 * i + 64, r.y);
    }
}
// GLS split of n scalars: out rows = 4 x (3 magnitude words, sign) u32
void hc_gls_split(const uint8_t *K, size_t n, uint32_t *out) {
    for (size_t i = 0; i < n; i++) {
        uint32_t k[8]; memcpy(k, K + 32 * i, 32);
        GlsSplit s; gls_split(s, k);
        for (int c = 0; c < 4; c++) { memcpy(out + 16 * i + 4 * c, s.k[c], 12); out[16 * i + 4 * c + 3] = s.neg[c]; }
    }
}
void hc_fp_mul(const uint8_t *A, const uint8_t *B, size_t n, uint8_t *out) {
    // gnark-form a (= x R) and b (= y R), R = 2^256: internal product of the converted operands is x y R' -> stored as x y R
    for (size_t i = 0; i < n; i++) fe_store(out + 32 * i, fe_mul(fe_load(A + 32 * i), fe_load(B + 32 * i)));
}
// out = a^-1 by the safegcd inversion, out2 = by the Fermat power (gnark-form in and out: (xR)^-1 stored as x^-1 R)
void hc_fp_inv(const uint8_t *A, size_t n, uint8_t *out, uint8_t *out2) {
    for (size_t i = 0; i < n; i++) {
        Fe a = fe_load(A + 32 * i);
        fe_store(out + 32 * i, fe_inv(a));
        fe_store(out2 + 32 * i, fe_inv_fermat(a));
    }
}
// (a / p) by the divstep-based symbol: 1, -1, or 0 where it does not decide within its 960 steps (and for a = 0)
void hc_fp_legendre(const uint8_t *A, size_t n, int8_t *out) {
    for (size_t i = 0; i < n; i++) out[i] = (int8_t)fe_legendre(fe_load(A + 32 * i));
}
void hc_gt_mul(const uint8_t *A, const uint8_t *B, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) { F12 a, b; f12_load(a, A + 384 * i); f12_load(b, B + 384 * i); f12_store(out + 384 * i, f12_mul(a, b)); }
}
// a / b as k_gt_binary computes it (norm-one divisors: conjugate instead of the Fp6 inversion)
void hc_gt_div(const uint8_t *A, const uint8_t *B, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        F12 a, b; f12_load(a, A + 384 * i); f12_load(b, B + 384 * i);
        f12_store(out + 384 * i, f12_mul(a, f12_inv_gt(b, [](bool c) { return c; })));
    }
}
void hc_gt_inv(const uint8_t *A, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) { F12 a; f12_load(a, A + 384 * i); f12_store(out + 384 * i, f12_inv(a)); }
}
void hc_gt_sqr(const uint8_t *A, size_t n, uint8_t *out, int cyclo) {
    for (size_t i = 0; i < n; i++) { F12 a; f12_load(a, A + 384 * i); f12_store(out + 384 * i, cyclo ? f12_cyclo_sqr(a) : f12_sqr(a)); }
}
// pair-lane forms: Miller accumulator fed by the single-lane line phase, and the final exponentiation
void hc_pair_lanes(const uint8_t *P, const uint8_t *Q, size_t n, uint8_t *out, int do_final_exp) {
    for (size_t i = 0; i < n; i++) {
        const uint8_t *p = P + 64 * i, *q = Q + 128 * i;
        G1A a{fe_load(p), fe_load(p + 32)};
        G2A b{f2_load(q), f2_load(q + 64)};
        LineS lines[MILLER_LINES];
        int cnt = 0;
        miller_lines(a, b, [&](const LineS &l) { lines[cnt++] = l; });
        PairRendezvous rv;
        auto lane = [&](bool odd) {
            PairHost x{odd, &rv};
            int k = 0;
            F6 h = miller_accumulate_pair(x, [&]() -> LineS { return lines[k++]; });
            if (do_final_exp) h = final_exp_pair(x, h);
            f6_store(out + 384 * i + (odd ? 192 : 0), h);
            stats_flush();
        };
        std::thread t1(lane, true);
        lane(false);
        t1.join();
    }
}
// product of the Miller functions of n pairs on ONE lane pair with shared squarings (k_miller_accumulate_chunks)
void hc_pair_lanes_multi(const uint8_t *P, const uint8_t *Q, size_t n, uint8_t *out) {
    std::vector<LineS> lines(n * MILLER_LINES);
    for (size_t i = 0; i < n; i++) {
        G1A a{fe_load(P + 64 * i), fe_load(P + 64 * i + 32)};
        G2A b{f2_load(Q + 128 * i), f2_load(Q + 128 * i + 64)};
        int cnt = 0;
        miller_lines(a, b, [&](const LineS &l) { lines[i * MILLER_LINES + cnt++] = l; });
    }
    PairRendezvous rv;
    auto lane = [&](bool odd) {
        PairHost x{odd, &rv};
        F6 h = miller_accumulate_multi(x, (int)n, [&](int p, int li) -> LineS { return lines[(size_t)p * MILLER_LINES + li]; });
        f6_store(out + (odd ? 192 : 0), h);
        stats_flush();
    };
    std::thread t1(lane, true);
    lane(false);
    t1.join();
}
// fixed-Q multi-pairing exactly as k_q_lines + k_q_lines_scale + k_g1_line_point + k_miller_accumulate_fixed_q do it: the RAW line
// coefficients of every Q_i (miller_lines_raw) scaled to c0 = 1, evaluated at (xP / yP, 1 / yP) inside the accumulator (one
// Fp x Fp2 product per lane, swapped), shared squarings on one lane pair; out = the Miller value of prod e(P_i, Q_i) up to a
// factor in Fp2 (do_final_exp = 0: the caller applies the final exponentiation, which removes it) or the GT value
void hc_pair_fixed_q(const uint8_t *P, const uint8_t *Q, size_t n, uint8_t *out, int do_final_exp) {
    std::vector<Line34> tab(n * MILLER_LINES);
    std::vector<G1A> pts(n);                                    // (x / y, 1 / y)
    for (size_t base = 0; base < n; base += 8)                  // k_g1_line_point: groups of 8 points share one inversion
        fe_batch_inverse<8>(n - base < 8 ? (int)(n - base) : 8, [&](int j) { return fe_load(P + 64 * (base + j) + 32); },
                            [&](int j, const Fe &yinv) { pts[base + j] = G1A{fe_mul(fe_load(P + 64 * (base + j)), yinv), yinv}; });
    for (size_t i = 0; i < n; i++) {
        G2A b{f2_load(Q + 128 * i), f2_load(Q + 128 * i + 64)};
        int cnt = 0;
        miller_lines_raw(b, [&](const LineE &l) {
            F2 inv = f2_inv(f2_norm(l.r0));
            tab[i * MILLER_LINES + cnt++] = Line34{f2_mul(f2_norm(l.r1), inv), f2_mul(f2_norm(l.r2), inv)};
        });
    }
    PairRendezvous rv;
    auto lane = [&](bool odd) {
        PairHost x{odd, &rv};
        F6 h = miller_accumulate_multi_34(x, (int)n, [&](int p, int li) -> Line34 {
            const Line34 &r = tab[(size_t)p * MILLER_LINES + li];
            const F2 mine = f2_mul_fe(x.odd ? r.c4 : r.c3, x.odd ? pts[p].y : pts[p].x), other = x.swap(mine);   // one product per lane, swapped
            return Line34{x.odd ? other : mine, x.odd ? mine : other};
        });
        if (do_final_exp) h = final_exp_pair(x, h);
        f6_store(out + (odd ? 192 : 0), h);
        stats_flush();
    };
    std::thread t1(lane, true);
    lane(false);
    t1.join();
}
void hc_gt_pair_ops(const uint8_t *A, const uint8_t *B, size_t n, uint8_t *mul, uint8_t *sqr, uint8_t *csqr, uint8_t *inv, uint8_t *frob1) {
    for (size_t i = 0; i < n; i++) {
        PairRendezvous rv;
        auto lane = [&](bool odd) {
            PairHost x{odd, &rv};
            size_t o = 384 * i + (odd ? 192 : 0);
            F6 a = f6_load(A + o), b = f6_load(B + o);
            f6_store(mul + o, f12p_mul(x, a, b));
            f6_store(sqr + o, f12p_sqr(x, a));
            f6_store(csqr + o, f12p_cyclo_sqr<true>(x, a));
            f6_store(inv + o, f12p_inv(x, a));
            f6_store(frob1 + o, f12p_frob(x, a, 1));
            stats_flush();
        };
        std::thread t1(lane, true);
        lane(false);
        t1.join();
    }
}
// GT.Exp on a lane pair (k_gt_exp): out = a^k for 256-bit little-endian k
void hc_gt_exp_pair(const uint8_t *A, const uint8_t *K, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        PairRendezvous rv;
        uint32_t k[8]; memcpy(k, K + 32 * i, 32);
        auto lane = [&](bool odd) {
            PairHost x{odd, &rv};
            size_t o = 384 * i + (odd ? 192 : 0);
            alignas(16) int32_t tab[GT_EXP_TAB_DWORDS];
            f6_store(out + o, f12p_exp256(x, f6_load(A + o), k, tab));
            stats_flush();
        };
        std::thread t1(lane, true);
        lane(false);
        t1.join();
    }
}
// wire formats (csrc/wire29.hip.hpp): kind 0 G1, 1 G2, 2 GT; the same per-element functions the kernels call
void hc_wire_encode(int kind, const uint8_t *in, size_t n, int compressed, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        if (kind == 0) g1_wire_encode(out + i * (compressed ? 32 : 64), in + 64 * i, compressed != 0);
        else if (kind == 1) g2_wire_encode(out + i * (compressed ? 64 : 128), in + 128 * i, compressed != 0);
        else gt_wire_encode(out + 384 * i, in + 384 * i);
    }
}
void hc_wire_decode(int kind, const uint8_t *in, int elem_bytes, size_t n, uint8_t *out, uint8_t *ok) {
    for (size_t i = 0; i < n; i++) {
        if (kind == 0) ok[i] = g1_wire_decode(out + 64 * i, in + (size_t)elem_bytes * i, elem_bytes);
        else if (kind == 1) ok[i] = g2_wire_decode(out + 128 * i, in + (size_t)elem_bytes * i, elem_bytes);
        else ok[i] = gt_wire_decode(out + 384 * i, in + 384 * i);
    }
}
// hash to curve, group part (csrc/h2c29.hip.hpp): U = n x 2 field elements (gnark fp.Element / E2), out = n affine points
void hc_map_fields(int g2, const uint8_t *U, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        if (!g2) {
            AffP<Fe> r;
            g1_map_fields(r, fe_load(U + 64 * i), fe_load(U + 64 * i + 32));
            fe_store(out + 64 * i, r.x); fe_store(out + 64 * i + 32, r.y);
        } else {
            AffP<F2> r;
            g2_map_fields(r, f2_load(U + 128 * i), f2_load(U + 128 * i + 64));
            f2_store(out + 128 * i, r.x); f2_store(out + 128 * i + 64, r.y);
        }
    }
}
// fixed-base MSM exactly as k_g1_fb_build / k_g1_fb_msm do it (G1): 8-bit window table of every base, then 32 mixed
// additions per term; out[m] = sum_j [K[m][j]] B[j]
void hc_g1_fb_msm(const uint8_t *B, size_t nbase, const uint8_t *K, size_t n_msm, uint8_t *out) {
    const size_t E = 32 * 255;
    std::vector<int32_t> table(nbase * E * 32 + 4);
    int32_t *tb0 = table.data();
    while (((uintptr_t)tb0) & 15) tb0++;
    std::vector<uint8_t> inf(nbase);
    for (size_t b = 0; b < nbase; b++) {
        AffP<Fe> base{fe_load(B + 64 * b), fe_load(B + 64 * b + 32), bytes_all_zero(B + 64 * b, 16)};
        inf[b] = base.inf;
        for (int w = 0; w < 32; w++)
            for (int d = 1; d <= 255; d += (w % 5 == 0 || d < 4 || d > 252) ? 1 : 37) {     // a subset of the rows is enough for the bounds proof
                uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                k[w >> 2] = (uint32_t)d << (8 * (w & 3));
                alignas(16) int32_t tab[glv_table_dwords<Fe>()];
                JacP<Fe> r; scalar_mul29_jac<Fe>(r, base, k, tab);
                AffP<Fe> a; jac_to_affine(a, r);
                tab_store(tb0 + ((b * 32 + w) * 255 + d - 1) * 32, 0, a);
            }
    }
    for (size_t m = 0; m < n_msm; m++) {
        JacP<Fe> acc; jac_set_inf(acc);
        for (size_t j = 0; j < nbase; j++) {
            if (inf[j]) continue;
            uint32_t k[8]; memcpy(k, K + 32 * (m * nbase + j), 32);
            for (int w = 0; w < 32; w++) {
                int d = (k[w >> 2] >> (8 * (w & 3))) & 255;
                if (!d) continue;
                AffP<Fe> e; tab_load(tb0 + ((j * 32 + w) * 255 + d - 1) * 32, 0, e);
                jac_add_mixed(acc, acc, e);
            }
        }
        AffP<Fe> a; jac_to_affine(a, acc);
        fe_store(out + 64 * m, a.x); fe_store(out + 64 * m + 32, a.y);
    }
}
void hc_msm(int g2, const uint8_t *B, const uint8_t *K, size_t n, int c, uint8_t *out) {
    if (!g2) msm_host<Fe>(B, K, n, c, out, 64,
                          [](const uint8_t *p) { return AffP<Fe>{fe_load(p), fe_load(p + 32), bytes_all_zero(p, 16)}; },
                          [](uint8_t *p, const AffP<Fe> &r) { fe_store(p, r.x); fe_store(p + 32, r.y); });
    else msm_host<F2>(B, K, n, c, out, 128,
                      [](const uint8_t *p) { return AffP<F2>{f2_load(p), f2_load(p + 64), bytes_all_zero(p, 32)}; },
                      [](uint8_t *p, const AffP<F2> &r) { f2_store(p, r.x); f2_store(p + 64, r.y); });
}
// worst-case figures since process start: [max |int64 column|, max limb bound, max value bound (units of p),
// #products (fe_mul + fe_mul2), #norms, #fe_mul2, #reduces]  (out must hold 7 doubles)
// v_mad_i64_i32 the device form of everything run since the last call would have executed (summed over the lanes of a pair)
double hc_mads_take(void) {
    stats_flush();
    std::lock_guard<std::mutex> lk(g_stats_mu);
    const double m = (double)g_stats_total.mads;
    g_stats_total.mads = 0;
    return m;
}
void hc_stats(double *out) {
    stats_flush();
    BoundStats &s = g_stats_total;
    out[0] = s.max_col; out[1] = s.max_limb; out[2] = s.max_vb; out[3] = (double)s.muls; out[4] = (double)s.norms; out[5] = (double)s.muls2; out[6] = (double)s.reduces;
}
}
