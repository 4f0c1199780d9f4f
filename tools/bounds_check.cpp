// Host build of the DEVICE arithmetic (csrc/fe29.cuh, tower29.cuh, curve29.cuh, pairing29.cuh) with -DGPBC_BOUNDS:
// every field element carries data-independent magnitude bounds and every product asserts that its int64 column
// accumulators cannot overflow (abort() on violation).  This is a verification harness for tests/ only — it is
// never loaded by the product path (which has no CPU fallback).
//
// build: g++ -O2 -std=c++17 -DGPBC_BOUNDS -shared -fPIC -o tools/libgpbc_bounds.so tools/bounds_check.cpp
#ifndef GPBC_BOUNDS
#define GPBC_BOUNDS
#endif
#include <cstring>
#include "../gopairingbasedcryptography_amd/csrc/curve29.cuh"
#include "../gopairingbasedcryptography_amd/csrc/pairing29.cuh"

using namespace gpbc;

extern "C" {

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
    for (size_t i = 0; i < n; i++) {
        AffP<Fe> b{fe_load(B + 64 * i), fe_load(B + 64 * i + 32), bytes_all_zero(B + 64 * i, 16)}, r;
        uint32_t k[8]; memcpy(k, K + 32 * i, 32);
        scalar_mul29<Fe>(r, b, k);
        fe_store(out + 64 * i, r.x); fe_store(out + 64 * i + 32, r.y);
    }
}
void hc_g2_mul(const uint8_t *B, const uint8_t *K, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        AffP<F2> b{f2_load(B + 128 * i), f2_load(B + 128 * i + 64), bytes_all_zero(B + 128 * i, 32)}, r;
        uint32_t k[8]; memcpy(k, K + 32 * i, 32);
        scalar_mul29<F2>(r, b, k);
        f2_store(out + 128 * i, r.x); f2_store(out + 128 * i + 64, r.y);
    }
}
void hc_fp_mul(const uint8_t *A, const uint8_t *B, size_t n, uint8_t *out) {
    // gnark-form a (= x R) and b (= y R), R = 2^256: internal product of the converted operands is x y R' -> stored as x y R
    for (size_t i = 0; i < n; i++) fe_store(out + 32 * i, fe_mul(fe_load(A + 32 * i), fe_load(B + 32 * i)));
}
void hc_gt_mul(const uint8_t *A, const uint8_t *B, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) { F12 a, b; f12_load(a, A + 384 * i); f12_load(b, B + 384 * i); f12_store(out + 384 * i, f12_mul(a, b)); }
}
void hc_gt_inv(const uint8_t *A, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) { F12 a; f12_load(a, A + 384 * i); f12_store(out + 384 * i, f12_inv(a)); }
}
void hc_gt_sqr(const uint8_t *A, size_t n, uint8_t *out, int cyclo) {
    for (size_t i = 0; i < n; i++) { F12 a; f12_load(a, A + 384 * i); f12_store(out + 384 * i, cyclo ? f12_cyclo_sqr(a) : f12_sqr(a)); }
}
// worst-case figures since process start: [max |int64 column|, max limb bound, max value bound (units of p),
// #products (fe_mul + fe_mul2), #norms, #fe_mul2, #reduces]  (out must hold 7 doubles)
void hc_stats(double *out) {
    BoundStats &s = bound_stats();
    out[0] = s.max_col; out[1] = s.max_limb; out[2] = s.max_vb; out[3] = (double)s.muls; out[4] = (double)s.norms; out[5] = (double)s.muls2; out[6] = (double)s.reduces;
}
}
