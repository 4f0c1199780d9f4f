/* CPU restatement of the BN254 hot path — TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference (mmsyan/GoPairingBasedCryptography) delegates all arithmetic to the
 * un-vendored module github.com/consensys/gnark-crypto v0.19.0 (reference go.mod:5) and holds no
 * known-answer vectors.  This library restates that module's published algorithm
 * (4x64 Montgomery Fp, E2/E6/E12 tower, projective-line optimal-ate Miller loop over the NAF of
 * 6u+2, Fuentes-Castaneda final exponentiation with the s-cofactor) and is pinned only against
 * oracle/bn254_py.py (textbook big-int route) through tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * All buffers use gnark-crypto in-memory layouts: fp.Element = 4 LE u64 limbs, Montgomery form;
 * G1Affine 64 B, G2Affine 128 B, GT 384 B, scalars 32 B little-endian plain integers.
 */
#ifndef GPBC_BN254_ORACLE_H
#define GPBC_BN254_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* n independent pairings (bn254.Pair with len==1 each); infinity in either slot -> GT one. */
void gpbc_oracle_pair_batch(const uint8_t *P, const uint8_t *Q, size_t n, uint8_t *gt_out, int threads);
/* k segments; segment j covers pairs [seg_off[j], seg_off[j+1]) — bn254.Pair with len>1. */
void gpbc_oracle_multi_pair(const uint8_t *P, const uint8_t *Q, const uint64_t *seg_off, size_t k,
                            uint8_t *gt_out, int threads);
/* stage-level entry points (Miller function before final exponentiation; FE alone) */
void gpbc_oracle_miller_loop(const uint8_t *P, const uint8_t *Q, size_t n, uint8_t *f_out, int threads);
void gpbc_oracle_final_exp(const uint8_t *f, size_t n, uint8_t *gt_out, int threads);
/* [s]A, affine output; nbase is 1 (shared base) or n */
void gpbc_oracle_g1_scalar_mul(const uint8_t *base, size_t nbase, const uint8_t *scalars, size_t n,
                               uint8_t *out, int threads);
void gpbc_oracle_g2_scalar_mul(const uint8_t *base, size_t nbase, const uint8_t *scalars, size_t n,
                               uint8_t *out, int threads);
/* affine point sums (used by the aggregate-verify tests) */
void gpbc_oracle_g1_sum(const uint8_t *pts, size_t n, uint8_t *out);
void gpbc_oracle_g2_sum(const uint8_t *pts, size_t n, uint8_t *out);
/* GT arithmetic */
void gpbc_oracle_gt_exp(const uint8_t *x, const uint8_t *k, size_t n, uint8_t *out, int threads);
void gpbc_oracle_gt_mul(const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out);
void gpbc_oracle_gt_div(const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out);
void gpbc_oracle_gt_inverse(const uint8_t *a, size_t n, uint8_t *out);
/* Fp-level entry points for kernel unit tests */
void gpbc_oracle_fp_mul(const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out);
void gpbc_oracle_fp_inv(const uint8_t *a, size_t n, uint8_t *out);
void gpbc_oracle_fp12_mul(const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out);
void gpbc_oracle_fp12_cyclotomic_square(const uint8_t *a, size_t n, uint8_t *out);
/* number of Fp Montgomery multiplications executed by the calling thread since the last reset
 * (all threads when built with -DGPBC_COUNT_MULS; 0 otherwise) */
uint64_t gpbc_oracle_fp_mul_count(int reset);

#ifdef __cplusplus
}
#endif
#endif
