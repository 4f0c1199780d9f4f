/* gpbc_bn254.h — C ABI of the MI355X batched BN254 pairing engine (libgpbc_bn254.so).
 *
 * Drop-in boundary for the gnark-crypto calls the reference (mmsyan/GoPairingBasedCryptography) makes on
 * its hot path.  The reference has no FFI of its own: it calls github.com/consensys/gnark-crypto v0.19.0
 * (go.mod:5) directly.  A cgo shim binds the functions below in place of those calls;
 * each entry names the reference call sites it replaces.
 *
 * Data layouts = gnark-crypto in-memory structs, so Go slices are passed without conversion:
 *   fp.Element   4 x uint64 little-endian limbs, Montgomery form (x * 2^256 mod p)          32 B
 *   G1Affine     {X, Y fp.Element}; point at infinity = all zero                             64 B
 *   G2Affine     {X, Y E2{A0, A1}}                                                          128 B
 *   GT (E12)     {C0, C1 E6{B0, B1, B2 E2{A0, A1}}}                                         384 B
 *   scalar       32-byte little-endian plain (non-Montgomery) integer, any value < 2^256
 *                (the shim fills it from big.Int; values >= r act as their residue mod r)
 *
 * Conventions: every function returns 0 on success and a negative gpbc_status on failure and never
 * aborts; gpbc_last_error() gives a thread-local message.  Buffers are caller-owned.  Functions are
 * thread-safe.  There is NO CPU fallback: without a usable gfx950 device every compute entry fails
 * with GPBC_ERR_NO_DEVICE.
 *
 * *_dev variants take device pointers (buffers already resident in HBM), enqueue on `stream`
 * (a hipStream_t passed as void*, NULL = default stream) and return without synchronising.
 */
#ifndef GPBC_BN254_H
#define GPBC_BN254_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GPBC_G1_BYTES 64
#define GPBC_G2_BYTES 128
#define GPBC_GT_BYTES 384
#define GPBC_SCALAR_BYTES 32
/* wire formats (big-endian canonical, gnark Marshal / Bytes) */
#define GPBC_G1_RAW_BYTES 64
#define GPBC_G1_COMPRESSED_BYTES 32
#define GPBC_G2_RAW_BYTES 128
#define GPBC_G2_COMPRESSED_BYTES 64

typedef enum {
    GPBC_OK = 0,
    GPBC_ERR_INVALID_ARG = -1,   /* null pointer, n == 0 where gnark errors ("invalid inputs sizes"), bad segment table */
    GPBC_ERR_NO_DEVICE = -2,     /* no HIP device / gpbc_init not successful */
    GPBC_ERR_HIP = -3,           /* a HIP runtime call failed; see gpbc_last_error() */
    GPBC_ERR_WORKSPACE = -4,     /* *_dev call given a workspace smaller than gpbc_*_workspace_bytes() */
    GPBC_ERR_COMM = -5,          /* RCCL missing or a collective failed; see gpbc_last_error() */
    GPBC_ERR_INTERNAL = -6       /* a self-check of the library failed (e.g. the multi-pairing kernels did not consume exactly the pairs of
                                    the caller's segment table); outputs are zeroed, nothing computed by the call may be used */
} gpbc_status;

/* ---- lifetime and devices ----------------------------------------------------------------------
 * The library is bound to a LIST of HIP devices (SURVEY.md §8b "init(devices)", §8e "one host thread + one stream per
 * device").  Every host thread has a current device — gpbc_set_device(index into the list), thread-local, default index
 * 0 — on which its *_dev calls and its un-sharded host calls run, so one Go / C++ process drives all the MI355X of a node:
 * one thread per device with device-resident buffers, or simply the host-pointer batch entries, which split their index
 * range [0, n) over all bound devices themselves (contiguous shards, sizes differing by at most one, one internal host
 * thread per device, no data-path collective) and return when every shard is back in the caller's buffer.  Results do not
 * depend on the number of devices. */
int gpbc_init(int device);                 /* = gpbc_init_devices(&device, 1); idempotent */
int gpbc_init_devices(const int *devices, int n_devices);   /* HIP ordinals, each gfx950; idempotent for an equal list.  An ordinal may
                                                             * appear twice (a one-GPU rig rehearsing the sharded paths: the slots
                                                             * then share that GPU); RCCL refuses such a list. */
int gpbc_num_devices(void);                /* devices bound by init (0 before) */
int gpbc_device_at(int index);             /* HIP ordinal of bound device `index`, <0 on error */
int gpbc_set_device(int index);            /* current device of the calling thread */
int gpbc_get_device(void);                 /* its index, <0 before init */
int gpbc_set_host_sharding(int on);        /* default 1; 0: host-pointer entries stay on the caller's current device */
int gpbc_shutdown(void);
/* Returns the library's grow-only device buffers (line / table workspaces and per-call scratch of every (device, stream): up to
 * several GB after BASELINE-size calls) to the driver; they come back on demand.  Synchronises the bound devices first.  For
 * long-lived processes that share HBM with other users; calls in flight on other threads must have returned. */
int gpbc_release_workspaces(void);
const char *gpbc_last_error(void);         /* thread-local, never NULL */
int gpbc_device_count(void);               /* number of visible HIP devices, <0 on error */
int gpbc_abi_version(void);

/* ---- collectives (RCCL over xGMI) --------------------------------------------------------------
 * The path has exactly one exchange step: an all-gather of byte rows — 192 B of partial sums per rank in
 * the aggregate-verify path (BASELINE config 3), n/G x 384 B of GT values per rank in batched decryption (config 5).
 * RCCL is opened with dlopen when a communicator is first requested (an RCCL already loaded in the process, e.g. PyTorch's,
 * is reused; the environment variable GPBC_RCCL_LIBRARY names one library file to open instead); without it these entries fail
 * with GPBC_ERR_COMM and everything else works.
 *   one process, all bound devices:   gpbc_comm_init_all()  -> rank = device index, gpbc_allgather_all_dev()
 *   one process per GPU (torchrun):   rank 0 calls gpbc_comm_get_unique_id(), the launcher broadcasts the 128 bytes, every
 *                                     process calls gpbc_comm_init_rank(id, n_ranks, rank) -> gpbc_allgather_dev()
 * recv holds n_ranks blocks of bytes_per_rank in rank order; enqueued on the stream(s), not synchronised. */
#define GPBC_COMM_ID_BYTES 128
int gpbc_comm_init_all(void);
int gpbc_comm_get_unique_id(void *id_out);
int gpbc_comm_init_rank(const void *id, int n_ranks, int rank);
int gpbc_comm_ranks(void);                 /* 0 when no communicator exists */
int gpbc_comm_rank(void);                  /* rank of the calling thread's current device */
int gpbc_comm_destroy(void);
int gpbc_allgather_dev(const void *d_send, size_t bytes_per_rank, void *d_recv, void *stream);
int gpbc_allgather_all_dev(const void *const *d_send, size_t bytes_per_rank, void *const *d_recv, void *const *streams);

/* ---- pairings ----------------------------------------------------------------------------------
 * bn254.Pair(P []G1Affine, Q []G2Affine) (GT, error) with len==1, n times
 *   (cpabe/bsw07/bsw07_cpabe.go:75,184; access/tree/access_tree_node.go:106,110,119;
 *    bibe/afp25_bibe/afp25_bibe.go:227,395,399,403; ibe/bb04_ibe/bb04_ibe.go:215,225; ...).
 * A pair with the point at infinity in either slot yields GT one. */
int gpbc_pair_batch(const void *P, const void *Q, size_t n, void *gt_out);
int gpbc_pair_batch_dev(const void *dP, const void *dQ, size_t n, void *d_gt_out, void *stream);

/* bn254.Pair with len>1, k times: segment j is pairs [seg_off[j], seg_off[j+1]); one Miller loop per
 * pair, one final exponentiation per segment (products assembled by callers: ibe/bb04_ibe/bb04_ibe.go:213-225,
 * access/tree/access_tree_node.go:106-157, bibe/afp25_bibe/afp25_bibe.go:395-413).
 * seg_off has k+1 non-decreasing entries, seg_off[0] == 0, seg_off[k] == number of pairs; an empty segment yields GT
 * one.  The host entry validates the table; the _dev entry cannot (the table is in device memory) and clamps every
 * offset to n_pairs instead, so a malformed table yields wrong products but never an out-of-bounds access. */
int gpbc_multi_pair(const void *P, const void *Q, const uint64_t *seg_off, size_t k, void *gt_out);
size_t gpbc_multi_pair_workspace_bytes(size_t n_pairs, size_t k);
int gpbc_multi_pair_dev(const void *dP, const void *dQ, const uint64_t *d_seg_off, size_t n_pairs, size_t k,
                        void *d_gt_out, void *d_workspace, size_t workspace_bytes, void *stream);
/* Validation of a device-resident table for gpbc_multi_pair_dev (k+1 entries read): one small kernel, synchronises `stream`,
 * returns GPBC_ERR_INVALID_ARG (first entry not 0, not monotone, last entry != n_pairs) or GPBC_OK.  Optional: the
 * multi-pairing itself stays asynchronous and safe (clamped) without it. */
int gpbc_check_segments_dev(const uint64_t *d_seg_off, size_t n_pairs, size_t k, void *stream);
/* Points and results in device memory, segment table on the HOST (validated like gpbc_multi_pair).  With the table at
 * hand the engine cuts segments (of four or more pairs on average) into chunks of up to 8 pairs whose Miller accumulators
 * SHARE their squarings (F <- F^2 * prod l_p, as gnark's own multi-pairing does): up to 35 % less accumulator work per
 * pair, bit-identical results.  gpbc_multi_pair and gpbc_pairing_check use the same path.  Synchronises
 * `stream` before returning. */
int gpbc_multi_pair_hostseg_dev(const void *dP, const void *dQ, const uint64_t *seg_off, size_t k, void *d_gt_out, void *stream);
/* k products over ONE shared list of G2 points: out[j] = Pair(P[j*m .. (j+1)*m), Q[0 .. m)) — a decryption key against k
 * ciphertexts (access/tree/access_tree_node.go:106-119 under cpabe/bsw07/bsw07_cpabe.go:172-195: the D_j of a key are the
 * same for every ciphertext), one public key against k signatures.  The line coefficients of each Q_i are computed once
 * and reused by all k segments (gnark: PrecomputeLines / MillerLoopFixedQ), on top of the shared squarings; results are
 * bit-identical to gpbc_multi_pair on the replicated list.  P: k*m points, segment-major.  The _dev form is asynchronous on
 * `stream`; its temporaries (line table, converted points, chunk values) stay in a per-stream scratch buffer between calls.
 * Calls of at most gpbc_set_latency_path's limit in total (k*m pairs; one ciphertext is 513) skip the line table — m lone lanes
 * would spend longer building it than it saves — and run one pairing per wavefront with the list indexed modulo m: 1.5 ms. */
int gpbc_multi_pair_fixed_q(const void *P, const void *Q, size_t m, size_t k, void *gt_out);
int gpbc_multi_pair_fixed_q_dev(const void *dP, const void *dQ, size_t m, size_t k, void *d_gt_out, void *stream);
/* Tuning / test knob of the multi-pairing paths: pairs per shared-squaring chunk, 1..64 (the general path, whose lines live in a
 * per-slot workspace, caps it at 8; the fixed-Q path takes all of it); 0 (default) = chosen from the batch size. */
int gpbc_set_multi_pair_chunk(int pairs_per_chunk);
/* Batches of up to 16 384 pairings run the two phases of the Miller loop concurrently in one launch (the accumulator consumes
 * the lines as they are produced: about a third off the latency of a small call).  0 switches to the two-kernel form used for
 * large batches (tests compare the two); default 1.  2 = pipelined with no waiting at all (a consumer that finds its line
 * missing computes its own lines: the bounded-wait fallback, for tests). */
int gpbc_set_pipelined_miller(int on);
/* The LATENCY path.  Every reference call site is ONE bn254.Pair / PairingCheck (cpabe/bsw07/bsw07_cpabe.go:184,
 * signature/bls01_signature/bls_signature.go:81, ...); in the throughput kernels one pairing is a chain of ~2 M dependent
 * instructions on one lane pair — ~6 ms per call whatever the batch size.  Calls of at most `max_pairs` Miller loops / final
 * exponentiations (default 2048: one wave per SIMD pair of the chip; final exponentiations alone up to twice that) run ONE PAIRING PER WAVEFRONT instead, the 64 lanes working
 * on the F2 products inside it (csrc/wide29.hip.hpp): same bits, about a fifth of the latency (1.2 ms per call).  The same limit
 * sends GT.Exp calls of at most 2 x `max_pairs` elements down a one-element-per-wavefront kernel (0.8 ms instead of 5 ms for one
 * new(GT).Exp, access/tree/access_tree_node.go:114), and multi-pairing calls of at most `max_pairs` pairs multiply the Miller
 * values of long segments on 8 or 16 wavefronts each (one 513-pair Pair: 1.5 ms).  0 switches the path off (tests compare the
 * two forms). */
int gpbc_set_latency_path(long max_pairs);      /* 0 .. 65 536 */
/* Fail-closed self-check of the host-table multi-pairings (gpbc_multi_pair, gpbc_pairing_check, gpbc_multi_pair_hostseg_dev): the
 * segment / chunk tables travel through library-owned pinned memory, the kernels echo the pairs they consumed per segment, and
 * the call returns GPBC_ERR_INTERNAL with zeroed outputs unless the echo equals the caller's table — a product over fewer pairs
 * than were passed (an empty product is GT one, i.e. "PairingCheck = true") can never come back as a result.  Test knob: the next
 * such call sends the device a table whose last segment is empty (what a stale table looks like) and must fail.  The knob answers
 * only in a process whose environment has GPBC_TEST_KNOBS=1 (GPBC_ERR_INVALID_ARG otherwise): no thread of a production process
 * can make another thread's verification fail. */
int gpbc_debug_stale_table_once(void);

/* bn254.PairingCheck(P, Q) (bool, error), k times (signature/bls01_signature/bls_signature.go:81):
 * ok_out[j] = 1 iff the product over segment j is GT one. */
int gpbc_pairing_check(const void *P, const void *Q, const uint64_t *seg_off, size_t k, uint8_t *ok_out);

/* stages of Pair, exposed for parity tests and profiling: the Miller function (before the final
 * exponentiation; defined up to factors the final exponentiation removes) and the exponentiation alone */
int gpbc_miller_loop_dev(const void *dP, const void *dQ, size_t n, void *d_f_out, void *stream);
int gpbc_final_exp_dev(const void *d_f, size_t n, void *d_gt_out, void *stream);
int gpbc_miller_loop(const void *P, const void *Q, size_t n, void *f_out);
int gpbc_final_exp(const void *f, size_t n, void *gt_out);

/* ---- scalar multiplication ---------------------------------------------------------------------
 * (*G1Affine).ScalarMultiplication(a, s) / ScalarMultiplicationBase(s), n times
 *   (signature/bls01_signature/bls_signature.go:45; cpabe/bsw07/bsw07_cpabe.go:69,149,157,160;
 *    bibe/afp25_bibe/afp25_bibe_utils.go:48,51).  nbase == n: one base per scalar; nbase == 1: shared base (from 16 384
 *    scalars on, served by a transient fixed-base window table: 5x the variable-base rate, same results). */
int gpbc_g1_scalar_mul_batch(const void *bases, size_t nbase, const void *scalars, size_t n, void *out);
int gpbc_g1_scalar_mul_batch_dev(const void *d_bases, size_t nbase, const void *d_scalars, size_t n, void *d_out, void *stream);
/* (*G2Affine).ScalarMultiplication(a, s) (signature/bls01_signature/bls_signature.go:63;
 *  cpabe/bsw07/bsw07_cpabe.go:73,83,103-121; bibe/afp25_bibe/afp25_bibe.go:215,250,251) */
int gpbc_g2_scalar_mul_batch(const void *bases, size_t nbase, const void *scalars, size_t n, void *out);
int gpbc_g2_scalar_mul_batch_dev(const void *d_bases, size_t nbase, const void *d_scalars, size_t n, void *d_out, void *stream);

/* sum of n affine points -> one affine point (G1Affine.Add chains: bibe/afp25_bibe/afp25_bibe_utils.go:52,
 * gka/agka09/asbb.go:193-220; the aggregate-verify path of BASELINE config 3) */
int gpbc_g1_sum(const void *pts, size_t n, void *out);
int gpbc_g2_sum(const void *pts, size_t n, void *out);
size_t gpbc_sum_workspace_bytes(size_t n, int is_g2);
int gpbc_g1_sum_dev(const void *d_pts, size_t n, void *d_out, void *d_workspace, size_t workspace_bytes, void *stream);
int gpbc_g2_sum_dev(const void *d_pts, size_t n, void *d_out, void *d_workspace, size_t workspace_bytes, void *stream);

/* sum_i [s_i] P_i — the verifier's side of BLS aggregate verification with random linear combination (BASELINE config 3:
 * A = sum rho_i pk_i in G1, B = sum rho_i sigma_i in G2; signature/bls01_signature/bls_signature.go:71-89 verifies one,
 * gka/agka09/asbb_test.go:203-238 has the aggregate shape).  nbase == n.
 * Host form: sharded over the bound devices like every batch entry; each device reduces its shard to one partial sum, the
 * partial sums come back to the host (one point per device) and the calling thread's device adds them — no collective: the
 * data of a host-pointer call is on the host anyway.
 * _dev form (one process per GPU): the calling rank's shard in device memory; with a communicator (gpbc_comm_init_rank)
 * the partial sums of all ranks are all-gathered and added, so every rank ends with the global sum; without one the
 * result is the local sum.  Stream-ordered, not synchronised. */
int gpbc_g1_scalar_mul_sum(const void *bases, const void *scalars, size_t n, void *out);
int gpbc_g2_scalar_mul_sum(const void *bases, const void *scalars, size_t n, void *out);
int gpbc_g1_scalar_mul_sum_dev(const void *d_bases, const void *d_scalars, size_t n, void *d_out, void *stream);
int gpbc_g2_scalar_mul_sum_dev(const void *d_bases, const void *d_scalars, size_t n, void *d_out, void *stream);
/* Diagnostics: from 16 384 terms on the sums above take the bucket (Pippenger) method unless the digit histogram shows a bucket far
 * longer than the mean (all scalars equal, small integers ...), in which case the terms are multiplied one by one.  Process-wide
 * counts of the two outcomes since the library was loaded (tests assert which path a workload took). */
int gpbc_msm_stats(uint64_t *bucket_runs_out, uint64_t *skewed_fallbacks_out);

/* ---- GT arithmetic -----------------------------------------------------------------------------
 * (*GT).Exp(x, k) for k >= 0 (access/tree/access_tree_node.go:123,156; bibe/afp25_bibe/afp25_bibe.go:258-259);
 * the shim maps a negative big.Int to Inverse followed by Exp(|k|), as gnark does. */
int gpbc_gt_exp_batch(const void *x, const void *k, size_t n, void *out);
int gpbc_gt_exp_batch_dev(const void *d_x, const void *d_k, size_t n, void *d_out, void *stream);
/* (*GT).Mul / Div / Inverse (access/tree/access_tree_node.go:114,157; cpabe/bsw07/bsw07_cpabe.go:189-190) */
int gpbc_gt_mul_batch(const void *a, const void *b, size_t n, void *out);
int gpbc_gt_div_batch(const void *a, const void *b, size_t n, void *out);
int gpbc_gt_inverse_batch(const void *a, size_t n, void *out);
int gpbc_gt_mul_batch_dev(const void *d_a, const void *d_b, size_t n, void *d_out, void *stream);
int gpbc_gt_div_batch_dev(const void *d_a, const void *d_b, size_t n, void *d_out, void *stream);
int gpbc_gt_inverse_batch_dev(const void *d_a, size_t n, void *d_out, void *stream);

/* ---- fixed-base tables and multi-scalar multiplication -----------------------------------------
 * Sums  out[m] = sum_j [s[m][j]] base_j  over a FIXED set of bases: (*G1Affine).ScalarMultiplicationBase (nbase == 1, the
 * generator: signature/bls01_signature/bls_signature.go:45, cpabe/bsw07/bsw07_cpabe.go:69-160) and the commitment loops
 * `for j { t.ScalarMultiplication(&srs[j], c_j); acc.Add(&acc, &t) }` (bibe/afp25_bibe/afp25_bibe_utils.go:44-55).
 * create builds 8-bit window tables [d * 2^(8w)] base_j (d = 1..255, w = 0..31) in HBM — gpbc_fixed_base_table_bytes():
 * 1 MB per G1 base, 2 MB per G2 base — after which a term costs 32 mixed additions and no doublings.  Scalars: n_msm rows
 * of nbase 32-byte little-endian integers (any value < 2^256; no reduction is needed).  The handle is bound to the device
 * it was built on; msm calls may run concurrently, destroy must not race with them. */
typedef struct gpbc_fixed_base gpbc_fixed_base;
size_t gpbc_fixed_base_table_bytes(size_t nbase, int is_g2);
int gpbc_g1_fixed_base_create(const void *bases, size_t nbase, gpbc_fixed_base **out);
int gpbc_g2_fixed_base_create(const void *bases, size_t nbase, gpbc_fixed_base **out);
int gpbc_fixed_base_create_dev(int is_g2, const void *d_bases, size_t nbase, void *stream, gpbc_fixed_base **out);
int gpbc_fixed_base_msm(const gpbc_fixed_base *table, const void *scalars, size_t n_msm, void *out);
size_t gpbc_fixed_base_msm_workspace_bytes(const gpbc_fixed_base *table, size_t n_msm);
int gpbc_fixed_base_msm_dev(const gpbc_fixed_base *table, const void *d_scalars, size_t n_msm, void *d_out,
                            void *d_workspace, size_t workspace_bytes, void *stream);
int gpbc_fixed_base_destroy(gpbc_fixed_base *table);

/* ---- wire formats ------------------------------------------------------------------------------
 * Big-endian canonical (non-Montgomery) encodings of gnark-crypto ecc/bn254 marshal.go; the two top bits of the first byte
 * select the form: 00 uncompressed (infinity = all zero), 01 compressed infinity, 10 / 11 compressed with the smaller /
 * larger Y.  G2 order: X.A1, X.A0[, Y.A1, Y.A0]; GT order: C1.B2.A1 ... C0.B0.A0.
 *
 * (G1Affine|G2Affine).Marshal() = RawBytes() (compressed == 0) and .Bytes() (compressed != 0), GT.Marshal() = Bytes()
 *   (serialization/serialization_curve.go:5-15; ibe/gentry06_ibe/gentry06_ibe.go:322-324; hash/hash_from_gt.go:5-8).
 * in: n gnark structs; out: n encodings of 64/32 (G1), 128/64 (G2), 384 (GT) bytes.  Not in place. */
int gpbc_g1_marshal_batch(const void *pts, size_t n, int compressed, void *out);
int gpbc_g2_marshal_batch(const void *pts, size_t n, int compressed, void *out);
int gpbc_gt_marshal_batch(const void *gt, size_t n, void *out);
int gpbc_g1_marshal_batch_dev(const void *d_pts, size_t n, int compressed, void *d_out, void *stream);
int gpbc_g2_marshal_batch_dev(const void *d_pts, size_t n, int compressed, void *d_out, void *stream);
int gpbc_gt_marshal_batch_dev(const void *d_gt, size_t n, void *d_out, void *stream);
/* (G1Affine|G2Affine|GT).Unmarshal() = SetBytes() (serialization/serialization_curve.go:17-33), n times: element i
 * occupies in[i*elem_bytes, (i+1)*elem_bytes) with elem_bytes = 32|64 (G1), 64|128 (G2), 384 (GT).  As in gnark the
 * form is read from the flag bits (a compressed encoding in the first half of a 64/128-byte slot is accepted, an
 * uncompressed flag in a 32/64-byte slot is a short buffer).  ok_out[i] = 1 where gnark returns no error and 0 where it
 * returns one — coordinate >= p, malformed infinity, no square root, point not on the curve / not in the order-r
 * subgroup (G2: decided, as gnark does, by the endomorphism identity [x+1]Q + psi([x]Q) + psi^2([x]Q) = psi^3([2x]Q), which holds
 * exactly on the points with [r]Q = infinity; tests/test_wire.py compares the two) — and then the output element is all zero.  The reference ignores that error
 * (serialization_curve.go:19,25,31); callers of this entry should not.  Not in place. */
int gpbc_g1_unmarshal_batch(const void *in, size_t elem_bytes, size_t n, void *pts_out, uint8_t *ok_out);
int gpbc_g2_unmarshal_batch(const void *in, size_t elem_bytes, size_t n, void *pts_out, uint8_t *ok_out);
int gpbc_gt_unmarshal_batch(const void *in, size_t n, void *gt_out, uint8_t *ok_out);
int gpbc_g1_unmarshal_batch_dev(const void *d_in, size_t elem_bytes, size_t n, void *d_pts_out, uint8_t *d_ok_out, void *stream);
int gpbc_g2_unmarshal_batch_dev(const void *d_in, size_t elem_bytes, size_t n, void *d_pts_out, uint8_t *d_ok_out, void *stream);
int gpbc_gt_unmarshal_batch_dev(const void *d_in, size_t n, void *d_gt_out, uint8_t *d_ok_out, void *stream);

/* ---- hash to curve, group part -----------------------------------------------------------------
 * bn254.HashToG1(msg, dst) / HashToG2(msg, dst) (hash/hash_to.go:113-119,169-175,204-210,271-277) after hash_to_field:
 *   u = fp.Hash(msg, dst, 2)  (G2: 2 E2 elements from 4 base-field elements; expand_message_xmd(SHA-256), L = 48 — computed by
 *   the caller for these entries; gpbc_hash_to_g1 / _g2 below do it on the device as well)
 *   out = MapToCurve(u[0]) + MapToCurve(u[1])            Shallue-van de Woestijne map, RFC 9380 Appendix F.1, Z = 1
 *   G2: out = ClearCofactor(out)                         [x]P + psi([3x]P) + psi^2([x]P) + psi^3(P)
 * u: n x 2 field elements in gnark's in-memory layout (G1: 2 x 32 B, G2: 2 x 64 B per output point); out: n points. */
int gpbc_g1_map_to_curve_batch(const void *u, size_t n, void *out);
int gpbc_g2_map_to_curve_batch(const void *u, size_t n, void *out);
int gpbc_g1_map_to_curve_batch_dev(const void *d_u, size_t n, void *d_out, void *stream);
int gpbc_g2_map_to_curve_batch_dev(const void *d_u, size_t n, void *d_out, void *stream);

/* ---- hash to curve, whole ------------------------------------------------------------------------
 * bn254.HashToG1(msg, dst) / HashToG2(msg, dst) and fp.Hash(msg, dst, count) for n messages in one call
 * (hash/hash_to.go:113-119 ToG1, :169-175 BytesToG1, :204-210 ToG2, :271-277 BytesToG2; bls01 signing hashes every message,
 * signature/bls01_signature/bls_signature.go:56-63): expand_message_xmd(SHA-256) and the reduction to field elements run on
 * the device too (one message per lane, csrc/xmd29.hip.hpp), followed by the map above.
 * msgs: the messages back to back; msg_off: n + 1 byte offsets (message i = msgs[msg_off[i], msg_off[i+1])); empty messages
 * are fine.  dst: HOST pointer in every form (it travels in the kernel arguments), at most 255 bytes — hash a longer one down
 * first as RFC 9380 section 5.3.3 says.  out: n G1 / G2 points, or n x count fp.Elements (count = 2 or 4; gnark's layout).
 * The _dev forms take device pointers for msgs / msg_off / out plus the size of the msgs buffer (offsets are clamped to it). */
int gpbc_hash_to_g1(const void *msgs, const uint64_t *msg_off, size_t n, const void *dst, size_t dst_len, void *out);
int gpbc_hash_to_g2(const void *msgs, const uint64_t *msg_off, size_t n, const void *dst, size_t dst_len, void *out);
int gpbc_hash_to_field(const void *msgs, const uint64_t *msg_off, size_t n, const void *dst, size_t dst_len, int count, void *out);
int gpbc_hash_to_g1_dev(const void *d_msgs, const uint64_t *d_msg_off, size_t msgs_bytes, size_t n, const void *dst, size_t dst_len, void *d_out, void *stream);
int gpbc_hash_to_g2_dev(const void *d_msgs, const uint64_t *d_msg_off, size_t msgs_bytes, size_t n, const void *dst, size_t dst_len, void *d_out, void *stream);
int gpbc_hash_to_field_dev(const void *d_msgs, const uint64_t *d_msg_off, size_t msgs_bytes, size_t n, const void *dst, size_t dst_len, int count, void *d_out, void *stream);

/* ---- per-kernel timing (measurement, bench.py) ---------------------------------------------------
 * Between begin and end every kernel launch of the pairing / scalar-multiplication entries made on `stream` is bracketed by
 * HIP events on that stream; end synchronises and returns, per kernel name (32-byte zero-padded rows), the summed duration
 * in ms and the number of launches.  One measurement at a time per process. */
int gpbc_profile_begin(void *stream);
int gpbc_profile_end(char *names_out, double *total_ms_out, int *launches_out, int max_kernels, int *n_kernels_out);
/* The roofline's denominator measured NOW on the current device ("measured peak v_mad_u64_u32 rate ... from a
 * dependency-free micro-benchmark kernel run on the box"): a ~2 ms kernel of independent v_mad_u64_u32 chains at 8 waves per
 * SIMD.  out[0] = MACs per second (lane operations), out[1] = shader clock in Hz while it ran (s_memtime against the 100 MHz
 * s_memrealtime), out[2] = SIMD cycles per wave-instruction at that clock, out[3] = kernel duration in ms.  Synchronises the
 * device's null stream.  bench.py reports roofline.frac against this same-run figure beside the fixed 34.9 T/s of round 1. */
int gpbc_valu_probe(double *out4);

/* ---- field-level entry (kernel unit tests) ------------------------------------------------------ */
int gpbc_fp_mul_batch(const void *a, const void *b, size_t n, void *out);

#ifdef __cplusplus
}
#endif
#endif
