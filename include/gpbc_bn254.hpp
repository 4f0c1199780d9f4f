// gpbc_bn254.hpp — header-only C++ mirror of the gnark-crypto bn254 surface the reference calls, over the C ABI
// (gpbc_bn254.h).  Same names, argument meaning and error behaviour as the Go API:
//
//   bn254.Pair(P []G1Affine, Q []G2Affine) (GT, error)          -> bn254::Pair(P, Q)            throws on size mismatch
//   bn254.PairingCheck(P, Q) (bool, error)                      -> bn254::PairingCheck(P, Q)
//   new(G1Affine).ScalarMultiplication(&a, s)                   -> G1Affine().ScalarMultiplication(a, s)
//   new(G1Affine).ScalarMultiplicationBase(s)                   -> G1Affine().ScalarMultiplicationBase(s)
//   new(GT).Exp(x, k) / Mul / Div / Inverse                     -> GT().Exp(x, k) ...
//   bn254.Generators()                                          -> bn254::Generators()
//   p.Marshal() / p.Bytes() / p.Unmarshal(buf)  (G1, G2, GT)    -> same names; Unmarshal throws where gnark errors
//
// Structs have gnark's in-memory layout (fp.Element = 4 LE u64 Montgomery limbs), so arrays of them are the ABI
// buffers.  Scalars are `Scalar` = 32-byte little-endian plain integers (what the cgo shim makes of *big.Int).
// The Go toolchain is absent from the build image; this header is the compiled-language host side used by
// tests/cpp/ (reference is Go, see INTEGRATION.md for the cgo shim).
#ifndef GPBC_BN254_HPP
#define GPBC_BN254_HPP
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "gpbc_bn254.h"

namespace bn254 {

struct Scalar {
    std::array<uint8_t, 32> le{};
    Scalar() = default;
    explicit Scalar(uint64_t v) { for (int i = 0; i < 8; i++) le[i] = (uint8_t)(v >> (8 * i)); }
};

inline void check(int rc) { if (rc < 0) throw std::runtime_error(std::string("gpbc: ") + gpbc_last_error()); }
inline void Init(int device = 0) { check(gpbc_init(device)); }
// Bind the process to several MI355X (HIP ordinals): host-pointer batch calls then shard over all of them, and a thread
// picks the device of its *_dev calls with SetDevice(index).  InitAllDevices() takes every visible device.
inline void Init(const std::vector<int> &devices) { check(gpbc_init_devices(devices.data(), (int)devices.size())); }
inline int InitAllDevices() {
    int n = gpbc_device_count();
    check(n);
    std::vector<int> d(n);
    for (int i = 0; i < n; i++) d[i] = i;
    Init(d);
    return n;
}
inline int NumDevices() { return gpbc_num_devices(); }
inline void SetDevice(int index) { check(gpbc_set_device(index)); }

struct fpElement { uint64_t l[4]; };
struct E2 { fpElement A0, A1; };

struct G1Affine {
    fpElement X{}, Y{};
    bool IsInfinity() const { uint64_t o = 0; for (int i = 0; i < 4; i++) o |= X.l[i] | Y.l[i]; return o == 0; }
    bool Equal(const G1Affine &b) const { return std::memcmp(this, &b, sizeof *this) == 0; }
    G1Affine &Neg(const G1Affine &a);
    G1Affine &ScalarMultiplication(const G1Affine &a, const Scalar &s) {
        check(gpbc_g1_scalar_mul_batch(&a, 1, s.le.data(), 1, this));
        return *this;
    }
    G1Affine &ScalarMultiplicationBase(const Scalar &s);
    // wire formats (gnark marshal.go): RawBytes 64 B, Bytes 32 B compressed; Unmarshal = SetBytes, returns bytes read
    std::vector<uint8_t> Marshal() const { std::vector<uint8_t> b(GPBC_G1_RAW_BYTES); check(gpbc_g1_marshal_batch(this, 1, 0, b.data())); return b; }
    std::array<uint8_t, GPBC_G1_COMPRESSED_BYTES> Bytes() const { std::array<uint8_t, GPBC_G1_COMPRESSED_BYTES> b; check(gpbc_g1_marshal_batch(this, 1, 1, b.data())); return b; }
    size_t Unmarshal(const std::vector<uint8_t> &buf) {
        if (buf.size() < GPBC_G1_COMPRESSED_BYTES) throw std::invalid_argument("short buffer");
        size_t eb = buf.size() >= GPBC_G1_RAW_BYTES ? GPBC_G1_RAW_BYTES : GPBC_G1_COMPRESSED_BYTES;
        uint8_t ok = 0;
        check(gpbc_g1_unmarshal_batch(buf.data(), eb, 1, this, &ok));
        if (!ok) throw std::invalid_argument("invalid G1 encoding");
        return (buf[0] & 0x80) || (buf[0] & 0xC0) == 0x40 ? GPBC_G1_COMPRESSED_BYTES : GPBC_G1_RAW_BYTES;
    }
};
// host-side field negation p - y on Montgomery limbs (gnark's G1Affine.Neg / G2Affine.Neg; not a hot-path operation)
inline fpElement fpNeg(const fpElement &y) {
    static const uint64_t P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    if ((y.l[0] | y.l[1] | y.l[2] | y.l[3]) == 0) return y;
    fpElement r;
    unsigned __int128 b = 0;
    for (int i = 0; i < 4; i++) {
        unsigned __int128 d = (unsigned __int128)P[i] - y.l[i] - b;
        r.l[i] = (uint64_t)d;
        b = (d >> 64) & 1;
    }
    return r;
}
struct G2Affine {
    E2 X{}, Y{};
    bool Equal(const G2Affine &b) const { return std::memcmp(this, &b, sizeof *this) == 0; }
    G2Affine &Neg(const G2Affine &a) { X = a.X; Y.A0 = fpNeg(a.Y.A0); Y.A1 = fpNeg(a.Y.A1); return *this; }
    G2Affine &ScalarMultiplication(const G2Affine &a, const Scalar &s) {
        check(gpbc_g2_scalar_mul_batch(&a, 1, s.le.data(), 1, this));
        return *this;
    }
    G2Affine &ScalarMultiplicationBase(const Scalar &s);
    std::vector<uint8_t> Marshal() const { std::vector<uint8_t> b(GPBC_G2_RAW_BYTES); check(gpbc_g2_marshal_batch(this, 1, 0, b.data())); return b; }
    std::array<uint8_t, GPBC_G2_COMPRESSED_BYTES> Bytes() const { std::array<uint8_t, GPBC_G2_COMPRESSED_BYTES> b; check(gpbc_g2_marshal_batch(this, 1, 1, b.data())); return b; }
    size_t Unmarshal(const std::vector<uint8_t> &buf) {          // includes gnark's subgroup check
        if (buf.size() < GPBC_G2_COMPRESSED_BYTES) throw std::invalid_argument("short buffer");
        size_t eb = buf.size() >= GPBC_G2_RAW_BYTES ? GPBC_G2_RAW_BYTES : GPBC_G2_COMPRESSED_BYTES;
        uint8_t ok = 0;
        check(gpbc_g2_unmarshal_batch(buf.data(), eb, 1, this, &ok));
        if (!ok) throw std::invalid_argument("invalid G2 encoding");
        return (buf[0] & 0x80) || (buf[0] & 0xC0) == 0x40 ? GPBC_G2_COMPRESSED_BYTES : GPBC_G2_RAW_BYTES;
    }
};
struct GT {
    E2 c[6]{};   // C0.B0, C0.B1, C0.B2, C1.B0, C1.B1, C1.B2
    bool Equal(const GT &b) const { return std::memcmp(this, &b, sizeof *this) == 0; }
    GT &Exp(const GT &x, const Scalar &k) { check(gpbc_gt_exp_batch(&x, k.le.data(), 1, this)); return *this; }
    GT &Mul(const GT &a, const GT &b) { check(gpbc_gt_mul_batch(&a, &b, 1, this)); return *this; }
    GT &Div(const GT &a, const GT &b) { check(gpbc_gt_div_batch(&a, &b, 1, this)); return *this; }
    GT &Inverse(const GT &a) { check(gpbc_gt_inverse_batch(&a, 1, this)); return *this; }
    std::array<uint8_t, GPBC_GT_BYTES> Bytes() const { std::array<uint8_t, GPBC_GT_BYTES> b; check(gpbc_gt_marshal_batch(this, 1, b.data())); return b; }
    std::vector<uint8_t> Marshal() const { auto b = Bytes(); return std::vector<uint8_t>(b.begin(), b.end()); }
    void Unmarshal(const std::vector<uint8_t> &buf) {
        if (buf.size() < GPBC_GT_BYTES) throw std::invalid_argument("short buffer");
        uint8_t ok = 0;
        check(gpbc_gt_unmarshal_batch(buf.data(), 1, this, &ok));
        if (!ok) throw std::invalid_argument("invalid GT encoding");
    }
};
static_assert(sizeof(G1Affine) == GPBC_G1_BYTES && sizeof(G2Affine) == GPBC_G2_BYTES && sizeof(GT) == GPBC_GT_BYTES, "gnark layouts");

// g1 = (1, 2), g2 = the alt_bn128 twist generator, Montgomery form (tools/gen_constants.py values)
inline void Generators(G1Affine &g1, G2Affine &g2) {
    g1.X = {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}};
    g1.Y = {{0xa6ba871b8b1e1b3aULL, 0x14f1d651eb8e167bULL, 0xccdd46def0f28c58ULL, 0x1c14ef83340fbe5eULL}};
    g2.X.A0 = {{0x8e83b5d102bc2026ULL, 0xdceb1935497b0172ULL, 0xfbb8264797811adfULL, 0x19573841af96503bULL}};
    g2.X.A1 = {{0xafb4737da84c6140ULL, 0x6043dd5a5802d8c4ULL, 0x09e950fc52a02f86ULL, 0x14fef0833aea7b6bULL}};
    g2.Y.A0 = {{0x619dfa9d886be9f6ULL, 0xfe7fd297f59e9b78ULL, 0xff9e1a62231b7dfeULL, 0x28fd7eebae9e4206ULL}};
    g2.Y.A1 = {{0x64095b56c71856eeULL, 0xdc57f922327d3cbbULL, 0x55f935be33351076ULL, 0x0da4a0e693fd6482ULL}};
}
inline G1Affine &G1Affine::Neg(const G1Affine &a) { X = a.X; Y = fpNeg(a.Y); return *this; }
// ScalarMultiplicationBase: fixed-base window tables of the two generators (1 MB / 2 MB of HBM), built on the first call and
// kept for the life of the process — 32 mixed additions per multiplication instead of the variable-base kernel's doublings.
namespace detail {
struct GeneratorTables {
    gpbc_fixed_base *g1 = nullptr, *g2 = nullptr;
    GeneratorTables() {
        G1Affine a; G2Affine b; Generators(a, b);
        check(gpbc_g1_fixed_base_create(&a, 1, &g1));
        check(gpbc_g2_fixed_base_create(&b, 1, &g2));
    }
};
inline const GeneratorTables &generator_tables() { static const GeneratorTables t; return t; }   // C++11: initialised once, thread-safe
}  // namespace detail
inline G1Affine &G1Affine::ScalarMultiplicationBase(const Scalar &s) { check(gpbc_fixed_base_msm(detail::generator_tables().g1, s.le.data(), 1, this)); return *this; }
inline G2Affine &G2Affine::ScalarMultiplicationBase(const Scalar &s) { check(gpbc_fixed_base_msm(detail::generator_tables().g2, s.le.data(), 1, this)); return *this; }

// bn254.Pair: product of pairings, one final exponentiation.  gnark's error: "invalid inputs sizes".
inline GT Pair(const std::vector<G1Affine> &P, const std::vector<G2Affine> &Q) {
    if (P.empty() || P.size() != Q.size()) throw std::invalid_argument("invalid inputs sizes");
    uint64_t seg[2] = {0, P.size()};
    GT out;
    check(gpbc_multi_pair(P.data(), Q.data(), seg, 1, &out));
    return out;
}
inline bool PairingCheck(const std::vector<G1Affine> &P, const std::vector<G2Affine> &Q) {
    if (P.empty() || P.size() != Q.size()) throw std::invalid_argument("invalid inputs sizes");
    uint64_t seg[2] = {0, P.size()};
    uint8_t ok = 0;
    check(gpbc_pairing_check(P.data(), Q.data(), seg, 1, &ok));
    return ok == 1;
}
// bn254.HashToG1(msg, dst) / HashToG2(msg, dst) (hash/hash_to.go:113-119,169-175,204-210,271-277; the BLS scheme hashes every
// message: signature/bls01_signature/bls_signature.go:56-63), hashing included; gnark errors on a DST longer than 255 bytes
inline std::vector<G1Affine> HashToG1Batch(const std::vector<std::string> &msgs, const std::string &dst) {
    if (dst.size() > 255) throw std::invalid_argument("invalid domain size (>255 bytes)");
    std::vector<G1Affine> out(msgs.size());
    std::string data;
    std::vector<uint64_t> off(msgs.size() + 1, 0);
    for (size_t i = 0; i < msgs.size(); i++) { data += msgs[i]; off[i + 1] = data.size(); }
    if (data.empty()) data.push_back('\0');
    check(gpbc_hash_to_g1(data.data(), off.data(), msgs.size(), dst.data(), dst.size(), out.data()));
    return out;
}
inline std::vector<G2Affine> HashToG2Batch(const std::vector<std::string> &msgs, const std::string &dst) {
    if (dst.size() > 255) throw std::invalid_argument("invalid domain size (>255 bytes)");
    std::vector<G2Affine> out(msgs.size());
    std::string data;
    std::vector<uint64_t> off(msgs.size() + 1, 0);
    for (size_t i = 0; i < msgs.size(); i++) { data += msgs[i]; off[i + 1] = data.size(); }
    if (data.empty()) data.push_back('\0');
    check(gpbc_hash_to_g2(data.data(), off.data(), msgs.size(), dst.data(), dst.size(), out.data()));
    return out;
}
inline G1Affine HashToG1(const std::string &msg, const std::string &dst) { return HashToG1Batch({msg}, dst)[0]; }
inline G2Affine HashToG2(const std::string &msg, const std::string &dst) { return HashToG2Batch({msg}, dst)[0]; }
// batched forms the engine adds
inline std::vector<GT> PairBatch(const std::vector<G1Affine> &P, const std::vector<G2Affine> &Q) {
    if (P.empty() || P.size() != Q.size()) throw std::invalid_argument("invalid inputs sizes");
    std::vector<GT> out(P.size());
    check(gpbc_pair_batch(P.data(), Q.data(), P.size(), out.data()));
    return out;
}
inline std::vector<G1Affine> G1ScalarMultiplicationBatch(const std::vector<G1Affine> &bases, const std::vector<Scalar> &s) {
    std::vector<G1Affine> out(s.size());
    check(gpbc_g1_scalar_mul_batch(bases.data(), bases.size(), s.data(), s.size(), out.data()));
    return out;
}
inline std::vector<G2Affine> G2ScalarMultiplicationBatch(const std::vector<G2Affine> &bases, const std::vector<Scalar> &s) {
    std::vector<G2Affine> out(s.size());
    check(gpbc_g2_scalar_mul_batch(bases.data(), bases.size(), s.data(), s.size(), out.data()));
    return out;
}
// k products against ONE list of G2 points (a decryption key against k ciphertexts): out[j] = Pair(P[j*m .. (j+1)*m), Q);
// the Miller lines of Q are computed once (gnark: PrecomputeLines / MillerLoopFixedQ)
inline std::vector<GT> PairFixedQ(const std::vector<G1Affine> &P, const std::vector<G2Affine> &Q) {
    if (Q.empty() || P.empty() || P.size() % Q.size()) throw std::invalid_argument("invalid inputs sizes");
    std::vector<GT> out(P.size() / Q.size());
    check(gpbc_multi_pair_fixed_q(P.data(), Q.data(), Q.size(), out.size(), out.data()));
    return out;
}
// Fixed-base window tables in HBM: ScalarMultiplicationBase-style batches and sums  sum_j [s_j] base_j  over fixed bases
class G1FixedBase {
public:
    explicit G1FixedBase(const std::vector<G1Affine> &bases) : n_(bases.size()) { check(gpbc_g1_fixed_base_create(bases.data(), bases.size(), &h_)); }
    ~G1FixedBase() { gpbc_fixed_base_destroy(h_); }
    G1FixedBase(const G1FixedBase &) = delete;
    G1FixedBase &operator=(const G1FixedBase &) = delete;
    // scalars: n_msm rows of NBases() scalars; one point per row
    std::vector<G1Affine> Msm(const std::vector<Scalar> &scalars) const {
        if (scalars.size() % n_) throw std::invalid_argument("need one scalar per base and sum");
        std::vector<G1Affine> out(scalars.size() / n_);
        check(gpbc_fixed_base_msm(h_, scalars.data(), out.size(), out.data()));
        return out;
    }
    size_t NBases() const { return n_; }
private:
    gpbc_fixed_base *h_ = nullptr;
    size_t n_;
};
static_assert(sizeof(Scalar) == GPBC_SCALAR_BYTES, "scalar layout");

}  // namespace bn254
#endif
