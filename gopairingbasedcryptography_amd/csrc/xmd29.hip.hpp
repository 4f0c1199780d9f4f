// hash_to_field on the device: expand_message_xmd with SHA-256 (RFC 9380 §5.3.1) and the reduction of its 48-byte strings to
// base-field elements (§5.2, L = 48) — the head of bn254.HashToG1 / HashToG2 and gnark's fp.Hash(msg, dst, count), which the
// reference reaches through hash/hash_to.go:113-119,169-175,204-210,271-277 [EXT, parity unpinned beyond the RFC's own K.1
// vectors for expand_message_xmd, which tests/test_hash_to_curve.py holds].  One message per lane; a lane's whole state lives in
// registers (eight chaining words, a sixteen-word block) and every byte of the hashed streams is produced by a position
// function, so there is no per-lane staging buffer:
//     b_0 = H(Z_pad(64 zero bytes) || msg || I2OSP(len_in_bytes, 2) || 0 || DST') — the zero block's state is a constant
//     b_1 = H(b_0 || 1 || DST'),   b_i = H((b_0 xor b_(i-1)) || i || DST'),   DST' = DST || I2OSP(len(DST), 1)
// DST (at most 255 bytes; longer ones are hashed down by the caller as the RFC prescribes) travels in the kernel arguments.
// Device only: the host build of the bounds harness has no use for it (the field operations it ends in are covered there).
#ifndef GPBC_XMD29_HIP_HPP
#define GPBC_XMD29_HIP_HPP
#include "wire29.hip.hpp"

namespace gpbc {

struct XmdDst { uint8_t b[256]; uint32_t len; };          // kernel argument: uniform, read with scalar loads

__device__ __forceinline__ uint32_t sha_rotr(uint32_t x, int n) { return __builtin_rotateright32(x, n); }
// one SHA-256 compression (FIPS 180-4 §6.2.2); w is consumed (rolling message schedule)
__device__ __noinline__ void sha256_compress(uint32_t (&st)[8], uint32_t (&w)[16]) {
    constexpr uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
        0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
        0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
        0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
        0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
        0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            w[i & 15] += (sha_rotr(w15, 7) ^ sha_rotr(w15, 18) ^ (w15 >> 3)) + w[(i - 7) & 15] + (sha_rotr(w2, 17) ^ sha_rotr(w2, 19) ^ (w2 >> 10));
        }
        const uint32_t t1 = h + (sha_rotr(e, 6) ^ sha_rotr(e, 11) ^ sha_rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i & 15];
        const uint32_t t2 = (sha_rotr(a, 2) ^ sha_rotr(a, 13) ^ sha_rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
__device__ __forceinline__ void sha256_iv(uint32_t (&st)[8]) {
    constexpr uint32_t IV[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
#pragma unroll
    for (int i = 0; i < 8; i++) st[i] = IV[i];
}

// Hashes the stream  byte_at(pos), pos < body_len,  then the SHA-256 padding for a message of done + body_len bytes of which
// `done` (a multiple of 64) are already in st.  STATIC_BLOCKS > 0 unrolls the block loop (every position is then a compile-time
// constant, so a byte_at that indexes registers stays in registers); 0 loops over the data-dependent number of blocks.
template <class ByteAt> __device__ __forceinline__ void sha256_block_from(uint32_t (&w)[16], uint64_t blk, uint64_t body_len, uint64_t n_blocks, uint64_t bits, ByteAt &&byte_at) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
        uint32_t word = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint64_t pos = blk * 64 + 4 * j + t;
            uint32_t v;
            if (pos < body_len) v = byte_at(pos);
            else if (pos == body_len) v = 0x80u;
            else if (pos >= n_blocks * 64 - 8) v = (uint32_t)(bits >> (8 * (n_blocks * 64 - 1 - pos))) & 0xffu;
            else v = 0;
            word = (word << 8) | v;
        }
        w[j] = word;
    }
}
template <int STATIC_BLOCKS, class ByteAt> __device__ __forceinline__ void sha256_tail(uint32_t (&st)[8], uint64_t done, uint64_t body_len, ByteAt &&byte_at) {
    const uint64_t n_blocks = (body_len + 9 + 63) / 64, bits = (done + body_len) * 8;
    uint32_t w[16];
    if (STATIC_BLOCKS > 0) {
#pragma unroll
        for (int blk = 0; blk < STATIC_BLOCKS; blk++)
            if ((uint64_t)blk < n_blocks) { sha256_block_from(w, (uint64_t)blk, body_len, n_blocks, bits, byte_at); sha256_compress(st, w); }
    } else {
        for (uint64_t blk = 0; blk < n_blocks; blk++) { sha256_block_from(w, blk, body_len, n_blocks, bits, byte_at); sha256_compress(st, w); }
    }
}

// uniform bytes of expand_message_xmd as N_DIGESTS x 8 big-endian words (len_in_bytes = 32 * N_DIGESTS... the callers use 96 and 192)
template <int N_DIGESTS> __device__ __forceinline__ void expand_message_xmd(uint32_t (&out)[N_DIGESTS * 8], const uint8_t *msg, uint64_t mlen, const XmdDst &dst) {
    constexpr uint32_t len_in_bytes = 32 * N_DIGESTS;
    const uint32_t dlen = dst.len;
    // b_0: the state after the 64 zero bytes of Z_pad is a constant
    uint32_t b0[8] = {0xda5698beu, 0x17b9b469u, 0x62335799u, 0x779fbecau, 0x8ce5d491u, 0xc0d26243u, 0xbafef9eau, 0x1837a9d8u};
    sha256_tail<0>(b0, 64, mlen + 4 + dlen, [&](uint64_t pos) -> uint32_t {
        if (pos < mlen) return msg[pos];
        const uint64_t q = pos - mlen;
        if (q == 0) return (len_in_bytes >> 8) & 0xffu;
        if (q == 1) return len_in_bytes & 0xffu;
        if (q == 2) return 0;
        return q - 3 < dlen ? dst.b[q - 3] : dlen;
    });
    uint32_t prev[8];
#pragma unroll
    for (int i = 0; i < 8; i++) prev[i] = 0;
#pragma unroll
    for (int idx = 1; idx <= N_DIGESTS; idx++) {
        uint32_t head[8], st[8];
#pragma unroll
        for (int i = 0; i < 8; i++) head[i] = b0[i] ^ prev[i];          // idx = 1: b_0 itself
        sha256_iv(st);
        sha256_tail<5>(st, 0, (uint64_t)34 + dlen, [&](uint64_t pos) -> uint32_t {        // 34 + 255 + 9 bytes: at most five blocks
            if (pos < 32) return (head[pos >> 2] >> (8 * (3 - (pos & 3)))) & 0xffu;
            if (pos == 32) return (uint32_t)idx;
            return pos - 33 < dlen ? dst.b[pos - 33] : dlen;
        });
#pragma unroll
        for (int i = 0; i < 8; i++) { prev[i] = st[i]; out[(idx - 1) * 8 + i] = st[i]; }
    }
}

// field element e of the uniform bytes: OS2IP(48 bytes) mod p, internal form.  v = hi 2^256 + lo with hi < 2^128.
template <int N_WORDS> __device__ __forceinline__ Fe xmd_field(const uint32_t (&u)[N_WORDS], int e) {
    uint32_t lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { lo[i] = u[12 * e + 11 - i]; hi[i] = i < 4 ? u[12 * e + 3 - i] : 0u; }
    const uint32_t two128[8] = {0, 0, 0, 0, 1, 0, 0, 0};
    const Fe r256 = fe_sqr(fe_from_plain_words(two128));                  // 2^256 as a field element
    return fe_reduce(fe_norm(fe_add(fe_from_plain_words(lo), fe_mul(fe_from_plain_words(hi), r256))));
}

}  // namespace gpbc
#endif
