// BLAKE3 (default hash mode, 32-byte output) for gfx950, written from the public specification
// (SURVEY.md Appendix C).  Replaces the `blake3` crate calls of /root/reference/crypto/src/hash/blake/mod.rs:27-59.
// One lane hashes one message; all 16 state words and 16 message words live in VGPRs, the message permutation is
// resolved at compile time (no register moves), rotations map to v_alignbit_b32.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wf {
namespace b3 {

enum : uint32_t { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };

// message word index used at position i of round r (sigma applied r times)
__host__ __device__ constexpr int sched(int r, int i) {
    constexpr int P[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
    int idx = i;
    for (int k = 0; k < r; k++) idx = P[idx];
    return idx;
}

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return __builtin_rotateright32(x, n); }

#define WF_B3_G(a, b, c, d, mx, my) \
    a = a + b + (mx);               \
    d = rotr(d ^ a, 16);            \
    c = c + d;                      \
    b = rotr(b ^ c, 12);            \
    a = a + b + (my);               \
    d = rotr(d ^ a, 8);             \
    c = c + d;                      \
    b = rotr(b ^ c, 7);

template <int R>
__device__ __forceinline__ void round_fn(uint32_t (&v)[16], const uint32_t (&m)[16]) {
    WF_B3_G(v[0], v[4], v[8], v[12], m[sched(R, 0)], m[sched(R, 1)])
    WF_B3_G(v[1], v[5], v[9], v[13], m[sched(R, 2)], m[sched(R, 3)])
    WF_B3_G(v[2], v[6], v[10], v[14], m[sched(R, 4)], m[sched(R, 5)])
    WF_B3_G(v[3], v[7], v[11], v[15], m[sched(R, 6)], m[sched(R, 7)])
    WF_B3_G(v[0], v[5], v[10], v[15], m[sched(R, 8)], m[sched(R, 9)])
    WF_B3_G(v[1], v[6], v[11], v[12], m[sched(R, 10)], m[sched(R, 11)])
    WF_B3_G(v[2], v[7], v[8], v[13], m[sched(R, 12)], m[sched(R, 13)])
    WF_B3_G(v[3], v[4], v[9], v[14], m[sched(R, 14)], m[sched(R, 15)])
}

// cv <- first 8 words of compress(cv, m, counter, block_len, flags)
__device__ __forceinline__ void compress(uint32_t (&cv)[8], const uint32_t (&m)[16], uint32_t counter_lo,
                                         uint32_t counter_hi, uint32_t block_len, uint32_t flags) {
    uint32_t v[16];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = cv[i];
    v[8] = 0x6A09E667u;
    v[9] = 0xBB67AE85u;
    v[10] = 0x3C6EF372u;
    v[11] = 0xA54FF53Au;
    v[12] = counter_lo;
    v[13] = counter_hi;
    v[14] = block_len;
    v[15] = flags;
    round_fn<0>(v, m);
    round_fn<1>(v, m);
    round_fn<2>(v, m);
    round_fn<3>(v, m);
    round_fn<4>(v, m);
    round_fn<5>(v, m);
    round_fn<6>(v, m);
#pragma unroll
    for (int i = 0; i < 8; i++) cv[i] = v[i] ^ v[i + 8];
}

__device__ __forceinline__ void set_iv(uint32_t (&cv)[8]) {
    cv[0] = 0x6A09E667u;
    cv[1] = 0xBB67AE85u;
    cv[2] = 0x3C6EF372u;
    cv[3] = 0xA54FF53Au;
    cv[4] = 0x510E527Fu;
    cv[5] = 0x9B05688Cu;
    cv[6] = 0x1F83D9ABu;
    cv[7] = 0x5BE0CD19u;
}

// Blake3_256::merge (blake/mod.rs:31-33): hash of two concatenated 32-byte digests = one compression.
__device__ __forceinline__ void merge(const uint32_t (&m)[16], uint32_t (&out)[8]) {
    set_iv(out);
    compress(out, m, 0, 0, 64, CHUNK_START | CHUNK_END | ROOT);
}

// The same for either hasher of the reference, on digests kept in 32-byte SLOTS (the device-side layout of leaves and nodes
// whatever the digest size; s = the two child slots as loaded): DW = 8 is Blake3_256::merge, DW = 6 Blake3_192::merge
// (blake/mod.rs:80-83: the hash of the 48 bytes of two 24-byte digests, truncated to 24 bytes -- words 6, 7 of the result
// slot are written as zeros, Digest::as_bytes pads the same way, crypto/src/hash/mod.rs:107-113).
template <int DW>
__device__ __forceinline__ void merge_slots(const uint32_t (&s)[16], uint32_t (&out)[8]) {
    if constexpr (DW == 8) {
        merge(s, out);
    } else {
        uint32_t m[16];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            m[i] = s[i];
            m[6 + i] = s[8 + i];
        }
        m[12] = m[13] = m[14] = m[15] = 0;
        set_iv(out);
        compress(out, m, 0, 0, 48, CHUNK_START | CHUNK_END | ROOT);
        out[6] = out[7] = 0;
    }
}

// ---- the same merge by FOUR lanes (a quad): lane q holds column q of the 4 x 4 state (a = v[q], b = v[4+q], c = v[8+q],
// d = v[12+q]); the column step is lane-local, the diagonal step reaches the neighbours' b, c, d through DPP quad permutes.
// A third of the instructions per lane (about 230 against 678): for the narrow top levels of a Merkle tree, where one
// compression per level is pure latency.  msg: the 16 message words (two child digests) in LDS; lane q returns the two
// words cv[q] and cv[4 + q] of the parent.
__host__ __device__ constexpr uint32_t quad_index_word(int q, int word) {  // 4-bit message indices, eight per word
    uint32_t w = 0;
    for (int n = 0; n < 8; n++) {
        const int j = word * 8 + n;  // j = 4 r + t: t = 0, 1 column step (mx, my); t = 2, 3 diagonal step
        if (j >= 28) break;
        const int r = j / 4, t = j % 4;
        const int pos = (t < 2 ? 0 : 8) + 2 * q + (t & 1);
        w |= (uint32_t)sched(r, pos) << (4 * n);
    }
    return w;
}

template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, false);
}

template <int DW = 8>
__device__ __forceinline__ void merge_quad(const uint32_t *msg, uint32_t q, uint32_t &out_lo, uint32_t &out_hi) {
    // this lane's 28 message indices: one of four compile-time tables, selected by q
    uint32_t iw[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t w0 = quad_index_word(0, w), w1 = quad_index_word(1, w), w2 = quad_index_word(2, w), w3 = quad_index_word(3, w);
        iw[w] = q == 0 ? w0 : q == 1 ? w1 : q == 2 ? w2 : w3;
    }
    uint32_t mw[28];
#pragma unroll
    for (int j = 0; j < 28; j++) {
        const uint32_t i = (iw[j / 8] >> (4 * (j % 8))) & 15u;
        if constexpr (DW == 8)
            mw[j] = msg[i];
        else  // 48-byte message: words 0..5 of the first slot, 0..5 of the second, zeros
            mw[j] = i < 12 ? msg[i < 6 ? i : i + 2] : 0u;
    }
    const uint32_t iv_lo = q == 0 ? 0x6A09E667u : q == 1 ? 0xBB67AE85u : q == 2 ? 0x3C6EF372u : 0xA54FF53Au;
    const uint32_t iv_hi = q == 0 ? 0x510E527Fu : q == 1 ? 0x9B05688Cu : q == 2 ? 0x1F83D9ABu : 0x5BE0CD19u;
    uint32_t a = iv_lo, b = iv_hi, c = iv_lo;
    uint32_t d = q == 2 ? (DW == 8 ? 64u : 48u) : q == 3 ? (uint32_t)(CHUNK_START | CHUNK_END | ROOT) : 0u;  // counter = 0, block_len, flags
#pragma unroll
    for (int r = 0; r < 7; r++) {
        WF_B3_G(a, b, c, d, mw[4 * r], mw[4 * r + 1])
        b = quad_perm<0x39>(b);  // lane q takes lane q + 1
        c = quad_perm<0x4E>(c);  //              q + 2
        d = quad_perm<0x93>(d);  //              q + 3
        WF_B3_G(a, b, c, d, mw[4 * r + 2], mw[4 * r + 3])
        b = quad_perm<0x93>(b);
        c = quad_perm<0x4E>(c);
        d = quad_perm<0x39>(d);
    }
    out_lo = a ^ c;
    out_hi = (DW == 6 && q >= 2) ? 0u : (b ^ d);  // (words 6, 7 of a 24-byte digest's slot are zero)
}

// Hash of a message of `len` <= 1024 bytes (one chunk; len a multiple of 4) delivered block by block:
// load(block_index, m) must fill the 16 words of 64-byte block `block_index`, zero-padded past `len`.
// Longer messages go through chunk_cv + merge_chunk_cvs below (one lane per chunk).
template <class LoadBlock>
__device__ __forceinline__ void hash_stream(uint64_t len, LoadBlock load, uint32_t (&out)[8]) {
    uint32_t m[16];
    set_iv(out);
    const uint32_t nblocks = len == 0 ? 1u : (uint32_t)((len + 63) >> 6);
    for (uint32_t b = 0; b < nblocks; b++) {
        load((uint64_t)b, m);
        const uint32_t blen = (uint32_t)(len - (uint64_t)b * 64 < 64 ? len - (uint64_t)b * 64 : 64);
        const uint32_t flags = (b == 0 ? CHUNK_START : 0u) | (b == nblocks - 1 ? (CHUNK_END | ROOT) : 0u);
        compress(out, m, 0, 0, blen, flags);
    }
}

// Chaining value of chunk `c` (clen bytes, 1..1024) of a multi-chunk message: load(block_in_chunk, m) fills a block.
template <class LoadBlock>
__device__ __forceinline__ void chunk_cv(uint64_t c, uint32_t clen, LoadBlock load, uint32_t (&cv)[8]) {
    uint32_t m[16];
    const uint32_t nblocks = (clen + 63) >> 6;
    set_iv(cv);
    for (uint32_t b = 0; b < nblocks; b++) {
        load(b, m);
        const uint32_t blen = clen - b * 64 < 64 ? clen - b * 64 : 64;
        const uint32_t flags = (b == 0 ? CHUNK_START : 0u) | (b == nblocks - 1 ? (uint32_t)CHUNK_END : 0u);
        compress(cv, m, (uint32_t)c, (uint32_t)(c >> 32), blen, flags);
    }
}

// Root of the BLAKE3 tree over n >= 2 chunk chaining values cvs[0..n) (8 words each, in memory): the left subtree
// of every node covers the largest power of two of chunks strictly smaller than the node's total.
template <int MAX_DEPTH = 24>
__device__ __forceinline__ void merge_chunk_cvs(const uint32_t *cvs, uint64_t n, uint32_t (&out)[8]) {
    uint32_t stack[MAX_DEPTH][8];
    uint32_t m[16], cv[8];
    int sp = 0;
    for (uint64_t c = 0; c < n; c++) {
#pragma unroll
        for (int i = 0; i < 8; i++) cv[i] = cvs[c * 8 + i];
        if (c + 1 < n) {
            uint64_t total = c + 1;
            while ((total & 1) == 0) {
                sp--;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    m[i] = stack[sp][i];
                    m[8 + i] = cv[i];
                }
                set_iv(cv);
                compress(cv, m, 0, 0, 64, PARENT);
                total >>= 1;
            }
#pragma unroll
            for (int i = 0; i < 8; i++) stack[sp][i] = cv[i];
            sp++;
        }
    }
    while (sp > 0) {
        sp--;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            m[i] = stack[sp][i];
            m[8 + i] = cv[i];
        }
        set_iv(cv);
        compress(cv, m, 0, 0, 64, PARENT | (sp == 0 ? (uint32_t)ROOT : 0u));
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = cv[i];
}

}  // namespace b3
}  // namespace wf
