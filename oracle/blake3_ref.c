/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * Portable restatement of the BLAKE3 hash function (default hash mode, 32-byte output) from the public BLAKE3
 * specification.  The reference does not contain this algorithm: it calls the third-party `blake3` crate
 * (`blake3 = "1.3"`, default-features = false, /root/reference/crypto/Cargo.toml:33) at
 * /root/reference/crypto/src/hash/blake/mod.rs:28,32,39,51 (blake3::hash) and :124-138 (blake3::Hasher).
 * Pinned against golden digests of the official C implementation (tests/golden/blake3_kat.json,
 * generator oracle/gen_golden.py).
 */
#include "blake3_ref.h"

#include <string.h>

enum { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };
enum { BLOCK_LEN = 64, CHUNK_LEN = 1024 };

static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                               0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t MSG_PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};

static inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

#define G(a, b, c, d, mx, my)        \
    do {                             \
        v[a] = v[a] + v[b] + (mx);   \
        v[d] = rotr32(v[d] ^ v[a], 16); \
        v[c] = v[c] + v[d];          \
        v[b] = rotr32(v[b] ^ v[c], 12); \
        v[a] = v[a] + v[b] + (my);   \
        v[d] = rotr32(v[d] ^ v[a], 8);  \
        v[c] = v[c] + v[d];          \
        v[b] = rotr32(v[b] ^ v[c], 7);  \
    } while (0)

/* out_cv = first 8 words of the compression output (scalar form: the definition, kept as the cross-check of the SIMD form) */
static void compress_scalar(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t block_len,
                     uint32_t flags, uint32_t out_cv[8]) {
    uint32_t v[16], m[16], t[16];
    for (int i = 0; i < 8; i++) v[i] = cv[i];
    for (int i = 0; i < 4; i++) v[8 + i] = IV[i];
    v[12] = (uint32_t)counter;
    v[13] = (uint32_t)(counter >> 32);
    v[14] = block_len;
    v[15] = flags;
    memcpy(m, block, sizeof(m));
    for (int r = 0; r < 7; r++) {
        G(0, 4, 8, 12, m[0], m[1]);
        G(1, 5, 9, 13, m[2], m[3]);
        G(2, 6, 10, 14, m[4], m[5]);
        G(3, 7, 11, 15, m[6], m[7]);
        G(0, 5, 10, 15, m[8], m[9]);
        G(1, 6, 11, 12, m[10], m[11]);
        G(2, 7, 8, 13, m[12], m[13]);
        G(3, 4, 9, 14, m[14], m[15]);
        for (int i = 0; i < 16; i++) t[i] = m[MSG_PERM[i]];
        memcpy(m, t, sizeof(m));
    }
    for (int i = 0; i < 8; i++) out_cv[i] = v[i] ^ v[i + 8];
}

#if defined(__SSSE3__)
/* The same compression with the four columns (then the four diagonals) of the state side by side in 128-bit vectors --
 * the arrangement the spec's round function is drawn in: rows a = v0..3, b = v4..7, c = v8..11, d = v12..15; a column
 * step is G on the rows, the diagonal step the same after rotating row b by one lane, c by two, d by three.  This is how
 * the `blake3` crate the reference links computes ONE compression on x86 (its published 62-106 ns per 2-to-1 hash,
 * /root/reference/crypto/README.md:69-75, are of such code); the scalar form above needs ~260 ns.  Written from the
 * specification; bit-equality with the scalar form and with LLVM's BLAKE3 is tested (tests/test_oracle.py). */
#include <immintrin.h>

static const uint8_t MSG_SCHEDULE[7][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8},
    {3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1}, {10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6},
    {12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4}, {9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7},
    {11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13}};

static inline __m128i rot_shift(__m128i x, int n) { return _mm_or_si128(_mm_srli_epi32(x, n), _mm_slli_epi32(x, 32 - n)); }

static void compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t block_len,
                     uint32_t flags, uint32_t out_cv[8]) {
    const __m128i r16 = _mm_set_epi8(13, 12, 15, 14, 9, 8, 11, 10, 5, 4, 7, 6, 1, 0, 3, 2);
    const __m128i r8 = _mm_set_epi8(12, 15, 14, 13, 8, 11, 10, 9, 4, 7, 6, 5, 0, 3, 2, 1);
    __m128i a = _mm_loadu_si128((const __m128i *)cv), b = _mm_loadu_si128((const __m128i *)(cv + 4));
    __m128i c = _mm_loadu_si128((const __m128i *)IV);
    __m128i d = _mm_set_epi32((int)flags, (int)block_len, (int)(uint32_t)(counter >> 32), (int)(uint32_t)counter);
    const uint32_t *m = block;
#define B3_G(mx, my)                                     \
    a = _mm_add_epi32(_mm_add_epi32(a, b), (mx));        \
    d = _mm_shuffle_epi8(_mm_xor_si128(d, a), r16);      \
    c = _mm_add_epi32(c, d);                             \
    b = rot_shift(_mm_xor_si128(b, c), 12);              \
    a = _mm_add_epi32(_mm_add_epi32(a, b), (my));        \
    d = _mm_shuffle_epi8(_mm_xor_si128(d, a), r8);       \
    c = _mm_add_epi32(c, d);                             \
    b = rot_shift(_mm_xor_si128(b, c), 7)
    for (int r = 0; r < 7; r++) {
        const uint8_t *s = MSG_SCHEDULE[r];
        B3_G(_mm_set_epi32((int)m[s[6]], (int)m[s[4]], (int)m[s[2]], (int)m[s[0]]),
             _mm_set_epi32((int)m[s[7]], (int)m[s[5]], (int)m[s[3]], (int)m[s[1]]));
        b = _mm_shuffle_epi32(b, _MM_SHUFFLE(0, 3, 2, 1)); /* v5 v6 v7 v4 */
        c = _mm_shuffle_epi32(c, _MM_SHUFFLE(1, 0, 3, 2)); /* v10 v11 v8 v9 */
        d = _mm_shuffle_epi32(d, _MM_SHUFFLE(2, 1, 0, 3)); /* v15 v12 v13 v14 */
        B3_G(_mm_set_epi32((int)m[s[14]], (int)m[s[12]], (int)m[s[10]], (int)m[s[8]]),
             _mm_set_epi32((int)m[s[15]], (int)m[s[13]], (int)m[s[11]], (int)m[s[9]]));
        b = _mm_shuffle_epi32(b, _MM_SHUFFLE(2, 1, 0, 3));
        c = _mm_shuffle_epi32(c, _MM_SHUFFLE(1, 0, 3, 2));
        d = _mm_shuffle_epi32(d, _MM_SHUFFLE(0, 3, 2, 1));
    }
#undef B3_G
    _mm_storeu_si128((__m128i *)out_cv, _mm_xor_si128(a, c));
    _mm_storeu_si128((__m128i *)(out_cv + 4), _mm_xor_si128(b, d));
}
#else
#define compress compress_scalar
#endif

/* test hook: one compression through both forms (tests/test_oracle.py) */
void orc_blake3_compress_both(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t block_len,
                              uint32_t flags, uint32_t out_simd[8], uint32_t out_scalar[8]) {
    compress(cv, block, counter, block_len, flags, out_simd);
    compress_scalar(cv, block, counter, block_len, flags, out_scalar);
}

static void load_block(const uint8_t *p, size_t len, uint32_t w[16]) {
    if (len == 64 && __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__) {
        memcpy(w, p, 64);
        return;
    }
    uint8_t buf[64];
    memset(buf, 0, sizeof(buf));
    memcpy(buf, p, len);
    for (int i = 0; i < 16; i++)
        w[i] = (uint32_t)buf[4 * i] | ((uint32_t)buf[4 * i + 1] << 8) | ((uint32_t)buf[4 * i + 2] << 16) |
               ((uint32_t)buf[4 * i + 3] << 24);
}

/* chaining value of one chunk (len in 1..=1024, or 0 only for the empty input); extra_flags carries ROOT */
static void chunk_cv(const uint8_t *p, size_t len, uint64_t chunk_counter, uint32_t extra_flags, uint32_t out[8]) {
    uint32_t cv[8], w[16];
    memcpy(cv, IV, sizeof(cv));
    size_t nblocks = len == 0 ? 1 : (len + BLOCK_LEN - 1) / BLOCK_LEN;
    for (size_t b = 0; b < nblocks; b++) {
        size_t off = b * BLOCK_LEN;
        size_t blen = len - off < BLOCK_LEN ? len - off : BLOCK_LEN;
        uint32_t flags = 0;
        if (b == 0) flags |= CHUNK_START;
        if (b == nblocks - 1) flags |= CHUNK_END | extra_flags;
        load_block(p + off, blen, w);
        compress(cv, w, chunk_counter, (uint32_t)blen, flags, cv);
    }
    memcpy(out, cv, sizeof(cv));
}

static void parent_cv(const uint32_t l[8], const uint32_t r[8], uint32_t extra_flags, uint32_t out[8]) {
    uint32_t w[16];
    memcpy(w, l, 32);
    memcpy(w + 8, r, 32);
    compress(IV, w, 0, BLOCK_LEN, PARENT | extra_flags, out);
}

void orc_blake3_hash(const uint8_t *in, size_t len, uint8_t out[32]) {
    uint32_t cv[8];
    if (len <= CHUNK_LEN) {
        chunk_cv(in, len, 0, ROOT, cv);
    } else {
        uint32_t stack[64][8];
        int sp = 0;
        size_t nchunks = (len + CHUNK_LEN - 1) / CHUNK_LEN;
        /* all chunks but the last: push and merge completed subtrees */
        for (size_t c = 0; c + 1 < nchunks; c++) {
            uint32_t x[8];
            chunk_cv(in + c * CHUNK_LEN, CHUNK_LEN, c, 0, x);
            size_t total = c + 1;
            while ((total & 1) == 0) {
                parent_cv(stack[--sp], x, 0, x);
                total >>= 1;
            }
            memcpy(stack[sp++], x, 32);
        }
        size_t last = nchunks - 1;
        chunk_cv(in + last * CHUNK_LEN, len - last * CHUNK_LEN, last, 0, cv);
        while (sp > 0) {
            sp--;
            parent_cv(stack[sp], cv, sp == 0 ? ROOT : 0, cv);
        }
    }
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)cv[i];
        out[4 * i + 1] = (uint8_t)(cv[i] >> 8);
        out[4 * i + 2] = (uint8_t)(cv[i] >> 16);
        out[4 * i + 3] = (uint8_t)(cv[i] >> 24);
    }
}
