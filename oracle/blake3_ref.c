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

/* out_cv = first 8 words of the compression output */
static void compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t block_len,
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

static void load_block(const uint8_t *p, size_t len, uint32_t w[16]) {
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
