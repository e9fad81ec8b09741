/* C99 client of include/wf_lde.h: proves the header is plain C (gcc -std=c99 -Wall -Wextra -Werror -pedantic) and, on
 * a box with a GPU, runs one small commitment through the ABI the way a C host would:
 *   Prover::build_trace_commitment (/root/reference/prover/src/lib.rs:615-670) on a 2^8 x 3 f64 trace, blowup 4,
 *   then checks what any client can check without a second implementation: nodes[0] is the zero digest, the root is
 *   nodes[1], the resident form returns the same root, a queried position returns the row the copy-out form wrote
 *   and a Merkle path of depth + 1 digests whose first entry is that row's leaf; bad parameters come back as the
 *   documented status codes.  Exit code 0 = ok, 77 = no HIP device (skipped), anything else = failure.
 * Build: gcc -std=c99 -Wall -Wextra -Werror -pedantic -I include tests/c/test_abi.c -L <csrc> -lwf_lde -o test_abi */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "wf_lde.h"

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            fprintf(stderr, "%s:%d: %s failed (%s)\n", __FILE__, __LINE__, #cond, wf_last_error()); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(void) {
    enum { LOG_R = 8, LOG_B = 2, COLS = 3, R = 1 << LOG_R, N = R << LOG_B };
    const uint64_t P = 0xFFFFFFFF00000001ULL;
    wf_params p;
    wf_ctx *ctx = NULL;
    uint64_t *cols[COLS], *polys[COLS], *lde;
    const void *col_ptrs[COLS];
    void *poly_ptrs[COLS], *lde_ptrs[1];
    uint8_t *leaves, *nodes, root[32], root2[32], zero[32], path[(LOG_R + LOG_B + 1) * 32];
    uint64_t x = 0x9E3779B97F4A7C15ULL, pos[1], *row, n_rows, row_elems;
    uint32_t depth;
    wf_commitment *com = NULL;
    size_t rw;
    int i, j, rc;

    memset(&p, 0, sizeof p);
    p.field = WF_FIELD_F64;
    p.ext_degree = 1;
    p.log2_trace_len = LOG_R;
    p.log2_blowup = LOG_B;
    p.n_cols = COLS;
    p.n_traces = 1;
    p.digest_bytes = 32;
    p.domain_offset[0] = 7;

    /* no device needed: validation and sizes */
    CHECK(wf_params_check(&p, 0) == WF_OK);
    CHECK(wf_row_width(&p) == 8 && wf_column_bytes(&p) == R * 8 && wf_lde_bytes(&p) == (size_t)N * 8 * 8);
    CHECK(wf_digests_bytes(&p) == (size_t)N * 32 && wf_elem_bytes(WF_FIELD_F128) == 16);
    p.log2_blowup = 0;
    CHECK(wf_params_check(&p, 0) == WF_ERR_BLOWUP);
    p.log2_blowup = LOG_B;
    p.n_traces = 2;
    CHECK(wf_params_check(&p, 1) == WF_ERR_TRACES);
    p.n_traces = 1;
    {
        uint32_t first, count, dig[4];
        CHECK(wf_shard_cosets(8, 3, 4, &first, &count) == WF_OK && first == 6 && count == 2);
        CHECK(wf_shard_cosets(8, 0, 3, &first, &count) == WF_ERR_ARG);
        CHECK(wf_plan_digits(WF_FIELD_F64, 20, 1, dig) == 2 && dig[0] == 10 && dig[1] == 10);
    }

    if (wf_device_count() < 1) {
        CHECK(wf_ctx_create(0, &ctx) == WF_ERR_HIP); /* no CPU fallback: fails loudly */
        printf("test_abi: header and host-side entry points ok; no HIP device, compute skipped\n");
        return 77;
    }

    rw = wf_row_width(&p);
    for (i = 0; i < COLS; i++) {
        cols[i] = (uint64_t *)malloc(R * 8);
        polys[i] = (uint64_t *)malloc(R * 8);
        CHECK(cols[i] && polys[i]);
        for (j = 0; j < R; j++) { /* any residue below p is a valid Montgomery form */
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            cols[i][j] = x >= P ? x - P : x;
        }
        col_ptrs[i] = cols[i];
        poly_ptrs[i] = polys[i];
    }
    lde = (uint64_t *)malloc(wf_lde_bytes(&p));
    leaves = (uint8_t *)malloc(wf_digests_bytes(&p));
    nodes = (uint8_t *)malloc(wf_digests_bytes(&p));
    row = (uint64_t *)malloc(COLS * 8);
    CHECK(lde && leaves && nodes && row);
    lde_ptrs[0] = lde;
    memset(zero, 0, 32);

    CHECK(wf_ctx_create(0, &ctx) == WF_OK);
    rc = wf_trace_commit(ctx, &p, col_ptrs, poly_ptrs, lde_ptrs, leaves, nodes, root);
    CHECK(rc == WF_OK);
    CHECK(memcmp(nodes, zero, 32) == 0 && memcmp(nodes + 32, root, 32) == 0);
    for (j = 0; j < N; j++) /* padding lanes of every row are zero (segments.rs:65-72) */
        for (i = COLS; i < (int)rw; i++) CHECK(lde[(size_t)j * rw + i] == 0);

    CHECK(wf_trace_commit_resident(ctx, &p, col_ptrs, NULL, &com) == WF_OK);
    CHECK(wf_commitment_root(com, root2) == WF_OK && memcmp(root, root2, 32) == 0);
    CHECK(wf_commitment_info(com, &n_rows, &row_elems, &depth) == WF_OK);
    CHECK(n_rows == N && row_elems == COLS && depth == LOG_R + LOG_B);
    pos[0] = 777;
    CHECK(wf_commitment_read_rows(com, pos, 1, row) == WF_OK);
    CHECK(memcmp(row, lde + pos[0] * rw, COLS * 8) == 0);
    CHECK(wf_commitment_prove(com, pos[0], path) == WF_OK);
    CHECK(memcmp(path, leaves + pos[0] * 32, 32) == 0 && memcmp(path + 32, leaves + (pos[0] ^ 1) * 32, 32) == 0);
    pos[0] = N;
    CHECK(wf_commitment_read_rows(com, pos, 1, row) == WF_ERR_LEAVES);
    { /* a range of the resident matrix as RowMatrix::data holds it, and the DEEP composition of its polynomials */
        uint64_t width = 0, z = 12345, cc[COLS] = {3, 5, 7}, zero_z = 0;
        uint64_t *piece = (uint64_t *)malloc((size_t)64 * 8 * 8), *deep = (uint64_t *)malloc((size_t)R * 8),
                 *deep2 = (uint64_t *)malloc((size_t)R * 8);
        const wf_commitment *handles[1];
        CHECK(wf_commitment_read_lde(com, 0, 100, 64, piece, &width) == WF_OK && width == rw);
        CHECK(memcmp(piece, lde + 100 * rw, (size_t)64 * rw * 8) == 0);
        CHECK(wf_commitment_read_lde(com, 1, 0, 1, piece, NULL) == WF_ERR_TRACES);
        handles[0] = com;
        CHECK(wf_deep_compose(ctx, handles, 1, NULL, &z, 1, cc, NULL, deep, NULL, 0) == WF_OK);
        CHECK(deep[R - 1] == 0 && deep[R - 2] != 0); /* degree R - 2 (composer/mod.rs:151) */
        CHECK(wf_deep_compose(ctx, handles, 1, NULL, &z, 1, cc, NULL, deep2, NULL, 0) == WF_OK);
        CHECK(memcmp(deep, deep2, (size_t)R * 8) == 0);
        CHECK(wf_deep_compose(ctx, handles, 1, NULL, &zero_z, 1, cc, NULL, deep, NULL, 0) == WF_ERR_ARG);
        CHECK(wf_deep_compose(ctx, handles, 1, NULL, &z, 1, cc, NULL, NULL, NULL, 0) == WF_ERR_ARG);
        free(piece); free(deep); free(deep2);
    }
    wf_commitment_destroy(com);
    {   /* a stream of proofs (asynchronous form) and the second hasher, from plain C */
        wf_commitment *a[3] = {NULL, NULL, NULL};
        uint8_t r3[32], leaves24[2 * 24], nodes24[2 * 24];
        int i;
        for (i = 0; i < 3; i++) CHECK(wf_trace_commit_resident_async(ctx, &p, col_ptrs, &a[i]) == WF_OK);
        for (i = 0; i < 3; i++) {
            CHECK(wf_commitment_wait(a[i]) == WF_OK);
            CHECK(wf_commitment_root(a[i], r3) == WF_OK && memcmp(root, r3, 32) == 0);
            wf_commitment_destroy(a[i]);
        }
        CHECK(wf_commitment_wait(NULL) == WF_ERR_ARG);
        CHECK(wf_ctx_set_digest_bytes(ctx, 20) == WF_ERR_DIGEST && wf_ctx_set_digest_bytes(ctx, 24) == WF_OK);
        memset(leaves24, 7, sizeof(leaves24));
        CHECK(wf_merkle_build(ctx, leaves24, 2, nodes24) == WF_OK);   /* Blake3_192: 24-byte entries in and out */
        for (i = 0; i < 24; i++) CHECK(nodes24[i] == 0);               /* nodes[0] = Digest::default() */
        CHECK(wf_ctx_set_digest_bytes(ctx, 32) == WF_OK);
        p.digest_bytes = 24;
        CHECK(wf_params_check(&p, 0) == WF_OK && wf_digests_bytes(&p) == (size_t)N * 24);
        p.digest_bytes = 32;
    }

    p.n_cols = 0;
    CHECK(wf_trace_commit(ctx, &p, col_ptrs, NULL, NULL, NULL, NULL, root) == WF_ERR_WIDTH);
    wf_ctx_destroy(ctx);
    for (i = 0; i < COLS; i++) { free(cols[i]); free(polys[i]); }
    free(lde); free(leaves); free(nodes); free(row);
    printf("test_abi: ok\n");
    return 0;
}
