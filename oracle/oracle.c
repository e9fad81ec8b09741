/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement of the reference hot path: trace / constraint low-degree extension + BLAKE3 Merkle commitment
 * (/root/reference/prover/src/lib.rs:615-715).  See oracle.h for the exported functions and fft_generic.inc,
 * field_f64.h, field_f128.h, blake3_ref.c for the pieces.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (starkpack-winterfell_amd/) never does.
 *
 * Pinning status: the reference cannot be built here (Rust, no toolchain) and its tests hold no literal
 * expected outputs for this path, so there are no reference-run vectors.  The oracle is pinned instead by
 * (1) every definitional / known-answer check the reference's tests make for the path, re-run on it
 * (tests/test_oracle_*.py), (2) independent Python big-integer arithmetic (oracle/pyref.py) and (3) golden
 * digests of the official BLAKE3 C implementation (tests/golden/, oracle/gen_golden.py).
 */
#include "oracle.h"

#include <omp.h>
#include <stdlib.h>
#include <string.h>

#include "blake3_ref.h"
#include "field_f128.h"
#include "field_f64.h"

/* ------------------------------------------------------------------ extension-field products
 * f64 quadratic  x^2 - x + 2 : math/src/field/f64/mod.rs:401-417
 * f64 cubic      x^3 - x - 1 : math/src/field/f64/mod.rs:446-472
 * f128 quadratic x^2 - x - 1 : math/src/field/f128/mod.rs:273-279 (no cubic extension: :296-314) */
static inline void f64_ext_mul(size_t ext, const uint64_t *a, const uint64_t *b, uint64_t *out) {
    if (ext == 1) {
        out[0] = f64_mul(a[0], b[0]);
    } else if (ext == 2) {
        uint64_t a0b0 = f64_mul(a[0], b[0]), a1b1 = f64_mul(a[1], b[1]);
        uint64_t o0 = f64_sub(a0b0, f64_add(a1b1, a1b1));
        uint64_t o1 = f64_sub(f64_mul(f64_add(a[0], a[1]), f64_add(b[0], b[1])), a0b0);
        out[0] = o0;
        out[1] = o1;
    } else {
        uint64_t a0b0 = f64_mul(a[0], b[0]), a1b1 = f64_mul(a[1], b[1]), a2b2 = f64_mul(a[2], b[2]);
        uint64_t s01 = f64_mul(f64_add(a[0], a[1]), f64_add(b[0], b[1]));
        uint64_t s02 = f64_mul(f64_add(a[0], a[2]), f64_add(b[0], b[2]));
        uint64_t s12 = f64_mul(f64_add(a[1], a[2]), f64_add(b[1], b[2]));
        uint64_t d01 = f64_sub(a0b0, a1b1);
        uint64_t o0 = f64_sub(f64_add(s12, d01), a2b2);
        uint64_t o1 = f64_sub(f64_sub(f64_add(s01, s12), f64_add(a1b1, a1b1)), a0b0);
        uint64_t o2 = f64_sub(s02, d01);
        out[0] = o0;
        out[1] = o1;
        out[2] = o2;
    }
}
static inline void f128_ext_mul(size_t ext, const f128e *a, const f128e *b, f128e *out) {
    if (ext == 1) {
        out[0] = f128_mul(a[0], b[0]);
    } else {
        f128e z = f128_mul(a[0], b[0]);
        f128e o0 = f128_add(z, f128_mul(a[1], b[1]));
        f128e o1 = f128_sub(f128_mul(f128_add(a[0], a[1]), f128_add(b[0], b[1])), z);
        out[0] = o0;
        out[1] = o1;
    }
}

/* ------------------------------------------------------------------ f64 instantiation */
#define FE uint64_t
#define EXT_MUL(ext, a, b, out) f64_ext_mul((ext), (a), (b), (out))
#define FN(x) orc_f64_##x
#define FE_ADD(a, b) f64_add((a), (b))
#define FE_SUB(a, b) f64_sub((a), (b))
#define FE_MUL(a, b) f64_mul((a), (b))
#define FE_ONE f64_new(1)
#define FE_ZERO ((uint64_t)0)
#define FE_FROM_U64(v) f64_new(v)
#define FE_INV(a) f64_inv(a)
#define FE_EXP_U64(b, e) f64_exp((b), (e))
#define FE_ROOT(n) f64_root_of_unity(n)
#define FE_TWO_ADICITY F64_TWO_ADICITY
#include "fft_generic.inc"
#undef FE
#undef FN
#undef FE_ADD
#undef FE_SUB
#undef FE_MUL
#undef FE_ONE
#undef FE_ZERO
#undef FE_FROM_U64
#undef FE_INV
#undef FE_EXP_U64
#undef FE_ROOT
#undef FE_TWO_ADICITY
#undef EXT_MUL

/* ------------------------------------------------------------------ f128 instantiation */
#define FE f128e
#define EXT_MUL(ext, a, b, out) f128_ext_mul((ext), (a), (b), (out))
#define FN(x) orc_f128_##x
#define FE_ADD(a, b) f128_add((a), (b))
#define FE_SUB(a, b) f128_sub((a), (b))
#define FE_MUL(a, b) f128_mul((a), (b))
#define FE_ONE ((f128e)1)
#define FE_ZERO ((f128e)0)
#define FE_FROM_U64(v) ((f128e)(v))
#define FE_INV(a) f128_inv(a)
#define FE_EXP_U64(b, e) f128_exp((b), (f128e)(e))
#define FE_ROOT(n) f128_root_of_unity(n)
#define FE_TWO_ADICITY F128_TWO_ADICITY
#include "fft_generic.inc"
#undef FE
#undef FN

/* ------------------------------------------------------------------ scalar field API (for the parity tests) */
uint64_t orc_f64_new(uint64_t v) { return f64_new(v); }
uint64_t orc_f64_as_int(uint64_t x) { return f64_as_int(x); }
uint64_t orc_f64_add(uint64_t a, uint64_t b) { return f64_add(a, b); }
uint64_t orc_f64_sub(uint64_t a, uint64_t b) { return f64_sub(a, b); }
uint64_t orc_f64_mul(uint64_t a, uint64_t b) { return f64_mul(a, b); }
uint64_t orc_f64_inv(uint64_t a) { return f64_inv(a); }
uint64_t orc_f64_exp(uint64_t a, uint64_t e) { return f64_exp(a, e); }
uint64_t orc_f64_get_root_of_unity(uint32_t n) { return f64_root_of_unity(n); }

void orc_f128_add(const void *a, const void *b, void *out) {
    f128e x, y;
    memcpy(&x, a, 16);
    memcpy(&y, b, 16);
    x = f128_add(x, y);
    memcpy(out, &x, 16);
}
void orc_f128_sub(const void *a, const void *b, void *out) {
    f128e x, y;
    memcpy(&x, a, 16);
    memcpy(&y, b, 16);
    x = f128_sub(x, y);
    memcpy(out, &x, 16);
}
void orc_f128_mul(const void *a, const void *b, void *out) {
    f128e x, y;
    memcpy(&x, a, 16);
    memcpy(&y, b, 16);
    x = f128_mul(x, y);
    memcpy(out, &x, 16);
}
void orc_f128_inv(const void *a, void *out) {
    f128e x;
    memcpy(&x, a, 16);
    x = f128_inv(x);
    memcpy(out, &x, 16);
}
void orc_f128_get_root_of_unity(uint32_t n, void *out) {
    f128e x = f128_root_of_unity(n);
    memcpy(out, &x, 16);
}

/* ------------------------------------------------------------------ hashing */

/* Blake3_256::hash_elements, crypto/src/hash/blake/mod.rs:46-59.
 * f128 (IS_CANONICAL): raw element bytes.  f64: every element's canonical as_int() as 8 LE bytes
 * (f64/mod.rs:605-610 via utils/core/src/serde/mod.rs:38-42,82-86; no length prefix). */
/* Digest size of the hasher: 32 = Blake3_256 (blake/mod.rs:20-59), 24 = Blake3_192 (blake/mod.rs:68-114: the same BLAKE3
 * output truncated to its first 24 bytes; merge hashes the 48 bytes of two digests; leaves / nodes are arrays of
 * ByteDigest<24>, 24 bytes apart).  Process-wide setting of the checker (tests select it around a call). */
static size_t g_db = 32;
int orc_set_digest_bytes(int db) {
    if (db != 24 && db != 32) return -1;
    g_db = (size_t)db;
    return 0;
}
static void hash_trunc(const uint8_t *in, size_t len, uint8_t *out) {
    uint8_t full[32];
    orc_blake3_hash(in, len, full);
    memcpy(out, full, g_db);
}

void orc_hash_elements(int field, const void *elems, size_t n_base, uint8_t *out) {
    if (field == ORC_FIELD_F128) {
        hash_trunc((const uint8_t *)elems, n_base * 16, out);
        return;
    }
    const uint64_t *e = (const uint64_t *)elems;
    uint8_t stackbuf[2048];
    uint8_t *buf = n_base * 8 <= sizeof(stackbuf) ? stackbuf : (uint8_t *)malloc(n_base * 8);
    for (size_t i = 0; i < n_base; i++) {
        uint64_t v = f64_as_int(e[i]);
        for (int b = 0; b < 8; b++) buf[8 * i + b] = (uint8_t)(v >> (8 * b));
    }
    hash_trunc(buf, n_base * 8, out);
    if (buf != stackbuf) free(buf);
}

/* Blake3_256::merge, blake/mod.rs:31-33 */
void orc_merge(const uint8_t *left, const uint8_t *right, uint8_t *out) {
    uint8_t buf[64];
    memcpy(buf, left, g_db);
    memcpy(buf + g_db, right, g_db);
    hash_trunc(buf, 2 * g_db, out);
}

/* Blake3_256::merge_with_int, blake/mod.rs:35-40 */
void orc_merge_with_int(const uint8_t *seed, uint64_t value, uint8_t *out) {
    uint8_t buf[40];
    memcpy(buf, seed, g_db);
    for (int b = 0; b < 8; b++) buf[g_db + b] = (uint8_t)(value >> (8 * b));
    hash_trunc(buf, g_db + 8, out);
}

/* build_merkle_nodes, crypto/src/merkle/mod.rs:350-374 (threads<=1) and merkle/concurrent.rs:21-70 (threads>1: the first
 * row of internal nodes in parallel, then one SUB-TREE per batch -- num_subtrees = threads.next_power_of_two() -- each walked
 * from its widest level to its sub-root by one thread (:44-62), then the tip of the tree serially (:65-67); below
 * MIN_CONCURRENT_LEAVES = 1024 leaves the reference takes the serial form, merkle/mod.rs:126-130).
 * nodes has n_leaves digests: nodes[0] = zero digest, nodes[1] = root. */
int orc_build_merkle_nodes(const uint8_t *leaves, size_t n_leaves, uint8_t *nodes, int threads) {
    if (n_leaves < 2) return -1;                    /* merkle/mod.rs:118-120 */
    if (n_leaves & (n_leaves - 1)) return -2;       /* :121-123 */
    size_t n = n_leaves / 2;
    const size_t db = g_db; /* two adjacent digests ARE the merge input (merkle/mod.rs:350-374: merge(&[l, r])) */
    memset(nodes, 0, db);
    size_t num_subtrees = 1;
    while (num_subtrees < (size_t)(threads > 1 ? threads : 1)) num_subtrees *= 2;
    if (threads <= 1 || n_leaves < 1024 || n / num_subtrees < 2) {
        for (size_t i = 0; i < n; i++) hash_trunc(leaves + 2 * db * i, 2 * db, nodes + db * (n + i));
        for (size_t i = n - 1; i >= 1; i--) hash_trunc(nodes + 2 * db * i, 2 * db, nodes + db * i);
        return 0;
    }
#pragma omp parallel for num_threads(threads) schedule(static)
    for (size_t i = 0; i < n; i++) hash_trunc(leaves + 2 * db * i, 2 * db, nodes + db * (n + i));
    const size_t batch0 = n / num_subtrees;
#pragma omp parallel for num_threads(threads) schedule(static, 1)
    for (size_t s = 0; s < num_subtrees; s++) {
        size_t batch = batch0 / 2, start = n / 2 + batch * s;
        while (start >= num_subtrees) {
            for (size_t k = start + batch; k-- > start;) hash_trunc(nodes + 2 * db * k, 2 * db, nodes + db * k);
            start /= 2;
            batch /= 2;
        }
    }
    for (size_t i = num_subtrees - 1; i >= 1; i--) hash_trunc(nodes + 2 * db * i, 2 * db, nodes + db * i);
    return 0;
}

/* Wall clock of the phases of the last orc_build_trace_commitment / orc_build_constraint_commitment of this process
 * (the reference logs the same phases with debug!, prover/src/lib.rs:628-667): interpolate, evaluate, hash rows, tree.
 * For bench.py's cpu_baseline record; not thread-safe across concurrent commitments. */
static double g_phase_ms[4];
static double now_ms(void) { return omp_get_wtime() * 1e3; }
void orc_last_phase_ms(double out[4]) { memcpy(out, g_phase_ms, sizeof(g_phase_ms)); }

/* RowMatrix::commit_to_comb_rows (row_matrix.rs:204-238); with n_traces == 1 it is commit_to_rows (:183-203).
 * lde[t]: row-major matrices of n_rows x row_width base values; a row contributes its first elements_per_row. */
int orc_commit_to_comb_rows(int field, const void *const *lde, size_t n_traces, size_t n_rows, size_t row_width,
                            size_t elements_per_row, uint8_t *leaves, uint8_t *nodes, int threads) {
    size_t eb = field == ORC_FIELD_F128 ? 16 : 8;
    if (threads < 1) threads = 1;
    int err = 0;
    const double t_hash0 = now_ms();
#pragma omp parallel num_threads(threads) if (threads > 1)
    {
        uint8_t *comb = (uint8_t *)malloc(n_traces * elements_per_row * eb);
        if (!comb) err = 1;
#pragma omp for schedule(static)
        for (size_t i = 0; i < n_rows; i++) {
            if (!comb) continue;
            for (size_t t = 0; t < n_traces; t++)
                memcpy(comb + t * elements_per_row * eb, (const uint8_t *)lde[t] + i * row_width * eb,
                       elements_per_row * eb);
            orc_hash_elements(field, comb, n_traces * elements_per_row, leaves + g_db * i);
        }
        free(comb);
    }
    if (err) return -3;
    const double t_tree0 = now_ms();
    g_phase_ms[2] = t_tree0 - t_hash0;
    int rc = orc_build_merkle_nodes(leaves, n_rows, nodes, threads);
    g_phase_ms[3] = now_ms() - t_tree0;
    return rc;
}

/* ------------------------------------------------------------------ the path */

static int check_params(int field, size_t ext, unsigned log2_R, unsigned log2_blowup, size_t n_cols) {
    if (field != ORC_FIELD_F64 && field != ORC_FIELD_F128) return -10;
    if (ext < 1 || ext > 3 || (field == ORC_FIELD_F128 && ext == 3)) return -11; /* f128/mod.rs:296-314 */
    if (log2_R < 3) return -12;                           /* air/src/air/trace_info.rs:35 */
    if (log2_blowup < 1 || log2_blowup > 7) return -13;   /* air/src/options.rs:19-20 */
    unsigned adicity = field == ORC_FIELD_F64 ? F64_TWO_ADICITY : F128_TWO_ADICITY;
    if (log2_R + log2_blowup > adicity) return -14;       /* fft/mod.rs:196-200 */
    if (n_cols < 1 || n_cols > 255) return -15;           /* trace_info.rs:37 */
    return 0;
}

/* Prover::build_trace_commitment, prover/src/lib.rs:615-670.
 * trace_cols[t*n_cols + c]: R elements (ext coordinates each).  polys_out likewise.  lde_out[t]: N x row_width. */
int orc_build_trace_commitment(int field, size_t ext, unsigned log2_R, unsigned log2_blowup, size_t n_cols,
                               size_t n_traces, const uint8_t offset_le[16], const void *const *trace_cols,
                               void *const *polys_out, void *const *lde_out, uint8_t *leaves, uint8_t *nodes,
                               int threads) {
    int rc = check_params(field, ext, log2_R, log2_blowup, n_cols);
    if (rc) return rc;
    if (n_traces < 1) return -16;
    size_t R = (size_t)1 << log2_R, blowup = (size_t)1 << log2_blowup;
    size_t base_cols = n_cols * ext, row_width = 8 * ((base_cols + 7) / 8);
    f128e off128;
    memcpy(&off128, offset_le, 16);
    if (off128 == 0) return -17; /* fft/mod.rs:201 */
    g_phase_ms[0] = g_phase_ms[1] = 0.0;
    for (size_t t = 0; t < n_traces; t++) {
        const double t_int0 = now_ms();
        if (field == ORC_FIELD_F64) {
            orc_f64_interpolate_columns((const uint64_t *const *)(trace_cols + t * n_cols), n_cols, ext, R,
                                        (uint64_t *const *)(polys_out + t * n_cols), threads);
            g_phase_ms[0] += now_ms() - t_int0;
            rc = orc_f64_evaluate_polys_over((const uint64_t *const *)(polys_out + t * n_cols), n_cols, ext, R, blowup,
                                             f64_new((uint64_t)off128), (uint64_t *)lde_out[t], threads);
        } else {
            orc_f128_interpolate_columns((const f128e *const *)(trace_cols + t * n_cols), n_cols, ext, R,
                                         (f128e *const *)(polys_out + t * n_cols), threads);
            g_phase_ms[0] += now_ms() - t_int0;
            rc = orc_f128_evaluate_polys_over((const f128e *const *)(polys_out + t * n_cols), n_cols, ext, R, blowup,
                                              off128, (f128e *)lde_out[t], threads);
        }
        if (rc) return rc;
        g_phase_ms[1] += now_ms() - t_int0;
    }
    g_phase_ms[1] -= g_phase_ms[0];
    return orc_commit_to_comb_rows(field, (const void *const *)lde_out, n_traces, R * blowup, row_width, base_cols,
                                   leaves, nodes, threads);
}

/* Prover::build_constraint_commitment, prover/src/lib.rs:680-715: composition-poly columns (coefficients) ->
 * row-major LDE -> commit_to_rows. */
int orc_build_constraint_commitment(int field, size_t ext, unsigned log2_R, unsigned log2_blowup, size_t n_cols,
                                    const uint8_t offset_le[16], const void *const *poly_cols, void *lde_out,
                                    uint8_t *leaves, uint8_t *nodes, int threads) {
    int rc = check_params(field, ext, log2_R, log2_blowup, n_cols);
    if (rc) return rc;
    size_t R = (size_t)1 << log2_R, blowup = (size_t)1 << log2_blowup;
    size_t base_cols = n_cols * ext, row_width = 8 * ((base_cols + 7) / 8);
    f128e off128;
    memcpy(&off128, offset_le, 16);
    if (off128 == 0) return -17;
    if (field == ORC_FIELD_F64)
        rc = orc_f64_evaluate_polys_over((const uint64_t *const *)poly_cols, n_cols, ext, R, blowup,
                                         f64_new((uint64_t)off128), (uint64_t *)lde_out, threads);
    else
        rc = orc_f128_evaluate_polys_over((const f128e *const *)poly_cols, n_cols, ext, R, blowup, off128,
                                          (f128e *)lde_out, threads);
    if (rc) return rc;
    const void *l = lde_out;
    return orc_commit_to_comb_rows(field, &l, 1, R * blowup, row_width, base_cols, leaves, nodes, threads);
}

/* ------------------------------------------------------------------ ctypes-friendly wrappers (u128 by pointer) */
int orc_f128_get_twiddles_p(void *out, size_t n, int inverse) { return orc_f128_get_twiddles((f128e *)out, n, inverse); }
void orc_f128_evaluate_poly_with_offset_p(const void *p, size_t n, size_t ext, const void *tw, const void *off,
                                          size_t blowup, void *result) {
    f128e o;
    memcpy(&o, off, 16);
    orc_f128_evaluate_poly_with_offset((const f128e *)p, n, ext, (const f128e *)tw, o, blowup, (f128e *)result);
}
void orc_f128_interpolate_poly_with_offset_p(void *v, size_t n, size_t ext, const void *inv_tw, const void *off) {
    f128e o;
    memcpy(&o, off, 16);
    orc_f128_interpolate_poly_with_offset((f128e *)v, n, ext, (const f128e *)inv_tw, o);
}
int orc_f128_evaluate_polys_over_p(const void *const *polys, size_t n_cols, size_t ext, size_t R, size_t blowup,
                                   const void *off, void *out, int threads) {
    f128e o;
    memcpy(&o, off, 16);
    return orc_f128_evaluate_polys_over((const f128e *const *)polys, n_cols, ext, R, blowup, o, (f128e *)out, threads);
}

/* extension products and FRI helpers, pointer-typed for ctypes */
void orc_ext_mul(int field, size_t ext, const void *a, const void *b, void *out) {
    if (field == ORC_FIELD_F64)
        f64_ext_mul(ext, (const uint64_t *)a, (const uint64_t *)b, (uint64_t *)out);
    else
        f128_ext_mul(ext, (const f128e *)a, (const f128e *)b, (f128e *)out);
}
/* polynom::syn_div_in_place(p, 1, b) in place on n coefficients of `ext` coordinates */
void orc_syn_div(int field, size_t ext, void *p, size_t n, const void *b) {
    if (field == ORC_FIELD_F64)
        orc_f64_syn_div((uint64_t *)p, ext, n, (const uint64_t *)b);
    else
        orc_f128_syn_div((f128e *)p, ext, n, (const f128e *)b);
}
void orc_transpose_slice(int field, const void *src, size_t n, size_t ext, size_t N, void *out) {
    if (field == ORC_FIELD_F64)
        orc_f64_transpose_slice((const uint64_t *)src, n, ext, N, (uint64_t *)out);
    else
        orc_f128_transpose_slice((const f128e *)src, n, ext, N, (f128e *)out);
}
void orc_apply_drp(int field, const void *values, size_t rows, size_t ext, size_t N, const uint8_t offset_le[16],
                   const void *alpha, void *out, int threads) {
    f128e off;
    memcpy(&off, offset_le, 16);
    if (field == ORC_FIELD_F64)
        orc_f64_apply_drp((const uint64_t *)values, rows, ext, N, f64_new((uint64_t)off), (const uint64_t *)alpha,
                          (uint64_t *)out, threads);
    else
        orc_f128_apply_drp((const f128e *)values, rows, ext, N, off, (const f128e *)alpha, (f128e *)out, threads);
}

void orc_eval_column_at(int field, const void *poly, size_t n, size_t ext_c, const void *z, size_t ext_z, void *out) {
    if (field == ORC_FIELD_F64)
        orc_f64_eval_column_at((const uint64_t *)poly, n, ext_c, (const uint64_t *)z, ext_z, (uint64_t *)out);
    else
        orc_f128_eval_column_at((const f128e *)poly, n, ext_c, (const f128e *)z, ext_z, (f128e *)out);
}

void orc_deep_compose(int field, size_t ext, size_t n, size_t n_tables, const size_t *cols_per_table, const void *const *cols,
                      const size_t *col_ext, const void *ood_z, const void *ood_zg, const void *cc_traces,
                      size_t n_constraint_cols, const void *const *constraint_cols, const void *ood_constraints,
                      const void *cc_constraints, const void *z, void *out) {
    if (field == ORC_FIELD_F64)
        orc_f64_deep_compose(ext, n, n_tables, cols_per_table, (const uint64_t *const *)cols, col_ext, (const uint64_t *)ood_z,
                             (const uint64_t *)ood_zg, (const uint64_t *)cc_traces, n_constraint_cols,
                             (const uint64_t *const *)constraint_cols, (const uint64_t *)ood_constraints,
                             (const uint64_t *)cc_constraints, (const uint64_t *)z, (uint64_t *)out);
    else
        orc_f128_deep_compose(ext, n, n_tables, cols_per_table, (const f128e *const *)cols, col_ext, (const f128e *)ood_z,
                              (const f128e *)ood_zg, (const f128e *)cc_traces, n_constraint_cols,
                              (const f128e *const *)constraint_cols, (const f128e *)ood_constraints,
                              (const f128e *)cc_constraints, (const f128e *)z, (f128e *)out);
}

void orc_scale_acc(int field, void *acc, const void *src, size_t ext, size_t n, const void *final_coeff, size_t power) {
    if (field == ORC_FIELD_F64)
        orc_f64_scale_acc((uint64_t *)acc, (const uint64_t *)src, ext, n, (const uint64_t *)final_coeff, power);
    else
        orc_f128_scale_acc((f128e *)acc, (const f128e *)src, ext, n, (const f128e *)final_coeff, power);
}

void orc_acc_column(int field, const void *column, size_t ext, size_t ce, size_t a, const void *b, const void *exemptions, size_t n_ex,
                    const uint8_t offset_le[16], void *result) {
    f128e off;
    memcpy(&off, offset_le, 16);
    if (field == ORC_FIELD_F64) {
        orc_f64_acc_column((const uint64_t *)column, ext, ce, a, *(const uint64_t *)b, (const uint64_t *)exemptions, n_ex,
                           f64_new((uint64_t)off), (uint64_t *)result);
    } else {
        f128e bb;
        memcpy(&bb, b, 16);
        orc_f128_acc_column((const f128e *)column, ext, ce, a, bb, (const f128e *)exemptions, n_ex, off, (f128e *)result);
    }
}

int orc_max_threads(void) { return omp_get_max_threads(); }
