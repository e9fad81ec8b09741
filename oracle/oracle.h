/*
 * ORACLE (test infrastructure, NOT product code) -- exported functions of oracle/liboracle.so.
 * CPU restatement of the reference's LDE + BLAKE3 Merkle commitment path; every function cites the reference
 * lines it follows in oracle.c / fft_generic.inc.  f64 values are Montgomery residues (as the reference keeps them
 * in memory); f128 values are canonical little-endian u128.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_FIELD_F64 = 1, ORC_FIELD_F128 = 2 };

/* --- f64 scalar ops (math/src/field/f64/mod.rs) */
uint64_t orc_f64_new(uint64_t canonical);
uint64_t orc_f64_as_int(uint64_t mont);
uint64_t orc_f64_add(uint64_t a, uint64_t b);
uint64_t orc_f64_sub(uint64_t a, uint64_t b);
uint64_t orc_f64_mul(uint64_t a, uint64_t b);
uint64_t orc_f64_inv(uint64_t a);
uint64_t orc_f64_exp(uint64_t a, uint64_t e);
uint64_t orc_f64_get_root_of_unity(uint32_t n);

/* --- f128 scalar ops (math/src/field/f128/mod.rs); 16-byte little-endian operands by pointer */
void orc_f128_add(const void *a, const void *b, void *out);
void orc_f128_sub(const void *a, const void *b, void *out);
void orc_f128_mul(const void *a, const void *b, void *out);
void orc_f128_inv(const void *a, void *out);
void orc_f128_get_root_of_unity(uint32_t n, void *out);

/* --- math::fft over f64 (ext = coordinates per element) */
int orc_f64_get_twiddles(uint64_t *out, size_t n, int inverse);
void orc_f64_permute(uint64_t *v, size_t n, size_t ext);
void orc_f64_fft_in_place(uint64_t *v, size_t n, size_t ext, const uint64_t *tw);
void orc_f64_evaluate_poly(uint64_t *p, size_t n, size_t ext, const uint64_t *tw);
void orc_f64_evaluate_poly_with_offset(const uint64_t *p, size_t n, size_t ext, const uint64_t *tw,
                                       uint64_t domain_offset, size_t blowup, uint64_t *result);
void orc_f64_interpolate_poly(uint64_t *v, size_t n, size_t ext, const uint64_t *inv_tw);
void orc_f64_interpolate_poly_with_offset(uint64_t *v, size_t n, size_t ext, const uint64_t *inv_tw,
                                          uint64_t domain_offset);
void orc_f64_eval_many(const uint64_t *p, size_t n, const uint64_t *xs, size_t m, uint64_t *out);
void orc_f64_interpolate_columns(const uint64_t *const *cols, size_t n_cols, size_t ext, size_t R,
                                 uint64_t *const *out, int threads);
int orc_f64_evaluate_polys_over(const uint64_t *const *polys, size_t n_cols, size_t ext, size_t R, size_t blowup,
                                uint64_t domain_offset, uint64_t *out, int threads);

/* --- math::fft over f128 (pointer-typed wrappers; the elements are 16-byte LE integers) */
int orc_f128_get_twiddles_p(void *out, size_t n, int inverse);
void orc_f128_permute(unsigned __int128 *v, size_t n, size_t ext);
void orc_f128_fft_in_place(unsigned __int128 *v, size_t n, size_t ext, const unsigned __int128 *tw);
void orc_f128_evaluate_poly(unsigned __int128 *p, size_t n, size_t ext, const unsigned __int128 *tw);
void orc_f128_evaluate_poly_with_offset_p(const void *p, size_t n, size_t ext, const void *tw, const void *off,
                                          size_t blowup, void *result);
void orc_f128_interpolate_poly(unsigned __int128 *v, size_t n, size_t ext, const unsigned __int128 *inv_tw);
void orc_f128_interpolate_poly_with_offset_p(void *v, size_t n, size_t ext, const void *inv_tw, const void *off);
void orc_f128_eval_many(const unsigned __int128 *p, size_t n, const unsigned __int128 *xs, size_t m,
                        unsigned __int128 *out);
void orc_f128_interpolate_columns(const unsigned __int128 *const *cols, size_t n_cols, size_t ext, size_t R,
                                  unsigned __int128 *const *out, int threads);
int orc_f128_evaluate_polys_over_p(const void *const *polys, size_t n_cols, size_t ext, size_t R, size_t blowup,
                                   const void *off, void *out, int threads);

/* --- crypto::hash::Blake3_256 and crypto::merkle */
/* digest size of every hashing function below: 32 (Blake3_256, default) or 24 (Blake3_192); outputs and the leaf / node
 * arrays hold digests of that many bytes, that many bytes apart */
int orc_set_digest_bytes(int digest_bytes);
void orc_hash_elements(int field, const void *elems, size_t n_base, uint8_t *out);
void orc_merge(const uint8_t *left, const uint8_t *right, uint8_t *out);
void orc_merge_with_int(const uint8_t *seed, uint64_t value, uint8_t *out);
int orc_build_merkle_nodes(const uint8_t *leaves, size_t n_leaves, uint8_t *nodes, int threads);
int orc_commit_to_comb_rows(int field, const void *const *lde, size_t n_traces, size_t n_rows, size_t row_width,
                            size_t elements_per_row, uint8_t *leaves, uint8_t *nodes, int threads);

/* --- the path: Prover::build_trace_commitment / build_constraint_commitment (prover/src/lib.rs:615-715) */
int orc_build_trace_commitment(int field, size_t ext, unsigned log2_R, unsigned log2_blowup, size_t n_cols,
                               size_t n_traces, const uint8_t offset_le[16], const void *const *trace_cols,
                               void *const *polys_out, void *const *lde_out, uint8_t *leaves, uint8_t *nodes,
                               int threads);
int orc_build_constraint_commitment(int field, size_t ext, unsigned log2_R, unsigned log2_blowup, size_t n_cols,
                                    const uint8_t offset_le[16], const void *const *poly_cols, void *lde_out,
                                    uint8_t *leaves, uint8_t *nodes, int threads);

/* --- extension fields and FRI layer pieces (fri/src/prover/mod.rs:191-226, fri/src/folding/mod.rs:85-117) */
void orc_ext_mul(int field, size_t ext, const void *a, const void *b, void *out);
void orc_syn_div(int field, size_t ext, void *p, size_t n, const void *b);
void orc_transpose_slice(int field, const void *src, size_t n, size_t ext, size_t N, void *out);
void orc_apply_drp(int field, const void *values, size_t rows, size_t ext, size_t N, const uint8_t offset_le[16],
                   const void *alpha, void *out, int threads);
void orc_f64_transpose_slice(const uint64_t *src, size_t n, size_t ext, size_t N, uint64_t *out);
void orc_f64_apply_drp(const uint64_t *values, size_t rows, size_t ext, size_t N, uint64_t domain_offset,
                       const uint64_t *alpha, uint64_t *out, int threads);
void orc_f128_transpose_slice(const unsigned __int128 *src, size_t n, size_t ext, size_t N, unsigned __int128 *out);
void orc_f128_apply_drp(const unsigned __int128 *values, size_t rows, size_t ext, size_t N,
                        unsigned __int128 domain_offset, const unsigned __int128 *alpha, unsigned __int128 *out,
                        int threads);

/* --- out-of-domain evaluation (prover/src/trace/poly_table.rs:60-73, matrix/col_matrix.rs:249-254) */
void orc_eval_column_at(int field, const void *poly, size_t n, size_t ext_c, const void *z, size_t ext_z, void *out);
void orc_f64_eval_column_at(const uint64_t *poly, size_t n, size_t ext_c, const uint64_t *z, size_t ext_z,
                            uint64_t *out);
void orc_f128_eval_column_at(const unsigned __int128 *poly, size_t n, size_t ext_c, const unsigned __int128 *z,
                             size_t ext_z, unsigned __int128 *out);

/* --- DEEP composition polynomial (prover/src/composer/mod.rs:62-193) */
void orc_deep_compose(int field, size_t ext, size_t n, size_t n_tables, const size_t *cols_per_table, const void *const *cols,
                      const size_t *col_ext, const void *ood_z, const void *ood_zg, const void *cc_traces,
                      size_t n_constraint_cols, const void *const *constraint_cols, const void *ood_constraints,
                      const void *cc_constraints, const void *z, void *out);
void orc_f64_syn_div(uint64_t *p, size_t ext, size_t n, const uint64_t *b);
void orc_f128_syn_div(unsigned __int128 *p, size_t ext, size_t n, const unsigned __int128 *b);
void orc_f64_deep_compose(size_t ext, size_t n, size_t n_tables, const size_t *cols_per_table, const uint64_t *const *cols,
                          const size_t *col_ext, const uint64_t *ood_z, const uint64_t *ood_zg, const uint64_t *cc_traces,
                          size_t n_constraint_cols, const uint64_t *const *constraint_cols, const uint64_t *ood_constraints,
                          const uint64_t *cc_constraints, const uint64_t *z, uint64_t *out);
void orc_f128_deep_compose(size_t ext, size_t n, size_t n_tables, const size_t *cols_per_table,
                           const unsigned __int128 *const *cols, const size_t *col_ext, const unsigned __int128 *ood_z,
                           const unsigned __int128 *ood_zg, const unsigned __int128 *cc_traces, size_t n_constraint_cols,
                           const unsigned __int128 *const *constraint_cols, const unsigned __int128 *ood_constraints,
                           const unsigned __int128 *cc_constraints, const unsigned __int128 *z, unsigned __int128 *out);

/* --- STARKPack combination of composition polynomials (prover/src/lib.rs:442-453) */
void orc_scale_acc(int field, void *acc, const void *src, size_t ext, size_t n, const void *final_coeff, size_t power);
void orc_f64_scale_acc(uint64_t *acc, const uint64_t *src, size_t ext, size_t n, const uint64_t *final_coeff, size_t power);
void orc_f128_scale_acc(unsigned __int128 *acc, const unsigned __int128 *src, size_t ext, size_t n,
                        const unsigned __int128 *final_coeff, size_t power);

/* --- into_comb_poly's division by the divisors (prover/src/constraints/evaluation_table.rs:335-426); b and the exemptions are
 * base-field elements in memory representation */
void orc_acc_column(int field, const void *column, size_t ext, size_t ce, size_t a, const void *b, const void *exemptions, size_t n_ex,
                    const uint8_t offset_le[16], void *result);
void orc_f64_acc_column(const uint64_t *column, size_t ext, size_t ce, size_t a, uint64_t b, const uint64_t *exemptions, size_t n_ex,
                        uint64_t offset, uint64_t *result);
void orc_f128_acc_column(const unsigned __int128 *column, size_t ext, size_t ce, size_t a, unsigned __int128 b,
                         const unsigned __int128 *exemptions, size_t n_ex, unsigned __int128 offset, unsigned __int128 *result);

int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
/* phases of the last commitment of this process, ms: interpolate, evaluate, hash rows, tree (bench.py's record) */
void orc_last_phase_ms(double out[4]);
/* one BLAKE3 compression through the SIMD and the scalar form (test hook) */
void orc_blake3_compress_both(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t block_len,
                              uint32_t flags, uint32_t out_simd[8], uint32_t out_scalar[8]);

#endif
