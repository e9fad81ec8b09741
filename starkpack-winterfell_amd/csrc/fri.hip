// libwf_lde.so, unit 4 of 6 -- FRI: layer commitments, the degree-respecting projection, and the resident
// FRI prover (FriProver::build_layers / build_layer / set_remainder, /root/reference/fri/src/prover/mod.rs:172-227).
#include "wf_internal.hpp"

#include "kernels.hpp"
#include "fri_kernels.hpp"
#include "tables.hpp"

using namespace wf;

static int check_fri_args(wf_ctx *ctx, uint32_t field, uint32_t ext, size_t n, uint32_t folding, uint32_t *logn) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    if (field != WF_FIELD_F64 && field != WF_FIELD_F128) return fail(WF_ERR_FIELD, "unknown field id %u", field);
    if (ext < 1 || ext > 3 || (field == WF_FIELD_F128 && ext == 3)) return fail(WF_ERR_EXTENSION, "unsupported extension degree %u", ext);
    if (folding != 2 && folding != 4 && folding != 8 && folding != 16)
        return fail(WF_ERR_ARG, "folding factor %u is not supported", folding);  // fri/src/prover/mod.rs:178-185
    if (n < 2 * (size_t)folding || (n & (n - 1))) return fail(WF_ERR_TRACE_LENGTH, "domain size must be a power of two >= 2 * folding factor");
    uint32_t l = 0;
    while (((size_t)1 << l) < n) l++;
    const uint32_t adicity = field == WF_FIELD_F64 ? F64::TWO_ADICITY : F128::TWO_ADICITY;
    if (l > adicity) return fail(WF_ERR_DOMAIN, "no multiplicative subgroup of size 2^%u in this field", l);
    *logn = l;
    return 0;
}

template <class F>
static int fri_layer_commit_dev(wf_ctx *ctx, hipStream_t st, uint32_t ext, const void *d_evals, size_t n,
                                uint32_t folding, void *d_transposed, void *d_leaves, void *d_nodes) {
    typedef typename F::T T;
    const uint64_t rows = n / folding;
    prof_mark(ctx, st, "fri.transpose");
    hipLaunchKernelGGL(k_fri_transpose<F>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (const T *)d_evals,
                       (T *)d_transposed, rows, folding, ext);
    HIP_TRY(hipGetLastError());
    prof_mark(ctx, st, "fri.hash_values");
    int rc = path_hash_rows(ctx, st, F::FIELD_ID == 1 ? WF_FIELD_F64 : WF_FIELD_F128, d_transposed, 0, rows, folding * ext, folding * ext, 1, d_leaves, ctx->digest_bytes);
    if (rc) return rc;
    prof_mark(ctx, st, "fri.merkle");
    rc = path_merkle(ctx, st, d_leaves, rows, d_nodes, ctx->digest_bytes);
    prof_mark(ctx, st, "between_calls");
    return rc;
}

template <class F, int W>
static int fri_drp_launch(hipStream_t st, uint32_t folding, const DrpArgs<F> &a) {
    const dim3 grid((uint32_t)((a.rows + 127) / 128)), block(128);
    switch (folding) {
        case 2: hipLaunchKernelGGL((k_fri_drp<F, W, 2>), grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL((k_fri_drp<F, W, 4>), grid, block, 0, st, a); break;
        case 8: hipLaunchKernelGGL((k_fri_drp<F, W, 8>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((k_fri_drp<F, W, 16>), grid, block, 0, st, a); break;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

template <class F>
static int fri_apply_drp_dev(wf_ctx *ctx, hipStream_t st, uint32_t ext, const void *d_transposed, size_t rows,
                             uint32_t folding, const uint8_t offset16[16], const void *alpha_host, void *d_out) {
    typedef typename F::T T;
    u128 off;
    memcpy(&off, offset16, 16);
    if (off == 0 || off >= FieldInfo<F>::modulus()) return fail(WF_ERR_OFFSET, "domain offset must be a non-zero field element");
    uint32_t logn = 0, logf = 0;
    while (((size_t)1 << logn) < rows * folding) logn++;
    while ((1u << logf) < folding) logf++;
    TableSet *ginv;
    int rc = root_tables<F>(ctx, logn, true, &ginv);
    if (rc) return rc;
    DrpArgs<F> a;
    memset(&a, 0, sizeof(a));
    a.values = (const T *)d_transposed;
    a.out = (T *)d_out;
    a.rows = rows;
    a.ginv = as_pow2l<F>(*ginv);
    rc = digit_table<F>(ctx, logf, true, &a.tw);
    if (rc) return rc;
    a.sinv = f_inv<F>(F::from_u128_canonical(off));
    a.ninv = f_inv<F>(F::from_u128_canonical((u128)folding));
    memcpy(a.alpha, alpha_host, ext * sizeof(T));
    for (uint32_t w = 0; w < ext; w++)
        if (!F::is_valid(a.alpha[w])) return fail(WF_ERR_ARG, "alpha is not a valid field element");
    prof_mark(ctx, st, "fri.apply_drp");
    if (ext == 1) rc = fri_drp_launch<F, 1>(st, folding, a);
    else if (ext == 2) rc = fri_drp_launch<F, 2>(st, folding, a);
    else {
        if constexpr (F::FIELD_ID == 1) rc = fri_drp_launch<F, 3>(st, folding, a);
        else rc = fail(WF_ERR_EXTENSION, "f128 has no cubic extension");
    }
    prof_mark(ctx, st, "between_calls");
    return rc;
}

extern "C" {

int wf_fri_layer_commit_dev(wf_ctx *ctx, uint32_t field, uint32_t ext, const void *d_evals, size_t n, uint32_t folding,
                            void *d_transposed, void *d_leaves, void *d_nodes, void *stream) {
    uint32_t l;
    int rc = check_fri_args(ctx, field, ext, n, folding, &l);
    if (rc) return rc;
    if (!d_evals || !d_transposed || !d_leaves || !d_nodes) return fail(WF_ERR_ARG, "null device buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    WF_ENTER(ctx, st);
    return field == WF_FIELD_F64 ? fri_layer_commit_dev<F64>(ctx, st, ext, d_evals, n, folding, d_transposed, d_leaves, d_nodes)
                                 : fri_layer_commit_dev<F128>(ctx, st, ext, d_evals, n, folding, d_transposed, d_leaves, d_nodes);
}

int wf_fri_apply_drp_dev(wf_ctx *ctx, uint32_t field, uint32_t ext, const void *d_transposed, size_t rows,
                         uint32_t folding, const uint8_t domain_offset[16], const void *alpha, void *d_out,
                         void *stream) {
    uint32_t l;
    int rc = check_fri_args(ctx, field, ext, rows * folding, folding, &l);
    if (rc) return rc;
    if (!d_transposed || !d_out || !alpha || !domain_offset) return fail(WF_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    WF_ENTER(ctx, st);
    return field == WF_FIELD_F64 ? fri_apply_drp_dev<F64>(ctx, st, ext, d_transposed, rows, folding, domain_offset, alpha, d_out)
                                 : fri_apply_drp_dev<F128>(ctx, st, ext, d_transposed, rows, folding, domain_offset, alpha, d_out);
}

int wf_fri_layer_commit(wf_ctx *ctx, uint32_t field, uint32_t ext, const void *evals, size_t n, uint32_t folding,
                        void *transposed_out, uint8_t *leaves_out, uint8_t *nodes_out, uint8_t *root_out) {
    uint32_t l;
    int rc = check_fri_args(ctx, field, ext, n, folding, &l);
    if (rc) return rc;
    if (!evals) return fail(WF_ERR_ARG, "evaluations pointer is null");
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    const size_t bytes = n * ext * wf_elem_bytes(field), rows = n / folding;
    if ((rc = ensure(ctx, ctx->io[0], bytes))) return rc;
    if ((rc = ensure(ctx, ctx->io[2], bytes))) return rc;
    if ((rc = ensure(ctx, ctx->io[3], rows * 32))) return rc;
    if ((rc = ensure(ctx, ctx->io[4], rows * 32))) return rc;
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(ctx->io[0].p, evals, bytes, hipMemcpyHostToDevice, st));
    rc = wf_fri_layer_commit_dev(ctx, field, ext, ctx->io[0].p, n, folding, ctx->io[2].p, ctx->io[3].p, ctx->io[4].p, st);
    if (rc) return rc;
    if (transposed_out) HIP_TRY(hipMemcpyAsync(transposed_out, ctx->io[2].p, bytes, hipMemcpyDeviceToHost, st));
    if (leaves_out) {
        if ((rc = path_digests_to_host(ctx, st, ctx->io[3].p, leaves_out, rows, ctx->digest_bytes))) return rc;
        if (ctx->digest_bytes != 32) HIP_TRY(hipStreamSynchronize(st));  // (one packing buffer for both arrays)
    }
    if (nodes_out && (rc = path_digests_to_host(ctx, st, ctx->io[4].p, nodes_out, rows, ctx->digest_bytes))) return rc;
    if (root_out) HIP_TRY(hipMemcpyAsync(root_out, (char *)ctx->io[4].p + 32, 32, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int wf_fri_apply_drp(wf_ctx *ctx, uint32_t field, uint32_t ext, const void *transposed, size_t rows, uint32_t folding,
                     const uint8_t domain_offset[16], const void *alpha, void *out) {
    uint32_t l;
    int rc = check_fri_args(ctx, field, ext, rows * folding, folding, &l);
    if (rc) return rc;
    if (!transposed || !out) return fail(WF_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    const size_t eb = ext * wf_elem_bytes(field);
    if ((rc = ensure(ctx, ctx->io[0], rows * folding * eb))) return rc;
    if ((rc = ensure(ctx, ctx->io[1], rows * eb))) return rc;
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(ctx->io[0].p, transposed, rows * folding * eb, hipMemcpyHostToDevice, st));
    rc = wf_fri_apply_drp_dev(ctx, field, ext, ctx->io[0].p, rows, folding, domain_offset, alpha, ctx->io[1].p, st);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, ctx->io[1].p, rows * eb, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------- resident FRI prover
// (struct wf_fri_prover: wf_internal.hpp)
static void *fri_arena_take(wf_fri_prover *pr, size_t bytes) {
    const size_t off = (pr->arena_used + 255) & ~(size_t)255;
    if (!pr->arena.p || off + bytes > pr->arena.cap) return nullptr;
    pr->arena_used = off + bytes;
    return (char *)pr->arena.p + off;
}

// room for a whole proof over n evaluations: the first layer's evaluations, then per layer the transposed values, leaves,
// nodes and folded evaluations (a hipMalloc / hipFree pair for the 128 MiB of a 2^23-point first layer cost 0.2 ms per proof)
static int fri_arena_reserve(wf_fri_prover *pr, size_t n) {
    const size_t eb = (size_t)pr->ext * wf_elem_bytes(pr->field);
    size_t total = n * eb + 256;
    for (size_t m = n; m >= pr->folding; m /= pr->folding) {
        const size_t rows = m / pr->folding;
        total += (m * eb + 256) + 2 * (rows * 32 + 256) + (rows * eb + 256);
    }
    pr->arena_used = 0;
    return ensure(pr->ctx, pr->arena, total);
}

// the first layer's evaluation buffer: from the arena (right after fri_arena_reserve), else an allocation of its own
static int fri_take_evals(wf_fri_prover *pr, size_t bytes) {
    pr->evals = fri_arena_take(pr, bytes);
    pr->evals_borrowed = pr->evals != nullptr;
    if (!pr->evals) {
        hipError_t e = dev_malloc(pr->ctx, &pr->evals, bytes);
        if (e != hipSuccess) {
            pr->evals = nullptr;
            return fail(WF_ERR_HIP, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        }
    }
    return 0;
}
static void fri_drop_evals(wf_fri_prover *pr) {
    if (pr->evals && !pr->evals_borrowed) (void)hipFree(pr->evals);
    pr->evals = nullptr;
    pr->evals_borrowed = false;
}

static void fri_prover_clear(wf_fri_prover *pr) {
    if (ctx_alive(pr->ctx, pr->ctx_generation)) {
        (void)hipSetDevice(pr->ctx->device);
        (void)hipStreamSynchronize(pr->ctx->stream);
    }
    if (pr->evals && !pr->evals_borrowed) (void)hipFree(pr->evals);
    pr->evals = nullptr;
    pr->evals_borrowed = false;
    pr->arena_used = 0;
    pr->n = 0;
    free_commitment(pr->pending);
    pr->pending = nullptr;
    for (wf_commitment *c : pr->layers) free_commitment(c);
    pr->layers.clear();
}

extern "C" {

size_t wf_fri_num_layers(uint32_t folding, uint32_t blowup, uint32_t remainder_max_degree, size_t domain_size) {
    if (folding < 2) return 0;
    size_t result = 0;
    const size_t max_remainder_size = ((size_t)remainder_max_degree + 1) * blowup;  // fri/src/options.rs:87
    while (domain_size > max_remainder_size) {
        domain_size /= folding;
        result++;
    }
    return result;
}

int wf_fri_fold_positions(const uint64_t *positions, size_t n, size_t source_domain_size, uint32_t folding, uint64_t *out,
                          size_t *n_out) {
    if (!positions || !out || !n_out) return fail(WF_ERR_ARG, "null argument");
    if (folding == 0 || source_domain_size < folding) return fail(WF_ERR_ARG, "invalid domain size / folding factor");
    const size_t target = source_domain_size / folding;
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        const uint64_t pos = positions[i] % target;
        bool seen = false;
        for (size_t j = 0; j < m && !seen; j++) seen = out[j] == pos;
        if (!seen) out[m++] = pos;
    }
    *n_out = m;
    return 0;
}

int wf_fri_prover_create(wf_ctx *ctx, uint32_t field, uint32_t ext, uint32_t folding, uint32_t blowup,
                         uint32_t remainder_max_degree, const uint8_t domain_offset[16], wf_fri_prover **out) {
    if (!out) return fail(WF_ERR_ARG, "out is null");
    uint32_t l;
    int rc = check_fri_args(ctx, field, ext, 2 * (size_t)16, folding, &l);  // field / extension / folding factor
    if (rc) return rc;
    if (blowup < 1 || (blowup & (blowup - 1))) return fail(WF_ERR_BLOWUP, "blowup factor must be a power of two");
    if (!domain_offset) return fail(WF_ERR_ARG, "domain offset is null");
    u128 off;
    memcpy(&off, domain_offset, 16);
    const u128 mod = field == WF_FIELD_F64 ? (u128)F64::P : F128::P();
    if (off == 0 || off >= mod) return fail(WF_ERR_OFFSET, "domain offset must be a non-zero field element");
    wf_fri_prover *pr = new wf_fri_prover();
    pr->ctx = ctx;
    pr->ctx_generation = ctx->generation;
    pr->field = field;
    pr->ext = ext;
    pr->folding = folding;
    pr->blowup = blowup;
    pr->remainder_max_degree = remainder_max_degree;
    memcpy(pr->offset, domain_offset, 16);
    *out = pr;
    return 0;
}

void wf_fri_prover_destroy(wf_fri_prover *pr) {
    if (!pr) return;
    fri_prover_clear(pr);
    if (pr->arena.p) (void)hipFree(pr->arena.p);
    delete pr;
}

int wf_fri_prover_reset(wf_fri_prover *pr) {
    if (!pr) return fail(WF_ERR_ARG, "prover is null");
    fri_prover_clear(pr);
    return 0;
}

static int fri_prover_begin(wf_fri_prover *pr, const void *src, size_t n, bool on_device, hipStream_t st) {
    if (!pr || !src) return fail(WF_ERR_ARG, "null argument");
    if (!pr->layers.empty() || pr->pending || pr->evals)
        return fail(WF_ERR_ARG, "a prior proof generation request has not been completed yet");  // prover/mod.rs:173-176
    if (n < 2 || (n & (n - 1))) return fail(WF_ERR_TRACE_LENGTH, "number of evaluations must be a power of two");
    uint32_t l = 0;
    while (((size_t)1 << l) < n) l++;
    if (l > (pr->field == WF_FIELD_F64 ? F64::TWO_ADICITY : F128::TWO_ADICITY))
        return fail(WF_ERR_DOMAIN, "no multiplicative subgroup of size 2^%u in this field", l);
    HIP_TRY(hipSetDevice(pr->ctx->device));
    WF_ENTER(pr->ctx, st);
    const size_t bytes = n * pr->ext * wf_elem_bytes(pr->field);
    int rca = fri_arena_reserve(pr, n);
    if (rca) return rca;
    if ((rca = fri_take_evals(pr, bytes))) return rca;
    hipError_t e = hipMemcpyAsync(pr->evals, src, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {  // leave the prover as it was: no half-started proof
        fri_drop_evals(pr);
        return fail(WF_ERR_HIP, "copying the evaluations failed: %s", hipGetErrorString(e));
    }
    pr->n = n;
    return 0;
}

int wf_fri_prover_begin(wf_fri_prover *pr, const void *evals, size_t n) {
    return fri_prover_begin(pr, evals, n, false, pr ? pr->ctx->stream : nullptr);
}

int wf_fri_prover_begin_dev(wf_fri_prover *pr, const void *d_evals, size_t n, void *stream) {
    return fri_prover_begin(pr, d_evals, n, true, stream ? (hipStream_t)stream : (pr ? pr->ctx->stream : nullptr));
}

}  // extern "C"

// `poly`: n coefficients of E in host memory, or (poly_on_device) in device memory of this context -- wf_deep_compose
// hands over the polynomial it has just built in ctx->io[0]
int fri_begin_poly_impl(wf_fri_prover *pr, const void *poly, bool poly_on_device, size_t n, size_t lde_blowup) {
    if (!pr || !poly) return fail(WF_ERR_ARG, "null argument");
    if (!pr->layers.empty() || pr->pending || pr->evals)
        return fail(WF_ERR_ARG, "a prior proof generation request has not been completed yet");
    if (n < 8 || (n & (n - 1))) return fail(WF_ERR_TRACE_LENGTH, "polynomial size must be a power of two >= 8");
    if (lde_blowup < 2 || lde_blowup > 128 || (lde_blowup & (lde_blowup - 1)))
        return fail(WF_ERR_BLOWUP, "blowup must be a power of two in [2,128]");
    wf_ctx *ctx = pr->ctx;
    wf_params p;
    memset(&p, 0, sizeof(p));
    p.field = pr->field;
    p.ext_degree = pr->ext;
    while (((size_t)1 << p.log2_trace_len) < n) p.log2_trace_len++;
    while (((size_t)1 << p.log2_blowup) < lde_blowup) p.log2_blowup++;
    p.n_cols = 1;
    p.n_traces = 1;
    p.digest_bytes = 32;
    memcpy(p.domain_offset, pr->offset, 16);
    int rc = check_params(&p, true);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    const size_t eb = wf_elem_bytes(pr->field), rows = n * lde_blowup, rw = wf_row_width(&p);
    if ((rc = ensure(ctx, ctx->io[0], wf_column_bytes(&p)))) return rc;
    const bool dense = path_dense_column_ok(&p);
    if (!dense && (rc = ensure(ctx, ctx->io[2], wf_lde_bytes(&p)))) return rc;
    if ((rc = fri_arena_reserve(pr, rows))) return rc;
    if ((rc = fri_take_evals(pr, rows * pr->ext * eb))) return rc;
    hipStream_t st = ctx->stream;
    if (poly != ctx->io[0].p)
        rc = hipMemcpyAsync(ctx->io[0].p, poly, wf_column_bytes(&p), poly_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st) == hipSuccess
                 ? 0
                 : fail(WF_ERR_HIP, "upload failed");
    if (rc == 0 && dense) {  // the evaluation writes the dense vector itself
        rc = path_constraint_commit(ctx, &p, ctx->io[0].p, pr->evals, nullptr, nullptr, st, true);
    } else if (rc == 0) {
        // one column of E evaluated to row-major (row width 8), then its ext_degree live lanes gathered into a dense vector
        rc = wf_constraint_commit_dev(ctx, &p, ctx->io[0].p, ctx->io[2].p, nullptr, nullptr, st);
        if (rc == 0 && hipMemcpy2DAsync(pr->evals, pr->ext * eb, ctx->io[2].p, rw * eb, pr->ext * eb, rows, hipMemcpyDeviceToDevice, st) != hipSuccess)
            rc = fail(WF_ERR_HIP, "gathering the evaluations failed");
    }
    if (rc == 0 && hipStreamSynchronize(st) != hipSuccess) rc = fail(WF_ERR_HIP, "stream synchronisation failed");
    if (rc) {
        fri_drop_evals(pr);
        return rc;
    }
    pr->n = rows;
    return 0;
}

extern "C" {

int wf_fri_prover_begin_poly(wf_fri_prover *pr, const void *poly, size_t n, size_t lde_blowup) {
    return fri_begin_poly_impl(pr, poly, false, n, lde_blowup);
}

int wf_fri_prover_commit_layer(wf_fri_prover *pr, uint8_t root_out[32]) {
    if (!pr || !root_out) return fail(WF_ERR_ARG, "null argument");
    if (!pr->evals) return fail(WF_ERR_ARG, "no evaluations: call wf_fri_prover_begin first");
    if (pr->pending) return fail(WF_ERR_ARG, "the committed layer has not been folded yet");
    uint32_t l;
    int rc = check_fri_args(pr->ctx, pr->field, pr->ext, pr->n, pr->folding, &l);
    if (rc) return rc;
    wf_ctx *ctx = pr->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    const size_t eb = wf_elem_bytes(pr->field), rows = pr->n / pr->folding;
    wf_commitment *c = commitment_new(ctx);
    c->p.field = pr->field;
    c->p.ext_degree = pr->ext;
    c->p.n_cols = pr->folding;
    c->p.n_traces = 1;
    c->p.digest_bytes = ctx->digest_bytes;  // the context's hasher (wf_ctx_set_digest_bytes)
    memcpy(c->p.domain_offset, pr->offset, 16);
    c->n_rows = rows;
    c->row_width = c->epr = c->row_elems = (uint64_t)pr->folding * pr->ext;
    for (uint64_t t = rows; t > 1; t >>= 1) c->depth++;
    c->p.log2_trace_len = c->depth;
    const size_t used0 = pr->arena_used;
    c->lde = fri_arena_take(pr, pr->n * pr->ext * eb);
    c->leaves = c->lde ? fri_arena_take(pr, rows * 32) : nullptr;
    c->nodes = c->leaves ? fri_arena_take(pr, rows * 32) : nullptr;
    c->borrowed = c->nodes != nullptr;
    if (!c->borrowed) {  // (arena too small: cannot happen after fri_arena_reserve, kept as a fallback)
        pr->arena_used = used0;
        c->lde = c->leaves = c->nodes = nullptr;
        hipError_t e = dev_malloc(ctx, &c->lde, pr->n * pr->ext * eb);
        if (e == hipSuccess) e = dev_malloc(ctx, &c->leaves, rows * 32);
        if (e == hipSuccess) e = dev_malloc(ctx, &c->nodes, rows * 32);
        if (e != hipSuccess) {
            free_commitment(c);
            return fail(WF_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
        }
    }
    hipStream_t st = ctx->stream;
    rc = pr->field == WF_FIELD_F64
             ? fri_layer_commit_dev<F64>(ctx, st, pr->ext, pr->evals, pr->n, pr->folding, c->lde, c->leaves, c->nodes)
             : fri_layer_commit_dev<F128>(ctx, st, pr->ext, pr->evals, pr->n, pr->folding, c->lde, c->leaves, c->nodes);
    if (rc == 0 && hipMemcpyAsync(c->root, (const char *)c->nodes + 32, 32, hipMemcpyDeviceToHost, st) != hipSuccess)
        rc = fail(WF_ERR_HIP, "copying the layer root failed");
    if (rc == 0 && hipStreamSynchronize(st) != hipSuccess) rc = fail(WF_ERR_HIP, "stream synchronisation failed");
    if (rc) {
        free_commitment(c);
        return rc;
    }
    memcpy(root_out, c->root, 32);
    pr->pending = c;
    return 0;
}

int wf_fri_prover_fold(wf_fri_prover *pr, const void *alpha) {
    if (!pr || !alpha) return fail(WF_ERR_ARG, "null argument");
    if (!pr->pending) return fail(WF_ERR_ARG, "no committed layer to fold: call wf_fri_prover_commit_layer first");
    wf_ctx *ctx = pr->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    const size_t rows = pr->n / pr->folding;
    void *next = fri_arena_take(pr, rows * pr->ext * wf_elem_bytes(pr->field));
    const bool next_borrowed = next != nullptr;
    if (!next && dev_malloc(ctx, &next, rows * pr->ext * wf_elem_bytes(pr->field)) != hipSuccess)
        return fail(WF_ERR_HIP, "hipMalloc failed for the folded layer");
    hipStream_t st = ctx->stream;
    // (asynchronous: the folded layer is consumed by the next call on the same stream; alpha is copied at launch)
    int rc = pr->field == WF_FIELD_F64
                 ? fri_apply_drp_dev<F64>(ctx, st, pr->ext, pr->pending->lde, rows, pr->folding, pr->offset, alpha, next)
                 : fri_apply_drp_dev<F128>(ctx, st, pr->ext, pr->pending->lde, rows, pr->folding, pr->offset, alpha, next);
    if (rc) {
        if (!next_borrowed) (void)hipFree(next);
        return rc;
    }
    if (!pr->evals_borrowed) {  // the caller's first layer: its own allocation (hipFree waits for the fold that reads it)
        (void)hipFree(pr->evals);
    }
    pr->evals = next;
    pr->evals_borrowed = next_borrowed;
    pr->n = rows;
    pr->layers.push_back(pr->pending);
    pr->pending = nullptr;
    return 0;
}

int wf_fri_prover_set_remainder(wf_fri_prover *pr, void *remainder_out, size_t capacity, size_t *len_out,
                                uint8_t commitment_out[32]) {
    if (!pr || !remainder_out || !len_out || !commitment_out) return fail(WF_ERR_ARG, "null argument");
    if (!pr->evals) return fail(WF_ERR_ARG, "no evaluations: call wf_fri_prover_begin first");
    if (pr->pending) return fail(WF_ERR_ARG, "the committed layer has not been folded yet");
    const size_t len = pr->n / pr->blowup;
    if (len == 0) return fail(WF_ERR_BLOWUP, "fewer evaluations (%zu) than the blowup factor (%u)", pr->n, pr->blowup);
    if (len > capacity) return fail(WF_ERR_ARG, "remainder has %zu coefficients, buffer holds %zu", len, capacity);
    wf_ctx *ctx = pr->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    const size_t eb = wf_elem_bytes(pr->field), bytes = pr->n * pr->ext * eb;
    std::vector<unsigned char> host(bytes);
    HIP_TRY(hipMemcpyAsync(host.data(), pr->evals, bytes, hipMemcpyDeviceToHost, ctx->stream));  // (after the last fold)
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    // the remainder layer is tiny ((remainder_max_degree + 1) * blowup evaluations): through the host-buffer entry points
    int rc = wf_fft_interpolate_poly_with_offset(ctx, pr->field, pr->ext, host.data(), pr->n, pr->offset);
    if (rc) return rc;
    memcpy(remainder_out, host.data(), len * pr->ext * eb);
    rc = wf_hash_rows(ctx, pr->field, remainder_out, 1, len * pr->ext, commitment_out);  // hash_elements(&remainder_poly)
    if (rc) return rc;
    *len_out = len;
    if (!pr->evals_borrowed) (void)hipFree(pr->evals);
    pr->evals = nullptr;
    pr->evals_borrowed = false;
    pr->n = 0;
    return 0;
}

size_t wf_fri_prover_num_layers(const wf_fri_prover *pr) { return pr ? pr->layers.size() : 0; }

int wf_fri_prover_layer(const wf_fri_prover *pr, size_t i, const wf_commitment **out) {
    if (!pr || !out) return fail(WF_ERR_ARG, "null argument");
    if (i >= pr->layers.size()) return fail(WF_ERR_ARG, "layer %zu does not exist (%zu layers)", i, pr->layers.size());
    *out = pr->layers[i];
    return 0;
}

}  // extern "C"

