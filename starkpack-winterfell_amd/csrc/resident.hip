// libwf_lde.so, unit 3 of 6 -- resident commitments: the LDE, the tree and the polynomials stay in HBM (what TraceCommitment
// / ConstraintCommitment own in the reference: prover/src/trace/commitment.rs:21-26, constraints/commitment.rs:21-24);
// synchronous and asynchronous construction from host columns, the query service (rows + BatchMerkleProof), the
// out-of-domain evaluation, and the resident form of the coset-sharded commitment with its collective queries.
#include "wf_internal.hpp"

#include "kernels.hpp"
#include "fri_kernels.hpp"

using namespace wf;

// completes an asynchronous commitment: waits for its kernels, fetches the root from its pinned slot, gives the slot back.  May run on a thread that holds no call on the context (a finaliser destroying a
// pending handle): the context is touched only under the registry's lock (CtxPin), its slot table under pool_mutex.
static int commitment_wait(wf_commitment *c) {
    if (!c->pending) return 0;
    int rc = 0;
    (void)hipSetDevice(c->device);
    if (c->done) {
        const hipError_t e = hipEventSynchronize(c->done);  // (the event is the handle's own: no context needed to wait for it)
        if (e != hipSuccess) rc = fail(WF_ERR_HIP, "the commitment's kernels failed: %s", hipGetErrorString(e));
    }
    {
        CtxPin pin(c->ctx, c->ctx_generation);
        if (pin.alive && c->root_slot1) {
            const uint8_t *slot = c->ctx->root_pin + (size_t)(c->root_slot1 - 1) * WF_ROOT_SLOT_BYTES;
            if (rc == 0) memcpy(c->root, slot, 32);
            std::lock_guard<std::mutex> lock(c->ctx->pool_mutex);
            c->ctx->root_used[c->root_slot1 - 1] = 0;
        }
    }
    if (c->done) (void)hipEventDestroy(c->done);
    c->done = nullptr;
    c->root_slot1 = 0;
    c->pending = false;
    return rc;
}

wf_commitment *commitment_new(wf_ctx *ctx) {
    wf_commitment *c = new wf_commitment();
    memset(c, 0, sizeof(*c));
    c->ctx = ctx;
    c->ctx_generation = ctx->generation;
    c->device = ctx->device;
    return c;
}

void free_commitment(wf_commitment *c) {
    if (!c) return;
    if (ctx_alive(c->ctx, c->ctx_generation)) (void)hipSetDevice(c->ctx->device);
    if (c->pending) (void)commitment_wait(c);
    if (!c->borrowed) {
        pool_free(c->ctx, c->ctx_generation, c->lde, c->lde_bytes);
        pool_free(c->ctx, c->ctx_generation, c->leaves, c->dig_bytes);
        pool_free(c->ctx, c->ctx_generation, c->nodes, c->dig_bytes);
    }
    pool_free(c->ctx, c->ctx_generation, c->polys, c->polys_bytes);
    delete c;
}

// A resident commitment's handle with its buffers (LDE, leaves, nodes, polynomials) taken from the context's pool
int commitment_alloc(wf_ctx *ctx, const wf_params *p, bool constraint, wf_commitment **out, bool *dense_out) {
    const size_t colb = wf_column_bytes(p), ldeb = wf_lde_bytes(p);
    const size_t digb = ((size_t)1 << (p->log2_trace_len + p->log2_blowup)) * 32;  // 32-byte slots whatever the digest size
    const size_t TC = (size_t)p->n_cols * p->n_traces;
    wf_commitment *c = commitment_new(ctx);
    c->p = *p;
    c->n_rows = (uint64_t)1 << (p->log2_trace_len + p->log2_blowup);
    c->epr = (uint64_t)p->n_cols * p->ext_degree;
    // a resident constraint commitment of a narrow matrix keeps its rows dense (no padding to 8 elements: a quarter of
    // the LDE bytes for one E column); nothing outside this library sees the row stride of a resident LDE
    const bool dense = constraint && path_dense_matrix_ok(p);
    c->row_width = dense ? c->epr : wf_row_width(p);
    c->row_elems = c->epr * p->n_traces;
    c->depth = p->log2_trace_len + p->log2_blowup;
    hipError_t e;
    c->lde_bytes = dense ? c->n_rows * c->row_width * wf_elem_bytes(p->field) : ldeb * p->n_traces;
    c->dig_bytes = digb;
    c->polys_bytes = TC * colb;
    if ((e = pool_alloc(ctx, &c->lde, c->lde_bytes)) != hipSuccess || (e = pool_alloc(ctx, &c->leaves, digb)) != hipSuccess ||
        (e = pool_alloc(ctx, &c->nodes, digb)) != hipSuccess || (e = pool_alloc(ctx, &c->polys, c->polys_bytes)) != hipSuccess) {
        free_commitment(c);
        return fail(WF_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    *out = c;
    *dense_out = dense;
    return 0;
}

static int commit_resident(wf_ctx *ctx, const wf_params *p, bool constraint, const void *const *cols_in,
                           void *const *polys_out, wf_commitment **out) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    if (!out) return fail(WF_ERR_ARG, "out is null");
    int rc = check_params(p, constraint);
    if (rc) return rc;
    if (!cols_in) return fail(WF_ERR_ARG, "column pointer array is null");
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    const size_t colb = wf_column_bytes(p);
    const size_t TC = (size_t)p->n_cols * p->n_traces;
    for (size_t i = 0; i < TC; i++)
        if (!cols_in[i]) return fail(WF_ERR_ARG, "column %zu is null", i);
    wf_commitment *c = nullptr;
    bool dense = false;
    if ((rc = commitment_alloc(ctx, p, constraint, &c, &dense))) return rc;
    rc = ensure(ctx, ctx->io[0], TC * colb);
    if (rc) {
        free_commitment(c);
        return rc;
    }
    hipStream_t st = ctx->stream;
    void *stage = constraint ? c->polys : ctx->io[0].p;  // composition polys are the input themselves
    const bool pipelined = !constraint && path_pipelined_upload_ok(ctx, p, colb);
    if (pipelined) {
        rc = path_trace_commit_pipelined(ctx, p, cols_in, stage, c->polys, c->lde, c->leaves, c->nodes, st, polys_out);
    } else if ((rc = upload_columns(ctx, stage, cols_in, TC, colb, st))) {
        free_commitment(c);
        return rc;
    }
    if (rc) {
        free_commitment(c);
        return rc;
    }
    if (pipelined)
        ;
    else if (constraint)
        rc = path_constraint_commit(ctx, p, c->polys, c->lde, c->leaves, c->nodes, st, dense);
    else
        rc = wf_trace_commit_dev(ctx, p, ctx->io[0].p, c->polys, c->lde, c->leaves, c->nodes, st);
    if (rc) {
        free_commitment(c);
        return rc;
    }
    if (polys_out && !constraint && !pipelined && (rc = download_columns(ctx, polys_out, c->polys, TC, colb, st))) {
        free_commitment(c);
        return rc;
    }
    hipError_t e = hipMemcpyAsync(c->root, (char *)c->nodes + 32, 32, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        free_commitment(c);
        return fail(WF_ERR_HIP, "commitment failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return 0;
}

extern "C" {

int wf_trace_commit_resident(wf_ctx *ctx, const wf_params *p, const void *const *trace_cols, void *const *polys_out,
                             wf_commitment **out) {
    return commit_resident(ctx, p, false, trace_cols, polys_out, out);
}

int wf_constraint_commit_resident(wf_ctx *ctx, const wf_params *p, const void *const *poly_cols, wf_commitment **out) {
    return commit_resident(ctx, p, true, poly_cols, nullptr, out);
}

// A stream of proofs from host columns: Prover::build_trace_commitment (prover/src/lib.rs:615-670) of STARKPack's many
// proofs (examples/src/lib.rs:97-135, winterfell/src/main.rs:105-160) one after the other, the upload of proof k + 1 on
// the copy stream under the kernels of proof k.  Returns once the columns are on their way (pageable memory: once they
// are staged; pinned memory: at once -- the caller keeps pinned columns alive until wf_commitment_wait).
static int commit_resident_async(wf_ctx *ctx, const wf_params *p, const void *const *cols_in, wf_commitment **out) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    if (!out) return fail(WF_ERR_ARG, "out is null");
    int rc = check_params(p, false);
    if (rc) return rc;
    if (!cols_in) return fail(WF_ERR_ARG, "column pointer array is null");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    WF_ENTER(ctx, st);
    const size_t colb = wf_column_bytes(p), TC = (size_t)p->n_cols * p->n_traces;
    for (size_t i = 0; i < TC; i++)
        if (!cols_in[i]) return fail(WF_ERR_ARG, "column %zu is null", i);
    if (!ctx->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (!ctx->root_pin) {
        HIP_TRY(hipHostMalloc((void **)&ctx->root_pin, WF_ROOT_SLOTS * WF_ROOT_SLOT_BYTES, hipHostMallocDefault));
        memset(ctx->root_pin, 0, WF_ROOT_SLOTS * WF_ROOT_SLOT_BYTES);
        std::lock_guard<std::mutex> lock(ctx->pool_mutex);
        ctx->root_used.assign(WF_ROOT_SLOTS, 0);
    }
    for (int i = 0; i < 2; i++) {
        if (!ctx->stage_free[i]) HIP_TRY(hipEventCreateWithFlags(&ctx->stage_free[i], hipEventDisableTiming));
        if (!ctx->upload_done[i]) HIP_TRY(hipEventCreateWithFlags(&ctx->upload_done[i], hipEventDisableTiming));
    }
    uint32_t slot1 = 0;
    {
        std::lock_guard<std::mutex> lock(ctx->pool_mutex);  // (slots come back from other threads: commitment_wait)
        for (size_t i = 0; i < WF_ROOT_SLOTS && !slot1; i++)
            if (!ctx->root_used[i]) {
                slot1 = (uint32_t)i + 1;
                ctx->root_used[i] = 1;  // taken here; handed back below if the call fails
            }
    }
    auto give_back = [&]() {
        std::lock_guard<std::mutex> lock(ctx->pool_mutex);
        ctx->root_used[slot1 - 1] = 0;
    };
    if (!slot1) return fail(WF_ERR_BUSY, "%zu asynchronous commitments are in flight: wait for (or destroy) some first", WF_ROOT_SLOTS);
    const int sb = (int)(ctx->async_seq & 1);
    if ((rc = ensure(ctx, ctx->stage[sb], TC * colb))) {
        give_back();
        return rc;
    }
    wf_commitment *c = nullptr;
    bool dense = false;
    if ((rc = commitment_alloc(ctx, p, false, &c, &dense))) {
        give_back();
        return rc;
    }
    hipError_t e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    if (e != hipSuccess) {
        give_back();
        free_commitment(c);
        return fail(WF_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e));
    }
    // the staging buffer is free once the layout kernel of the commitment that used it two calls ago has read it
    if (ctx->stage_busy[sb]) e = hipStreamWaitEvent(ctx->copy_stream, ctx->stage_free[sb], 0);
    if (e == hipSuccess) {
        rc = upload_columns(ctx, ctx->stage[sb].p, cols_in, TC, colb, ctx->copy_stream);
        if (rc == 0) e = hipEventRecord(ctx->upload_done[sb], ctx->copy_stream);
    }
    if (rc == 0 && e == hipSuccess) e = hipStreamWaitEvent(st, ctx->upload_done[sb], 0);
    if (rc == 0 && e == hipSuccess)
        rc = path_trace_commit(ctx, p, ctx->stage[sb].p, c->polys, c->lde, c->leaves, c->nodes, st, ctx->stage_free[sb]);
    uint8_t *slot = ctx->root_pin + (size_t)(slot1 - 1) * WF_ROOT_SLOT_BYTES;
    if (rc == 0 && e == hipSuccess) e = hipMemcpyAsync(slot, (char *)c->nodes + 32, 32, hipMemcpyDeviceToHost, st);
    if (rc == 0 && e == hipSuccess) e = hipEventRecord(c->done, st);
    if (rc || e != hipSuccess) {
        // whatever was queued from the caller's columns or into this handle's buffers must have drained before either goes away
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamSynchronize(st);
        ctx->stage_busy[sb] = false;
        give_back();
        free_commitment(c);
        return rc ? rc : fail(WF_ERR_HIP, "queueing the commitment failed: %s", hipGetErrorString(e));
    }
    ctx->stage_busy[sb] = true;
    ctx->async_seq++;
    c->root_slot1 = slot1;
    c->pending = true;
    *out = c;
    return 0;
}

int wf_trace_commit_resident_async(wf_ctx *ctx, const wf_params *p, const void *const *trace_cols, wf_commitment **out) {
    return commit_resident_async(ctx, p, trace_cols, out);
}

int wf_commitment_wait(wf_commitment *c) {
    if (!c) return fail(WF_ERR_ARG, "commitment is null");
    return commitment_wait(c);
}

void wf_commitment_destroy(wf_commitment *c) { free_commitment(c); }

int wf_commitment_root(const wf_commitment *c, uint8_t root_out[32]) {
    if (!c || !root_out) return fail(WF_ERR_ARG, "null argument");
    if (c->pending) {  // an asynchronous commitment asked for its root: this is where the host waits for it
        int rc = commitment_wait(const_cast<wf_commitment *>(c));
        if (rc) return rc;
    }
    memcpy(root_out, c->root, 32);
    return 0;
}

int wf_commitment_info(const wf_commitment *c, uint64_t *n_rows, uint64_t *row_elems, uint32_t *depth) {
    if (!c) return fail(WF_ERR_ARG, "commitment is null");
    if (n_rows) *n_rows = c->n_rows;
    if (row_elems) *row_elems = c->row_elems;
    if (depth) *depth = c->depth;
    return 0;
}

static int check_positions(const wf_commitment *c, const uint64_t *positions, size_t n) {
    if (!c || !positions) return fail(WF_ERR_ARG, "null argument");
    if (n == 0) return fail(WF_ERR_ARG, "at least one position is required");                       // TooFewLeafIndexes
    if (n > 255) return fail(WF_ERR_ARG, "number of positions cannot exceed 255 (got %zu)", n);      // MAX_PATHS
    for (size_t i = 0; i < n; i++)
        if (positions[i] >= c->n_rows)
            return fail(WF_ERR_LEAVES, "position %llu is out of bounds (%llu rows)", (unsigned long long)positions[i],
                        (unsigned long long)c->n_rows);                                              // LeafIndexOutOfBounds
    return 0;
}

int wf_commitment_read_rows(const wf_commitment *c, const uint64_t *positions, size_t n, void *rows_out) {
    int rc = check_positions(c, positions, n);
    if (rc) return rc;
    if (!rows_out) return fail(WF_ERR_ARG, "rows_out is null");
    wf_ctx *ctx = c->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    const size_t eb = wf_elem_bytes(c->p.field);
    const size_t out_bytes = n * c->row_elems * eb;
    if ((rc = ensure(ctx, ctx->io[3], n * 8))) return rc;
    if ((rc = ensure(ctx, ctx->io[4], out_bytes))) return rc;
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(ctx->io[3].p, positions, n * 8, hipMemcpyHostToDevice, st));
    const uint64_t trace_elems = c->n_rows * c->row_width;
    if (c->p.field == WF_FIELD_F64)
        hipLaunchKernelGGL(k_gather_rows<F64>, dim3((uint32_t)n, c->p.n_traces), dim3(64), 0, st, (const uint64_t *)c->lde,
                           trace_elems, (uint32_t)c->row_width, (uint32_t)c->epr, (const uint64_t *)ctx->io[3].p,
                           (uint64_t *)ctx->io[4].p);
    else
        hipLaunchKernelGGL(k_gather_rows<F128>, dim3((uint32_t)n, c->p.n_traces), dim3(64), 0, st, (const U128 *)c->lde,
                           trace_elems, (uint32_t)c->row_width, (uint32_t)c->epr, (const uint64_t *)ctx->io[3].p,
                           (U128 *)ctx->io[4].p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rows_out, ctx->io[4].p, out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int wf_commitment_read_lde(const wf_commitment *c, uint32_t trace, uint64_t row_begin, uint64_t n_rows, void *rows_out,
                           uint64_t *row_width_out) {
    if (!c) return fail(WF_ERR_ARG, "commitment is null");
    if (row_width_out) *row_width_out = c->row_width;
    if (n_rows == 0) return 0;
    if (!rows_out) return fail(WF_ERR_ARG, "rows_out is null");
    if (!c->lde) return fail(WF_ERR_ARG, "this commitment holds no rows");
    if (trace >= c->p.n_traces) return fail(WF_ERR_TRACES, "trace %u of %u", trace, c->p.n_traces);
    if (row_begin >= c->n_rows || n_rows > c->n_rows - row_begin)
        return fail(WF_ERR_LEAVES, "rows [%llu, %llu) are outside the %llu rows of the matrix", (unsigned long long)row_begin,
                    (unsigned long long)(row_begin + n_rows), (unsigned long long)c->n_rows);
    wf_ctx *ctx = c->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    const size_t rb = c->row_width * wf_elem_bytes(c->p.field);
    const char *src = (const char *)c->lde + ((size_t)trace * c->n_rows + row_begin) * rb;
    HIP_TRY(hipMemcpyAsync(rows_out, src, n_rows * rb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"

// rows row_begin + k * stride (k < n_rows) of a row-major matrix packed next to each other; 16 bytes per thread
__global__ void __launch_bounds__(256) k_gather_strided_rows(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint64_t n_rows,
                                                             uint64_t stride_q, uint32_t row_q) {  // *_q: in 16-byte units
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_rows * row_q) return;
    const uint64_t r = idx / row_q, q = idx - r * row_q;
    dst[idx] = src[r * stride_q + q];
}

extern "C" {

int wf_commitment_read_lde_strided(const wf_commitment *c, uint32_t trace, uint64_t row_begin, uint64_t n_rows, uint64_t row_stride,
                                   void *rows_out, uint64_t *row_width_out) {
    if (!c) return fail(WF_ERR_ARG, "commitment is null");
    if (row_stride <= 1) return wf_commitment_read_lde(c, trace, row_begin, n_rows, rows_out, row_width_out);
    if (row_width_out) *row_width_out = c->row_width;
    if (n_rows == 0) return 0;
    if (!rows_out) return fail(WF_ERR_ARG, "rows_out is null");
    if (!c->lde) return fail(WF_ERR_ARG, "this commitment holds no rows");
    if (trace >= c->p.n_traces) return fail(WF_ERR_TRACES, "trace %u of %u", trace, c->p.n_traces);
    if (row_begin >= c->n_rows || (n_rows - 1) > (c->n_rows - 1 - row_begin) / row_stride)
        return fail(WF_ERR_LEAVES, "rows %llu + k * %llu, k < %llu, leave the %llu rows of the matrix", (unsigned long long)row_begin,
                    (unsigned long long)row_stride, (unsigned long long)n_rows, (unsigned long long)c->n_rows);
    wf_ctx *ctx = c->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    const size_t rb = c->row_width * wf_elem_bytes(c->p.field);
    if (rb % 16) return fail(WF_ERR_ARG, "rows of %zu bytes cannot be gathered in 16-byte pieces", rb);  // (dense one-column f64 rows)
    int rc = ensure(ctx, ctx->io[4], n_rows * rb);
    if (rc) return rc;
    const char *src = (const char *)c->lde + ((size_t)trace * c->n_rows + row_begin) * rb;
    const uint64_t quads = n_rows * (rb / 16);
    hipLaunchKernelGGL(k_gather_strided_rows, dim3((uint32_t)((quads + 255) / 256)), dim3(256), 0, ctx->stream, (const uint4 *)src,
                       (uint4 *)ctx->io[4].p, n_rows, row_stride * (rb / 16), (uint32_t)(rb / 16));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rows_out, ctx->io[4].p, n_rows * rb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

// fetch digests by id (id < n_rows: leaf; else node id - n_rows) into host memory
static int fetch_digests(const wf_commitment *c, const std::vector<uint64_t> &ids, uint8_t *out) {
    if (ids.empty()) return 0;
    wf_ctx *ctx = c->ctx;
    int rc;
    if ((rc = ensure(ctx, ctx->io[3], ids.size() * 8))) return rc;
    if ((rc = ensure(ctx, ctx->io[4], ids.size() * 32))) return rc;
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(ctx->io[3].p, ids.data(), ids.size() * 8, hipMemcpyHostToDevice, st));
    const uint32_t n = (uint32_t)ids.size();
    hipLaunchKernelGGL(k_gather_digests, dim3((2 * n + 255) / 256), dim3(256), 0, st, (const uint4 *)c->leaves,
                       (const uint4 *)c->nodes, c->n_rows, (const uint64_t *)ctx->io[3].p, n, (uint4 *)ctx->io[4].p);
    HIP_TRY(hipGetLastError());
    const uint32_t db = c->p.digest_bytes ? c->p.digest_bytes : 32;
    std::vector<uint8_t> slots(db == 32 ? 0 : ids.size() * 32);
    HIP_TRY(hipMemcpyAsync(db == 32 ? (void *)out : (void *)slots.data(), ctx->io[4].p, ids.size() * 32, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (db != 32) copy_digests_out(out, slots.data(), ids.size(), db);
    return 0;
}

int wf_commitment_prove(const wf_commitment *c, uint64_t index, uint8_t *path_out) {
    if (!c || !path_out) return fail(WF_ERR_ARG, "null argument");
    if (index >= c->n_rows) return fail(WF_ERR_LEAVES, "leaf index out of bounds");  // merkle/mod.rs:193-198
    HIP_TRY(hipSetDevice(c->ctx->device));
    WF_ENTER(c->ctx, c->ctx->stream);
    std::vector<uint64_t> ids{index, index ^ 1};
    for (uint64_t i = (index + c->n_rows) >> 1; i > 1; i >>= 1) ids.push_back(c->n_rows + (i ^ 1));
    return fetch_digests(c, ids, path_out);
}

// The digests a BatchMerkleProof of `positions` consists of (MerkleTree::prove_batch, merkle/mod.rs:222-284), as ids for
// k_gather_digests (id < n_rows: leaf, else node id - n_rows): vec_ids[i] = the nodes vector of the i-th normalised index.
static int batch_proof_ids(const wf_commitment *c, const uint64_t *positions, size_t n, std::vector<std::vector<uint64_t>> &vec_ids,
                           size_t &total) {
    // map_indexes (merkle/mod.rs:376-395): duplicates are an error
    std::map<uint64_t, size_t> index_map;
    for (size_t i = 0; i < n; i++) index_map[positions[i]] = i;
    if (index_map.size() != n) return fail(WF_ERR_LEAVES, "list of positions contains duplicates");  // DuplicateLeafIndex
    // normalize_indexes (:397-403): sorted set of even-aligned indexes
    std::vector<uint64_t> idx;
    for (auto &kv : index_map) {
        uint64_t e = kv.first - (kv.first & 1);
        if (idx.empty() || idx.back() != e) idx.push_back(e);
    }
    // ids of the digests of each vector, in the order prove_batch pushes them (:238-276)
    vec_ids.assign(idx.size(), {});
    std::vector<uint64_t> next;
    const uint64_t nl = c->n_rows;
    for (size_t i = 0; i < idx.size(); i++) {
        for (uint64_t j = idx[i]; j < idx[i] + 2; j++)
            if (!index_map.count(j)) vec_ids[i].push_back(j);  // leaf id
        next.push_back((idx[i] + nl) >> 1);
    }
    for (uint32_t lvl = 1; lvl < c->depth; lvl++) {
        std::vector<uint64_t> cur = next;
        next.clear();
        size_t i = 0;
        while (i < cur.size()) {
            const uint64_t sibling = cur[i] ^ 1;
            if (i + 1 < cur.size() && cur[i + 1] == sibling)
                i += 1;
            else
                vec_ids[i].push_back(nl + sibling);  // note: indexed by position in the current list, as the reference does
            next.push_back(sibling >> 1);
            i += 1;
        }
    }
    total = 0;
    for (auto &v : vec_ids) total += v.size();
    return 0;
}

// Several commitments of one context queried in one host round trip: every id list goes up in one copy from pinned
// memory, the gathers of all commitments are queued, one copy brings rows and digests back, one synchronisation.
// (A proof queries the trace tree, the constraint tree and every FRI layer: ten calls of ~0.15 ms each otherwise.)
static int query_many_impl(wf_query *q, size_t nq) {
    if (!q || nq == 0) return fail(WF_ERR_ARG, "no queries");
    wf_ctx *ctx = nullptr;
    for (size_t i = 0; i < nq; i++) {
        int rc = check_positions(q[i].commitment, q[i].positions, q[i].n);
        if (rc) return rc;
        if (!q[i].leaves_out || !q[i].nodes_out || !q[i].node_counts) return fail(WF_ERR_ARG, "query %zu: null output", i);
        if (!ctx) ctx = q[i].commitment->ctx;
        if (q[i].commitment->ctx != ctx) return fail(WF_ERR_ARG, "query %zu: the commitments belong to different contexts", i);
        if (q[i].rows_out && !q[i].commitment->lde) return fail(WF_ERR_LEAVES, "query %zu: this commitment holds no rows", i);
    }
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    struct Part {
        std::vector<std::vector<uint64_t>> vec_ids;
        size_t total = 0, n_ids = 0, ids_off = 0, rows_off = 0, rows_bytes = 0, dig_off = 0;
    };
    std::vector<Part> parts(nq);
    size_t ids_total = 0, out_total = 0;
    for (size_t i = 0; i < nq; i++) {
        const wf_commitment *c = q[i].commitment;
        Part &pt = parts[i];
        int rc = batch_proof_ids(c, q[i].positions, q[i].n, pt.vec_ids, pt.total);
        if (rc) return rc;
        if (pt.total > q[i].nodes_capacity) return fail(WF_ERR_ARG, "query %zu: nodes_out too small: %zu digests needed", i, pt.total);
        pt.n_ids = q[i].n + pt.total;  // the queried leaves first: also the positions of the row gather
        pt.ids_off = ids_total;
        ids_total += pt.n_ids;
        pt.rows_bytes = q[i].rows_out ? q[i].n * c->row_elems * wf_elem_bytes(c->p.field) : 0;
        pt.rows_off = out_total;
        out_total += (pt.rows_bytes + 255) & ~(size_t)255;
        pt.dig_off = out_total;
        out_total += (pt.n_ids * 32 + 255) & ~(size_t)255;
    }
    const size_t ids_bytes = (ids_total * 8 + 255) & ~(size_t)255;
    int rc;
    if ((rc = ensure(ctx, ctx->io[3], ids_bytes))) return rc;
    if ((rc = ensure(ctx, ctx->io[4], out_total))) return rc;
    if (ctx->qpin_cap < ids_bytes + out_total) {
        if (ctx->qpin) (void)hipHostFree(ctx->qpin);
        ctx->qpin = nullptr;
        ctx->qpin_cap = 0;
        const size_t want = std::max<size_t>(2 * (ids_bytes + out_total), (size_t)1 << 20);
        if (hipHostMalloc(&ctx->qpin, want, hipHostMallocDefault) != hipSuccess)
            return fail(WF_ERR_HIP, "hipHostMalloc failed: %s", hipGetErrorString(hipGetLastError()));
        ctx->qpin_cap = want;
    }
    uint64_t *h_ids = (uint64_t *)ctx->qpin;
    char *h_out = (char *)ctx->qpin + ids_bytes;
    for (size_t i = 0; i < nq; i++) {
        uint64_t *d = h_ids + parts[i].ids_off;
        memcpy(d, q[i].positions, q[i].n * 8);
        d += q[i].n;
        for (auto &v : parts[i].vec_ids) {
            memcpy(d, v.data(), v.size() * 8);
            d += v.size();
        }
    }
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(ctx->io[3].p, h_ids, ids_total * 8, hipMemcpyHostToDevice, st));
    for (size_t i = 0; i < nq; i++) {
        const wf_commitment *c = q[i].commitment;
        const Part &pt = parts[i];
        const uint64_t *d_ids = (const uint64_t *)ctx->io[3].p + pt.ids_off;
        const uint32_t n = (uint32_t)q[i].n;
        if (q[i].rows_out) {
            const uint64_t trace_elems = c->n_rows * c->row_width;
            if (c->p.field == WF_FIELD_F64)
                hipLaunchKernelGGL(k_gather_rows<F64>, dim3(n, c->p.n_traces), dim3(64), 0, st, (const uint64_t *)c->lde, trace_elems,
                                   (uint32_t)c->row_width, (uint32_t)c->epr, d_ids, (uint64_t *)((char *)ctx->io[4].p + pt.rows_off));
            else
                hipLaunchKernelGGL(k_gather_rows<F128>, dim3(n, c->p.n_traces), dim3(64), 0, st, (const U128 *)c->lde, trace_elems,
                                   (uint32_t)c->row_width, (uint32_t)c->epr, d_ids, (U128 *)((char *)ctx->io[4].p + pt.rows_off));
            HIP_TRY(hipGetLastError());
        }
        const uint32_t nid = (uint32_t)pt.n_ids;
        hipLaunchKernelGGL(k_gather_digests, dim3((2 * nid + 255) / 256), dim3(256), 0, st, (const uint4 *)c->leaves,
                           (const uint4 *)c->nodes, c->n_rows, d_ids, nid, (uint4 *)((char *)ctx->io[4].p + pt.dig_off));
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(h_out, ctx->io[4].p, out_total, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t i = 0; i < nq; i++) {
        const Part &pt = parts[i];
        if (q[i].rows_out) memcpy(q[i].rows_out, h_out + pt.rows_off, pt.rows_bytes);
        const uint32_t db = q[i].commitment->p.digest_bytes ? q[i].commitment->p.digest_bytes : 32;  // entries of the caller's arrays
        copy_digests_out(q[i].leaves_out, h_out + pt.dig_off, q[i].n, db);
        copy_digests_out(q[i].nodes_out, h_out + pt.dig_off + q[i].n * 32, pt.total, db);
        for (size_t v = 0; v < pt.vec_ids.size(); v++) q[i].node_counts[v] = (uint32_t)pt.vec_ids[v].size();
        q[i].n_vectors = pt.vec_ids.size();
        q[i].n_nodes = pt.total;
        q[i].depth = q[i].commitment->depth;
    }
    return 0;
}

// rows_out == nullptr: the proof only
static int query_impl(const wf_commitment *c, const uint64_t *positions, size_t n, void *rows_out, uint8_t *leaves_out,
                      uint8_t *nodes_out, size_t nodes_capacity, uint32_t *node_counts, size_t *n_vectors, size_t *n_nodes,
                      uint32_t *depth_out) {
    if (!n_vectors || !n_nodes) return fail(WF_ERR_ARG, "null argument");
    wf_query q;
    memset(&q, 0, sizeof(q));
    q.commitment = c;
    q.positions = positions;
    q.n = n;
    q.rows_out = rows_out;
    q.leaves_out = leaves_out;
    q.nodes_out = nodes_out;
    q.nodes_capacity = nodes_capacity;
    q.node_counts = node_counts;
    int rc = query_many_impl(&q, 1);
    if (rc) return rc;
    *n_vectors = q.n_vectors;
    *n_nodes = q.n_nodes;
    if (depth_out) *depth_out = q.depth;
    return 0;
}

int wf_commitment_query_many(wf_query *queries, size_t n_queries) { return query_many_impl(queries, n_queries); }

int wf_commitment_prove_batch(const wf_commitment *c, const uint64_t *positions, size_t n, uint8_t *leaves_out,
                              uint8_t *nodes_out, size_t nodes_capacity, uint32_t *node_counts, size_t *n_vectors,
                              size_t *n_nodes, uint32_t *depth_out) {
    return query_impl(c, positions, n, nullptr, leaves_out, nodes_out, nodes_capacity, node_counts, n_vectors, n_nodes, depth_out);
}

int wf_commitment_query(const wf_commitment *c, const uint64_t *positions, size_t n, void *rows_out, uint8_t *leaves_out,
                        uint8_t *nodes_out, size_t nodes_capacity, uint32_t *node_counts, size_t *n_vectors, size_t *n_nodes,
                        uint32_t *depth_out) {
    if (!rows_out) return fail(WF_ERR_ARG, "rows_out is null");
    return query_impl(c, positions, n, rows_out, leaves_out, nodes_out, nodes_capacity, node_counts, n_vectors, n_nodes, depth_out);
}

}  // extern "C"

template <class F, int WZ>
static void eval_fill_powers(EvalAtArgs<F> &a, uint32_t q, const typename F::T *z) {
    Ext<F, WZ> y;
    for (int w = 0; w < WZ; w++) y.c[w] = z[w];
    for (int s = 0; s < EVAL_POWERS; s++) {  // y = z^(2^s)
        for (int w = 0; w < WZ; w++) a.pw[q][s][w] = y.c[w];
        y = ext_mul<F, WZ>(y, y);
    }
}

template <class F>
static int eval_columns_at_dev(wf_ctx *ctx, hipStream_t st, const void *d_polys, size_t n_cols, size_t n, uint32_t ext_c,
                               const void *z_host, uint32_t ext_z, void *d_out, uint32_t n_points = 1) {
    typedef typename F::T T;
    const uint32_t key = ext_c * 10 + ext_z;
    if (key != 11 && key != 12 && key != 22 && !(F::FIELD_ID == 1 && (key == 13 || key == 33))) {
        if (F::FIELD_ID != 1 && (key == 13 || key == 33)) return fail(WF_ERR_EXTENSION, "f128 has no cubic extension");
        return fail(WF_ERR_EXTENSION, "cannot evaluate degree-%u extension coefficients at a degree-%u extension point", ext_c, ext_z);
    }
    const uint32_t n_blocks = (uint32_t)((n + EVAL_BLOCK - 1) / EVAL_BLOCK);
    int rcp = ensure(ctx, ctx->hash_tmp, (size_t)n_points * n_cols * n_blocks * ext_z * sizeof(T));  // (block values; no hashing runs alongside)
    if (rcp) return rcp;
    for (uint32_t q0 = 0; q0 < n_points; q0 += EVAL_POINTS) {  // two points per launch
        const uint32_t np = std::min<uint32_t>(EVAL_POINTS, n_points - q0);
        EvalAtArgs<F> a;
        memset(&a, 0, sizeof(a));
        a.polys = (const T *)d_polys;
        a.n = n;
        a.n_cols = (uint32_t)n_cols;
        a.n_blocks = n_blocks;
        a.partial = (T *)ctx->hash_tmp.p + (size_t)q0 * n_cols * n_blocks * ext_z;
        a.out = (T *)d_out + (size_t)q0 * n_cols * ext_z;
        for (uint32_t q = 0; q < np; q++) {
            const T *z = (const T *)z_host + (size_t)(q0 + q) * ext_z;
            for (uint32_t w = 0; w < ext_z; w++)
                if (!F::is_valid(z[w])) return fail(WF_ERR_ARG, "z is not a valid field element");
            switch (ext_z) {
                case 1: eval_fill_powers<F, 1>(a, q, z); break;
                case 2: eval_fill_powers<F, 2>(a, q, z); break;
                default:
                    if constexpr (F::FIELD_ID == 1) eval_fill_powers<F, 3>(a, q, z);
                    break;
            }
        }
        const dim3 grid(a.n_blocks, (uint32_t)n_cols, np), grid2((uint32_t)n_cols, np), block(256);
        prof_mark(ctx, st, "ood.evaluate_columns_at");
        switch (key) {
            case 11: hipLaunchKernelGGL((k_eval_columns_at<F, 1, 1>), grid, block, 0, st, a); break;
            case 12: hipLaunchKernelGGL((k_eval_columns_at<F, 1, 2>), grid, block, 0, st, a); break;
            case 22: hipLaunchKernelGGL((k_eval_columns_at<F, 2, 2>), grid, block, 0, st, a); break;
            case 13:
                if constexpr (F::FIELD_ID == 1) hipLaunchKernelGGL((k_eval_columns_at<F, 1, 3>), grid, block, 0, st, a);
                break;
            default:
                if constexpr (F::FIELD_ID == 1) hipLaunchKernelGGL((k_eval_columns_at<F, 3, 3>), grid, block, 0, st, a);
                break;
        }
        HIP_TRY(hipGetLastError());
        switch (ext_z) {
            case 1: hipLaunchKernelGGL((k_eval_columns_sum<F, 1>), grid2, block, 0, st, a); break;
            case 2: hipLaunchKernelGGL((k_eval_columns_sum<F, 2>), grid2, block, 0, st, a); break;
            default:
                if constexpr (F::FIELD_ID == 1) hipLaunchKernelGGL((k_eval_columns_sum<F, 3>), grid2, block, 0, st, a);
                break;
        }
        HIP_TRY(hipGetLastError());
        prof_mark(ctx, st, "between_calls");
    }
    return 0;
}

extern "C" {

int wf_commitment_evaluate_polys_at(const wf_commitment *c, const void *z, uint32_t z_ext_degree, void *out) {
    if (!c || !z || !out) return fail(WF_ERR_ARG, "null argument");
    if (!c->polys) return fail(WF_ERR_ARG, "this commitment holds no polynomials (FRI layer)");
    wf_ctx *ctx = c->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    const size_t n_cols = (size_t)c->p.n_cols * c->p.n_traces, n = (size_t)1 << c->p.log2_trace_len;
    const size_t out_bytes = n_cols * z_ext_degree * wf_elem_bytes(c->p.field);
    int rc = ensure(ctx, ctx->io[4], out_bytes);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    rc = c->p.field == WF_FIELD_F64
             ? eval_columns_at_dev<F64>(ctx, st, c->polys, n_cols, n, c->p.ext_degree, z, z_ext_degree, ctx->io[4].p)
             : eval_columns_at_dev<F128>(ctx, st, c->polys, n_cols, n, c->p.ext_degree, z, z_ext_degree, ctx->io[4].p);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, ctx->io[4].p, out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int wf_commitment_evaluate_polys_at_points(const wf_commitment *c, const void *points, uint32_t n_points, uint32_t z_ext_degree,
                                           void *out) {
    if (!c || !points || !out) return fail(WF_ERR_ARG, "null argument");
    if (n_points < 1 || n_points > 4) return fail(WF_ERR_ARG, "1 to 4 points per call (got %u)", n_points);
    if (!c->polys) return fail(WF_ERR_ARG, "this commitment holds no polynomials (FRI layer)");
    wf_ctx *ctx = c->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    const size_t n_cols = (size_t)c->p.n_cols * c->p.n_traces, n = (size_t)1 << c->p.log2_trace_len;
    const size_t out_bytes = (size_t)n_points * n_cols * z_ext_degree * wf_elem_bytes(c->p.field);
    int rc = ensure(ctx, ctx->io[4], out_bytes);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    rc = c->p.field == WF_FIELD_F64
             ? eval_columns_at_dev<F64>(ctx, st, c->polys, n_cols, n, c->p.ext_degree, points, z_ext_degree, ctx->io[4].p, n_points)
             : eval_columns_at_dev<F128>(ctx, st, c->polys, n_cols, n, c->p.ext_degree, points, z_ext_degree, ctx->io[4].p, n_points);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, ctx->io[4].p, out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int wf_evaluate_columns_at(wf_ctx *ctx, uint32_t field, uint32_t ext_degree, const void *const *poly_cols,
                           size_t n_cols, size_t n, const void *z, uint32_t z_ext_degree, void *out) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    if (field != WF_FIELD_F64 && field != WF_FIELD_F128) return fail(WF_ERR_FIELD, "unknown field id %u", field);
    if (n < 2 || (n & (n - 1))) return fail(WF_ERR_TRACE_LENGTH, "size must be a power of two >= 2");
    if (!poly_cols || !z || !out || n_cols == 0) return fail(WF_ERR_ARG, "null argument");
    int rc;
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    const size_t colb = n * ext_degree * wf_elem_bytes(field), out_bytes = n_cols * z_ext_degree * wf_elem_bytes(field);
    if ((rc = ensure(ctx, ctx->io[0], n_cols * colb))) return rc;
    if ((rc = ensure(ctx, ctx->io[4], out_bytes))) return rc;
    hipStream_t st = ctx->stream;
    for (size_t i = 0; i < n_cols; i++)
        if (!poly_cols[i]) return fail(WF_ERR_ARG, "column %zu is null", i);
    if ((rc = upload_columns(ctx, ctx->io[0].p, poly_cols, n_cols, colb, st))) return rc;
    rc = field == WF_FIELD_F64
             ? eval_columns_at_dev<F64>(ctx, st, ctx->io[0].p, n_cols, n, ext_degree, z, z_ext_degree, ctx->io[4].p)
             : eval_columns_at_dev<F128>(ctx, st, ctx->io[0].p, n_cols, n, ext_degree, z, z_ext_degree, ctx->io[4].p);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, ctx->io[4].p, out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}


}  // extern "C"
// ---- resident form of the sharded commitment + its query service -------------------------------------------------------
struct wf_sharded_commitment {
    wf_comm *comm;
    uint64_t ctx_generation;
    wf_params p;
    void *lde_shard, *leaves, *nodes, *polys;  // this rank's rows (its cosets), leaf range, sub-tree; all polynomials
    size_t lde_bytes, dig_bytes, polys_bytes;
    uint64_t n_rows, row_width, epr, row_elems;  // of the WHOLE commitment
    uint32_t depth, per;
    std::vector<uint8_t> top;  // nodes 0 .. 2 W - 1 of the whole tree (host copy, identical on every rank)
    wf_commitment polys_view;  // the polynomials as a wf_commitment (out-of-domain evaluation); holds no rows
};

static void free_sharded(wf_sharded_commitment *c) {
    if (!c) return;
    wf_ctx *ctx = c->comm->ctx;  // (the communicator outlives its commitments: wf_comm_destroy comes after)
    if (ctx_alive(ctx, c->ctx_generation)) (void)hipSetDevice(ctx->device);
    pool_free(ctx, c->ctx_generation, c->lde_shard, c->lde_bytes);
    pool_free(ctx, c->ctx_generation, c->leaves, c->dig_bytes);
    pool_free(ctx, c->ctx_generation, c->nodes, c->dig_bytes);
    pool_free(ctx, c->ctx_generation, c->polys, c->polys_bytes);
    delete c;
}

// where digest `id` of the whole tree lives (id < N: leaf id; else node id - N): owner rank and its index in that rank's
// gather space (index < N / W: its leaves; else its sub-tree nodes + N / W), or owner = -1: a top node, replicated
static void locate_digest(uint64_t id, uint64_t N, uint32_t W, int *owner, uint64_t *local) {
    const uint64_t nl = N / W;
    if (id < N) {
        *owner = (int)(id / nl);
        *local = id % nl;
        return;
    }
    const uint64_t i = id - N;  // node index, 1 <= i < N
    uint64_t n = 1;
    while (2 * n <= i) n *= 2;  // level of n nodes: n <= i < 2 n
    if (n < W) {
        *owner = -1;
        *local = i;
        return;
    }
    const uint64_t r = (i - n) / (n / W);
    *owner = (int)r;
    *local = nl + (i - n - r * (n / W)) + n / W;
}

extern "C" {

int wf_trace_commit_sharded_resident(wf_comm *comm, const wf_params *p, const void *const *trace_cols,
                                     wf_sharded_commitment **out) {
    if (!comm || !out) return fail(WF_ERR_ARG, "null argument");
    int rc = check_params(p, false);
    if (rc) return rc;
    if (!trace_cols) return fail(WF_ERR_ARG, "column pointer array is null");
    uint32_t c0, per;
    rc = wf_shard_cosets(1u << p->log2_blowup, (uint32_t)comm->rank, (uint32_t)comm->world, &c0, &per);
    if (rc) return rc;
    const uint32_t W = (uint32_t)comm->world;
    const uint64_t R = (uint64_t)1 << p->log2_trace_len, N = R << p->log2_blowup;
    if (N / W < 2 || R < W) return fail(WF_ERR_ARG, "trace too short to split over %u ranks", W);
    wf_ctx *ctx = comm->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    const size_t colb = wf_column_bytes(p), TC = (size_t)p->n_cols * p->n_traces;
    for (size_t i = 0; i < TC; i++)
        if (!trace_cols[i]) return fail(WF_ERR_ARG, "column %zu is null", i);
    wf_sharded_commitment *c = new wf_sharded_commitment();
    c->comm = comm;
    c->ctx_generation = ctx->generation;
    c->p = *p;
    c->lde_shard = c->leaves = c->nodes = c->polys = nullptr;
    c->n_rows = N;
    c->row_width = wf_row_width(p);
    c->epr = (uint64_t)p->n_cols * p->ext_degree;
    c->row_elems = c->epr * p->n_traces;
    c->depth = p->log2_trace_len + p->log2_blowup;
    c->per = per;
    c->lde_bytes = (size_t)p->n_traces * R * per * c->row_width * wf_elem_bytes(p->field);
    c->dig_bytes = (size_t)(N / W) * 32;
    c->polys_bytes = TC * colb;
    hipStream_t st = ctx->stream;
    // local stage (allocations, upload), then the ranks agree that all of them got this far before the first exchange
    void *top_pin = nullptr;
    const int local_rc = [&]() -> int {
        hipError_t e;
        if ((e = pool_alloc(ctx, &c->lde_shard, c->lde_bytes)) != hipSuccess || (e = pool_alloc(ctx, &c->leaves, c->dig_bytes)) != hipSuccess ||
            (e = pool_alloc(ctx, &c->nodes, c->dig_bytes)) != hipSuccess || (e = pool_alloc(ctx, &c->polys, c->polys_bytes)) != hipSuccess)
            return fail(WF_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
        int rl;
        if ((rl = ensure(ctx, ctx->io[0], TC * colb)) || (rl = ensure(ctx, ctx->io[4], (size_t)2 * W * 32))) return rl;
        if ((rl = ensure(ctx, comm->stage, 2 * (size_t)R * per * 32))) return rl;  // (the exchange staging of trace_commit_sharded)
        if ((rl = comm_pinned(comm, (size_t)2 * W * 32, &top_pin))) return rl;      // (allocated while nothing is queued: see comm_pinned)
        return upload_columns(ctx, ctx->io[0].p, trace_cols, TC, colb, st);
    }();
    if ((rc = comm_agree(comm, local_rc, "wf_trace_commit_sharded_resident"))) {
        free_sharded(c);
        return rc;
    }
    rc = path_trace_commit_sharded(comm, p, ctx->io[0].p, c->polys, c->lde_shard, c->leaves, c->nodes, ctx->io[4].p, st);
    c->top.resize((size_t)2 * W * 32);
    // the top levels come back through the communicator's pinned memory: a copy into the (pageable) vector would hold the
    // host inside hipMemcpyAsync until the exchanges in front of it have run -- out of the watchdog's reach -- and would
    // write into freed memory if it were still queued when a time-out frees the handle
    if (rc == 0 && hipMemcpyAsync(top_pin, ctx->io[4].p, c->top.size(), hipMemcpyDeviceToHost, st) != hipSuccess)
        rc = fail(WF_ERR_HIP, "commitment failed: %s", hipGetErrorString(hipGetLastError()));
    if (rc == 0) rc = comm_wait(comm, st);  // (with the watchdog: a peer that failed inside the exchanges never arrives)
    if (rc) {
        // (after a time-out the kernels behind the dead exchange may still hold these buffers: they go back to the driver
        // through hipFree, which waits for the device, not into the pool)
        if (comm->dead) c->lde_bytes = c->dig_bytes = c->polys_bytes = 0;
        free_sharded(c);
        return rc;
    }
    memcpy(c->top.data(), top_pin, c->top.size());
    memset(&c->polys_view, 0, sizeof(c->polys_view));
    c->polys_view.ctx = ctx;
    c->polys_view.ctx_generation = ctx->generation;
    c->polys_view.p = *p;
    c->polys_view.polys = c->polys;
    c->polys_view.borrowed = true;
    memcpy(c->polys_view.root, c->top.data() + 32, 32);
    *out = c;
    return 0;
}

void wf_sharded_commitment_destroy(wf_sharded_commitment *c) { free_sharded(c); }

int wf_sharded_commitment_root(const wf_sharded_commitment *c, uint8_t root_out[32]) {
    if (!c || !root_out) return fail(WF_ERR_ARG, "null argument");
    memcpy(root_out, c->top.data() + 32, 32);
    return 0;
}

int wf_sharded_commitment_polys(const wf_sharded_commitment *c, const wf_commitment **out) {
    if (!c || !out) return fail(WF_ERR_ARG, "null argument");
    *out = &c->polys_view;
    return 0;
}

int wf_sharded_commitment_query(wf_sharded_commitment *c, const uint64_t *positions, size_t n, void *rows_out,
                                uint8_t *leaves_out, uint8_t *nodes_out, size_t nodes_capacity, uint32_t *node_counts,
                                size_t *n_vectors, size_t *n_nodes, uint32_t *depth_out) {
    if (!c || !positions) return fail(WF_ERR_ARG, "null argument");
    if (!rows_out || !leaves_out || !nodes_out || !node_counts || !n_vectors || !n_nodes) return fail(WF_ERR_ARG, "null argument");
    // the position checks and the digest ids of the batch proof are those of the unsharded commitment
    wf_commitment shape;
    memset(&shape, 0, sizeof(shape));
    shape.n_rows = c->n_rows;
    shape.depth = c->depth;
    int rc = check_positions(&shape, positions, n);
    if (rc) return rc;
    std::vector<std::vector<uint64_t>> vec_ids;
    size_t total = 0;
    if ((rc = batch_proof_ids(&shape, positions, n, vec_ids, total))) return rc;
    if (total > nodes_capacity) return fail(WF_ERR_ARG, "nodes_out too small: %zu digests needed", total);
    std::vector<uint64_t> ids(positions, positions + n);
    for (auto &v : vec_ids) ids.insert(ids.end(), v.begin(), v.end());

    wf_comm *comm = c->comm;
    // refused at once, BEFORE the local stage below: it synchronises with the stream, and on a communicator that died in a
    // timed-out collective the stream may still sit behind the exchange that never completed (round 5: this call used to block there
    // until the stream drained and only then report the dead communicator)
    if (comm->dead) return fail(WF_ERR_COMM, "the communicator was aborted after a failed or timed-out collective");
    wf_ctx *ctx = comm->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    const uint32_t W = (uint32_t)comm->world, me = (uint32_t)comm->rank, blowup = 1u << c->p.log2_blowup;
    const uint64_t N = c->n_rows, nl = N / W;
    const size_t eb = wf_elem_bytes(c->p.field), row_bytes = c->row_elems * eb;

    // what this rank owns: digests (gathered from its leaves / sub-tree) and rows (its cosets)
    std::vector<uint64_t> my_dig_local, my_row_local;
    std::vector<size_t> my_dig_at, my_row_at;
    for (size_t k = 0; k < ids.size(); k++) {
        int owner;
        uint64_t local;
        locate_digest(ids[k], N, W, &owner, &local);
        if (owner == (int)me) {
            my_dig_local.push_back(local);
            my_dig_at.push_back(k);
        }
    }
    for (size_t i = 0; i < n; i++) {
        const uint64_t k = positions[i] / blowup;
        const uint32_t cst = (uint32_t)(positions[i] % blowup);
        if (cst / c->per == me) {
            my_row_local.push_back(k * c->per + cst % c->per);
            my_row_at.push_back(i);
        }
    }
    // message of a rank: [ids.size()][32] digests then [n] rows, zero where it owns nothing; exchanged with one all-gather
    const size_t msg = ((ids.size() * 32 + n * row_bytes + 255) / 256) * 256;
    std::vector<uint8_t> mine(msg, 0);
    const size_t nd = my_dig_local.size(), nr = my_row_local.size();
    const size_t idx_bytes = (nd + nr) * 8, dig_off = (idx_bytes + 255) / 256 * 256, row_off = dig_off + (nd * 32 + 255) / 256 * 256;
    hipStream_t st = ctx->stream;
    char *d_msg = nullptr;
    void *all_pin = nullptr;
    // local stage: nothing below the agreement may fail on one rank alone
    const int local_rc = [&]() -> int {
        int rl;
        if ((rl = ensure(ctx, ctx->io[3], row_off + nr * row_bytes + 256))) return rl;
        if ((rl = ensure(ctx, ctx->io[4], msg * (W + 1)))) return rl;
        if ((rl = comm_pinned(comm, msg * W, &all_pin))) return rl;  // (allocated while nothing is queued: see comm_pinned)
        char *work = (char *)ctx->io[3].p;
        if (nd + nr) {
            std::vector<uint64_t> idx(my_dig_local);
            idx.insert(idx.end(), my_row_local.begin(), my_row_local.end());
            HIP_TRY(hipMemcpyAsync(work, idx.data(), idx_bytes, hipMemcpyHostToDevice, st));
            if (nd) {
                hipLaunchKernelGGL(k_gather_digests, dim3((uint32_t)((2 * nd + 255) / 256)), dim3(256), 0, st, (const uint4 *)c->leaves,
                                   (const uint4 *)c->nodes, nl, (const uint64_t *)work, (uint32_t)nd, (uint4 *)(work + dig_off));
                HIP_TRY(hipGetLastError());
            }
            if (nr) {
                const uint64_t trace_elems = (N / blowup) * c->per * c->row_width;  // one trace's shard
                if (c->p.field == WF_FIELD_F64)
                    hipLaunchKernelGGL(k_gather_rows<F64>, dim3((uint32_t)nr, c->p.n_traces), dim3(64), 0, st, (const uint64_t *)c->lde_shard,
                                       trace_elems, (uint32_t)c->row_width, (uint32_t)c->epr, (const uint64_t *)work + nd,
                                       (uint64_t *)(work + row_off));
                else
                    hipLaunchKernelGGL(k_gather_rows<F128>, dim3((uint32_t)nr, c->p.n_traces), dim3(64), 0, st, (const U128 *)c->lde_shard,
                                       trace_elems, (uint32_t)c->row_width, (uint32_t)c->epr, (const uint64_t *)work + nd,
                                       (U128 *)(work + row_off));
                HIP_TRY(hipGetLastError());
            }
            std::vector<uint8_t> got(nd * 32 + nr * row_bytes);
            if (nd) HIP_TRY(hipMemcpyAsync(got.data(), work + dig_off, nd * 32, hipMemcpyDeviceToHost, st));
            if (nr) HIP_TRY(hipMemcpyAsync(got.data() + nd * 32, work + row_off, nr * row_bytes, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            for (size_t q = 0; q < nd; q++) memcpy(&mine[my_dig_at[q] * 32], &got[q * 32], 32);
            for (size_t q = 0; q < nr; q++) memcpy(&mine[ids.size() * 32 + my_row_at[q] * row_bytes], &got[nd * 32 + q * row_bytes], row_bytes);
        }
        d_msg = (char *)ctx->io[4].p;
        HIP_TRY(hipMemcpyAsync(d_msg, mine.data(), msg, hipMemcpyHostToDevice, st));
        return 0;
    }();
    if ((rc = comm_agree(comm, local_rc, "wf_sharded_commitment_query"))) return rc;
    if ((rc = comm_all_gather(comm, d_msg, d_msg + msg, msg, st))) return rc;
    // the merged messages land in the communicator's pinned memory (see comm_pinned) and are read after the wait succeeded
    HIP_TRY(hipMemcpyAsync(all_pin, d_msg + msg, msg * W, hipMemcpyDeviceToHost, st));
    if ((rc = comm_wait(comm, st))) return rc;
    const uint8_t *all = (const uint8_t *)all_pin;

    // every entry from its owner's message
    std::vector<uint8_t> dig(ids.size() * 32);
    for (size_t k = 0; k < ids.size(); k++) {
        int owner;
        uint64_t local;
        locate_digest(ids[k], N, W, &owner, &local);
        if (owner < 0)
            memcpy(&dig[k * 32], &c->top[local * 32], 32);
        else
            memcpy(&dig[k * 32], &all[(size_t)owner * msg + k * 32], 32);
    }
    for (size_t i = 0; i < n; i++) {
        const uint32_t owner = (uint32_t)(positions[i] % blowup) / c->per;
        memcpy((char *)rows_out + i * row_bytes, &all[(size_t)owner * msg + ids.size() * 32 + i * row_bytes], row_bytes);
    }
    copy_digests_out(leaves_out, dig.data(), n, c->p.digest_bytes);
    copy_digests_out(nodes_out, dig.data() + n * 32, total, c->p.digest_bytes);
    for (size_t i = 0; i < vec_ids.size(); i++) node_counts[i] = (uint32_t)vec_ids[i].size();
    *n_vectors = vec_ids.size();
    *n_nodes = total;
    if (depth_out) *depth_out = c->depth;
    return 0;
}

}  // extern "C"
