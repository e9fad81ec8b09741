// Multi-GPU layer of libwf_lde.so (included by wf_lde.hip): one process per GPU, the path's exchanges behind the C ABI.
//
// The reference has no distributed code (/root/reference/README.md:43 lists a distributed prover as planned only); the
// sharding follows SURVEY.md §8e:
//   * independent proofs, one per GPU: no data-path collective, ONE all-gather of the 32-byte roots
//     (wf_comm_all_gather_roots);
//   * one STARKPack commitment (commit_to_comb_rows, prover/src/matrix/row_matrix.rs:204-238) sharded by coset:
//     coset c of the LDE domain owns the rows j = k * blowup + c, a leaf needs only its own row of every trace, so a
//     rank evaluates and hashes its cosets alone and the ranks exchange DIGESTS, never rows.
// Transport: RCCL (librccl.so.1, resolved with dlopen so that the library loads on hosts without it and shares the
// copy a PyTorch process has already mapped), or a caller-supplied table of two collectives (wf_transport) -- a host
// with its own fabric code (MPI, gloo in the rehearsal tests) drives exactly the same partitioning and kernels.
#pragma once

#include <dlfcn.h>

#include <chrono>
#include <mutex>
#include <rccl/rccl.h>

namespace wfcomm {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllToAll) AllToAll = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;                  // optional: the watchdog of the blocking calls
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;  // optional
    char path[512] = "";         // the file the symbols came from (dladdr): which copy of RCCL a process really runs
    char load_error[256] = "";   // dlerror() of the failed load, kept (dlerror() itself reports an error only once)
};

static bool rccl_load(Rccl &r) {
    const char *names[] = {getenv("WF_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
        const char *e = dlerror();
        snprintf(r.load_error, sizeof(r.load_error), "%s", e ? e : "dlopen failed");
    }
    if (!h) return false;
#define WF_SYM(field, sym)                                                          \
    r.field = (decltype(r.field))dlsym(h, #sym);                                    \
    if (!r.field) {                                                                 \
        snprintf(r.load_error, sizeof(r.load_error), "symbol %s not found", #sym);  \
        dlclose(h);                                                                 \
        return false;                                                               \
    }
    WF_SYM(GetUniqueId, ncclGetUniqueId)
    WF_SYM(CommInitRank, ncclCommInitRank)
    WF_SYM(CommDestroy, ncclCommDestroy)
    WF_SYM(AllGather, ncclAllGather)
    WF_SYM(AllToAll, ncclAllToAll)
    WF_SYM(GetErrorString, ncclGetErrorString)
    WF_SYM(GetVersion, ncclGetVersion)
#undef WF_SYM
    r.CommAbort = (decltype(r.CommAbort))dlsym(h, "ncclCommAbort");
    r.CommGetAsyncError = (decltype(r.CommGetAsyncError))dlsym(h, "ncclCommGetAsyncError");
    Dl_info info;
    if (dladdr((const void *)r.AllGather, &info) && info.dli_fname) snprintf(r.path, sizeof(r.path), "%s", info.dli_fname);
    r.load_error[0] = 0;
    r.handle = h;
    return true;
}

// loaded once, by whichever thread needs it first (several host threads may create communicators at the same time)
static Rccl &rccl_state() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { (void)rccl_load(r); });
    return r;
}
static Rccl *rccl() {
    Rccl &r = rccl_state();
    return r.handle ? &r : nullptr;
}
static const char *rccl_load_error() {
    const Rccl &r = rccl_state();
    return r.load_error[0] ? r.load_error : "no error recorded";
}

}  // namespace wfcomm

struct wf_comm {
    wf_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    ncclComm_t nccl = nullptr;  // RCCL transport
    bool custom = false;        // caller-supplied transport
    wf_transport tr{};
    DevBuf stage;  // receive staging of the leaf exchanges ([world][...] rank-major, before the interleave)
    DevBuf small;  // barrier / reduction words
    // Watchdog of the host-blocking calls (barrier, reductions, collective queries, wf_comm_stream_wait): a rank that died
    // or never arrives must not leave the others waiting for ever.  WF_COMM_TIMEOUT_S, read once at creation.
    double timeout_s = 300.0;
    bool dead = false;  // a collective timed out or failed asynchronously: the communicator was aborted
};

static double comm_timeout_from_env() {
    const char *e = getenv("WF_COMM_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 300.0;
}

// Host-blocking wait for everything queued on `st`, with the communicator's watchdog: polls the stream (and RCCL's
// asynchronous error state); on expiry the communicator is aborted (ncclCommAbort ends the kernels of a collective whose
// peer never came) and WF_ERR_COMM is returned.  A dead communicator refuses further collectives.
static int comm_wait(wf_comm *c, hipStream_t st) {
    using clock = std::chrono::steady_clock;
    const auto t0 = clock::now();
    wfcomm::Rccl *R = (c->nccl && !c->custom) ? wfcomm::rccl() : nullptr;
    unsigned spins = 0;
    for (;;) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) return fail(WF_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(q));
        if (R && R->CommGetAsyncError && (spins & 1023) == 1023) {
            ncclResult_t ae = ncclSuccess;
            if (R->CommGetAsyncError(c->nccl, &ae) == ncclSuccess && ae != ncclSuccess && ae != ncclInProgress) {
                if (R->CommAbort) (void)R->CommAbort(c->nccl);
                c->nccl = nullptr;
                c->dead = true;
                return fail(WF_ERR_COMM, "RCCL reported an asynchronous error on rank %d: %s", c->rank, R->GetErrorString(ae));
            }
        }
        const double waited = std::chrono::duration<double>(clock::now() - t0).count();
        if (waited > c->timeout_s) {
            if (R && R->CommAbort && c->nccl) {
                (void)R->CommAbort(c->nccl);
                c->nccl = nullptr;
            }
            c->dead = true;
            return fail(WF_ERR_COMM, "collective timed out after %.0f s on rank %d of %d (a peer died or never arrived); the communicator was aborted",
                        waited, c->rank, c->world);
        }
        if (++spins < 2000)
            std::this_thread::yield();
        else
            std::this_thread::sleep_for(std::chrono::microseconds(spins < 20000 ? 20 : 500));
    }
}

#define RCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t _r = (expr);                                                                        \
        if (_r != ncclSuccess) return fail(WF_ERR_COMM, "%s failed: %s", #expr, R->GetErrorString(_r)); \
    } while (0)

// every rank contributes `bytes` at d_send; d_recv receives world * bytes, rank-major
static int comm_all_gather(wf_comm *c, const void *d_send, void *d_recv, size_t bytes, hipStream_t st) {
    if (c->dead) return fail(WF_ERR_COMM, "the communicator was aborted after a failed or timed-out collective");
    if (c->world == 1) {
        if (d_send != d_recv) HIP_TRY(hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, st));
        return 0;
    }
    if (c->custom) {
        const int rc = c->tr.all_gather(c->tr.user, d_send, d_recv, bytes, (void *)st);
        return rc ? fail(WF_ERR_COMM, "transport all_gather failed with %d", rc) : 0;
    }
    wfcomm::Rccl *R = wfcomm::rccl();
    RCCL_TRY(R->AllGather(d_send, d_recv, bytes, ncclUint8, c->nccl, st));
    return 0;
}

// block s (`bytes` bytes at d_send + s * bytes) of rank r lands at d_recv + r * bytes on rank s
static int comm_all_to_all(wf_comm *c, const void *d_send, void *d_recv, size_t bytes, hipStream_t st) {
    if (c->dead) return fail(WF_ERR_COMM, "the communicator was aborted after a failed or timed-out collective");
    if (c->world == 1) {
        if (d_send != d_recv) HIP_TRY(hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, st));
        return 0;
    }
    if (c->custom) {
        const int rc = c->tr.all_to_all(c->tr.user, d_send, d_recv, bytes, (void *)st);
        return rc ? fail(WF_ERR_COMM, "transport all_to_all failed with %d", rc) : 0;
    }
    wfcomm::Rccl *R = wfcomm::rccl();
    RCCL_TRY(R->AllToAll(d_send, d_recv, bytes, ncclUint8, c->nccl, st));
    return 0;
}

namespace wf {

// Leaf digests as they arrive from an exchange -- src[rank s][k][local coset lc], k < n_k, lc < per -- to the order of the
// tree: dst[k * world * per + s * per + lc] (natural LDE row order inside the k-range).  One 16-byte half digest per lane,
// consecutive lanes write consecutive bytes.
__global__ void __launch_bounds__(256) k_interleave_leaves(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint64_t n_k,
                                                           uint32_t world, uint32_t per) {
    const uint64_t total = n_k * world * per * 2;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t leaf = g >> 1;
        const uint32_t beta = world * per;
        const uint64_t k = leaf / beta;
        const uint32_t c = (uint32_t)(leaf - k * beta), s = c / per, lc = c - s * per;
        dst[g] = src[((s * n_k + k) * per + lc) * 2 + (g & 1)];
    }
}

}  // namespace wf

static int run_interleave(hipStream_t st, const void *src, void *dst, uint64_t n_k, uint32_t world, uint32_t per) {
    const uint64_t total = n_k * world * per * 2;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(wf::k_interleave_leaves, dim3(grid), dim3(256), 0, st, (const uint4 *)src, (uint4 *)dst, n_k, world, per);
    HIP_TRY(hipGetLastError());
    return 0;
}

static bool pow2_u32(uint32_t v) { return v && !(v & (v - 1)); }

// Segment-sharded interpolation + coset-sharded evaluation of one packed commitment; see wf_trace_commit_sharded_dev.
template <class F>
static int trace_commit_sharded(wf_comm *c, const wf_params *p, const void *d_trace, void *d_polys, void *d_lde_shard,
                                void *d_leaves, void *d_nodes, void *d_top, hipStream_t st) {
    typedef typename F::T T;
    wf_ctx *ctx = c->ctx;
    const uint32_t W = (uint32_t)c->world, r = (uint32_t)c->rank;
    const uint32_t blowup = 1u << p->log2_blowup, per = blowup / W;
    const uint64_t R = (uint64_t)1 << p->log2_trace_len, N = R << p->log2_blowup;
    PathBufs<F> b;
    int rc = path_buffers<F>(ctx, p, b, per);
    if (rc) return rc;
    constexpr uint32_t S = SegCfg<F>::S;
    const size_t seg_bytes = (size_t)R * S * sizeof(T);

    // K1, sharded by segment when the segments divide evenly: rank r interpolates segments [r * n_seg / W, ..) and the
    // coefficients are all-gathered straight into the segment layout (rank-major == segment-major: no reordering).
    // Otherwise (fewer segments than ranks) every rank interpolates everything: no exchange.
    const bool shard_k1 = W > 1 && b.n_seg % W == 0;
    const uint32_t seg_cnt = shard_k1 ? b.n_seg / W : b.n_seg, seg0 = shard_k1 ? r * seg_cnt : 0;
    rc = run_xpose<F>(ctx, st, true, d_trace, b.segA, R, p->ext_degree, b.total_base_cols, b.n_seg, seg0, seg_cnt);
    if (rc) return rc;
    SegDesc<F> d;
    memset(&d, 0, sizeof(d));
    d.in = b.segA + (size_t)seg0 * R * S;
    d.work = (T *)d.in;
    d.out = b.segB + (size_t)seg0 * R * S;
    d.logN = p->log2_trace_len;
    d.n_seg = seg_cnt;
    d.n_cosets = 1;
    d.rows_out = false;
    rc = run_seg_transform<F>(ctx, st, d);
    if (rc) return rc;
    if (shard_k1) {
        prof_mark(ctx, st, "exchange.polys");
        rc = comm_all_gather(c, d.out, b.segB, seg_bytes * seg_cnt, st);
        if (rc) return rc;
    }
    if (d_polys) {
        rc = run_xpose<F>(ctx, st, false, b.segB, d_polys, R, p->ext_degree, b.total_base_cols, b.n_seg, 0, b.n_seg);
        if (rc) return rc;
    }

    // K2 + K3 on this rank's cosets: rows k * per + lc of the shard, leaves in the same order, into the send staging
    const size_t shard_digests = (size_t)R * per * 32;
    rc = ensure(c->ctx, c->stage, 2 * shard_digests);
    if (rc) return rc;
    uint8_t *send = (uint8_t *)c->stage.p, *recv = send + shard_digests;
    rc = evaluate_and_commit<F>(ctx, st, p, b, d_lde_shard, send, nullptr, r * per, per);
    if (rc) return rc;

    // The one exchange of the data path: rank s keeps the tree over the leaf range [s * N / W, (s + 1) * N / W), i.e. the
    // k-range [s * R / W, ..) of every coset -- a contiguous piece of every rank's shard -- so an all-to-all of
    // R / W * per digests per pair (1 / W of an all-gather's bytes) brings every rank exactly its leaves.
    prof_mark(ctx, st, "exchange.leaves");
    rc = comm_all_to_all(c, send, recv, shard_digests / W, st);
    if (rc) return rc;
    prof_mark(ctx, st, "merkle");
    rc = run_interleave(st, recv, d_leaves, R / W, W, per);
    if (rc) return rc;
    const uint64_t n_local = N / W;
    if (n_local >= 2) {
        rc = run_merkle(st, d_leaves, n_local, d_nodes);  // local layout: d_nodes[1] = this rank's sub-root
        if (rc) return rc;
    }
    // the top log2(W) levels: all-gather of the W sub-roots (32 * W bytes), folded by every rank
    uint8_t *top = (uint8_t *)d_top;
    const void *sub_root = n_local >= 2 ? (const uint8_t *)d_nodes + 32 : (const uint8_t *)d_leaves;
    prof_mark(ctx, st, "exchange.sub_roots");
    if (W == 1) {
        HIP_TRY(hipMemcpyAsync(top, d_nodes, 64, hipMemcpyDeviceToDevice, st));  // [0] = zero digest, [1] = root
    } else {
        rc = comm_all_gather(c, sub_root, top + (size_t)W * 32, 32, st);
        if (rc) return rc;
        rc = run_merkle(st, top + (size_t)W * 32, W, top);
        if (rc) return rc;
    }
    prof_mark(ctx, st, "between_calls");
    return 0;
}

extern "C" {

int wf_comm_unique_id(uint8_t id_out[WF_COMM_ID_BYTES]) {
    if (!id_out) return fail(WF_ERR_ARG, "id_out is null");
    wfcomm::Rccl *R = wfcomm::rccl();
    if (!R) return fail(WF_ERR_COMM, "RCCL (librccl.so.1) could not be loaded: %s", wfcomm::rccl_load_error());
    static_assert(sizeof(ncclUniqueId) == WF_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    RCCL_TRY(R->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

static int comm_common(wf_ctx *ctx, int rank, int world, wf_comm **out) {
    if (!ctx || !out) return fail(WF_ERR_ARG, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(WF_ERR_ARG, "rank %d is not inside a world of %d", rank, world);
    return 0;
}

int wf_comm_create(wf_ctx *ctx, const uint8_t id[WF_COMM_ID_BYTES], int rank, int world, wf_comm **out) {
    int rc = comm_common(ctx, rank, world, out);
    if (rc) return rc;
    if (!id) return fail(WF_ERR_ARG, "id is null");
    wfcomm::Rccl *R = wfcomm::rccl();
    if (!R) return fail(WF_ERR_COMM, "RCCL (librccl.so.1) could not be loaded: %s", wfcomm::rccl_load_error());
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    wf_comm *c = new wf_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    c->timeout_s = comm_timeout_from_env();
    ncclResult_t e = R->CommInitRank(&c->nccl, world, uid, rank);
    if (e != ncclSuccess) {
        delete c;
        return fail(WF_ERR_COMM, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, R->GetErrorString(e));
    }
    *out = c;
    return 0;
}

int wf_comm_create_with_transport(wf_ctx *ctx, const wf_transport *t, int rank, int world, wf_comm **out) {
    int rc = comm_common(ctx, rank, world, out);
    if (rc) return rc;
    if (!t || !t->all_gather || !t->all_to_all) return fail(WF_ERR_ARG, "transport table is incomplete");
    wf_comm *c = new wf_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    c->timeout_s = comm_timeout_from_env();
    c->custom = true;
    c->tr = *t;
    *out = c;
    return 0;
}

void wf_comm_destroy(wf_comm *c) {
    if (!c) return;
    if (ctx_alive(c->ctx)) {
        (void)hipSetDevice(c->ctx->device);
        if (!c->dead) (void)hipStreamSynchronize(c->ctx->stream);
    }
    if (c->nccl) {
        wfcomm::Rccl *R = wfcomm::rccl();
        if (R) (void)R->CommDestroy(c->nccl);
    }
    if (c->stage.p) (void)hipFree(c->stage.p);
    if (c->small.p) (void)hipFree(c->small.p);
    delete c;
}

int wf_comm_rank(const wf_comm *c) { return c ? c->rank : -1; }
int wf_comm_world(const wf_comm *c) { return c ? c->world : 0; }

int wf_comm_rccl_version(void) {
    wfcomm::Rccl *R = wfcomm::rccl();
    int v = 0;
    if (!R || R->GetVersion(&v) != ncclSuccess) return 0;
    return v;
}

const char *wf_comm_rccl_path(void) {
    wfcomm::Rccl *R = wfcomm::rccl();
    return R ? R->path : "";
}

int wf_comm_stream_wait(wf_comm *c, void *stream) {
    if (!c) return fail(WF_ERR_ARG, "comm is null");
    HIP_TRY(hipSetDevice(c->ctx->device));
    return comm_wait(c, stream ? (hipStream_t)stream : c->ctx->stream);
}

int wf_comm_all_gather(wf_comm *c, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream) {
    if (!c || !d_send || !d_recv) return fail(WF_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(c->ctx->device));
    return comm_all_gather(c, d_send, d_recv, bytes_per_rank, stream ? (hipStream_t)stream : c->ctx->stream);
}

int wf_comm_all_gather_roots(wf_comm *c, const void *d_roots, size_t n_roots, void *d_all, void *stream) {
    return wf_comm_all_gather(c, d_roots, d_all, n_roots * 32, stream);
}

// all-gather of one 8-byte word per rank through the device, result on the host (blocking)
static int comm_gather_words(wf_comm *c, uint64_t mine, std::vector<uint64_t> &all) {
    HIP_TRY(hipSetDevice(c->ctx->device));
    int rc = ensure(c->ctx, c->small, 8 * (size_t)(c->world + 1));
    if (rc) return rc;
    hipStream_t st = c->ctx->stream;
    uint64_t *d = (uint64_t *)c->small.p;
    HIP_TRY(hipMemcpyAsync(d, &mine, 8, hipMemcpyHostToDevice, st));
    rc = comm_all_gather(c, d, d + 1, 8, st);
    if (rc) return rc;
    all.resize(c->world);
    HIP_TRY(hipMemcpyAsync(all.data(), d + 1, 8 * (size_t)c->world, hipMemcpyDeviceToHost, st));
    return comm_wait(c, st);
}

// Collective entry points agree on a status word before their data exchange: a rank that failed locally (allocation, a
// HIP error in its gathers) reports it here, and EVERY rank returns an error instead of the others waiting inside RCCL
// for a peer that has already left the call.
static int comm_agree(wf_comm *c, int local_rc, const char *what) {
    char local_msg[sizeof(g_err)];
    snprintf(local_msg, sizeof(local_msg), "%s", g_err);
    std::vector<uint64_t> all;
    int rc = comm_gather_words(c, (uint64_t)(uint32_t)local_rc, all);
    if (rc) return rc;
    if (local_rc) return fail(local_rc, "%s", local_msg);
    for (size_t r = 0; r < all.size(); r++)
        if (all[r]) return fail(WF_ERR_COMM, "%s failed on rank %zu with status %d", what, r, (int)(int32_t)(uint32_t)all[r]);
    return 0;
}

int wf_comm_barrier(wf_comm *c) {
    if (!c) return fail(WF_ERR_ARG, "comm is null");
    std::vector<uint64_t> all;
    return comm_gather_words(c, (uint64_t)c->rank, all);
}

int wf_comm_max_f64(wf_comm *c, double *value) {
    if (!c || !value) return fail(WF_ERR_ARG, "null argument");
    uint64_t bits;
    memcpy(&bits, value, 8);
    std::vector<uint64_t> all;
    int rc = comm_gather_words(c, bits, all);
    if (rc) return rc;
    double m = *value;
    for (uint64_t w : all) {
        double v;
        memcpy(&v, &w, 8);
        if (v > m) m = v;
    }
    *value = m;
    return 0;
}

int wf_shard_proofs(uint32_t n_proofs, uint32_t rank, uint32_t world, uint32_t *first, uint32_t *count) {
    if (!first || !count || world == 0 || rank >= world) return fail(WF_ERR_ARG, "invalid rank / world");
    const uint32_t base = n_proofs / world, rem = n_proofs % world;
    *first = rank * base + std::min(rank, rem);
    *count = base + (rank < rem ? 1 : 0);
    return 0;
}

int wf_shard_cosets(uint32_t blowup, uint32_t rank, uint32_t world, uint32_t *first, uint32_t *count) {
    if (!first || !count || world == 0 || rank >= world) return fail(WF_ERR_ARG, "invalid rank / world");
    if (!pow2_u32(blowup) || !pow2_u32(world) || blowup % world)
        return fail(WF_ERR_ARG, "the world size %u must be a power of two dividing the blowup factor %u", world, blowup);
    *count = blowup / world;
    *first = rank * *count;
    return 0;
}

int wf_shard_route(uint32_t log2_lde_rows, uint32_t blowup, uint32_t world, uint64_t position, uint32_t *row_rank,
                   uint64_t *row_local, uint32_t *tree_rank, uint64_t *leaf_local) {
    if (!pow2_u32(blowup) || !pow2_u32(world) || blowup % world || log2_lde_rows > 40)
        return fail(WF_ERR_ARG, "the world size %u must be a power of two dividing the blowup factor %u", world, blowup);
    const uint64_t N = (uint64_t)1 << log2_lde_rows;
    if (position >= N || N < blowup) return fail(WF_ERR_LEAVES, "position %llu is outside the domain", (unsigned long long)position);
    const uint32_t per = blowup / world;
    const uint64_t k = position / blowup;
    const uint32_t cst = (uint32_t)(position % blowup);
    if (row_rank) *row_rank = cst / per;
    if (row_local) *row_local = k * per + cst % per;
    if (tree_rank) *tree_rank = (uint32_t)(position / (N / world));
    if (leaf_local) *leaf_local = position % (N / world);
    return 0;
}

int wf_comm_all_gather_leaf_shards(wf_comm *c, const void *d_leaves_shard, size_t trace_len, uint32_t cosets_per_rank,
                                   void *d_leaves, void *stream) {
    if (!c || !d_leaves_shard || !d_leaves) return fail(WF_ERR_ARG, "null argument");
    if (trace_len == 0 || cosets_per_rank == 0) return fail(WF_ERR_ARG, "empty shard");
    HIP_TRY(hipSetDevice(c->ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->ctx->stream;
    const size_t bytes = trace_len * cosets_per_rank * 32;
    int rc = ensure(c->ctx, c->stage, bytes * c->world);
    if (rc) return rc;
    rc = comm_all_gather(c, d_leaves_shard, c->stage.p, bytes, st);
    if (rc) return rc;
    return run_interleave(st, c->stage.p, d_leaves, trace_len, (uint32_t)c->world, cosets_per_rank);
}

int wf_trace_commit_sharded_dev(wf_comm *c, const wf_params *p, const void *d_trace, void *d_polys, void *d_lde_shard,
                                void *d_leaves, void *d_nodes, void *d_top, void *stream) {
    if (!c) return fail(WF_ERR_ARG, "comm is null");
    int rc = check_params(p, false);
    if (rc) return rc;
    if (!d_trace || !d_lde_shard || !d_leaves || !d_nodes || !d_top) return fail(WF_ERR_ARG, "null device buffer");
    uint32_t c0, cn;
    rc = wf_shard_cosets(1u << p->log2_blowup, (uint32_t)c->rank, (uint32_t)c->world, &c0, &cn);
    if (rc) return rc;
    if (((uint64_t)1 << p->log2_trace_len) < (uint64_t)c->world)
        return fail(WF_ERR_ARG, "trace too short to split its rows over %d ranks", c->world);
    HIP_TRY(hipSetDevice(c->ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->ctx->stream;
    CallGuard guard(c->ctx, st);
    if (guard.rc) return guard.rc;
    if (p->field == WF_FIELD_F64) return trace_commit_sharded<F64>(c, p, d_trace, d_polys, d_lde_shard, d_leaves, d_nodes, d_top, st);
    return trace_commit_sharded<F128>(c, p, d_trace, d_polys, d_lde_shard, d_leaves, d_nodes, d_top, st);
}

}  // extern "C"

// ---- resident form of the sharded commitment + its query service -------------------------------------------------------
struct wf_sharded_commitment {
    wf_comm *comm;
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
    if (ctx_alive(ctx)) (void)hipSetDevice(ctx->device);
    pool_free(ctx, c->lde_shard, c->lde_bytes);
    pool_free(ctx, c->leaves, c->dig_bytes);
    pool_free(ctx, c->nodes, c->dig_bytes);
    pool_free(ctx, c->polys, c->polys_bytes);
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
    const int local_rc = [&]() -> int {
        hipError_t e;
        if ((e = pool_alloc(ctx, &c->lde_shard, c->lde_bytes)) != hipSuccess || (e = pool_alloc(ctx, &c->leaves, c->dig_bytes)) != hipSuccess ||
            (e = pool_alloc(ctx, &c->nodes, c->dig_bytes)) != hipSuccess || (e = pool_alloc(ctx, &c->polys, c->polys_bytes)) != hipSuccess)
            return fail(WF_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
        int rl;
        if ((rl = ensure(ctx, ctx->io[0], TC * colb)) || (rl = ensure(ctx, ctx->io[4], (size_t)2 * W * 32))) return rl;
        if ((rl = ensure(ctx, comm->stage, 2 * (size_t)R * per * 32))) return rl;  // (the exchange staging of trace_commit_sharded)
        return upload_columns(ctx, ctx->io[0].p, trace_cols, TC, colb, st);
    }();
    if ((rc = comm_agree(comm, local_rc, "wf_trace_commit_sharded_resident"))) {
        free_sharded(c);
        return rc;
    }
    rc = p->field == WF_FIELD_F64
             ? trace_commit_sharded<F64>(comm, p, ctx->io[0].p, c->polys, c->lde_shard, c->leaves, c->nodes, ctx->io[4].p, st)
             : trace_commit_sharded<F128>(comm, p, ctx->io[0].p, c->polys, c->lde_shard, c->leaves, c->nodes, ctx->io[4].p, st);
    c->top.resize((size_t)2 * W * 32);
    if (rc == 0 && hipMemcpyAsync(c->top.data(), ctx->io[4].p, c->top.size(), hipMemcpyDeviceToHost, st) != hipSuccess)
        rc = fail(WF_ERR_HIP, "commitment failed: %s", hipGetErrorString(hipGetLastError()));
    if (rc == 0) rc = comm_wait(comm, st);  // (with the watchdog: a peer that failed inside the exchanges never arrives)
    if (rc) {
        free_sharded(c);
        return rc;
    }
    memset(&c->polys_view, 0, sizeof(c->polys_view));
    c->polys_view.ctx = ctx;
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
    std::vector<uint8_t> mine(msg, 0), all(msg * W);
    const size_t nd = my_dig_local.size(), nr = my_row_local.size();
    const size_t idx_bytes = (nd + nr) * 8, dig_off = (idx_bytes + 255) / 256 * 256, row_off = dig_off + (nd * 32 + 255) / 256 * 256;
    hipStream_t st = ctx->stream;
    char *d_msg = nullptr;
    // local stage: nothing below the agreement may fail on one rank alone
    const int local_rc = [&]() -> int {
        int rl;
        if ((rl = ensure(ctx, ctx->io[3], row_off + nr * row_bytes + 256))) return rl;
        if ((rl = ensure(ctx, ctx->io[4], msg * (W + 1)))) return rl;
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
    HIP_TRY(hipMemcpyAsync(all.data(), d_msg + msg, msg * W, hipMemcpyDeviceToHost, st));
    if ((rc = comm_wait(comm, st))) return rc;

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
    memcpy(leaves_out, dig.data(), n * 32);
    memcpy(nodes_out, dig.data() + n * 32, total * 32);
    for (size_t i = 0; i < vec_ids.size(); i++) node_counts[i] = (uint32_t)vec_ids[i].size();
    *n_vectors = vec_ids.size();
    *n_nodes = total;
    if (depth_out) *depth_out = c->depth;
    return 0;
}

}  // extern "C"
