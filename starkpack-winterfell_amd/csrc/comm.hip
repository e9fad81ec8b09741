// libwf_lde.so, unit 5 of 6 -- wf_comm: one process per GPU, the path's exchanges behind the C ABI.
//
// The reference has no distributed code (/root/reference/README.md:43 lists a distributed prover as planned only); the
// sharding follows SURVEY.md §8e:
//   * independent proofs, one per GPU: no data-path collective, ONE all-gather of the 32-byte roots
//     (wf_comm_all_gather_roots);
//   * one STARKPack commitment (commit_to_comb_rows, prover/src/matrix/row_matrix.rs:204-238) sharded by coset:
//     coset c of the LDE domain owns the rows j = k * blowup + c, a leaf needs only its own row of every trace, so a
//     rank evaluates and hashes its cosets alone and the ranks exchange DIGESTS, never rows (path.hip:
//     path_trace_commit_sharded; resident.hip: the resident form and its collective queries).
// Transport: RCCL (librccl.so.1, resolved with dlopen so that the library loads on hosts without it and shares the
// copy a PyTorch process has already mapped), or a caller-supplied table of two collectives (wf_transport) -- a host
// with its own fabric code (MPI, gloo in the rehearsal tests) drives exactly the same partitioning and kernels.
#include "wf_internal.hpp"

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
    decltype(&ncclCommCount) CommCount = nullptr;                  // optional: wf_comm_info
    decltype(&ncclCommUserRank) CommUserRank = nullptr;            // optional
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;            // optional
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
    r.CommCount = (decltype(r.CommCount))dlsym(h, "ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))dlsym(h, "ncclCommUserRank");
    r.CommCuDevice = (decltype(r.CommCuDevice))dlsym(h, "ncclCommCuDevice");
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

static double comm_timeout_from_env() {
    const char *e = getenv("WF_COMM_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 300.0;
}

// Host-blocking wait for everything queued on `st`, with the communicator's watchdog: polls the stream (and RCCL's
// asynchronous error state); on expiry the communicator is aborted (ncclCommAbort ends the kernels of a collective whose
// peer never came) and WF_ERR_COMM is returned.  A dead communicator refuses further collectives.
int comm_wait(wf_comm *c, hipStream_t st) {
    using clock = std::chrono::steady_clock;
    const auto t0 = clock::now();
    wfcomm::Rccl *R = (((ncclComm_t &)c->nccl) && !c->custom) ? wfcomm::rccl() : nullptr;
    unsigned spins = 0;
    for (;;) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) return fail(WF_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(q));
        if (R && R->CommGetAsyncError && (spins & 1023) == 1023) {
            ncclResult_t ae = ncclSuccess;
            if (R->CommGetAsyncError(((ncclComm_t &)c->nccl), &ae) == ncclSuccess && ae != ncclSuccess && ae != ncclInProgress) {
                if (R->CommAbort) (void)R->CommAbort(((ncclComm_t &)c->nccl));
                ((ncclComm_t &)c->nccl) = nullptr;
                c->dead = true;
                return fail(WF_ERR_COMM, "RCCL reported an asynchronous error on rank %d: %s", c->rank, R->GetErrorString(ae));
            }
        }
        const double waited = std::chrono::duration<double>(clock::now() - t0).count();
        if (waited > c->timeout_s) {
            if (R && R->CommAbort && ((ncclComm_t &)c->nccl)) {
                (void)R->CommAbort(((ncclComm_t &)c->nccl));
                ((ncclComm_t &)c->nccl) = nullptr;
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
int comm_all_gather(wf_comm *c, const void *d_send, void *d_recv, size_t bytes, hipStream_t st) {
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
    RCCL_TRY(R->AllGather(d_send, d_recv, bytes, ncclUint8, ((ncclComm_t &)c->nccl), st));
    return 0;
}

// block s (`bytes` bytes at d_send + s * bytes) of rank r lands at d_recv + r * bytes on rank s
int comm_all_to_all(wf_comm *c, const void *d_send, void *d_recv, size_t bytes, hipStream_t st) {
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
    RCCL_TRY(R->AllToAll(d_send, d_recv, bytes, ncclUint8, ((ncclComm_t &)c->nccl), st));
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

int comm_interleave(hipStream_t st, const void *src, void *dst, uint64_t n_k, uint32_t world, uint32_t per) {
    const uint64_t total = n_k * world * per * 2;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(wf::k_interleave_leaves, dim3(grid), dim3(256), 0, st, (const uint4 *)src, (uint4 *)dst, n_k, world, per);
    HIP_TRY(hipGetLastError());
    return 0;
}


static bool pow2_u32(uint32_t v) { return v && !(v & (v - 1)); }

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
    c->ctx_generation = ctx->generation;
    c->rank = rank;
    c->world = world;
    c->timeout_s = comm_timeout_from_env();
    ncclResult_t e = R->CommInitRank(&((ncclComm_t &)c->nccl), world, uid, rank);
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
    c->ctx_generation = ctx->generation;
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
    if (ctx_alive(c->ctx, c->ctx_generation)) {
        (void)hipSetDevice(c->ctx->device);
        if (!c->dead) (void)hipStreamSynchronize(c->ctx->stream);
    }
    if (((ncclComm_t &)c->nccl)) {
        wfcomm::Rccl *R = wfcomm::rccl();
        if (R) (void)R->CommDestroy(((ncclComm_t &)c->nccl));
    }
    if (c->stage.p) (void)hipFree(c->stage.p);
    if (c->small.p) (void)hipFree(c->small.p);
    // (hipHostFree waits for the device: a copy into these buffers that a timed-out collective left queued has either run
    // or been dropped with its aborted communicator's kernels before the memory goes away)
    if (c->pin) (void)hipHostFree(c->pin);
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

}  // extern "C"

// Pinned host memory owned by the communicator for what its host-blocking calls copy back (the gathered words of
// barrier / max / agree, the top levels of a sharded commitment, the merged messages of a collective query).  The
// destination of such a copy must not be pageable (hipMemcpyAsync into pageable memory holds the HOST until the
// collective in front of it has run: a missing peer would hang the caller before the watchdog is ever reached) and must
// outlive a timed-out call (the copy may still be queued when WF_ERR_COMM is returned): it lives here until
// wf_comm_destroy, and is read only after comm_wait has succeeded.
int comm_pinned(wf_comm *c, size_t bytes, void **out) {
    if (bytes > c->pin_cap) {
        // (growing means freeing: only while nothing of an earlier, failed call can still be in flight)
        if (c->dead) return fail(WF_ERR_COMM, "the communicator was aborted after a failed or timed-out collective");
        if (c->pin) HIP_TRY(hipHostFree(c->pin));
        c->pin = nullptr;
        c->pin_cap = 0;
        const size_t cap = std::max<size_t>(bytes, 4096);
        HIP_TRY(hipHostMalloc(&c->pin, cap, hipHostMallocDefault));
        c->pin_cap = cap;
    }
    *out = c->pin;
    return 0;
}

// all-gather of one 8-byte word per rank through the device, result on the host (blocking, under the watchdog)
static int comm_gather_words(wf_comm *c, uint64_t mine, std::vector<uint64_t> &all) {
    HIP_TRY(hipSetDevice(c->ctx->device));
    int rc = ensure(c->ctx, c->small, 8 * (size_t)(c->world + 1));
    if (rc) return rc;
    void *pin;
    if ((rc = comm_pinned(c, 8 * (size_t)(c->world + 1), &pin))) return rc;
    uint64_t *h = (uint64_t *)pin;  // [0] = this rank's word on its way up, [1 ..] = everybody's on their way back
    hipStream_t st = c->ctx->stream;
    uint64_t *d = (uint64_t *)c->small.p;
    h[0] = mine;
    HIP_TRY(hipMemcpyAsync(d, h, 8, hipMemcpyHostToDevice, st));
    rc = comm_all_gather(c, d, d + 1, 8, st);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h + 1, d + 1, 8 * (size_t)c->world, hipMemcpyDeviceToHost, st));
    if ((rc = comm_wait(c, st))) return rc;  // on a time-out the copy may still be queued: its destination stays allocated
    all.assign(h + 1, h + 1 + c->world);
    return 0;
}

// Collective entry points agree on a status word before their data exchange: a rank that failed locally (allocation, a
// HIP error in its gathers) reports it here, and EVERY rank returns an error instead of the others waiting inside RCCL
// for a peer that has already left the call.
int comm_agree(wf_comm *c, int local_rc, const char *what) {
    char local_msg[512];
    snprintf(local_msg, sizeof(local_msg), "%s", last_error_text());
    std::vector<uint64_t> all;
    int rc = comm_gather_words(c, (uint64_t)(uint32_t)local_rc, all);
    if (rc) return rc;
    if (local_rc) return fail(local_rc, "%s", local_msg);
    for (size_t r = 0; r < all.size(); r++)
        if (all[r]) return fail(WF_ERR_COMM, "%s failed on rank %zu with status %d", what, r, (int)(int32_t)(uint32_t)all[r]);
    return 0;
}

extern "C" {

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

int wf_comm_gather_f64(wf_comm *c, double value, double *all_out) {
    if (!c || !all_out) return fail(WF_ERR_ARG, "null argument");
    uint64_t bits;
    memcpy(&bits, &value, 8);
    std::vector<uint64_t> all;
    int rc = comm_gather_words(c, bits, all);
    if (rc) return rc;
    memcpy(all_out, all.data(), 8 * all.size());
    return 0;
}

int wf_comm_info(const wf_comm *c, int *transport, int *nccl_count, int *nccl_user_rank, int *nccl_device) {
    if (!c) return fail(WF_ERR_ARG, "comm is null");
    int count = c->world, user = c->rank, dev = c->ctx->device;
    if (transport) *transport = c->custom ? 1 : 0;
    if (!c->custom && ((ncclComm_t &)const_cast<wf_comm *>(c)->nccl)) {
        // what RCCL itself says about this communicator (not what the caller passed to wf_comm_create)
        wfcomm::Rccl *R = wfcomm::rccl();
        ncclComm_t comm = (ncclComm_t &)const_cast<wf_comm *>(c)->nccl;
        if (R && R->CommCount) RCCL_TRY(R->CommCount(comm, &count));
        if (R && R->CommUserRank) RCCL_TRY(R->CommUserRank(comm, &user));
        if (R && R->CommCuDevice) RCCL_TRY(R->CommCuDevice(comm, &dev));
    }
    if (nccl_count) *nccl_count = count;
    if (nccl_user_rank) *nccl_user_rank = user;
    if (nccl_device) *nccl_device = dev;
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
    return comm_interleave(st, c->stage.p, d_leaves, trace_len, (uint32_t)c->world, cosets_per_rank);
}

}  // extern "C"
