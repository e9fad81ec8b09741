// Shared host-side declarations of libwf_lde.so's translation units (nothing here is part of the C ABI):
//   context.hip   errors, wf_ctx, call guard, buffer pool, profiling marks, column upload / download, validation, sizes
//   path.hip      planner, twiddle tables, every launch of the commitment path, the C ABI of the path and of math::fft
//   resident.hip  wf_commitment (resident and asynchronous forms), query service, out-of-domain evaluation,
//                 the resident sharded commitment
//   fri.hip       FRI layer commitments and the resident FRI prover
//   comm.hip      wf_comm: RCCL / caller transport, partition rules
//   deep.hip      DEEP composition polynomial
// A function that launches kernels lives in exactly one unit (kernels are templates / static functions of the headers, so
// each is compiled once, where it is launched); the other units reach it through the non-template entry points below.
#pragma once

#include "../../include/wf_lde.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <tuple>
#include <type_traits>
#include <vector>

// ------------------------------------------------------------------------------------------------- errors (context.hip)
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
const char *last_error_text();
uint64_t fail_epoch();  // advanced by every fail(): see ensure_tickets (path.hip)

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(WF_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// ------------------------------------------------------------------------------------------------- context
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct TableSet {  // device-resident Pow2L tables
    void *lo = nullptr, *hi = nullptr;
    uint32_t s = 0, mask = 0;
    uint64_t lo_stride = 0, hi_stride = 0;  // per-coset strides (elements)
};

// Test / tuning switches (WF_EXP_* environment variables): they force the planner's and launcher's ALTERNATIVES (all
// product code: every one is the default for some shape) so that the parity suites reach each of them on small inputs.
// Read ONCE, when a context is created, and only when WF_EXP_ENABLE=1 is set as well (a stray WF_EXP_* variable in a
// production environment changes nothing); values are range-checked.  The launch path consults ctx->tune and never the
// environment.  All off by default; tests/test_gpu_plans.py forces each one.
struct wf_tuning {
    uint32_t max_digit = 0;           // WF_EXP_MAX_DIGIT: cap the digit size (more, smaller passes); 0 = planner's own
    bool full_tiles = false;          // WF_EXP_FULL_TILES: allow plans of two maximal digits (one work-group per CU in both passes)
    bool no_specialized = false;      // WF_EXP_NO_SPECIALIZED: generic strided pass instead of the 2^10-row instantiation
    bool no_fused_hash = false;       // WF_EXP_NO_FUSED_HASH: leaves always from k_hash_rows
    bool no_chunked = false;          // WF_EXP_NO_CHUNKED: long rows hashed by the separate chunk kernels
    bool persistent_always = false;   // WF_EXP_PERSISTENT_ALWAYS: the ticket kernel on every shape it can run
    bool no_persistent = false;       // WF_EXP_NO_PERSISTENT: one work-group per tile instead of the ticket kernel
    uint32_t merkle_l2_min = 18;      // WF_EXP_MERKLE_L2_MIN: log2 of the narrowest level the two-level launches take
    bool no_pipeline = false;         // WF_EXP_NO_PIPELINE: host columns uploaded in front of the kernels, never under them
    size_t pipeline_min_bytes = (size_t)1 << 20;  // WF_EXP_PIPELINE_MIN_BYTES
    bool no_tail_pack = false;        // WF_EXP_NO_TAIL_PACK: a half-empty last segment evaluated like the others (not coset-packed)
    bool no_coset_inner = false;      // WF_EXP_NO_COSET_INNER: first strided evaluation pass with the coset as the outermost tile index
    bool no_gtab1_wide = false;       // WF_EXP_NO_GTAB1_WIDE: the first WIDE strided evaluation pass (three-pass plans) rebuilds its output factors per tile
    bool no_ftab = false;             // WF_EXP_NO_FTAB: single-pass f128 evaluations rebuild their input factors per tile
    bool no_gtab1 = false;            // WF_EXP_NO_GTAB1: the first strided evaluation pass rebuilds its output factors per tile (h_c^i on the output side)
    bool gtab1_f64 = false;           // WF_EXP_GTAB1_F64: ... from the global table for f64 too (default: f128 only, where it measures faster)
    bool no_staged_chunks = false;    // WF_EXP_NO_STAGED_CHUNKS: rows longer than a BLAKE3 chunk hashed by k_hash_chunks (a lane walks its own row) instead of k_hash_chunks_staged
    bool no_gtab = false;             // WF_EXP_NO_GTAB: later wide strided passes rebuild their output factors in LDS per tile
    uint32_t wide_ti = 0;             // WF_EXP_WIDE_TI: inner positions per tile of the wide strided pass of 3+-pass plans (2, 4, 8); 1 = never; 0 = planner's own (2)
    int fail_after_segment = -1;      // WF_EXP_FAIL_AFTER_SEGMENT: the pipelined upload fails after that many segments (error-path test)
};
wf_tuning tuning_from_env();

struct wf_ctx {
    int device = 0;
    int num_cus = 256;  // compute units of the device: sizes the persistent grid of k_seg_last_hash
    wf_tuning tune;
    // the hasher of the entry points that take no wf_params (wf_hash_rows, wf_merkle_build*, wf_fri_*): 32 = Blake3_256
    // (default), 24 = Blake3_192; wf_ctx_set_digest_bytes.  Commitments carry theirs in wf_params::digest_bytes.
    uint32_t digest_bytes = 32;
    DevBuf pack_tmp;  // 24-byte digests on their way between host arrays and the device's 32-byte slots
    hipStream_t stream = nullptr;
    // key: (field, logN, kind, aux, offset lo, offset hi); kind 0 = forward root, 1 = inverse root,
    // 2 = coset bases (aux = log blowup), 3 = output series for interpolate_with_offset
    std::map<std::tuple<int, int, int, int, uint64_t, uint64_t>, TableSet> tables;
    // optional per-launch timing (wf_ctx_profile_*): an event is recorded in front of every kernel launch
    int prof_level = 0;  // 0 off, 1 one event per logical kernel (interpolate / evaluate / hash_rows / merkle), 2 per launch
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;
    std::vector<const char *> prof_name;
    size_t prof_n = 0;
    DevBuf scratch;   // evaluation intermediate [cosets][columns][R]
    DevBuf io[5];     // staging for the host-buffer API: trace, polys, lde, leaves, nodes
    DevBuf hash_tmp;  // chunk chaining values of rows longer than one BLAKE3 chunk
    DevBuf tickets;   // per-XCD ticket + exit counters of the persistent last passes (self-resetting; cleared again after any reported failure)
    uint64_t tickets_epoch = 0;  // fail_epoch() when the counters were last cleared
    // Buffers of destroyed resident commitments, kept for the next commitment of the same shape (four hipFree + four
    // hipMalloc of 64..512 MiB cost about as much as the commitment itself); released by wf_ctx_release_cached / destroy.
    // Guarded by pool_mutex: a handle may be destroyed by another thread (a finaliser, Rust's Drop) while a call runs.
    std::mutex pool_mutex;
    std::vector<std::pair<void *, size_t>> pool;
    size_t pool_bytes = 0, pool_cap = 0;  // pool_cap: a quarter of the device's memory (set at creation)
    hipStream_t copy_stream = nullptr;  // uploads that run under kernels (trace_commit_pipelined, the asynchronous form)
    std::vector<hipEvent_t> seg_events;
    void *pin = nullptr;  // pinned host staging for uploads of many small columns (upload_columns)
    size_t pin_cap = 0;
    void *qpin = nullptr;  // pinned staging of the query service (ids up, rows and digests back)
    size_t qpin_cap = 0;
    // One call at a time: the thread inside an entry point (0 = none) and its nesting depth.  A second thread entering
    // while a call is in progress gets WF_ERR_BUSY instead of corrupting scratch and ticket counters.
    std::atomic<uintptr_t> owner{0};
    int depth = 0;
    // The stream the last asynchronous call was issued on.  Scratch, chunk chaining values and ticket counters belong
    // to the context, so a call on ANOTHER stream first waits (on the device) for everything queued on that one.
    hipStream_t last_stream = nullptr;
    hipEvent_t order_ev = nullptr;
    // A stream of proofs from host memory (wf_trace_commit_resident_async): two input staging buffers, so that proof
    // k + 1 goes up on the copy stream while the kernels of proof k run; stage_free[i] is recorded on the compute stream
    // behind the one kernel that reads staging buffer i, upload_done[i] on the copy stream behind its upload.  The roots
    // come back through a ring of pinned 32-byte slots (a copy into pageable memory would block the host until the
    // kernels in front of it have finished).
    DevBuf stage[2];
    hipEvent_t stage_free[2] = {nullptr, nullptr}, upload_done[2] = {nullptr, nullptr};
    bool stage_busy[2] = {false, false};
    uint64_t async_seq = 0;
    uint8_t *root_pin = nullptr;       // WF_ROOT_SLOTS slots of WF_ROOT_SLOT_BYTES: the root of an asynchronous commitment (pinned: wf_commitment_wait reads it without a copy)
    std::vector<uint8_t> root_used;    // guarded by pool_mutex (slots are handed back by whatever thread completes a handle)
    uint64_t generation = 0;  // distinguishes this context from an earlier one at the same address (stale handles)
};
static constexpr size_t WF_ROOT_SLOTS = 256, WF_ROOT_SLOT_BYTES = 64;

// RAII entry of every ctx-taking entry point (see the two comments above).  Re-entrant for the owning thread: the
// host-buffer forms call the device-buffer forms.
struct CallGuard {
    wf_ctx *ctx = nullptr;
    int rc = 0;
    static uintptr_t self() {
        static thread_local char token;
        return (uintptr_t)&token;
    }
    CallGuard(wf_ctx *c, hipStream_t st) {
        uintptr_t expected = 0;
        if (c->owner.compare_exchange_strong(expected, self()))
            c->depth = 1;
        else if (expected == self())
            c->depth++;
        else {
            rc = fail(WF_ERR_BUSY, "the context is in use by another thread (a wf_ctx serves one call at a time)");
            return;
        }
        ctx = c;
        if (c->depth == 1 && st) {
            if (c->last_stream && c->last_stream != st) {
                // (a stream the caller has destroyed since is refused by hipEventRecord; ROCm drains a stream when it
                // destroys it, so there is nothing left to wait for)
                hipError_t e = hipSuccess;
                if (!c->order_ev) e = hipEventCreateWithFlags(&c->order_ev, hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventRecord(c->order_ev, c->last_stream);
                if (e == hipSuccess) e = hipStreamWaitEvent(st, c->order_ev, 0);
                if (e != hipSuccess) (void)hipGetLastError();
            }
            c->last_stream = st;
        }
    }
    void release() {
        if (ctx && --ctx->depth == 0) ctx->owner.store(0);
        ctx = nullptr;
    }
    ~CallGuard() { release(); }
};
// stream == nullptr: the call does not touch the device (or synchronises before it returns on the context's stream)
#define WF_ENTER(ctx_, st_)              \
    CallGuard _guard((ctx_), (st_));     \
    if (_guard.rc) return _guard.rc

// Contexts that exist.  The rule of the ABI is "destroy commitments and provers first, their context last"; a handle
// destroyed after its context (hosts with garbage collectors do this at shutdown) must not touch the dead context's
// pool or stream: its destroy function checks here -- by address AND generation, so that a new context that happens to
// sit at the old address is not mistaken for the handle's own -- and frees its device buffers directly.
bool ctx_alive(const wf_ctx *ctx);
bool ctx_alive(const wf_ctx *ctx, uint64_t generation);
// holds the registry's lock for its lifetime; `alive`: the context (address AND generation) exists and cannot be retired
// by wf_ctx_destroy while this object lives
struct CtxPin {
    std::unique_lock<std::mutex> lock;
    bool alive;
    CtxPin(const wf_ctx *ctx, uint64_t generation);
};

hipError_t dev_malloc(wf_ctx *ctx, void **p, size_t bytes);
hipError_t pool_alloc(wf_ctx *ctx, void **p, size_t bytes);
void pool_free(wf_ctx *ctx, uint64_t generation, void *p, size_t bytes);
void prof_mark(wf_ctx *ctx, hipStream_t st, const char *name);
int ensure(wf_ctx *ctx, DevBuf &b, size_t bytes);
int upload_columns(wf_ctx *ctx, void *dst, const void *const *cols, size_t n, size_t colb, hipStream_t st);
int download_columns(wf_ctx *ctx, void *const *cols, const void *src, size_t n, size_t colb, hipStream_t st);
int check_params(const wf_params *p, bool constraint);

// ------------------------------------------------------------------------------------------------- comm.hip
struct wf_comm {
    wf_ctx *ctx = nullptr;
    uint64_t ctx_generation = 0;
    int rank = 0, world = 1;
    void *nccl = nullptr;       // ncclComm_t of the RCCL transport
    bool custom = false;        // caller-supplied transport
    wf_transport tr{};
    DevBuf stage;  // receive staging of the leaf exchanges ([world][...] rank-major, before the interleave)
    DevBuf small;  // barrier / reduction words
    void *pin = nullptr;  // pinned host memory the host-blocking calls copy back into (comm_pinned); freed by wf_comm_destroy only
    size_t pin_cap = 0;
    // Watchdog of the host-blocking calls (barrier, reductions, collective queries, wf_comm_stream_wait): a rank that died
    // or never arrives must not leave the others waiting for ever.  WF_COMM_TIMEOUT_S, read once at creation.
    double timeout_s = 300.0;
    bool dead = false;  // a collective timed out or failed asynchronously: the communicator was aborted
};
// every rank contributes `bytes` at d_send; d_recv receives world * bytes, rank-major
int comm_all_gather(wf_comm *c, const void *d_send, void *d_recv, size_t bytes, hipStream_t st);
// block s (`bytes` bytes at d_send + s * bytes) of rank r lands at d_recv + r * bytes on rank s
int comm_all_to_all(wf_comm *c, const void *d_send, void *d_recv, size_t bytes, hipStream_t st);
int comm_wait(wf_comm *c, hipStream_t st);                         // host-blocking, under the watchdog
int comm_pinned(wf_comm *c, size_t bytes, void **out);             // communicator-owned pinned memory of at least `bytes`
int comm_agree(wf_comm *c, int local_rc, const char *what);        // every rank returns an error if one of them failed locally
int comm_interleave(hipStream_t st, const void *src, void *dst, uint64_t n_k, uint32_t world, uint32_t per);

// ------------------------------------------------------------------------------------------------- the path (path.hip)
// Non-template entry points of the commitment pipeline for the other units (field dispatch inside).
// Prover::build_trace_commitment on device buffers; input_read (optional) is recorded behind the one kernel that reads d_trace
int path_trace_commit(wf_ctx *ctx, const wf_params *p, const void *d_trace, void *d_polys, void *d_lde, void *d_leaves,
                      void *d_nodes, hipStream_t st, hipEvent_t input_read = nullptr);
// the same from HOST columns of a matrix of several segments, the upload running under the kernels
int path_trace_commit_pipelined(wf_ctx *ctx, const wf_params *p, const void *const *cols_in, void *d_stage, void *d_polys,
                                void *d_lde, void *d_leaves, void *d_nodes, hipStream_t st, void *const *polys_out);
bool path_pipelined_upload_ok(const wf_ctx *ctx, const wf_params *p, size_t colb);
// Prover::build_constraint_commitment on device buffers (dense_rows: rows of exactly n_cols * ext elements)
int path_constraint_commit(wf_ctx *ctx, const wf_params *p, const void *d_polys, void *d_lde, void *d_leaves, void *d_nodes,
                           hipStream_t st, bool dense_rows = false);
bool path_dense_column_ok(const wf_params *p);
bool path_dense_matrix_ok(const wf_params *p);
// (digest_bytes: 32 or 24; device leaves / nodes are 32-byte slots either way, a 24-byte digest in the first 24 bytes)
int path_hash_rows(wf_ctx *ctx, hipStream_t st, uint32_t field, const void *lde, uint64_t trace_elems, uint64_t n_rows,
                   uint32_t row_width, uint32_t epr, uint32_t n_traces, void *leaves, uint32_t digest_bytes);
int path_merkle(wf_ctx *ctx, hipStream_t st, const void *leaves, uint64_t n_leaves, void *nodes, uint32_t digest_bytes);
// n digests from device slots to a HOST array of digest_bytes-sized entries (asynchronous on st; 24-byte digests are packed
// on the device first) and the way back (host entries -> device slots)
int path_digests_to_host(wf_ctx *ctx, hipStream_t st, const void *d_slots, void *host, size_t n, uint32_t digest_bytes);
int path_digests_from_host(wf_ctx *ctx, hipStream_t st, const void *host, void *d_slots, size_t n, uint32_t digest_bytes);
// host-side copy of n digests out of a buffer of 32-byte slots
static inline void copy_digests_out(void *dst, const void *slots, size_t n, uint32_t digest_bytes) {
    if (digest_bytes == 32) {
        memcpy(dst, slots, n * 32);
        return;
    }
    for (size_t i = 0; i < n; i++) memcpy((char *)dst + i * digest_bytes, (const char *)slots + i * 32, digest_bytes);
}
// one packed commitment sharded over the ranks of a communicator (segment-sharded interpolation, coset-sharded evaluation)
int path_trace_commit_sharded(wf_comm *c, const wf_params *p, const void *d_trace, void *d_polys, void *d_lde_shard,
                              void *d_leaves, void *d_nodes, void *d_top, hipStream_t st);

// ------------------------------------------------------------------------------------------------- resident.hip
struct wf_commitment {
    wf_ctx *ctx;
    uint64_t ctx_generation;
    wf_params p;
    void *lde, *leaves, *nodes, *polys;
    uint64_t n_rows, row_width, epr, row_elems;
    uint32_t depth;
    uint8_t root[32];
    bool borrowed;  // lde / leaves / nodes live in an arena of their owner (FRI layers): not freed one by one
    size_t lde_bytes, dig_bytes, polys_bytes;  // allocation sizes when they come from the context's buffer pool (else 0)
    // wf_trace_commit_resident_async: the kernels may still be running; `done` is recorded behind the copy of the root into
    // pinned slot root_slot1 - 1 of the context; wf_commitment_wait (or the first wf_commitment_root) completes the handle
    bool pending;
    hipEvent_t done;
    uint32_t root_slot1;
    int device;  // of its context (a handle may be completed after its context is gone)
};
void free_commitment(wf_commitment *c);
int commitment_alloc(wf_ctx *ctx, const wf_params *p, bool constraint, wf_commitment **out, bool *dense_out);
wf_commitment *commitment_new(wf_ctx *ctx);  // zeroed handle bound to ctx (FRI layers fill it themselves)

// ------------------------------------------------------------------------------------------------- fri.hip
// (the DEEP composition hands its polynomial straight to a prover: deep.hip reads its field / extension / offset)
struct wf_fri_prover {
    wf_ctx *ctx = nullptr;
    uint64_t ctx_generation = 0;
    uint32_t field = 0, ext = 0, folding = 0, blowup = 0, remainder_max_degree = 0;
    uint8_t offset[16] = {0};
    void *evals = nullptr;  // evaluations of the current layer (device)
    bool evals_borrowed = false;
    size_t n = 0;
    wf_commitment *pending = nullptr;  // committed, not yet folded
    std::vector<wf_commitment *> layers;
    // One allocation for all layers of a proof (hipMalloc / hipFree per layer cost more than the layers' kernels from the
    // third layer on); kept across wf_fri_prover_reset, released by wf_fri_prover_destroy.
    DevBuf arena;
    size_t arena_used = 0;
};
// `poly`: n coefficients of E in host memory, or (poly_on_device) in device memory of the prover's context
int fri_begin_poly_impl(wf_fri_prover *pr, const void *poly, bool poly_on_device, size_t n, size_t lde_blowup);

