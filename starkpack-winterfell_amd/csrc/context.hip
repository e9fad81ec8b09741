// libwf_lde.so, unit 1 of 6 -- the context: errors, the registry of live contexts, the buffer pool, profiling marks,
// column upload / download, parameter validation and the size helpers of the C ABI (include/wf_lde.h).
// There is deliberately no CPU fallback: every compute entry point needs a HIP device.
#include "wf_internal.hpp"

#include <atomic>

#include "field.hpp"

using namespace wf;

// ------------------------------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";

// Every reported failure advances this epoch.  The per-XCD ticket counters of the persistent last passes reset themselves only when
// every work-group of a launch signs off; a context whose call failed anywhere in between (a launch error after the counters were
// handed out, a HIP error of a later launch) clears them before their next use instead of trusting them: ensure_tickets compares.
std::atomic<uint64_t> g_fail_epoch{0};
uint64_t fail_epoch() { return g_fail_epoch.load(std::memory_order_relaxed); }

int fail(int code, const char *fmt, ...) {
    g_fail_epoch.fetch_add(1, std::memory_order_relaxed);
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

const char *last_error_text() { return g_err; }

// WF_EXP_* switches: read here, once per context, and only under WF_EXP_ENABLE=1 (see wf_tuning); out-of-range values are
// ignored (the planner holds at most four digits: a digit cap below 4 bits cannot plan 2^16 rows and up)
wf_tuning tuning_from_env() {
    wf_tuning t;
    const char *on = getenv("WF_EXP_ENABLE");
    if (!on || strcmp(on, "1") != 0) return t;
    if (const char *e = getenv("WF_EXP_MAX_DIGIT")) {
        const int v = atoi(e);
        if (v >= 4 && v <= 11) t.max_digit = (uint32_t)v;
    }
    t.no_specialized = getenv("WF_EXP_NO_SPECIALIZED") != nullptr;
    t.full_tiles = getenv("WF_EXP_FULL_TILES") != nullptr;
    t.no_fused_hash = getenv("WF_EXP_NO_FUSED_HASH") != nullptr;
    t.no_chunked = getenv("WF_EXP_NO_CHUNKED") != nullptr;
    t.persistent_always = getenv("WF_EXP_PERSISTENT_ALWAYS") != nullptr;
    t.no_persistent = getenv("WF_EXP_NO_PERSISTENT") != nullptr;
    if (const char *e = getenv("WF_EXP_MERKLE_L2_MIN")) {
        const int v = atoi(e);
        if (v >= 10 && v <= 30) t.merkle_l2_min = (uint32_t)v;
    }
    t.no_pipeline = getenv("WF_EXP_NO_PIPELINE") != nullptr;
    t.no_tail_pack = getenv("WF_EXP_NO_TAIL_PACK") != nullptr;
    t.no_coset_inner = getenv("WF_EXP_NO_COSET_INNER") != nullptr;
    t.no_gtab = getenv("WF_EXP_NO_GTAB") != nullptr;
    t.no_staged_chunks = getenv("WF_EXP_NO_STAGED_CHUNKS") != nullptr;
    t.no_gtab1 = getenv("WF_EXP_NO_GTAB1") != nullptr;
    t.no_ftab = getenv("WF_EXP_NO_FTAB") != nullptr;
    t.no_gtab1_wide = getenv("WF_EXP_NO_GTAB1_WIDE") != nullptr;
    t.gtab1_f64 = getenv("WF_EXP_GTAB1_F64") != nullptr;
    if (const char *e = getenv("WF_EXP_WIDE_TI")) {
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4 || v == 8) t.wide_ti = (uint32_t)v;
    }
    if (const char *e = getenv("WF_EXP_PIPELINE_MIN_BYTES")) {
        const long long v = atoll(e);
        if (v >= 0 && v <= (1ll << 40)) t.pipeline_min_bytes = (size_t)v;
    }
    if (const char *e = getenv("WF_EXP_FAIL_AFTER_SEGMENT")) {  // error-path test of the pipelined upload
        const int v = atoi(e);
        if (v >= 0 && v < 4096) t.fail_after_segment = v;
    }
    return t;
}

// Contexts that exist.  The rule of the ABI is "destroy commitments and provers first, their context last"; a handle
// destroyed after its context (hosts with garbage collectors do this at shutdown) must not touch the dead context's
// pool or stream: its destroy function checks here and frees its device buffers directly.
static std::mutex g_ctx_mutex;
static std::set<const wf_ctx *> g_live_ctx;
static uint64_t g_next_generation = 1;
bool ctx_alive(const wf_ctx *ctx) {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    return g_live_ctx.count(ctx) != 0;
}
bool ctx_alive(const wf_ctx *ctx, uint64_t generation) {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    return g_live_ctx.count(ctx) != 0 && ctx->generation == generation;
}
// The registry's lock, for the few places that must keep a context alive while they touch it from a thread that does not
// own a call on it (a handle destroyed or waited for on a finaliser thread): wf_ctx_destroy takes the same lock to retire
// the context, so whoever holds it and finds the context in the registry may use the context's pool and pinned slots.
CtxPin::CtxPin(const wf_ctx *ctx, uint64_t generation) : lock(g_ctx_mutex) {
    alive = g_live_ctx.count(ctx) != 0 && ctx->generation == generation;
}

// hipMalloc that gives the context's parked buffers back to the driver and retries once when the device is full
hipError_t dev_malloc(wf_ctx *ctx, void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess && ctx) {
        std::unique_lock<std::mutex> lock(ctx->pool_mutex);
        if (!ctx->pool.empty()) {
            (void)hipGetLastError();
            for (auto &b : ctx->pool) (void)hipFree(b.first);
            ctx->pool.clear();
            ctx->pool_bytes = 0;
            lock.unlock();
            e = hipMalloc(p, bytes);
        }
    }
    if (e != hipSuccess) (void)hipGetLastError();  // the failure is reported through the return value; leave no sticky error
    return e;
}

hipError_t pool_alloc(wf_ctx *ctx, void **p, size_t bytes) {
    {
        std::lock_guard<std::mutex> lock(ctx->pool_mutex);
        for (size_t i = 0; i < ctx->pool.size(); i++)
            if (ctx->pool[i].second == bytes) {
                *p = ctx->pool[i].first;
                ctx->pool.erase(ctx->pool.begin() + i);
                ctx->pool_bytes -= bytes;
                return hipSuccess;
            }
    }
    return dev_malloc(ctx, p, bytes);
}

// Parks a buffer for the next commitment of the same shape.  The pool is bounded by entries and by bytes (pool_cap: a
// quarter of the device's memory, set when the context is created): the oldest entries are released first, so buffers of
// shapes that never come back do not pile up.  Takes the pool's own mutex: handles are destroyed by whatever thread the
// host's finalisers run on, possibly while a call of another thread is in progress on the context.
void pool_free(wf_ctx *ctx, uint64_t generation, void *p, size_t bytes) {
    if (!p) return;
    std::vector<void *> drop;
    {
        // the registry stays locked while the context is touched: a concurrent wf_ctx_destroy waits until the buffer is parked
        // (and then releases it with the rest of the pool) or has already retired the context (the buffer is freed directly)
        CtxPin pin(ctx, generation);
        if (!pin.alive || !bytes || bytes > ctx->pool_cap) {  // (not alive: the handle outlived its context)
            drop.push_back(p);
        } else {
            std::lock_guard<std::mutex> lock(ctx->pool_mutex);
            ctx->pool.emplace_back(p, bytes);
            ctx->pool_bytes += bytes;
            while (!ctx->pool.empty() && (ctx->pool.size() > 16 || ctx->pool_bytes > ctx->pool_cap)) {
                drop.push_back(ctx->pool.front().first);
                ctx->pool_bytes -= ctx->pool.front().second;
                ctx->pool.erase(ctx->pool.begin());
            }
        }
    }
    for (void *d : drop) (void)hipFree(d);
}

// logical kernel of a mark: the text before the first '.', with the layout changes counted as interpolation
static int prof_group(const char *name) {
    if (!strncmp(name, "layout", 6) || !strncmp(name, "interpolate", 11)) return 1;
    if (!strncmp(name, "evaluate", 8)) return 2;
    if (!strncmp(name, "hash_rows", 9)) return 3;
    if (!strncmp(name, "merkle", 6)) return 4;
    if (!strncmp(name, "between_calls", 13)) return 5;
    return 6 + (int)(unsigned char)name[0] + 256 * (int)(unsigned char)name[4];
}

void prof_mark(wf_ctx *ctx, hipStream_t st, const char *name) {
    if (!ctx->prof_on) return;
    if (ctx->prof_level == 1 && ctx->prof_n > 0 && prof_group(ctx->prof_name[ctx->prof_n - 1]) == prof_group(name)) return;
    if (ctx->prof_level == 1) {  // coarse marks carry the logical kernel's name
        switch (prof_group(name)) {
            case 1: name = "interpolate"; break;
            case 2: name = "evaluate"; break;
            default: break;
        }
    }
    if (ctx->prof_n == ctx->prof_ev.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        ctx->prof_ev.push_back(e);
        ctx->prof_name.push_back(name);
    }
    ctx->prof_name[ctx->prof_n] = name;
    (void)hipEventRecord(ctx->prof_ev[ctx->prof_n], st);
    ctx->prof_n++;
}

int ensure(wf_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) {
        HIP_TRY(hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    hipError_t e = dev_malloc(ctx, &b.p, bytes);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(WF_ERR_HIP, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    }
    b.cap = bytes;
    return 0;
}

// Host columns ([n] separate allocations of `colb` bytes, the reference's Vec<Vec<E>>) -> one device buffer.  Large
// columns go straight through hipMemcpyAsync (pageable copies of MiBs run at PCIe speed); thousands of small ones (512
// packed traces of 2^10 steps: 5120 columns of 16 KiB) cost ~4.5 us per call that way -- they are gathered into a
// pinned staging buffer by a few host threads and sent in 32 MiB pieces (24 -> ~7 ms for that case).
int upload_columns(wf_ctx *ctx, void *dst, const void *const *cols, size_t n, size_t colb, hipStream_t st) {
    if (colb >= ((size_t)1 << 20) || n < 16) {
        for (size_t i = 0; i < n; i++)
            if (hipMemcpyAsync((char *)dst + i * colb, cols[i], colb, hipMemcpyHostToDevice, st) != hipSuccess)
                return fail(WF_ERR_HIP, "hipMemcpyAsync failed: %s", hipGetErrorString(hipGetLastError()));
        return 0;
    }
    const size_t piece = (size_t)32 << 20;
    if (!ctx->pin) {
        if (hipHostMalloc(&ctx->pin, 2 * piece, hipHostMallocDefault) != hipSuccess)
            return fail(WF_ERR_HIP, "hipHostMalloc failed: %s", hipGetErrorString(hipGetLastError()));
        ctx->pin_cap = 2 * piece;
    }
    const size_t per = std::max<size_t>(1, piece / colb);  // columns per piece
    hipEvent_t done[2] = {nullptr, nullptr};
    int rc = 0;
    for (size_t i0 = 0, k = 0; i0 < n && rc == 0; i0 += per, k++) {
        const size_t cnt = std::min(per, n - i0), half = k & 1;
        char *stage = (char *)ctx->pin + half * piece;
        if (done[half]) (void)hipEventSynchronize(done[half]);  // the piece sent from this half two rounds ago has left
        const unsigned nt = (unsigned)std::min<size_t>(8, std::max<size_t>(1, cnt * colb >> 20));
        auto work = [&](unsigned t) {
            for (size_t j = t; j < cnt; j += nt) memcpy(stage + j * colb, cols[i0 + j], colb);
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        if (hipMemcpyAsync((char *)dst + i0 * colb, stage, cnt * colb, hipMemcpyHostToDevice, st) != hipSuccess)
            rc = fail(WF_ERR_HIP, "hipMemcpyAsync failed: %s", hipGetErrorString(hipGetLastError()));
        if (rc == 0 && !done[half] && hipEventCreateWithFlags(&done[half], hipEventDisableTiming) != hipSuccess)
            rc = fail(WF_ERR_HIP, "hipEventCreate failed");
        if (rc == 0) (void)hipEventRecord(done[half], st);
    }
    for (auto e : done)
        if (e) {
            (void)hipEventSynchronize(e);  // the staging buffer is free again when this returns
            (void)hipEventDestroy(e);
        }
    return rc;
}

// The way back (polynomial columns to the caller's separate allocations; null entries are skipped).  Synchronous for the
// staged route (the scatter into the caller's columns happens on the host), asynchronous on `st` for large columns.
int download_columns(wf_ctx *ctx, void *const *cols, const void *src, size_t n, size_t colb, hipStream_t st) {
    if (colb >= ((size_t)1 << 20) || n < 16) {
        for (size_t i = 0; i < n; i++)
            if (cols[i] && hipMemcpyAsync(cols[i], (const char *)src + i * colb, colb, hipMemcpyDeviceToHost, st) != hipSuccess)
                return fail(WF_ERR_HIP, "hipMemcpyAsync failed: %s", hipGetErrorString(hipGetLastError()));
        return 0;
    }
    const size_t piece = (size_t)32 << 20;
    if (!ctx->pin) {
        if (hipHostMalloc(&ctx->pin, 2 * piece, hipHostMallocDefault) != hipSuccess)
            return fail(WF_ERR_HIP, "hipHostMalloc failed: %s", hipGetErrorString(hipGetLastError()));
        ctx->pin_cap = 2 * piece;
    }
    const size_t per = std::max<size_t>(1, piece / colb);
    for (size_t i0 = 0; i0 < n; i0 += per) {
        const size_t cnt = std::min(per, n - i0);
        if (hipMemcpyAsync(ctx->pin, (const char *)src + i0 * colb, cnt * colb, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            return fail(WF_ERR_HIP, "download failed: %s", hipGetErrorString(hipGetLastError()));
        const unsigned nt = (unsigned)std::min<size_t>(8, std::max<size_t>(1, cnt * colb >> 20));
        auto work = [&](unsigned t) {
            for (size_t j = t; j < cnt; j += nt)
                if (cols[i0 + j]) memcpy(cols[i0 + j], (const char *)ctx->pin + j * colb, colb);
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------- validation
int check_params(const wf_params *p, bool constraint) {
    if (!p) return fail(WF_ERR_ARG, "params is null");
    if (p->field != WF_FIELD_F64 && p->field != WF_FIELD_F128) return fail(WF_ERR_FIELD, "unknown field id %u", p->field);
    if (p->ext_degree < 1 || p->ext_degree > 3 || (p->field == WF_FIELD_F128 && p->ext_degree == 3))
        return fail(WF_ERR_EXTENSION, "unsupported extension degree %u for field %u", p->ext_degree, p->field);
    if (p->log2_trace_len < 3) return fail(WF_ERR_TRACE_LENGTH, "trace length must be at least 8");
    if (p->log2_blowup < 1 || p->log2_blowup > 7) return fail(WF_ERR_BLOWUP, "blowup must be a power of two in [2,128]");
    const uint32_t adicity = p->field == WF_FIELD_F64 ? F64::TWO_ADICITY : F128::TWO_ADICITY;
    if (p->log2_trace_len + p->log2_blowup > adicity)
        return fail(WF_ERR_DOMAIN, "no multiplicative subgroup of size 2^%u in this field", p->log2_trace_len + p->log2_blowup);
    if (p->n_cols < 1 || p->n_cols > 255) return fail(WF_ERR_WIDTH, "number of columns must be in [1,255]");
    if (p->n_traces < 1 || (constraint && p->n_traces != 1)) return fail(WF_ERR_TRACES, "invalid number of traces %u", p->n_traces);
    if (p->digest_bytes != 32 && p->digest_bytes != 24)
        return fail(WF_ERR_DIGEST, "digest_bytes must be 32 (Blake3_256) or 24 (Blake3_192), got %u", p->digest_bytes);
    if (p->reserved != 0) return fail(WF_ERR_ARG, "reserved field must be zero");
    u128 off;
    memcpy(&off, p->domain_offset, 16);
    const u128 mod = p->field == WF_FIELD_F64 ? (u128)F64::P : F128::P();
    if (off == 0 || off >= mod) return fail(WF_ERR_OFFSET, "domain offset must be a non-zero field element");
    return 0;
}

// ------------------------------------------------------------------------------------------------- C ABI
extern "C" {

const char *wf_last_error(void) { return g_err; }

int wf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int wf_ctx_create(int device, wf_ctx **out) {
    if (!out) return fail(WF_ERR_ARG, "out is null");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(WF_ERR_HIP, "HIP device %d not available (%d visible)", device, n);
    HIP_TRY(hipSetDevice(device));
    wf_ctx *c = new wf_ctx();
    c->device = device;
    c->tune = tuning_from_env();
    size_t mem_free = 0, mem_total = 0;
    c->pool_cap = hipMemGetInfo(&mem_free, &mem_total) == hipSuccess && mem_total ? mem_total / 4 : (size_t)16 << 30;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->num_cus = cus;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(WF_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        c->generation = g_next_generation++;
        g_live_ctx.insert(c);
    }
    *out = c;
    return 0;
}

void wf_ctx_destroy(wf_ctx *ctx) {
    if (!ctx) return;
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        if (!g_live_ctx.erase(ctx)) return;  // not a live context (destroyed twice)
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->tables) {
        if (kv.second.lo) (void)hipFree(kv.second.lo);
        if (kv.second.hi) (void)hipFree(kv.second.hi);
    }
    if (ctx->scratch.p) (void)hipFree(ctx->scratch.p);
    for (auto &b : ctx->io)
        if (b.p) (void)hipFree(b.p);
    if (ctx->hash_tmp.p) (void)hipFree(ctx->hash_tmp.p);
    if (ctx->tickets.p) (void)hipFree(ctx->tickets.p);
    if (ctx->pack_tmp.p) (void)hipFree(ctx->pack_tmp.p);
    {
        std::lock_guard<std::mutex> lock(ctx->pool_mutex);  // (nobody can be inside pool_free any more: the context is retired)
        for (auto &b : ctx->pool) (void)hipFree(b.first);
        ctx->pool.clear();
    }
    for (hipEvent_t e : ctx->seg_events) (void)hipEventDestroy(e);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->qpin) (void)hipHostFree(ctx->qpin);
    if (ctx->root_pin) (void)hipHostFree(ctx->root_pin);
    for (int i = 0; i < 2; i++) {
        if (ctx->stage[i].p) (void)hipFree(ctx->stage[i].p);
        if (ctx->stage_free[i]) (void)hipEventDestroy(ctx->stage_free[i]);
        if (ctx->upload_done[i]) (void)hipEventDestroy(ctx->upload_done[i]);
    }
    for (auto e : ctx->prof_ev) (void)hipEventDestroy(e);
    if (ctx->order_ev) (void)hipEventDestroy(ctx->order_ev);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int wf_ctx_release_cached(wf_ctx *ctx) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);
    for (auto &b : ctx->pool) (void)hipFree(b.first);
    ctx->pool.clear();
    ctx->pool_bytes = 0;
    return 0;
}

int wf_ctx_synchronize(wf_ctx *ctx) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

void *wf_ctx_stream(wf_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int wf_ctx_profile_enable(wf_ctx *ctx, int on) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    ctx->prof_on = on != 0;
    ctx->prof_level = on == 1 ? 1 : 2;
    ctx->prof_n = 0;
    return 0;
}

int wf_ctx_profile_read(wf_ctx *ctx, int max_entries, const char **names, float *ms) {
    if (!ctx || !names || !ms) return fail(WF_ERR_ARG, "null argument");
    if (ctx->prof_n < 2) return 0;
    HIP_TRY(hipEventSynchronize(ctx->prof_ev[ctx->prof_n - 1]));
    int n = 0;
    for (size_t i = 0; i + 1 < ctx->prof_n && n < max_entries; i++, n++) {
        names[n] = ctx->prof_name[i];
        HIP_TRY(hipEventElapsedTime(&ms[n], ctx->prof_ev[i], ctx->prof_ev[i + 1]));
    }
    ctx->prof_n = 0;
    return n;
}

int wf_params_check(const wf_params *p, int is_constraint) { return check_params(p, is_constraint != 0); }

size_t wf_elem_bytes(uint32_t field) { return field == WF_FIELD_F64 ? 8 : (field == WF_FIELD_F128 ? 16 : 0); }
size_t wf_row_width(const wf_params *p) { return 8 * (((size_t)p->n_cols * p->ext_degree + 7) / 8); }
size_t wf_column_bytes(const wf_params *p) {
    return ((size_t)1 << p->log2_trace_len) * p->ext_degree * wf_elem_bytes(p->field);
}
size_t wf_lde_bytes(const wf_params *p) {
    return ((size_t)1 << (p->log2_trace_len + p->log2_blowup)) * wf_row_width(p) * wf_elem_bytes(p->field);
}
size_t wf_digests_bytes(const wf_params *p) { return ((size_t)1 << (p->log2_trace_len + p->log2_blowup)) * p->digest_bytes; }

int wf_ctx_set_digest_bytes(wf_ctx *ctx, uint32_t digest_bytes) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    if (digest_bytes != 32 && digest_bytes != 24)
        return fail(WF_ERR_DIGEST, "digest_bytes must be 32 (Blake3_256) or 24 (Blake3_192), got %u", digest_bytes);
    WF_ENTER(ctx, nullptr);
    ctx->digest_bytes = digest_bytes;
    return 0;
}


}  // extern "C"
