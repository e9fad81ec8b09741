// libwf_lde.so -- host side of the C ABI declared in include/wf_lde.h: context, twiddle tables, pass planner and
// kernel launches.  Replaces Prover::build_trace_commitment / build_constraint_commitment
// (/root/reference/prover/src/lib.rs:615-715) and the math::fft / crypto building blocks they call.
// There is deliberately no CPU fallback: every compute entry point needs a HIP device.
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

#include "kernels.hpp"
#include "seg_kernels.hpp"
#include "fri_kernels.hpp"

using namespace wf;

// ------------------------------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

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

struct wf_ctx {
    int device = 0;
    int num_cus = 256;  // compute units of the device: sizes the persistent grid of k_seg_last_hash
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
    DevBuf tickets;   // per-XCD tile counters of the persistent last pass
    // Buffers of destroyed resident commitments, kept for the next commitment of the same shape (four hipFree + four
    // hipMalloc of 64..512 MiB cost about as much as the commitment itself); released by wf_ctx_release_cached / destroy.
    std::vector<std::pair<void *, size_t>> pool;
    size_t pool_bytes = 0, pool_cap = (size_t)64 << 30;
    hipStream_t copy_stream = nullptr;  // uploads that run under kernels (trace_commit_pipelined)
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
    uint8_t *root_pin = nullptr;
    std::vector<uint8_t> root_used;
};
static constexpr size_t WF_ROOT_SLOTS = 256;

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
// pool or stream: its destroy function checks here and frees its device buffers directly.
static std::mutex g_ctx_mutex;
static std::set<const wf_ctx *> g_live_ctx;
static bool ctx_alive(const wf_ctx *ctx) {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    return g_live_ctx.count(ctx) != 0;
}

// hipMalloc that gives the context's parked buffers back to the driver and retries once when the device is full
static hipError_t dev_malloc(wf_ctx *ctx, void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess && ctx && !ctx->pool.empty()) {
        (void)hipGetLastError();
        for (auto &b : ctx->pool) (void)hipFree(b.first);
        ctx->pool.clear();
        ctx->pool_bytes = 0;
        e = hipMalloc(p, bytes);
    }
    if (e != hipSuccess) (void)hipGetLastError();  // the failure is reported through the return value; leave no sticky error
    return e;
}

static hipError_t pool_alloc(wf_ctx *ctx, void **p, size_t bytes) {
    for (size_t i = 0; i < ctx->pool.size(); i++)
        if (ctx->pool[i].second == bytes) {
            *p = ctx->pool[i].first;
            ctx->pool.erase(ctx->pool.begin() + i);
            ctx->pool_bytes -= bytes;
            return hipSuccess;
        }
    return dev_malloc(ctx, p, bytes);
}

// Parks a buffer for the next commitment of the same shape.  The pool is bounded by entries and by bytes (a quarter of
// the device memory): the oldest entries are released first, so buffers of shapes that never come back do not pile up.
static void pool_free(wf_ctx *ctx, void *p, size_t bytes) {
    if (!p) return;
    if (!ctx_alive(ctx)) {  // (see g_live_ctx)
        (void)hipFree(p);
        return;
    }
    if (!bytes || bytes > ctx->pool_cap) {
        (void)hipFree(p);
        return;
    }
    ctx->pool.emplace_back(p, bytes);
    ctx->pool_bytes += bytes;
    while (!ctx->pool.empty() && (ctx->pool.size() > 16 || ctx->pool_bytes > ctx->pool_cap)) {
        (void)hipFree(ctx->pool.front().first);
        ctx->pool_bytes -= ctx->pool.front().second;
        ctx->pool.erase(ctx->pool.begin());
    }
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

static void prof_mark(wf_ctx *ctx, hipStream_t st, const char *name) {
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

static int ensure(wf_ctx *ctx, DevBuf &b, size_t bytes) {
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
static int upload_columns(wf_ctx *ctx, void *dst, const void *const *cols, size_t n, size_t colb, hipStream_t st) {
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
static int download_columns(wf_ctx *ctx, void *const *cols, const void *src, size_t n, size_t colb, hipStream_t st) {
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

// ------------------------------------------------------------------------------------------------- tables (host)
template <class F>
static int upload_pow2l(wf_ctx *ctx, const std::vector<typename F::T> &bases, uint32_t logN, TableSet &ts) {
    // for each base g: lo[e] = g^e (e < 2^s), hi[h] = g^(h * 2^s) (h < 2^(logN - s))
    typedef typename F::T T;
    const uint32_t s = (logN + 1) / 2;
    const size_t nlo = (size_t)1 << s, nhi = (size_t)1 << (logN - s);
    std::vector<T> lo(nlo * bases.size()), hi(nhi * bases.size());
    for (size_t bi = 0; bi < bases.size(); bi++) {
        T g = bases[bi], acc = F::one();
        for (size_t e = 0; e < nlo; e++) {
            lo[bi * nlo + e] = acc;
            acc = F::mul(acc, g);
        }
        T gs = acc;  // g^(2^s)
        acc = F::one();
        for (size_t h = 0; h < nhi; h++) {
            hi[bi * nhi + h] = acc;
            acc = F::mul(acc, gs);
        }
    }
    HIP_TRY(hipMalloc(&ts.lo, lo.size() * sizeof(T)));
    HIP_TRY(hipMalloc(&ts.hi, hi.size() * sizeof(T)));
    HIP_TRY(hipMemcpyAsync(ts.lo, lo.data(), lo.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ts.hi, hi.data(), hi.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // host vectors die at return
    ts.s = s;
    ts.mask = (uint32_t)(nlo - 1);
    ts.lo_stride = nlo;
    ts.hi_stride = nhi;
    return 0;
}

template <class F>
static Pow2L<F> as_pow2l(const TableSet &ts) {
    Pow2L<F> p;
    p.lo = (const typename F::T *)ts.lo;
    p.hi = (const typename F::T *)ts.hi;
    p.s = ts.s;
    p.mask = ts.mask;
    return p;
}

// powers of the 2^logN-th root of unity (or its inverse): get_twiddles / get_inv_twiddles of the reference
// (math/src/fft/mod.rs:466-522) without the bit-reversal, in two-level form
template <class F>
static int root_tables(wf_ctx *ctx, uint32_t logN, bool inverse, TableSet **out) {
    auto key = std::make_tuple((int)F::FIELD_ID, (int)logN, inverse ? 1 : 0, 0, (uint64_t)0, (uint64_t)0);
    auto it = ctx->tables.find(key);
    if (it == ctx->tables.end()) {
        typename F::T w = f_root_of_unity<F>(logN);
        if (inverse) w = f_inv<F>(w);
        TableSet ts;
        int rc = upload_pow2l<F>(ctx, {w}, logN, ts);
        if (rc) return rc;
        it = ctx->tables.emplace(key, ts).first;
    }
    *out = &it->second;
    return 0;
}

// coset bases h_c = offset * g^c, c < blowup, g = root of unity of order R*blowup
// (get_evaluation_offsets, prover/src/matrix/row_matrix.rs:248-287, with natural coset numbering)
template <class F>
static int coset_tables(wf_ctx *ctx, uint32_t logR, uint32_t logB, typename F::T offset, uint64_t off_lo,
                        uint64_t off_hi, TableSet **out) {
    auto key = std::make_tuple((int)F::FIELD_ID, (int)logR, 2, (int)logB, off_lo, off_hi);
    auto it = ctx->tables.find(key);
    if (it == ctx->tables.end()) {
        typename F::T g = f_root_of_unity<F>(logR + logB);
        std::vector<typename F::T> bases((size_t)1 << logB);
        typename F::T h = offset;
        for (size_t c = 0; c < bases.size(); c++) {
            bases[c] = h;
            h = F::mul(h, g);
        }
        TableSet ts;
        int rc = upload_pow2l<F>(ctx, bases, logR, ts);
        if (rc) return rc;
        it = ctx->tables.emplace(key, ts).first;
    }
    *out = &it->second;
    return 0;
}

// output series for interpolate_poly_with_offset: coefficient k is multiplied by (1/n) * offset^-k
// (math/src/fft/serial.rs:78-93); 1/n is folded into the lo table
template <class F>
static int series_tables(wf_ctx *ctx, uint32_t logN, typename F::T offset, uint64_t off_lo, uint64_t off_hi,
                         TableSet **out) {
    auto key = std::make_tuple((int)F::FIELD_ID, (int)logN, 3, 0, off_lo, off_hi);
    auto it = ctx->tables.find(key);
    if (it == ctx->tables.end()) {
        typedef typename F::T T;
        T inv_off = f_inv<F>(offset);
        T inv_n = f_inv<F>(F::from_u128_canonical((u128)1 << logN));
        const uint32_t s = (logN + 1) / 2;
        const size_t nlo = (size_t)1 << s, nhi = (size_t)1 << (logN - s);
        std::vector<T> lo(nlo), hi(nhi);
        T acc = F::one();
        for (size_t e = 0; e < nlo; e++) {
            lo[e] = F::mul(acc, inv_n);
            acc = F::mul(acc, inv_off);
        }
        T gs = acc;
        acc = F::one();
        for (size_t h = 0; h < nhi; h++) {
            hi[h] = acc;
            acc = F::mul(acc, gs);
        }
        TableSet ts;
        HIP_TRY(hipMalloc(&ts.lo, nlo * sizeof(T)));
        HIP_TRY(hipMalloc(&ts.hi, nhi * sizeof(T)));
        HIP_TRY(hipMemcpy(ts.lo, lo.data(), nlo * sizeof(T), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ts.hi, hi.data(), nhi * sizeof(T), hipMemcpyHostToDevice));
        ts.s = s;
        ts.mask = (uint32_t)(nlo - 1);
        it = ctx->tables.emplace(key, ts).first;
    }
    *out = &it->second;
    return 0;
}

#ifdef WF_EXP_STAMPS
// diagnostic build (scripts/last_pass_phases.py): cycle sums per work-group and phase of the persistent last pass
static unsigned long long *g_exp_stamps = nullptr;
static unsigned long long *exp_stamps_buffer() {
    if (!g_exp_stamps && hipMalloc(&g_exp_stamps, 4096 * 8 * 8) == hipSuccess) (void)hipMemset(g_exp_stamps, 0, 4096 * 8 * 8);
    return g_exp_stamps;
}
extern "C" int wf_exp_stamps_read(unsigned long long *out, int clear) {
    if (!g_exp_stamps) return -1;
    if (hipMemcpy(out, g_exp_stamps, 4096 * 8 * 8, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    if (clear) (void)hipMemset(g_exp_stamps, 0, 4096 * 8 * 8);
    return 0;
}
#endif

// ------------------------------------------------------------------------------------------------- planner
struct Plan {
    int n_pass;
    uint32_t dig[4];
};

// digits of at most `max_digit` bits (what one work-group can hold in LDS: 11 for f64, 10 for f128), balanced.
// `avoid_full`: a plan of two maximal digits would run both passes with a single work-group per CU (the tile fills
// the LDS), which measures ~10 % slower than three passes over smaller tiles (2^22 f64, 2^20 f128).
static Plan make_plan(uint32_t L, uint32_t max_digit, bool avoid_full = false, bool few_tiles = false) {
    Plan p;
    p.n_pass = L <= 10 ? 1 : (int)((L + max_digit - 1) / max_digit);
    if (avoid_full && p.n_pass == 2 && L == 2 * max_digit) p.n_pass = 3;
    // segment kernels: a 2^11-row tile is one work-group per CU, and a transform of that size over a few segments only
    // a handful of them -- two passes of small tiles are faster then (2^11 x 8 f64: 0.123 -> 0.10 ms)
    if (few_tiles && p.n_pass == 1 && L > 10) p.n_pass = 2;
    uint32_t base = L / p.n_pass, rem = L % p.n_pass;
    for (int i = 0; i < p.n_pass; i++) p.dig[i] = base + (i < (int)rem ? 1 : 0);
    // a maximal digit goes last: the last pass keeps one table less in LDS (an f128 2^10-row tile leaves room for two
    // work-groups per CU there, not in a strided pass) -- f128 2^19 x 10: 2.99 -> 2.81 ms, f64 2^21 x 64: 20.9 -> 19.8 ms
    if (avoid_full && p.n_pass >= 2 && p.dig[0] == max_digit && p.dig[p.n_pass - 1] < max_digit)
        std::swap(p.dig[0], p.dig[p.n_pass - 1]);
    return p;
}

// the plan of the segment kernels (run_seg_transform and the sizing of its work buffer must agree on it)
template <class F>
static Plan seg_plan(uint32_t logN, uint32_t n_seg) {
    uint32_t max_digit = F::BYTES == 8 ? 11 : 10;
    if (const char *e = getenv("WF_EXP_MAX_DIGIT")) {  // tuning experiment: force more, smaller passes
        const uint32_t v = (uint32_t)atoi(e);
        if (v >= 4 && v < max_digit && (logN + v - 1) / v <= 4) max_digit = v;  // Plan holds 4 digits
    }
    return make_plan(logN, max_digit, true, n_seg <= 8);
}

template <class F>
static uint32_t tile_target(uint32_t W) {  // adjacent elements so that a global chunk is ~64 bytes
    uint32_t t = 64 / (W * F::BYTES);
    return t < 1 ? 1 : t;
}

template <class F>
static int launch_dims(uint32_t logD, uint32_t V, uint32_t &threads, size_t &lds) {
    const size_t vals = ((size_t)1 << logD) * V;
    lds = (vals + ((size_t)1 << logD)) * sizeof(typename F::T);
    if (lds > 160 * 1024) return fail(WF_ERR_ARG, "internal: pass needs %zu bytes of LDS", lds);
    threads = vals >= 16384 ? 1024 : (vals >= 8192 ? 512 : 256);
    if (lds > 64 * 1024) {
        HIP_TRY(hipFuncSetAttribute((const void *)k_ntt_strided<F>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void *)k_ntt_last<F>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    return 0;
}

// One transform of a batch of columns in the caller's column layout (the stand-alone math::fft entry points):
// src = [batch] columns of N elements of W coordinates, dst likewise, natural order, scaled per scale_mode.
template <class F>
struct XformDesc {
    typedef typename F::T T;
    const T *src;
    T *dst;
    uint32_t logN, W, batch;
    bool inverse;
    uint32_t scale_mode;
    T scale;
    const TableSet *out_series;
};

template <class F>
static int run_transform(wf_ctx *ctx, hipStream_t st, const XformDesc<F> &d) {
    typedef typename F::T T;
    TableSet *tw;
    int rc = root_tables<F>(ctx, d.logN, d.inverse, &tw);
    if (rc) return rc;
    const Plan plan = make_plan(d.logN, F::BYTES == 8 ? 11 : 10);
    const uint64_t N = (uint64_t)1 << d.logN;

    NttArgs<F> a;
    memset(&a, 0, sizeof(a));
    a.logN = d.logN;
    a.W = d.W;
    a.col_elems = N;
    a.tw = as_pow2l<F>(*tw);
    // multi-pass transforms go  src -> scratch (first pass), scratch in place (middle), scratch -> dst (last pass):
    // the last pass scatters to natural order and therefore cannot run in place
    T *scratch = nullptr;
    if (plan.n_pass > 1) {
        rc = ensure(ctx, ctx->scratch, (size_t)d.batch * N * d.W * sizeof(T));
        if (rc) return rc;
        scratch = (T *)ctx->scratch.p;
    }
    uint32_t done_bits = 0;
    for (int pi = 0; pi + 1 < plan.n_pass; pi++) {
        a.logD = plan.dig[pi];
        a.O = (uint64_t)1 << done_bits;
        a.I = N >> (done_bits + a.logD);
        a.Tl = (uint32_t)std::min<uint64_t>(tile_target<F>(d.W), a.I);
        a.V = a.Tl * d.W;
        a.src = pi == 0 ? d.src : scratch;
        a.dst = scratch;
        uint32_t threads;
        size_t lds;
        rc = launch_dims<F>(a.logD, a.V, threads, lds);
        if (rc) return rc;
        const uint64_t grid = (uint64_t)d.batch * a.O * (a.I / a.Tl);
        if (grid > 0x7FFFFFFFull) return fail(WF_ERR_ARG, "problem too large for one launch (%llu groups)", (unsigned long long)grid);
        hipLaunchKernelGGL(k_ntt_strided<F>, dim3((uint32_t)grid), dim3(threads), lds, st, a);
        HIP_TRY(hipGetLastError());
        done_bits += a.logD;
    }
    {
        const int pi = plan.n_pass - 1;
        const bool single = plan.n_pass == 1;
        a.logD = plan.dig[pi];
        a.O = (uint64_t)1 << done_bits;
        a.I = 1;
        a.n_prev = plan.n_pass - 1;
        for (int i = 0; i < pi; i++) a.prev_log[i] = plan.dig[i];
        a.scale_mode = d.scale_mode;
        a.scale = d.scale;
        if (d.out_series) a.out_pow = as_pow2l<F>(*d.out_series);
        a.Tl = single ? 1 : std::min<uint32_t>(tile_target<F>(d.W), 1u << plan.dig[0]);
        a.V = a.Tl * d.W;
        a.src = single ? d.src : scratch;
        a.dst = d.dst;
        uint32_t threads;
        size_t lds;
        rc = launch_dims<F>(a.logD, a.V, threads, lds);
        if (rc) return rc;
        const uint64_t grid = (uint64_t)d.batch * (a.O / a.Tl);
        if (grid > 0x7FFFFFFFull) return fail(WF_ERR_ARG, "problem too large for one launch (%llu groups)", (unsigned long long)grid);
        hipLaunchKernelGGL(k_ntt_last<F>, dim3((uint32_t)grid), dim3(threads), lds, st, a);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------- segment pipeline
// powers of the 2^logD-th root (or its inverse), D entries, for the in-LDS transform of one digit
template <class F>
static int digit_table(wf_ctx *ctx, uint32_t logD, bool inverse, const typename F::T **out) {
    auto key = std::make_tuple((int)F::FIELD_ID, (int)logD, inverse ? 5 : 4, 0, (uint64_t)0, (uint64_t)0);
    auto it = ctx->tables.find(key);
    if (it == ctx->tables.end()) {
        typedef typename F::T T;
        T w = f_root_of_unity<F>(logD ? logD : 1);
        if (logD == 0) w = F::one();
        if (inverse) w = f_inv<F>(w);
        std::vector<T> tab((size_t)1 << logD);
        T acc = F::one();
        for (auto &v : tab) {
            v = acc;
            acc = F::mul(acc, w);
        }
        TableSet ts;
        HIP_TRY(hipMalloc(&ts.lo, tab.size() * sizeof(T)));
        HIP_TRY(hipMemcpy(ts.lo, tab.data(), tab.size() * sizeof(T), hipMemcpyHostToDevice));
        it = ctx->tables.emplace(key, ts).first;
    }
    *out = (const typename F::T *)it->second.lo;
    return 0;
}

template <class F>
static int seg_launch_dims(uint32_t logD, uint32_t &threads, size_t &lds, bool last_pass) {
    const size_t D = (size_t)1 << logD;
    // tile + digit twiddles (+ the factor table of a strided pass; a last pass keeps its input factors where the
    // twiddles go afterwards: a 2^10-row f128 tile is 80 KiB, two work-groups per CU)
    lds = (D * SegCfg<F>::S + (last_pass ? 1 : 2) * D) * sizeof(typename F::T);
    if (lds > 160 * 1024) return fail(WF_ERR_ARG, "internal: pass needs %zu bytes of LDS", lds);
    // one work item of the widest round per thread (radix-16 on 8 lanes for f64, radix-4 on lane pairs for f128: D/2
    // items either way), so that no wave idles through the transform rounds; a 2^11-row f64 tile fills the LDS of a CU
    // on its own and brings its 16 waves along
    threads = (uint32_t)std::min<size_t>(1024, std::max<size_t>(64, D / 2));
    if (lds > 64 * 1024) {
        HIP_TRY(hipFuncSetAttribute((const void *)k_seg_strided<F, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void *)k_seg_strided<F, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void *)k_seg_strided<F, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void *)k_seg_last<F, SEG_OUT_ROWS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void *)k_seg_last<F, SEG_OUT_SEG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void *)k_seg_last<F, SEG_OUT_ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        constexpr int L10 = F::BYTES == 8 ? 10 : 0;  // the tile-size-specialised instantiation whose tile exceeds 64 KiB
        HIP_TRY(hipFuncSetAttribute((const void *)k_seg_strided<F, 0, false, L10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void *)k_seg_strided<F, 1, false, L10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    return 0;
}

static void launch_merge_chunks(hipStream_t st, const void *cvs, uint32_t n_chunks, uint64_t n_rows, void *leaves);

// Coset packing of narrow evaluations (<= S/2 base columns in one segment, an even number of cosets): 2^cpr cosets of
// 2^lg lanes each share the lanes of a row.
template <class F>
static bool packed_shape(uint32_t n_seg, uint32_t total_base_cols, uint32_t base_cols, uint32_t n_cosets, uint32_t *cpr_log,
                         uint32_t *lg_log) {
    // several traces side by side: their rows go to different matrices, which the unpacked kernels write whole (store_rows_narrow)
    // and hash in the same pass -- packing only pays for them while at least 3/4 of the lanes would idle (measured)
    if (total_base_cols != base_cols && total_base_cols * 4 > SegCfg<F>::S) return false;
    if (n_seg != 1 || total_base_cols * 2 > SegCfg<F>::S || n_cosets < 2) return false;
    uint32_t lg = 0;
    while ((1u << lg) < total_base_cols) lg++;
    uint32_t cpr = 0;
    while ((2u << cpr) <= (SegCfg<F>::S >> lg) && n_cosets % (2u << cpr) == 0) cpr++;
    if (cpr == 0) return false;
    *cpr_log = cpr;
    *lg_log = lg;
    return true;
}

// One transform of n_seg segments (x n_cosets cosets).
//   interpolation (rows_out == false): in  = [n_seg][N][S] evaluations (overwritten when N needs > 1 pass),
//                                      out = [n_seg][N][S] coefficients, natural order, scaled by 1/N
//   evaluation    (rows_out == true) : in  = [n_seg][N][S] coefficients (read only), work = [cosets][n_seg][N][S],
//                                      out = row-major LDE matrices (see SegArgs)
template <class F>
struct SegDesc {
    typedef typename F::T T;
    const T *in;
    T *work;
    T *out;
    uint32_t logN, n_seg, n_cosets;
    bool rows_out;
    void *leaves = nullptr;     // rows_out: hash the leaves in the last pass when the shape allows (sets *fused)
    uint32_t hash_epr = 0;
    bool *fused = nullptr;
    const TableSet *pre;
    bool pad_traces = false;     // rows_out, unpacked: the lane with a trace's last column zeroes the rest of that row
    bool pad_in_kernel = false;  // rows_out: the last pass also writes the zero padding lanes of the rows
    uint32_t base_cols, total_base_cols, coset0;
    uint64_t row_width, trace_lde_elems;
    // a multi-pass transform in two calls (uploads of later segments run under the strided passes of earlier ones):
    // phase 1 = the strided passes of segments [seg0, seg0 + seg_cnt) only, phase 2 = the last pass (all segments) only
    uint32_t seg0 = 0, seg_cnt = 0;
    int phase = 0;
};

template <class F>
static int run_seg_transform(wf_ctx *ctx, hipStream_t st, const SegDesc<F> &d) {
    typedef typename F::T T;
    const bool inverse = !d.rows_out;
    TableSet *tw;
    int rc = root_tables<F>(ctx, d.logN, inverse, &tw);
    if (rc) return rc;
    const Plan plan = seg_plan<F>(d.logN, d.n_seg);
    const uint64_t N = (uint64_t)1 << d.logN;

    SegArgs<F> a;
    memset(&a, 0, sizeof(a));
    a.logN = d.logN;
    a.n_seg = d.n_seg;
    a.n_cosets = d.n_cosets;
    a.tw = as_pow2l<F>(*tw);
    if (d.pre) {
        a.pre = as_pow2l<F>(*d.pre);
        a.pre_lo_stride = d.pre->lo_stride;
        a.pre_hi_stride = d.pre->hi_stride;
    }
    a.base_cols = d.base_cols;
    a.total_base_cols = d.total_base_cols;
    a.store_cols = d.pad_in_kernel ? d.n_seg * SegCfg<F>::S : d.base_cols;
    a.total_store_cols = d.pad_in_kernel ? d.n_seg * SegCfg<F>::S : d.total_base_cols;
    a.pad_traces = d.pad_traces ? 1 : 0;
    a.tail_pad = d.pad_in_kernel ? (uint32_t)d.row_width - d.n_seg * SegCfg<F>::S : 0;
    a.coset0 = d.coset0;
    a.rows_per_k = d.n_cosets;
    a.row_width = d.row_width;
    a.trace_lde_elems = d.trace_lde_elems;
    const T inv_n = inverse ? f_inv<F>(F::from_u128_canonical((u128)1 << d.logN)) : F::one();
    // narrow matrices (evaluation of <= S/2 base columns: composition / DEEP polynomials): pack several cosets into the
    // lanes of a row instead of leaving them empty
    uint32_t n_groups = d.n_cosets;
    bool packed = false;
    if (d.rows_out && packed_shape<F>(d.n_seg, d.total_base_cols, d.base_cols, d.n_cosets, &a.cpr_log, &a.lg_log)) {
        packed = true;
        n_groups = d.n_cosets >> a.cpr_log;
        a.n_cosets = n_groups;
    }
    const char *tag_s = d.rows_out ? "evaluate.strided_pass" : "interpolate.strided_pass";
    const char *tag_l = d.rows_out ? "evaluate.last_pass" : "interpolate.last_pass";

    const uint32_t run_cnt = d.seg_cnt ? d.seg_cnt : d.n_seg;
    const size_t run_off = (size_t)d.seg0 * (N * SegCfg<F>::S);  // elements in front of segment seg0 within one coset
    a.seg_stride = d.n_seg;
    uint32_t done_bits = 0;
    for (int pi = 0; pi + 1 < plan.n_pass; pi++) {
        if (d.phase == 2) {  // strided passes already run
            done_bits += plan.dig[pi];
            continue;
        }
        const bool first = pi == 0;
        a.logD = plan.dig[pi];
        a.O = (uint64_t)1 << done_bits;
        a.I = N >> (done_bits + a.logD);
        rc = digit_table<F>(ctx, a.logD, inverse, &a.digit_tw);
        if (rc) return rc;
        if (d.rows_out) {
            a.src = first ? d.in : d.work;
            a.dst = d.work;
            a.src_shared = first ? 1 : 0;
            a.pre_on = first ? 1 : 0;
            a.scale_on = 0;
        } else {
            a.src = first ? d.in : d.work;  // interpolation: first pass in -> work, later passes in place
            a.dst = d.work;
            a.src_shared = 0;
            a.pre_on = 0;
            a.scale_on = first ? 1 : 0;     // 1/n rides on the first inter-pass twiddle table
            a.scale = inv_n;
        }
        a.src += run_off;
        a.dst += run_off;
        a.n_seg = run_cnt;
        uint32_t threads;
        size_t lds;
        rc = seg_launch_dims<F>(a.logD, threads, lds, false);
        if (rc) return rc;
        const uint64_t grid = (uint64_t)n_groups * run_cnt * a.O * a.I;
        if (grid > 0x7FFFFFFFull) return fail(WF_ERR_ARG, "problem too large for one launch (%llu groups)", (unsigned long long)grid);
        prof_mark(ctx, st, tag_s);
        // f64 tiles of 2^10 rows (the digits of the 2^19 .. 2^21 plans) run the tile-size-specialised instantiation
        // (seg_kernels.hpp, WF_TILE_BOUNDS: strided pass of cfg 2 0.347 -> 0.324 ms); everything else the generic kernel.
        // Measured and left out: 2^7 / 2^8-row tiles (2^22 x 64: 36.8 -> 37.5 ms, no gain), and the last passes, which
        // specialised for the tile size need more registers than two work-groups per CU allow (scratch spills).
        const bool spec_ok = F::BYTES == 8 && !packed && threads * 2 == (1u << a.logD) && getenv("WF_EXP_NO_SPECIALIZED") == nullptr;
        const void *kern = nullptr;
        if (spec_ok) {
            constexpr bool F8 = F::BYTES == 8;  // (the specialised instantiations exist for f64 only)
            switch (a.logD) {
                case 10: kern = d.rows_out ? (const void *)k_seg_strided<F, 1, false, F8 ? 10 : 0> : (const void *)k_seg_strided<F, 0, false, F8 ? 10 : 0>; break;
                default: break;
            }
        }
        if (kern) {
            void *kargs[] = {&a};
            HIP_TRY(hipLaunchKernel(kern, dim3((uint32_t)grid), dim3(threads), kargs, lds, st));
        } else if (d.rows_out && packed)
            hipLaunchKernelGGL((k_seg_strided<F, 1, true>), dim3((uint32_t)grid), dim3(threads), lds, st, a);
        else if (d.rows_out)
            hipLaunchKernelGGL((k_seg_strided<F, 1>), dim3((uint32_t)grid), dim3(threads), lds, st, a);
        else
            hipLaunchKernelGGL((k_seg_strided<F, 0>), dim3((uint32_t)grid), dim3(threads), lds, st, a);
        HIP_TRY(hipGetLastError());
        done_bits += a.logD;
    }
    a.n_seg = d.n_seg;
    if (d.phase == 1) return 0;
    {
        const int pi = plan.n_pass - 1;
        const bool single = plan.n_pass == 1;
        a.logD = plan.dig[pi];
        a.O = (uint64_t)1 << done_bits;
        a.I = 1;
        a.n_prev = plan.n_pass - 1;
        for (int i = 0; i < pi; i++) a.prev_log[i] = plan.dig[i];
        rc = digit_table<F>(ctx, a.logD, inverse, &a.digit_tw);
        if (rc) return rc;
        a.src = single ? d.in : d.work;
        a.dst = d.out;
        a.src_shared = (single && d.rows_out) ? 1 : 0;
        a.pre_on = (single && d.rows_out) ? 1 : 0;
        a.scale_on = (single && !d.rows_out) ? 1 : 0;
        a.scale = inv_n;
        // Leaf hashing rides on the last pass
        //  - in the persistent kernel k_seg_last_hash when the combined row of all traces is at most one BLAKE3 chunk
        //    (<= 16 segments) and the plan has several passes,
        //  - else in k_seg_last itself when a tile row is a whole matrix row of one trace (one segment),
        //  - else not at all: k_hash_rows reads the LDE back.
        uint32_t threads;
        size_t lds;
        rc = seg_launch_dims<F>(a.logD, threads, lds, true);
        if (rc) return rc;
        const uint64_t grid = (uint64_t)n_groups * d.n_seg * a.O;
        if (grid > 0x7FFFFFFFull) return fail(WF_ERR_ARG, "problem too large for one launch (%llu groups)", (unsigned long long)grid);
        const bool fuse_on = d.rows_out && d.leaves && getenv("WF_EXP_NO_FUSED_HASH") == nullptr;
        const bool may_fuse = fuse_on && !packed;
        // rows of more than 16 segments (one BLAKE3 chunk) are fused chunk by chunk: the pass leaves chunk chaining values
        const bool chunked = d.n_seg > 16 && getenv("WF_EXP_NO_CHUNKED") == nullptr;
        const uint32_t n_chunks = chunked ? (d.n_seg + 15) / 16 : 1;
        const uint64_t tickets = (uint64_t)d.n_cosets * a.O * n_chunks;
        const uint64_t launch_rows = (uint64_t)d.n_cosets << d.logN;
        // one segment of one trace: k_seg_last hashes its rows itself, and with tiles below 2^10 rows one work-group per
        // tile beats the ticket kernel (2^14..2^18 x 8: -3..-14 %, 2^22 x 8: -10 %; 2^20 x 8, 2^10-row tiles: +5 %)
        const bool one_seg = d.n_seg == 1 && d.total_base_cols == d.base_cols;
        const bool always = getenv("WF_EXP_PERSISTENT_ALWAYS") != nullptr;  // (tests: the ticket kernel on every shape it can run)
        const bool persistent = may_fuse && !single && (d.n_seg <= 16 || chunked) && threads * 2 == (1u << a.logD) &&
                                (always || !one_seg || a.logD >= 10) &&
                                // several segments of one trace, rows of one chunk: below 2^20 LDE rows the separate
                                // row-hash kernel costs less than the ticket kernel's small tiles (2^14 x 16: -15 %)
                                (always || one_seg || chunked || d.total_base_cols != d.base_cols || launch_rows >= (1ull << 20)) &&
                                tickets % 8 == 0 && tickets < (1ull << 31) && launch_rows * n_chunks * 32 < (1ull << 40) &&
                                getenv("WF_EXP_NO_PERSISTENT") == nullptr;
        // coset-packed rows are hashed in the pass where the separate kernel is the slower one (measured): f128, four
        // lanes per coset, or rows gathered from several traces; one- and two-lane f64 rows keep k_hash_rows
        const bool fuse_packed = fuse_on && packed && (F::BYTES == 16 || a.lg_log >= 2 || d.total_base_cols != d.base_cols);
        const bool fuse = persistent || (may_fuse && d.n_seg == 1 && d.total_base_cols == d.base_cols) || fuse_packed;
        a.leaves = fuse ? (uint32_t *)d.leaves : nullptr;
        a.hash_epr = d.hash_epr;
        if (d.fused) *d.fused = fuse;
        prof_mark(ctx, st, tag_l);
        if (persistent) {
            const bool multi = d.n_seg > 1 || d.total_base_cols != d.base_cols;
            const bool small = threads <= 256;  // the multi-segment variants without the 128-VGPR cap (2^22 x 64: last pass -3 %)
            const void *kern =
                chunked ? (a.pad_traces ? (small ? (const void *)k_seg_last_hash<F, true, true, true, true> : (const void *)k_seg_last_hash<F, true, true, true>)
                                        : (small ? (const void *)k_seg_last_hash<F, true, false, true, true> : (const void *)k_seg_last_hash<F, true, false, true>))
                : multi ? (a.pad_traces ? (small ? (const void *)k_seg_last_hash<F, true, true, false, true> : (const void *)k_seg_last_hash<F, true, true>)
                                        : (small ? (const void *)k_seg_last_hash<F, true, false, false, true> : (const void *)k_seg_last_hash<F, true, false>))
                        : (a.pad_traces ? (const void *)k_seg_last_hash<F, false, true> : (const void *)k_seg_last_hash<F, false, false>);
            if (chunked) {
                int rcc = ensure(ctx, ctx->hash_tmp, (size_t)launch_rows * n_chunks * 32);
                if (rcc) return rcc;
                a.chunk_cvs = (uint32_t *)ctx->hash_tmp.p;
                a.n_chunks = n_chunks;
            }
            if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            const size_t lds_p = lds;  // (the two ticket words live in the unused last twiddle slot)
            const uint64_t resident = (uint64_t)ctx->num_cus * std::max<size_t>(1, (160 * 1024) / lds_p);
            if (!ctx->tickets.p) {  // zeroed once: the kernel leaves its counters at zero
                int rcq = ensure(ctx, ctx->tickets, 64);
                if (rcq) return rcq;
                HIP_TRY(hipMemsetAsync(ctx->tickets.p, 0, 64, st));  // ordered on the launch stream (first use only)
            }
            a.tile_counters = (uint32_t *)ctx->tickets.p;
#ifdef WF_EXP_STAMPS
            a.stamps = exp_stamps_buffer();
#endif
            void *kargs[] = {&a};
            HIP_TRY(hipLaunchKernel(kern, dim3((uint32_t)std::min<uint64_t>(tickets, resident)), dim3(threads), kargs, lds_p, st));
            if (chunked) {
                HIP_TRY(hipGetLastError());
                launch_merge_chunks(st, ctx->hash_tmp.p, n_chunks, launch_rows, d.leaves);
            }
        } else if (d.rows_out && packed)
            hipLaunchKernelGGL((k_seg_last<F, SEG_OUT_ROWS, true>), dim3((uint32_t)grid), dim3(threads), lds, st, a);
        else if (d.rows_out)
            hipLaunchKernelGGL((k_seg_last<F, SEG_OUT_ROWS>), dim3((uint32_t)grid), dim3(threads), lds, st, a);
        else
            hipLaunchKernelGGL((k_seg_last<F, SEG_OUT_SEG>), dim3((uint32_t)grid), dim3(threads), lds, st, a);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

template <class F>
static int run_xpose(wf_ctx *ctx, hipStream_t st, bool to_seg, const void *src, void *dst, uint64_t R, uint32_t W,
                     uint32_t total_base_cols, uint32_t n_seg, uint32_t seg0 = 0, uint32_t seg_cnt = 0) {
    if (seg_cnt == 0) seg_cnt = n_seg - seg0;  // segments [seg0, seg0 + seg_cnt) of the n_seg of the matrix
    XposeArgs<F> x;
    x.seg0 = seg0;
    x.src = (const typename F::T *)src;
    x.dst = (typename F::T *)dst;
    x.R = R;
    x.W = W;
    x.total_base_cols = total_base_cols;
    constexpr uint32_t RPB = XPOSE_TILES * 256 / SegCfg<F>::S;
    const uint64_t grid = (uint64_t)seg_cnt * ((R + RPB - 1) / RPB);
    if (grid > 0x7FFFFFFFull) return fail(WF_ERR_ARG, "problem too large for one launch");
    prof_mark(ctx, st, to_seg ? "layout.cols_to_segments" : "layout.segments_to_cols");
    if (to_seg)
        hipLaunchKernelGGL(k_cols_to_seg<F>, dim3((uint32_t)grid), dim3(256), 0, st, x);
    else
        hipLaunchKernelGGL(k_seg_to_cols<F>, dim3((uint32_t)grid), dim3(256), 0, st, x);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------- hashing + tree
// leaves from per-row chunk chaining values ([row][n_chunks][8 words])
static void launch_merge_chunks(hipStream_t st, const void *cvs, uint32_t n_chunks, uint64_t n_rows, void *leaves) {
    // few, long rows: 16 lanes per row; otherwise one lane per row (measured: 8192 rows x 80 chunks 0.48 -> 0.37 ms for
    // chunks + merge, but 32768 x 20 and shorter rows are faster with a lane per row)
    if (n_chunks >= 32 && n_chunks <= 128 && n_rows <= 65536)
        hipLaunchKernelGGL(k_hash_merge_chunks_par, dim3((uint32_t)((n_rows + 15) / 16)), dim3(256), (size_t)16 * n_chunks * 32, st,
                           (const uint32_t *)cvs, n_chunks, n_rows, (uint32_t *)leaves);
    else
        hipLaunchKernelGGL(k_hash_merge_chunks, dim3((uint32_t)((n_rows + 255) / 256)), dim3(256), 0, st, (const uint32_t *)cvs,
                           n_chunks, n_rows, (uint32_t *)leaves);
}

template <class F>
static int run_hash_rows(wf_ctx *ctx, hipStream_t st, const void *lde, uint64_t trace_elems, uint64_t n_rows, uint32_t row_width,
                         uint32_t epr, uint32_t n_traces, void *leaves) {
    HashArgs<F> h;
    h.lde = (const typename F::T *)lde;
    h.trace_elems = trace_elems;
    h.n_rows = n_rows;
    h.row_width = row_width;
    h.epr = epr;
    h.n_traces = n_traces;
    h.leaves = (uint32_t *)leaves;
    const uint32_t threads = 256;
    const uint64_t grid = (n_rows + threads - 1) / threads;
    const uint64_t row_bytes = (uint64_t)n_traces * epr * F::BYTES;
    if (row_bytes <= 1024) {  // single BLAKE3 chunk: one lane per row, no subtree stack
        hipLaunchKernelGGL(k_hash_rows<F>, dim3((uint32_t)grid), dim3(threads), 0, st, h);
    } else {                  // one lane per (row, chunk), then one lane per row folds the chaining values
        const uint64_t chunks = (row_bytes + 1023) / 1024;
        if (chunks > 0xFFFFFFFFull || n_rows * chunks > 0x7FFFFFFFull * threads)
            return fail(WF_ERR_ARG, "rows too long for one launch");
        int rc = ensure(ctx, ctx->hash_tmp, (size_t)n_rows * chunks * 32);
        if (rc) return rc;
        const uint64_t g2 = (n_rows * chunks + threads - 1) / threads;
        hipLaunchKernelGGL(k_hash_chunks<F>, dim3((uint32_t)g2), dim3(threads), 0, st, h, (uint32_t)chunks,
                           (uint32_t *)ctx->hash_tmp.p);
        HIP_TRY(hipGetLastError());
        launch_merge_chunks(st, ctx->hash_tmp.p, (uint32_t)chunks, n_rows, leaves);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

static int run_merkle(hipStream_t st, const void *leaves, uint64_t n_leaves, void *nodes) {
    // (nodes[0] = Digest::default(), merkle/mod.rs:355, is written by the launch that produces the root)
    const uint32_t *children = (const uint32_t *)leaves;
    uint64_t n_children = n_leaves;
    while (n_children > 1) {
        const uint64_t n_par = n_children >> 1;
        const uint32_t threads = 256;
        const uint64_t grid = (n_par + threads - 1) / threads;
        // levels of >= 2^18 parents: two per launch; below that the LDS subtree kernel folds 9 levels per launch (one
        // launch less than switching at 2^16, 5 us at 2^23 leaves).  WF_EXP_MERKLE_L2_MIN: tuning switch
        static const uint32_t l2_min = [] {
            const char *e = getenv("WF_EXP_MERKLE_L2_MIN");
            const int v = e ? atoi(e) : 18;
            return (uint32_t)(v >= 10 && v <= 30 ? v : 18);
        }();
        if (n_par >= ((uint64_t)1 << l2_min)) {  // two levels that still fill the chip: one lane per grandparent
            const uint64_t n_grand = n_par >> 1;
            const uint64_t blocks2 = std::min<uint64_t>((n_grand + threads - 1) / threads, 256 * 8);  // grid-stride
            hipLaunchKernelGGL(k_merkle_level2, dim3((uint32_t)blocks2), dim3(threads), 0, st,
                               children, (uint32_t *)nodes + n_par * 8, (uint32_t *)nodes + n_grand * 8, n_grand);
            HIP_TRY(hipGetLastError());
            n_children = n_grand;
        } else if (n_par >= (1u << 15) && l2_min == 16) {  // a level that still fills the chip: one lane per node
            hipLaunchKernelGGL(k_merkle_level, dim3((uint32_t)grid), dim3(threads), 0, st, children,
                               (uint32_t *)nodes + n_par * 8, n_par);
            HIP_TRY(hipGetLastError());
            n_children = n_par;
        } else {                    // the top of the tree: up to 9 levels per launch through LDS
            uint32_t total_levels = 0;
            for (uint64_t t = n_children; t > 1; t >>= 1) total_levels++;
            const uint32_t levels = std::min<uint32_t>(9, total_levels);
            hipLaunchKernelGGL(k_merkle_subtree, dim3((uint32_t)grid), dim3(threads), 0, st, children,
                               (uint32_t *)nodes, n_children, levels);
            HIP_TRY(hipGetLastError());
            n_children >>= levels;
        }
        children = (const uint32_t *)nodes + n_children * 8;  // that level lives at nodes[n .. 2n)
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------- validation
static int check_params(const wf_params *p, bool constraint) {
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
    if (p->digest_bytes != 32) return fail(WF_ERR_DIGEST, "only 32-byte digests (Blake3_256) are supported");
    if (p->reserved != 0) return fail(WF_ERR_ARG, "reserved field must be zero");
    u128 off;
    memcpy(&off, p->domain_offset, 16);
    const u128 mod = p->field == WF_FIELD_F64 ? (u128)F64::P : F128::P();
    if (off == 0 || off >= mod) return fail(WF_ERR_OFFSET, "domain offset must be a non-zero field element");
    return 0;
}

template <class F>
static typename F::T offset_elem(const wf_params *p, uint64_t &lo, uint64_t &hi) {
    u128 off;
    memcpy(&off, p->domain_offset, 16);
    lo = (uint64_t)off;
    hi = (uint64_t)(off >> 64);
    return F::from_u128_canonical(off);
}

// ------------------------------------------------------------------------------------------------- the path (device)
// Scratch layout of the commitment path (context-owned, reused across calls):
//   segA [n_seg][R][S]          transposed input / interpolation work
//   segB [n_seg][R][S]          polynomial coefficients in segment layout (input of the evaluation)
//   work [cosets][n_seg][R][S]  evaluation intermediate (only when R needs more than one pass)
template <class F>
struct PathBufs {
    typename F::T *segA, *segB, *work;
    uint32_t n_seg, total_base_cols;
};

// n_cosets: the cosets this call evaluates (0 = all of them; a rank of a coset-sharded commitment has blowup / W)
template <class F>
static int path_buffers(wf_ctx *ctx, const wf_params *p, PathBufs<F> &b, uint32_t n_cosets = 0) {
    typedef typename F::T T;
    constexpr uint32_t S = SegCfg<F>::S;
    b.total_base_cols = p->n_cols * p->ext_degree * p->n_traces;
    b.n_seg = (b.total_base_cols + S - 1) / S;
    const size_t seg_vals = (size_t)b.n_seg * S << p->log2_trace_len;
    if (n_cosets == 0) n_cosets = 1u << p->log2_blowup;
    const size_t work_vals = seg_plan<F>(p->log2_trace_len, b.n_seg).n_pass > 1 ? seg_vals * n_cosets : 0;
    int rc = ensure(ctx, ctx->scratch, (2 * seg_vals + work_vals) * sizeof(T));
    if (rc) return rc;
    b.segA = (T *)ctx->scratch.p;
    b.segB = b.segA + seg_vals;
    b.work = b.segB + seg_vals;
    return 0;
}

// coefficients in segB -> row-major LDE -> leaves -> tree
template <class F>
static int evaluate_and_commit(wf_ctx *ctx, hipStream_t st, const wf_params *p, const PathBufs<F> &b, void *d_lde,
                               void *d_leaves, void *d_nodes, uint32_t coset0 = 0, uint32_t n_cosets = 0,
                               bool dense_rows = false, int phase = 0, uint32_t seg0 = 0, uint32_t seg_cnt = 0) {
    // phase 1: the strided evaluation passes of segments [seg0, seg0 + seg_cnt) only; phase 2: everything after them
    typedef typename F::T T;
    const uint32_t W = p->ext_degree, logR = p->log2_trace_len, logB = p->log2_blowup;
    if (n_cosets == 0) n_cosets = 1u << logB;  // all of them; otherwise a shard [coset0, coset0 + n_cosets)
    const uint64_t Nrows = (uint64_t)n_cosets << logR;
    const uint32_t base_cols = p->n_cols * W;
    // dense_rows: rows of exactly base_cols elements, no padding (a vector of evaluations rather than a RowMatrix)
    const uint64_t row_width = dense_rows ? base_cols : wf_row_width(p);

    uint64_t olo, ohi;
    T off = offset_elem<F>(p, olo, ohi);
    TableSet *cos;
    int rc = coset_tables<F>(ctx, logR, logB, off, olo, ohi, &cos);
    if (rc) return rc;
    // Zero padding lanes (segments.rs:65-72): a single trace whose segments cover the padded row, or all but its last S
    // elements (f128 rows are padded to 2 S),
    // gets them from the last evaluation pass; everything else is cleared up front.
    uint32_t cpr_unused, lg_unused;
    const uint64_t seg_lanes = (uint64_t)b.n_seg * SegCfg<F>::S;
    const bool pad_in_kernel = row_width != base_cols && p->n_traces == 1 &&
                               (seg_lanes == row_width || seg_lanes + SegCfg<F>::S == row_width) &&
                               (b.total_base_cols * 2 > SegCfg<F>::S ||
                                packed_shape<F>(b.n_seg, b.total_base_cols, base_cols, n_cosets, &cpr_unused, &lg_unused));
    // Other shapes (STARKPack traces side by side in the lanes, each with a padded row of its own): the lane holding a
    // trace's last column writes that row's zeros.  Only coset-packed multi-trace / f128 matrices are cleared up front.
    const bool pad_traces = row_width != base_cols && !pad_in_kernel &&
                            !packed_shape<F>(b.n_seg, b.total_base_cols, base_cols, n_cosets, &cpr_unused, &lg_unused);
    if (phase != 1 && row_width != base_cols && !pad_in_kernel && !pad_traces) {
        const uint64_t n16 = (uint64_t)p->n_traces * Nrows * row_width * sizeof(T) / 16;  // rows are multiples of 64 bytes
        hipLaunchKernelGGL(k_zero16, dim3((uint32_t)std::min<uint64_t>((n16 + 255) / 256, 256 * 32)), dim3(256), 0, st, (uint4 *)d_lde, n16);
        HIP_TRY(hipGetLastError());
    }

    SegDesc<F> d;
    memset(&d, 0, sizeof(d));
    d.in = b.segB;
    d.work = b.work;
    d.out = (T *)d_lde;
    d.logN = logR;
    d.n_seg = b.n_seg;
    d.n_cosets = n_cosets;
    d.coset0 = coset0;
    d.rows_out = true;
    d.pre = cos;
    d.base_cols = base_cols;
    d.total_base_cols = b.total_base_cols;
    d.row_width = row_width;
    d.trace_lde_elems = Nrows * row_width;
    d.pad_in_kernel = pad_in_kernel;
    d.pad_traces = pad_traces;
    bool hashed = false;  // leaves produced by the last evaluation pass itself (one segment, one trace)
    d.leaves = d_leaves;
    d.hash_epr = b.total_base_cols;  // the combined row of all traces (= base_cols for one trace)
    d.fused = &hashed;
    d.phase = phase;
    d.seg0 = seg0;
    d.seg_cnt = seg_cnt;
    rc = run_seg_transform<F>(ctx, st, d);
    if (rc) return rc;
    if (phase == 1) return 0;

    if (d_leaves) {
        if (!hashed) {
            prof_mark(ctx, st, "hash_rows");
            rc = run_hash_rows<F>(ctx, st, d_lde, Nrows * row_width, Nrows, (uint32_t)row_width, base_cols, p->n_traces, d_leaves);
            if (rc) return rc;
        }
        if (d_nodes) {
            prof_mark(ctx, st, "merkle");
            rc = run_merkle(st, d_leaves, Nrows, d_nodes);
            if (rc) return rc;
        }
    }
    prof_mark(ctx, st, "between_calls");
    return 0;
}

// Prover::build_trace_commitment on device buffers
template <class F>
static int trace_commit_dev(wf_ctx *ctx, const wf_params *p, const void *d_trace, void *d_polys, void *d_lde,
                            void *d_leaves, void *d_nodes, hipStream_t st, hipEvent_t input_read = nullptr) {
    PathBufs<F> b;
    int rc = path_buffers<F>(ctx, p, b);
    if (rc) return rc;
    const uint64_t R = (uint64_t)1 << p->log2_trace_len;
    // columns -> segments
    rc = run_xpose<F>(ctx, st, true, d_trace, b.segA, R, p->ext_degree, b.total_base_cols, b.n_seg);
    if (rc) return rc;
    if (input_read) HIP_TRY(hipEventRecord(input_read, st));  // nothing below reads d_trace: its buffer may be refilled
    // ColMatrix::interpolate_columns (col_matrix.rs:196-206)
    SegDesc<F> d;
    memset(&d, 0, sizeof(d));
    d.in = b.segA;
    d.work = b.segA;  // strided passes run in place
    d.out = b.segB;
    d.logN = p->log2_trace_len;
    d.n_seg = b.n_seg;
    d.n_cosets = 1;
    d.rows_out = false;
    rc = run_seg_transform<F>(ctx, st, d);
    if (rc) return rc;
    // the caller's copy of the polynomials, column layout
    rc = run_xpose<F>(ctx, st, false, b.segB, d_polys, R, p->ext_degree, b.total_base_cols, b.n_seg);
    if (rc) return rc;
    return evaluate_and_commit<F>(ctx, st, p, b, d_lde, d_leaves, d_nodes);
}

// build_trace_commitment from HOST columns of a matrix of several segments, the upload running under the kernels: segment g's
// eight columns go up on a copy stream while segment g - 1 is laid out, interpolated and taken through the strided
// evaluation passes of all cosets on the compute stream (those passes work on one segment at a time); only the last
// evaluation pass, which hashes whole rows, and the tree wait for the last segment.  Base-field matrices (a column is a
// base column); the results are the same launches' results in another order.
template <class F>
static int trace_commit_pipelined(wf_ctx *ctx, const wf_params *p, const void *const *cols_in, void *d_stage, void *d_polys,
                                  void *d_lde, void *d_leaves, void *d_nodes, hipStream_t st, void *const *polys_out) {
    typedef typename F::T T;
    constexpr uint32_t S = SegCfg<F>::S;
    PathBufs<F> b;
    int rc = path_buffers<F>(ctx, p, b);
    if (rc) return rc;
    const uint64_t R = (uint64_t)1 << p->log2_trace_len;
    const size_t colb = R * sizeof(T), TC = b.total_base_cols;
    if (!ctx->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    while (ctx->seg_events.size() < (size_t)b.n_seg + 1) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->seg_events.push_back(e);
    }
    // the staging buffer may still be read by what this context queued before
    HIP_TRY(hipEventRecord(ctx->seg_events[b.n_seg], st));
    HIP_TRY(hipStreamWaitEvent(ctx->copy_stream, ctx->seg_events[b.n_seg], 0));
    for (uint32_t g = 0; g < b.n_seg; g++) {
        for (size_t i = (size_t)g * S; i < std::min<size_t>(TC, (size_t)(g + 1) * S); i++)
            HIP_TRY(hipMemcpyAsync((char *)d_stage + i * colb, cols_in[i], colb, hipMemcpyHostToDevice, ctx->copy_stream));
        HIP_TRY(hipEventRecord(ctx->seg_events[g], ctx->copy_stream));
        HIP_TRY(hipStreamWaitEvent(st, ctx->seg_events[g], 0));
        rc = run_xpose<F>(ctx, st, true, d_stage, b.segA, R, 1, b.total_base_cols, b.n_seg, g, 1);
        if (rc) return rc;
        SegDesc<F> d;
        memset(&d, 0, sizeof(d));
        d.in = b.segA + (size_t)g * R * S;
        d.work = (T *)d.in;
        d.out = b.segB + (size_t)g * R * S;
        d.logN = p->log2_trace_len;
        d.n_seg = 1;
        d.n_cosets = 1;
        d.rows_out = false;
        if ((rc = run_seg_transform<F>(ctx, st, d))) return rc;
        if ((rc = evaluate_and_commit<F>(ctx, st, p, b, d_lde, d_leaves, d_nodes, 0, 0, false, 1, g, 1))) return rc;
    }
    if ((rc = run_xpose<F>(ctx, st, false, b.segB, d_polys, R, 1, b.total_base_cols, b.n_seg))) return rc;
    if (polys_out) HIP_TRY(hipEventRecord(ctx->seg_events[b.n_seg], st));  // the polynomials are complete here
    if ((rc = evaluate_and_commit<F>(ctx, st, p, b, d_lde, d_leaves, d_nodes, 0, 0, false, 2))) return rc;
    if (polys_out) {  // their way back to the host runs under the last evaluation pass and the tree
        HIP_TRY(hipStreamWaitEvent(ctx->copy_stream, ctx->seg_events[b.n_seg], 0));
        if ((rc = download_columns(ctx, polys_out, d_polys, TC, colb, ctx->copy_stream))) return rc;
        HIP_TRY(hipStreamSynchronize(ctx->copy_stream));
    }
    return 0;
}

static bool pipelined_upload_ok(const wf_params *p, size_t colb) {
    if (getenv("WF_EXP_NO_PIPELINE")) return false;
    if (p->ext_degree != 1) return false;
    const uint32_t S = p->field == WF_FIELD_F64 ? SegCfg<F64>::S : SegCfg<F128>::S;
    const uint32_t n_seg = (p->n_cols * p->n_traces + S - 1) / S;
    if (n_seg < 2) return false;
    const int n_pass = p->field == WF_FIELD_F64 ? seg_plan<F64>(p->log2_trace_len, n_seg).n_pass : seg_plan<F128>(p->log2_trace_len, n_seg).n_pass;
    if (n_pass < 2) return false;
    static const size_t min_bytes = [] {
        const char *e = getenv("WF_EXP_PIPELINE_MIN_BYTES");  // tests lower it; below ~1 MiB per column the events cost more than they hide
        return e ? (size_t)atoll(e) : (size_t)1 << 20;
    }();
    return colb >= min_bytes;
}

// Prover::build_constraint_commitment on device buffers
template <class F>
static int constraint_commit_dev(wf_ctx *ctx, const wf_params *p, const void *d_polys, void *d_lde, void *d_leaves,
                                 void *d_nodes, hipStream_t st, bool dense_rows = false) {
    PathBufs<F> b;
    int rc = path_buffers<F>(ctx, p, b);
    if (rc) return rc;
    rc = run_xpose<F>(ctx, st, true, d_polys, b.segB, (uint64_t)1 << p->log2_trace_len, p->ext_degree,
                      b.total_base_cols, b.n_seg);
    if (rc) return rc;
    return evaluate_and_commit<F>(ctx, st, p, b, d_lde, d_leaves, d_nodes, 0, 0, dense_rows);
}

// One column of E evaluated over the LDE domain straight into a dense vector of n * blowup elements (`d_out`): possible
// when the column goes through the coset-packed kernels (any even number of cosets), whose stores take any row stride.
template <class F>
static bool dense_column_ok(const wf_params *p) {
    uint32_t cpr, lg;
    return packed_shape<F>(1, p->ext_degree, p->ext_degree, 1u << p->log2_blowup, &cpr, &lg);
}

// The same for a whole narrow matrix of one trace (n_cols * ext_degree <= S/2 base columns): rows of exactly that many
// elements.  Used by resident constraint commitments, whose LDE only ever leaves the device through row queries.
static bool dense_matrix_ok(const wf_params *p) {
    if (p->n_traces != 1) return false;
    const uint32_t base = p->n_cols * p->ext_degree;
    uint32_t cpr, lg;
    return p->field == WF_FIELD_F64 ? packed_shape<F64>(1, base, base, 1u << p->log2_blowup, &cpr, &lg)
                                    : packed_shape<F128>(1, base, base, 1u << p->log2_blowup, &cpr, &lg);
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
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->num_cus = cus;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(WF_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
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
    for (auto &b : ctx->pool) (void)hipFree(b.first);
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

int wf_plan_digits(uint32_t field, uint32_t log2_n, uint32_t n_segments, uint32_t digits_out[4]) {
    if (!digits_out) return fail(WF_ERR_ARG, "digits_out is null");
    if (field != WF_FIELD_F64 && field != WF_FIELD_F128) return fail(WF_ERR_FIELD, "unknown field id %u", field);
    if (log2_n < 1 || log2_n > 40) return fail(WF_ERR_TRACE_LENGTH, "transform size out of range");
    const Plan p = field == WF_FIELD_F64 ? seg_plan<F64>(log2_n, n_segments) : seg_plan<F128>(log2_n, n_segments);
    for (int i = 0; i < 4; i++) digits_out[i] = i < p.n_pass ? p.dig[i] : 0;
    return p.n_pass;
}

int wf_ctx_release_cached(wf_ctx *ctx) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
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
size_t wf_digests_bytes(const wf_params *p) { return ((size_t)1 << (p->log2_trace_len + p->log2_blowup)) * 32; }

int wf_trace_commit_dev(wf_ctx *ctx, const wf_params *p, const void *d_trace, void *d_polys, void *d_lde,
                        void *d_leaves, void *d_nodes, void *stream) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    int rc = check_params(p, false);
    if (rc) return rc;
    if (!d_trace || !d_polys || !d_lde) return fail(WF_ERR_ARG, "null device buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    WF_ENTER(ctx, st);
    if (p->field == WF_FIELD_F64) return trace_commit_dev<F64>(ctx, p, d_trace, d_polys, d_lde, d_leaves, d_nodes, st);
    return trace_commit_dev<F128>(ctx, p, d_trace, d_polys, d_lde, d_leaves, d_nodes, st);
}

int wf_constraint_commit_dev(wf_ctx *ctx, const wf_params *p, const void *d_polys, void *d_lde, void *d_leaves,
                             void *d_nodes, void *stream) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    int rc = check_params(p, true);
    if (rc) return rc;
    if (!d_polys || !d_lde) return fail(WF_ERR_ARG, "null device buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    WF_ENTER(ctx, st);
    if (p->field == WF_FIELD_F64) return constraint_commit_dev<F64>(ctx, p, d_polys, d_lde, d_leaves, d_nodes, st);
    return constraint_commit_dev<F128>(ctx, p, d_polys, d_lde, d_leaves, d_nodes, st);
}

// coset-sharded form (one packed commitment spread over several GPUs) ------------------------------------
}  // extern "C"

template <class F>
static int trace_commit_shard_dev(wf_ctx *ctx, const wf_params *p, uint32_t coset0, uint32_t n_cosets,
                                  const void *d_trace, void *d_polys, void *d_lde, void *d_leaves, hipStream_t st) {
    PathBufs<F> b;
    int rc = path_buffers<F>(ctx, p, b, n_cosets);
    if (rc) return rc;
    const uint64_t R = (uint64_t)1 << p->log2_trace_len;
    rc = run_xpose<F>(ctx, st, true, d_trace, b.segA, R, p->ext_degree, b.total_base_cols, b.n_seg);
    if (rc) return rc;
    SegDesc<F> d;
    memset(&d, 0, sizeof(d));
    d.in = b.segA;
    d.work = b.segA;
    d.out = b.segB;
    d.logN = p->log2_trace_len;
    d.n_seg = b.n_seg;
    d.n_cosets = 1;
    d.rows_out = false;
    rc = run_seg_transform<F>(ctx, st, d);
    if (rc) return rc;
    if (d_polys) {
        rc = run_xpose<F>(ctx, st, false, b.segB, d_polys, R, p->ext_degree, b.total_base_cols, b.n_seg);
        if (rc) return rc;
    }
    return evaluate_and_commit<F>(ctx, st, p, b, d_lde, d_leaves, nullptr, coset0, n_cosets);
}

extern "C" {

int wf_trace_commit_shard_dev(wf_ctx *ctx, const wf_params *p, uint32_t coset_begin, uint32_t coset_count,
                              const void *d_trace, void *d_polys, void *d_lde_shard, void *d_leaves_shard,
                              void *stream) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    int rc = check_params(p, false);
    if (rc) return rc;
    if (!d_trace || !d_lde_shard || !d_leaves_shard) return fail(WF_ERR_ARG, "null device buffer");
    const uint32_t blowup = 1u << p->log2_blowup;
    if (coset_count == 0 || coset_begin >= blowup || coset_count > blowup - coset_begin)
        return fail(WF_ERR_ARG, "coset range [%u, %u) is not inside [0, %u)", coset_begin, coset_begin + coset_count, blowup);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    WF_ENTER(ctx, st);
    if (p->field == WF_FIELD_F64)
        return trace_commit_shard_dev<F64>(ctx, p, coset_begin, coset_count, d_trace, d_polys, d_lde_shard, d_leaves_shard, st);
    return trace_commit_shard_dev<F128>(ctx, p, coset_begin, coset_count, d_trace, d_polys, d_lde_shard, d_leaves_shard, st);
}

int wf_merkle_build_dev(wf_ctx *ctx, const void *d_leaves, size_t n_leaves, void *d_nodes, void *stream) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    if (!d_leaves || !d_nodes) return fail(WF_ERR_ARG, "null device buffer");
    if (n_leaves < 2) return fail(WF_ERR_LEAVES, "a tree must have at least 2 leaves");
    if (n_leaves & (n_leaves - 1)) return fail(WF_ERR_LEAVES, "number of leaves must be a power of two");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    WF_ENTER(ctx, st);
    prof_mark(ctx, st, "merkle");
    int rc = run_merkle(st, d_leaves, n_leaves, d_nodes);
    prof_mark(ctx, st, "between_calls");
    return rc;
}

// host-buffer form -------------------------------------------------------------------------------------------------
static int commit_host(wf_ctx *ctx, const wf_params *p, bool constraint, const void *const *cols_in,
                       void *const *polys_out, void *const *lde_out, uint8_t *leaves_out, uint8_t *nodes_out,
                       uint8_t *root_out) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    int rc = check_params(p, constraint);
    if (rc) return rc;
    if (!cols_in) return fail(WF_ERR_ARG, "column pointer array is null");
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    const size_t colb = wf_column_bytes(p), ldeb = wf_lde_bytes(p), digb = wf_digests_bytes(p);
    const size_t TC = (size_t)p->n_cols * p->n_traces;
    for (size_t i = 0; i < TC; i++)
        if (!cols_in[i]) return fail(WF_ERR_ARG, "column %zu is null", i);
    if ((rc = ensure(ctx, ctx->io[0], TC * colb))) return rc;
    if (!constraint && (rc = ensure(ctx, ctx->io[1], TC * colb))) return rc;
    if ((rc = ensure(ctx, ctx->io[2], ldeb * p->n_traces))) return rc;
    if ((rc = ensure(ctx, ctx->io[3], digb))) return rc;
    if ((rc = ensure(ctx, ctx->io[4], digb))) return rc;
    hipStream_t st = ctx->stream;
    if ((rc = upload_columns(ctx, ctx->io[0].p, cols_in, TC, colb, st))) return rc;
    void *d_polys = constraint ? ctx->io[0].p : ctx->io[1].p;
    if (constraint)
        rc = wf_constraint_commit_dev(ctx, p, ctx->io[0].p, ctx->io[2].p, ctx->io[3].p, ctx->io[4].p, st);
    else
        rc = wf_trace_commit_dev(ctx, p, ctx->io[0].p, ctx->io[1].p, ctx->io[2].p, ctx->io[3].p, ctx->io[4].p, st);
    if (rc) return rc;
    if (polys_out && (rc = download_columns(ctx, polys_out, d_polys, TC, colb, st))) return rc;
    if (lde_out)
        for (size_t t = 0; t < p->n_traces; t++)
            if (lde_out[t])
                HIP_TRY(hipMemcpyAsync(lde_out[t], (char *)ctx->io[2].p + t * ldeb, ldeb, hipMemcpyDeviceToHost, st));
    if (leaves_out) HIP_TRY(hipMemcpyAsync(leaves_out, ctx->io[3].p, digb, hipMemcpyDeviceToHost, st));
    if (nodes_out) HIP_TRY(hipMemcpyAsync(nodes_out, ctx->io[4].p, digb, hipMemcpyDeviceToHost, st));
    if (root_out) HIP_TRY(hipMemcpyAsync(root_out, (char *)ctx->io[4].p + 32, 32, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int wf_trace_commit(wf_ctx *ctx, const wf_params *p, const void *const *trace_cols, void *const *polys_out,
                    void *const *lde_out, uint8_t *leaves_out, uint8_t *nodes_out, uint8_t *root_out) {
    return commit_host(ctx, p, false, trace_cols, polys_out, lde_out, leaves_out, nodes_out, root_out);
}

int wf_constraint_commit(wf_ctx *ctx, const wf_params *p, const void *const *poly_cols, void *lde_out,
                         uint8_t *leaves_out, uint8_t *nodes_out, uint8_t *root_out) {
    void *lde_arr[1] = {lde_out};
    return commit_host(ctx, p, true, poly_cols, nullptr, lde_out ? lde_arr : nullptr, leaves_out, nodes_out, root_out);
}

// resident form ------------------------------------------------------------------------------------------------------
struct wf_commitment {
    wf_ctx *ctx;
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
};

// completes an asynchronous commitment: waits for its kernels, fetches the root, gives the pinned slot back
static int commitment_wait(wf_commitment *c) {
    if (!c->pending) return 0;
    int rc = 0;
    if (ctx_alive(c->ctx)) {
        (void)hipSetDevice(c->ctx->device);
        const hipError_t e = hipEventSynchronize(c->done);
        if (e != hipSuccess) rc = fail(WF_ERR_HIP, "the commitment's kernels failed: %s", hipGetErrorString(e));
        if (c->root_slot1) {
            if (rc == 0) memcpy(c->root, c->ctx->root_pin + (size_t)(c->root_slot1 - 1) * 32, 32);
            c->ctx->root_used[c->root_slot1 - 1] = 0;
        }
    }
    if (c->done) (void)hipEventDestroy(c->done);
    c->done = nullptr;
    c->root_slot1 = 0;
    c->pending = false;
    return rc;
}

static void free_commitment(wf_commitment *c) {
    if (!c) return;
    if (ctx_alive(c->ctx)) (void)hipSetDevice(c->ctx->device);
    if (c->pending) (void)commitment_wait(c);
    if (!c->borrowed) {
        pool_free(c->ctx, c->lde, c->lde_bytes);
        pool_free(c->ctx, c->leaves, c->dig_bytes);
        pool_free(c->ctx, c->nodes, c->dig_bytes);
    }
    pool_free(c->ctx, c->polys, c->polys_bytes);
    delete c;
}

// A resident commitment's handle with its buffers (LDE, leaves, nodes, polynomials) taken from the context's pool
static int commitment_alloc(wf_ctx *ctx, const wf_params *p, bool constraint, wf_commitment **out, bool *dense_out) {
    const size_t colb = wf_column_bytes(p), ldeb = wf_lde_bytes(p), digb = wf_digests_bytes(p);
    const size_t TC = (size_t)p->n_cols * p->n_traces;
    wf_commitment *c = new wf_commitment();
    memset(c, 0, sizeof(*c));
    c->ctx = ctx;
    c->p = *p;
    c->n_rows = (uint64_t)1 << (p->log2_trace_len + p->log2_blowup);
    c->epr = (uint64_t)p->n_cols * p->ext_degree;
    // a resident constraint commitment of a narrow matrix keeps its rows dense (no padding to 8 elements: a quarter of
    // the LDE bytes for one E column); nothing outside this library sees the row stride of a resident LDE
    const bool dense = constraint && dense_matrix_ok(p);
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
    const bool pipelined = !constraint && pipelined_upload_ok(p, colb);
    if (pipelined) {
        rc = p->field == WF_FIELD_F64
                 ? trace_commit_pipelined<F64>(ctx, p, cols_in, stage, c->polys, c->lde, c->leaves, c->nodes, st, polys_out)
                 : trace_commit_pipelined<F128>(ctx, p, cols_in, stage, c->polys, c->lde, c->leaves, c->nodes, st, polys_out);
        if (rc) (void)hipStreamSynchronize(ctx->copy_stream ? ctx->copy_stream : st);
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
        rc = p->field == WF_FIELD_F64 ? constraint_commit_dev<F64>(ctx, p, c->polys, c->lde, c->leaves, c->nodes, st, dense)
                                      : constraint_commit_dev<F128>(ctx, p, c->polys, c->lde, c->leaves, c->nodes, st, dense);
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
        HIP_TRY(hipHostMalloc((void **)&ctx->root_pin, WF_ROOT_SLOTS * 32, hipHostMallocDefault));
        ctx->root_used.assign(WF_ROOT_SLOTS, 0);
    }
    for (int i = 0; i < 2; i++) {
        if (!ctx->stage_free[i]) HIP_TRY(hipEventCreateWithFlags(&ctx->stage_free[i], hipEventDisableTiming));
        if (!ctx->upload_done[i]) HIP_TRY(hipEventCreateWithFlags(&ctx->upload_done[i], hipEventDisableTiming));
    }
    uint32_t slot1 = 0;
    for (size_t i = 0; i < WF_ROOT_SLOTS && !slot1; i++)
        if (!ctx->root_used[i]) slot1 = (uint32_t)i + 1;
    if (!slot1) return fail(WF_ERR_BUSY, "%zu asynchronous commitments are in flight: wait for (or destroy) some first", WF_ROOT_SLOTS);
    const int sb = (int)(ctx->async_seq & 1);
    if ((rc = ensure(ctx, ctx->stage[sb], TC * colb))) return rc;
    wf_commitment *c = nullptr;
    bool dense = false;
    if ((rc = commitment_alloc(ctx, p, false, &c, &dense))) return rc;
    hipError_t e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    if (e != hipSuccess) {
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
        rc = p->field == WF_FIELD_F64
                 ? trace_commit_dev<F64>(ctx, p, ctx->stage[sb].p, c->polys, c->lde, c->leaves, c->nodes, st, ctx->stage_free[sb])
                 : trace_commit_dev<F128>(ctx, p, ctx->stage[sb].p, c->polys, c->lde, c->leaves, c->nodes, st, ctx->stage_free[sb]);
    if (rc == 0 && e == hipSuccess)
        e = hipMemcpyAsync(ctx->root_pin + (size_t)(slot1 - 1) * 32, (char *)c->nodes + 32, 32, hipMemcpyDeviceToHost, st);
    if (rc == 0 && e == hipSuccess) e = hipEventRecord(c->done, st);
    if (rc || e != hipSuccess) {
        // whatever was queued from the caller's columns or into this handle's buffers must have drained before either goes away
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamSynchronize(st);
        ctx->stage_busy[sb] = false;
        free_commitment(c);
        return rc ? rc : fail(WF_ERR_HIP, "queueing the commitment failed: %s", hipGetErrorString(e));
    }
    ctx->stage_busy[sb] = true;
    ctx->async_seq++;
    ctx->root_used[slot1 - 1] = 1;
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
    HIP_TRY(hipMemcpyAsync(out, ctx->io[4].p, ids.size() * 32, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
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
        memcpy(q[i].leaves_out, h_out + pt.dig_off, q[i].n * 32);
        memcpy(q[i].nodes_out, h_out + pt.dig_off + q[i].n * 32, pt.total * 32);
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

// FRI layer pieces --------------------------------------------------------------------------------
}  // extern "C"

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
    int rc = run_hash_rows<F>(ctx, st, d_transposed, 0, rows, folding * ext, folding * ext, 1, d_leaves);
    if (rc) return rc;
    prof_mark(ctx, st, "fri.merkle");
    rc = run_merkle(st, d_leaves, rows, d_nodes);
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
    if (leaves_out) HIP_TRY(hipMemcpyAsync(leaves_out, ctx->io[3].p, rows * 32, hipMemcpyDeviceToHost, st));
    if (nodes_out) HIP_TRY(hipMemcpyAsync(nodes_out, ctx->io[4].p, rows * 32, hipMemcpyDeviceToHost, st));
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

// out-of-domain evaluation -------------------------------------------------------------------------
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

int wf_evaluate_polys_over(wf_ctx *ctx, const wf_params *p, const void *const *poly_cols, void *lde_out) {
    return wf_constraint_commit(ctx, p, poly_cols, lde_out, nullptr, nullptr, nullptr);
}

// math::fft building blocks -------------------------------------------------------------------------------------------
static int check_fft_args(wf_ctx *ctx, uint32_t field, uint32_t ext, const void *buf, size_t n, uint32_t *logn) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    if (field != WF_FIELD_F64 && field != WF_FIELD_F128) return fail(WF_ERR_FIELD, "unknown field id %u", field);
    if (ext < 1 || ext > 3 || (field == WF_FIELD_F128 && ext == 3)) return fail(WF_ERR_EXTENSION, "unsupported extension degree %u", ext);
    if (!buf) return fail(WF_ERR_ARG, "buffer is null");
    if (n < 2 || (n & (n - 1))) return fail(WF_ERR_TRACE_LENGTH, "size must be a power of two >= 2");  // fft/mod.rs:89-93
    uint32_t l = 0;
    while (((size_t)1 << l) < n) l++;
    const uint32_t adicity = field == WF_FIELD_F64 ? F64::TWO_ADICITY : F128::TWO_ADICITY;
    if (l > adicity) return fail(WF_ERR_DOMAIN, "no multiplicative subgroup of size 2^%u in this field", l);
    *logn = l;
    return 0;
}

}  // extern "C"

template <class F>
static int fft_host(wf_ctx *ctx, uint32_t ext, void *buf, uint32_t logn, bool inverse, const uint8_t *offset16) {
    typedef typename F::T T;
    const size_t bytes = ((size_t)1 << logn) * ext * sizeof(T);
    int rc;
    if ((rc = ensure(ctx, ctx->io[0], bytes))) return rc;
    if ((rc = ensure(ctx, ctx->io[1], bytes))) return rc;
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(ctx->io[0].p, buf, bytes, hipMemcpyHostToDevice, st));
    XformDesc<F> d;
    memset(&d, 0, sizeof(d));
    d.src = (const T *)ctx->io[0].p;
    d.dst = (T *)ctx->io[1].p;
    d.logN = logn;
    d.W = ext;
    d.batch = 1;
    d.inverse = inverse;
    if (inverse) {
        if (offset16) {
            u128 off;
            memcpy(&off, offset16, 16);
            if (off == 0 || off >= FieldInfo<F>::modulus()) return fail(WF_ERR_OFFSET, "domain offset must be a non-zero field element");
            TableSet *ser;
            rc = series_tables<F>(ctx, logn, F::from_u128_canonical(off), (uint64_t)off, (uint64_t)(off >> 64), &ser);
            if (rc) return rc;
            d.scale_mode = SCALE_SERIES;
            d.out_series = ser;
        } else {
            d.scale_mode = SCALE_CONST;
            d.scale = f_inv<F>(F::from_u128_canonical((u128)1 << logn));
        }
    }
    rc = run_transform<F>(ctx, st, d);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(buf, ctx->io[1].p, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

extern "C" {

int wf_fft_evaluate_poly(wf_ctx *ctx, uint32_t field, uint32_t ext, void *poly, size_t n) {
    uint32_t l;
    int rc = check_fft_args(ctx, field, ext, poly, n, &l);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    return field == WF_FIELD_F64 ? fft_host<F64>(ctx, ext, poly, l, false, nullptr) : fft_host<F128>(ctx, ext, poly, l, false, nullptr);
}

int wf_fft_interpolate_poly(wf_ctx *ctx, uint32_t field, uint32_t ext, void *evals, size_t n) {
    uint32_t l;
    int rc = check_fft_args(ctx, field, ext, evals, n, &l);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    return field == WF_FIELD_F64 ? fft_host<F64>(ctx, ext, evals, l, true, nullptr) : fft_host<F128>(ctx, ext, evals, l, true, nullptr);
}

int wf_fft_interpolate_poly_with_offset(wf_ctx *ctx, uint32_t field, uint32_t ext, void *evals, size_t n,
                                        const uint8_t domain_offset[16]) {
    uint32_t l;
    int rc = check_fft_args(ctx, field, ext, evals, n, &l);
    if (rc) return rc;
    if (!domain_offset) return fail(WF_ERR_ARG, "domain offset is null");
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    return field == WF_FIELD_F64 ? fft_host<F64>(ctx, ext, evals, l, true, domain_offset)
                                 : fft_host<F128>(ctx, ext, evals, l, true, domain_offset);
}

// evaluate_poly_with_offset: one column of E evaluated over the coset LDE domain -> natural-order vector.
// Implemented as the row-major evaluation with a single column (row_width 8) followed by a strided copy-out.
int wf_fft_evaluate_poly_with_offset(wf_ctx *ctx, uint32_t field, uint32_t ext, const void *poly, size_t n,
                                     const uint8_t domain_offset[16], size_t blowup, void *result) {
    uint32_t l;
    int rc = check_fft_args(ctx, field, ext, poly, n, &l);
    if (rc) return rc;
    if (!result || !domain_offset) return fail(WF_ERR_ARG, "null argument");
    // preconditions of the reference function itself (fft/mod.rs:181-201), not those of a trace: any power-of-two size
    // from 2 (periodic columns, periodic_table.rs:44-55) and any power-of-two blowup from 1
    if (blowup < 1 || (blowup & (blowup - 1))) return fail(WF_ERR_BLOWUP, "blowup must be a power of two");
    uint32_t lb = 0;
    while (((size_t)1 << lb) < blowup) lb++;
    if (l < 1) return fail(WF_ERR_TRACE_LENGTH, "polynomial size must be at least 2");
    if (lb > 7) return fail(WF_ERR_BLOWUP, "blowup must be at most 128");
    const uint32_t adicity = field == WF_FIELD_F64 ? F64::TWO_ADICITY : F128::TWO_ADICITY;
    if (l + lb > adicity) return fail(WF_ERR_DOMAIN, "no multiplicative subgroup of size 2^%u in this field", l + lb);
    u128 off;
    memcpy(&off, domain_offset, 16);
    if (off == 0 || off >= (field == WF_FIELD_F64 ? (u128)F64::P : F128::P()))
        return fail(WF_ERR_OFFSET, "domain offset must be a non-zero field element");
    wf_params p;
    memset(&p, 0, sizeof(p));
    p.field = field;
    p.ext_degree = ext;
    p.log2_trace_len = l;
    p.log2_blowup = lb;
    p.n_cols = 1;
    p.n_traces = 1;
    p.digest_bytes = 32;
    memcpy(p.domain_offset, domain_offset, 16);
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    const size_t eb = wf_elem_bytes(field), ldeb = wf_lde_bytes(&p), colb = wf_column_bytes(&p);
    if ((rc = ensure(ctx, ctx->io[0], colb))) return rc;
    if ((rc = ensure(ctx, ctx->io[2], ldeb))) return rc;
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(ctx->io[0].p, poly, colb, hipMemcpyHostToDevice, st));
    const bool dense = field == WF_FIELD_F64 ? dense_column_ok<F64>(&p) : dense_column_ok<F128>(&p);
    rc = field == WF_FIELD_F64 ? constraint_commit_dev<F64>(ctx, &p, ctx->io[0].p, ctx->io[2].p, nullptr, nullptr, st, dense)
                               : constraint_commit_dev<F128>(ctx, &p, ctx->io[0].p, ctx->io[2].p, nullptr, nullptr, st, dense);
    if (rc) return rc;
    const size_t rows = n * blowup, rw = wf_row_width(&p);
    if (dense)  // the device result is the vector itself: one contiguous copy instead of one 16..48-byte piece per row
        HIP_TRY(hipMemcpyAsync(result, ctx->io[2].p, rows * ext * eb, hipMemcpyDeviceToHost, st));
    else
        HIP_TRY(hipMemcpy2DAsync(result, ext * eb, ctx->io[2].p, rw * eb, ext * eb, rows, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int wf_hash_rows(wf_ctx *ctx, uint32_t field, const void *rows, size_t n_rows, size_t row_elems, uint8_t *digests_out) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    if (field != WF_FIELD_F64 && field != WF_FIELD_F128) return fail(WF_ERR_FIELD, "unknown field id %u", field);
    if (!digests_out || (!rows && n_rows * row_elems)) return fail(WF_ERR_ARG, "null argument");
    if (n_rows == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    const size_t bytes = n_rows * row_elems * wf_elem_bytes(field);
    int rc;
    if ((rc = ensure(ctx, ctx->io[2], bytes ? bytes : 16))) return rc;
    if ((rc = ensure(ctx, ctx->io[3], n_rows * 32))) return rc;
    hipStream_t st = ctx->stream;
    if (bytes) HIP_TRY(hipMemcpyAsync(ctx->io[2].p, rows, bytes, hipMemcpyHostToDevice, st));
    if (field == WF_FIELD_F64)
        rc = run_hash_rows<F64>(ctx, st, ctx->io[2].p, 0, n_rows, (uint32_t)row_elems, (uint32_t)row_elems, 1, ctx->io[3].p);
    else
        rc = run_hash_rows<F128>(ctx, st, ctx->io[2].p, 0, n_rows, (uint32_t)row_elems, (uint32_t)row_elems, 1, ctx->io[3].p);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(digests_out, ctx->io[3].p, n_rows * 32, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int wf_merkle_build(wf_ctx *ctx, const uint8_t *leaves, size_t n_leaves, uint8_t *nodes_out) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    if (!leaves || !nodes_out) return fail(WF_ERR_ARG, "null argument");
    if (n_leaves < 2) return fail(WF_ERR_LEAVES, "a tree must have at least 2 leaves");           // merkle/mod.rs:118-120
    if (n_leaves & (n_leaves - 1)) return fail(WF_ERR_LEAVES, "number of leaves must be a power of two");  // :121-123
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    int rc;
    if ((rc = ensure(ctx, ctx->io[3], n_leaves * 32))) return rc;
    if ((rc = ensure(ctx, ctx->io[4], n_leaves * 32))) return rc;
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(ctx->io[3].p, leaves, n_leaves * 32, hipMemcpyHostToDevice, st));
    rc = run_merkle(st, ctx->io[3].p, n_leaves, ctx->io[4].p);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(nodes_out, ctx->io[4].p, n_leaves * 32, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------- resident FRI prover
struct wf_fri_prover {
    wf_ctx *ctx = nullptr;
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
    if (ctx_alive(pr->ctx)) {
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
static int fri_begin_poly_impl(wf_fri_prover *pr, const void *poly, bool poly_on_device, size_t n, size_t lde_blowup) {
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
    const bool dense = pr->field == WF_FIELD_F64 ? dense_column_ok<F64>(&p) : dense_column_ok<F128>(&p);
    if (!dense && (rc = ensure(ctx, ctx->io[2], wf_lde_bytes(&p)))) return rc;
    if ((rc = fri_arena_reserve(pr, rows))) return rc;
    if ((rc = fri_take_evals(pr, rows * pr->ext * eb))) return rc;
    hipStream_t st = ctx->stream;
    if (poly != ctx->io[0].p)
        rc = hipMemcpyAsync(ctx->io[0].p, poly, wf_column_bytes(&p), poly_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st) == hipSuccess
                 ? 0
                 : fail(WF_ERR_HIP, "upload failed");
    if (rc == 0 && dense) {  // the evaluation writes the dense vector itself
        rc = pr->field == WF_FIELD_F64 ? constraint_commit_dev<F64>(ctx, &p, ctx->io[0].p, pr->evals, nullptr, nullptr, st, true)
                                       : constraint_commit_dev<F128>(ctx, &p, ctx->io[0].p, pr->evals, nullptr, nullptr, st, true);
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
    wf_commitment *c = new wf_commitment();
    memset(c, 0, sizeof(*c));
    c->ctx = ctx;
    c->p.field = pr->field;
    c->p.ext_degree = pr->ext;
    c->p.n_cols = pr->folding;
    c->p.n_traces = 1;
    c->p.digest_bytes = 32;
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

#include "comm.hpp"
#include "deep.hpp"
#include "constraint_poly.hpp"
