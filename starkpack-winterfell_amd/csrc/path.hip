// libwf_lde.so, unit 2 of 6 -- the commitment path: pass planner, every launch of the transform / hashing / tree kernels,
// and the C ABI of the path and of the math::fft building blocks.  Replaces Prover::build_trace_commitment /
// build_constraint_commitment (/root/reference/prover/src/lib.rs:615-715) and the math::fft / crypto functions they call.
#include "wf_internal.hpp"

#include "kernels.hpp"
#include "seg_kernels.hpp"
#include "col_kernels.hpp"
#include "fri_kernels.hpp"
#include "tables.hpp"

using namespace wf;

#if defined(WF_EXPERIMENTS) && defined(WF_EXP_STAMPS)
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
// the LDS), which measures ~10 % slower than three passes over smaller tiles (2^22 f64; f128 tiles of 2^10 rows are
// half a CU's LDS in both passes since the strided pass time-shares one table region: 2^20 f128 runs [10, 10],
// 12-15 % faster than [7, 7, 6]).  `max_last`: a maximal digit goes to the last pass.
static Plan make_plan(uint32_t L, uint32_t max_digit, bool avoid_full = false, bool few_tiles = false, bool max_last = false) {
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
    if (max_last && p.n_pass >= 2 && p.dig[0] == max_digit && p.dig[p.n_pass - 1] < max_digit)
        std::swap(p.dig[0], p.dig[p.n_pass - 1]);
    return p;
}

// the plan of the segment kernels (run_seg_transform and the sizing of its work buffer must agree on it)
template <class F>
static Plan seg_plan(uint32_t logN, uint32_t n_seg, uint32_t digit_cap = 0, bool full_tiles = false) {
    // f128: a tile of 2^11 rows (128 KiB) and ONE table region of 2^11 entries are exactly the 160 KiB of a CU.  Used where
    // it makes a transform of 2^11 rows a single pass over many segments (64 packed do_work traces of 2^11 steps: 0.725 ->
    // 0.513 ms); as a digit of a longer transform it loses to smaller tiles (2^21 x 10: [10, 11] 11.1 ms, [7, 7, 7] 10.0 ms).
    uint32_t max_digit = F::BYTES == 8 || (logN == 11 && n_seg > 8) ? 11 : 10;
    // wf_tuning::max_digit (tests / tuning): force more, smaller passes
    if (digit_cap >= 4 && digit_cap < max_digit && (logN + digit_cap - 1) / digit_cap <= 4) max_digit = digit_cap;  // Plan holds 4 digits
    return make_plan(logN, max_digit, !full_tiles && F::BYTES == 8, n_seg <= 8, !full_tiles);
}

// One transform of a batch of columns in the caller's column layout (the stand-alone math::fft entry points, the offset
// interpolation of the constraint side, the FRI remainder): src = [batch] columns of N elements of W coordinates, dst
// likewise, natural order, scaled per scale_mode.  col_kernels.hpp: the lanes of a tile row are adjacent inner positions
// of the one column, the in-LDS transform is the segment kernels' own.
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

template <class F, int W>
static int run_transform_w(wf_ctx *ctx, hipStream_t st, const XformDesc<F> &d) {
    typedef typename F::T T;
    constexpr uint32_t S = SegCfg<F>::S;
    TableSet *tw;
    int rc = root_tables<F>(ctx, d.logN, d.inverse, &tw);
    if (rc) return rc;
    const Plan plan = make_plan(d.logN, F::BYTES == 8 ? 11 : 10);
    const uint64_t N = (uint64_t)1 << d.logN;

    ColArgs<F> a;
    memset(&a, 0, sizeof(a));
    a.logN = d.logN;
    a.col_elems = N;
    a.tw = as_pow2l<F>(*tw);
    // multi-pass transforms go  src -> scratch (first pass), scratch in place (middle), scratch -> dst (last pass):
    // the last pass scatters to natural order and therefore cannot run in place
    T *scratch = nullptr;
    if (plan.n_pass > 1) {
        rc = ensure(ctx, ctx->scratch, (size_t)d.batch * N * W * sizeof(T));
        if (rc) return rc;
        scratch = (T *)ctx->scratch.p;
    }
    auto dims = [&](uint32_t logD, uint32_t &threads, size_t &lds, const void *kern) -> int {
        const size_t D = (size_t)1 << logD;
        lds = (D * S + D) * sizeof(T);
        if (lds > 160 * 1024) return fail(WF_ERR_ARG, "internal: pass needs %zu bytes of LDS", lds);
        threads = (uint32_t)std::min<size_t>(1024, std::max<size_t>(64, D / 2));
        if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        return 0;
    };
    const uint32_t tl_full = std::max<uint32_t>(1, S / W);  // adjacent positions that fill the S lanes of a tile row
    uint32_t done_bits = 0;
    for (int pi = 0; pi + 1 < plan.n_pass; pi++) {
        a.logD = plan.dig[pi];
        a.O = (uint64_t)1 << done_bits;
        a.I = N >> (done_bits + a.logD);
        a.Tl = (uint32_t)std::min<uint64_t>(tl_full, a.I);
        a.src = pi == 0 ? d.src : scratch;
        a.dst = scratch;
        rc = digit_table<F>(ctx, a.logD, d.inverse, &a.digit_tw);
        if (rc) return rc;
        const void *kern = d.inverse ? (const void *)k_col_strided<F, W, -1> : (const void *)k_col_strided<F, W, 1>;
        uint32_t threads;
        size_t lds;
        if ((rc = dims(a.logD, threads, lds, kern))) return rc;
        const uint64_t grid = (uint64_t)d.batch * a.O * (a.I / a.Tl);
        if (grid > 0x7FFFFFFFull) return fail(WF_ERR_ARG, "problem too large for one launch (%llu groups)", (unsigned long long)grid);
        prof_mark(ctx, st, d.inverse ? "fft.interpolate.strided_pass" : "fft.evaluate.strided_pass");
        void *kargs[] = {&a};
        HIP_TRY(hipLaunchKernel(kern, dim3((uint32_t)grid), dim3(threads), kargs, lds, st));
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
        a.Tl = single ? 1 : std::min<uint32_t>(tl_full, 1u << plan.dig[0]);
        a.src = single ? d.src : scratch;
        a.dst = d.dst;
        rc = digit_table<F>(ctx, a.logD, d.inverse, &a.digit_tw);
        if (rc) return rc;
        const void *kern = d.inverse ? (const void *)k_col_last<F, W, -1> : (const void *)k_col_last<F, W, 1>;
        uint32_t threads;
        size_t lds;
        if ((rc = dims(a.logD, threads, lds, kern))) return rc;
        const uint64_t grid = (uint64_t)d.batch * (a.O / a.Tl);
        if (grid > 0x7FFFFFFFull) return fail(WF_ERR_ARG, "problem too large for one launch (%llu groups)", (unsigned long long)grid);
        prof_mark(ctx, st, d.inverse ? "fft.interpolate.last_pass" : "fft.evaluate.last_pass");
        void *kargs[] = {&a};
        HIP_TRY(hipLaunchKernel(kern, dim3((uint32_t)grid), dim3(threads), kargs, lds, st));
        prof_mark(ctx, st, "between_calls");
    }
    return 0;
}

template <class F>
static int run_transform(wf_ctx *ctx, hipStream_t st, const XformDesc<F> &d) {
    switch (d.W) {
        case 1: return run_transform_w<F, 1>(ctx, st, d);
        case 2: return run_transform_w<F, 2>(ctx, st, d);
        case 3:
            if constexpr (F::FIELD_ID == 1) return run_transform_w<F, 3>(ctx, st, d);
            [[fallthrough]];
        default: return fail(WF_ERR_EXTENSION, "unsupported extension degree %u", d.W);
    }
}

template <class F>
static int seg_launch_dims(uint32_t logD, uint32_t &threads, size_t &lds, bool last_pass) {
    const size_t D = (size_t)1 << logD;
    // tile + digit twiddles (+ the factor table of a strided pass; a last pass keeps its input factors where the
    // twiddles go afterwards: a 2^10-row f128 tile is 80 KiB, two work-groups per CU)
    constexpr bool one_table = F::BYTES == 16;  // k_seg_strided, ONE_TABLE: the f128 strided pass time-shares one table region
    lds = (D * SegCfg<F>::S + (last_pass || one_table ? 1 : 2) * D) * sizeof(typename F::T);
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

static void launch_merge_chunks(hipStream_t st, const void *cvs, uint32_t n_chunks, uint64_t n_rows, void *leaves, uint32_t dw);

// per-XCD ticket / exit counters of the persistent kernels: zeroed at first use (the kernels leave them at zero: the last work-group
// to sign off resets them) and again whenever any call of the library has reported a failure since -- a launch that did not run
// to its sign-off (a failed launch behind it, a fault) must not leave its counters to the next commitment
static int ensure_tickets(wf_ctx *ctx, hipStream_t st) {
    const uint64_t epoch = fail_epoch();
    if (ctx->tickets.p && ctx->tickets_epoch == epoch) return 0;
    if (!ctx->tickets.p) {
        int rc = ensure(ctx, ctx->tickets, 128);
        if (rc) return rc;
    }
    HIP_TRY(hipMemsetAsync(ctx->tickets.p, 0, 128, st));  // ordered on the launch stream, behind whatever used the counters before
    ctx->tickets_epoch = epoch;
    return 0;
}

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
    uint32_t digest_words = 8;  // 6: Blake3_192 leaves
    bool *fused = nullptr;
    const TableSet *pre;
    const T *fin_tab = nullptr;  // rows_out, single-pass plan: [blowup][N] input factors h_c^k (FTAB, coset_row_factors); nullptr: rebuilt per tile
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
    const Plan plan = seg_plan<F>(d.logN, d.n_seg, ctx->tune.max_digit, ctx->tune.full_tiles);
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
    a.digest_words = d.digest_words ? d.digest_words : 8;  // (descriptors are memset to zero by their users)
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

    // Tail packing (k_seg_last_hash_tp): one trace of several segments whose last segment is at most half full, in the shapes
    // that take the fused persistent last pass anyway -- the tail segment is evaluated coset-packed (two cosets per tile row:
    // launch B of every strided pass, the PACKED instantiation) and finished by one tail tile per coset pair in the last pass.
    constexpr uint32_t S_ = SegCfg<F>::S;
    const uint32_t nf = d.n_seg - 1, tail_cols = d.total_base_cols - nf * S_;
    bool tailpack = false;
    if (d.rows_out && !packed && d.phase == 0 && d.n_seg >= 2 && d.n_seg <= 16 && d.total_base_cols == d.base_cols &&
        tail_cols >= 1 && tail_cols * 2 <= S_ && d.n_cosets % 2 == 0 && plan.n_pass >= 2 && d.leaves && !ctx->tune.no_fused_hash &&
        !ctx->tune.no_persistent && !ctx->tune.no_tail_pack) {
        const uint32_t logDl = plan.dig[plan.n_pass - 1];
        const uint64_t tickets_tp = (uint64_t)(d.n_cosets / 2) << (d.logN - logDl);
        const uint64_t launch_rows = (uint64_t)d.n_cosets << d.logN;
        // measured (profiles/r04_tail_pack.txt): f128 -3 .. -8 % on every shape tried; f64 -3 % with 2^9 / 2^10-row last tiles, +4 %
        // with the 2^7-row tiles of the 2^22 plan (their specialised instantiation exists for the unpacked kernel only)
        tailpack = logDl >= (ctx->tune.persistent_always || F::BYTES == 16 ? 7u : 9u) && logDl <= 10 && tickets_tp % 8 == 0 &&
                   tickets_tp < (1ull << 31) &&
                   (ctx->tune.persistent_always || launch_rows >= (1ull << 20));  // (as the ticket kernel's own rule for one trace)
    }
    // (the work buffer of EVERY caller holds n_cosets * n_seg * N * S elements -- path_buffers sizes it so, with n_cosets the cosets of this call:
    // all of them, or a rank's share -- and the tail region ends at n_cosets * (nf + 1/2) * N * S <= that; the pipelined upload (phase 1 / 2)
    // never takes this route: tests/test_gpu_tail_pack.py runs it through the unsharded and the coset-sharded entry points)
    T *work_tail = tailpack ? d.work + (size_t)d.n_cosets * nf * N * S_ : nullptr;  // [coset pair][N][S] behind the full segments

    const uint32_t run_cnt = tailpack ? nf : (d.seg_cnt ? d.seg_cnt : d.n_seg);
    const size_t run_off = (size_t)d.seg0 * (N * SegCfg<F>::S);  // elements in front of segment seg0 within one coset
    a.seg_stride = tailpack ? nf : d.n_seg;
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
        // first pass of a coset evaluation: the cosets of a tile back to back on one XCD (coset_inner_split, seg_kernels.hpp)
        auto coset_inner = [&](uint32_t cosets, uint64_t tiles_per_coset) -> uint32_t {
            if (!(d.rows_out && first) || ctx->tune.no_coset_inner || cosets < 2 || (cosets & (cosets - 1)) || (tiles_per_coset & 63)) return 0u;
            uint32_t lc = 0;
            while ((1u << lc) < cosets) lc++;
            return lc + 1;
        };
        prof_mark(ctx, st, tag_s);
        // f64 tiles of 2^10 and 2^9 rows (the digits of the 2^17 .. 2^21 plans) run the tile-size-specialised instantiations
        // (seg_kernels.hpp, WF_TILE_BOUNDS: strided pass of cfg 2 0.347 -> 0.324 ms); everything else the generic kernel
        // (2^7 / 2^8-row tiles and the f128 tiles measured slower or the same specialised: docs/EXPERIMENTS.md).
        const bool spec_ok = F::BYTES == 8 && !packed && threads * 2 == (1u << a.logD) && !ctx->tune.no_specialized;
        const void *kern = nullptr;
        if (spec_ok) {
            constexpr bool F8 = F::BYTES == 8;  // (the specialised instantiations exist for f64 only)
            switch (a.logD) {
                case 10: kern = d.rows_out ? (const void *)k_seg_strided<F, 1, false, F8 ? 10 : 0> : (const void *)k_seg_strided<F, 0, false, F8 ? 10 : 0>; break;
                case 9: kern = d.rows_out ? (const void *)k_seg_strided<F, 1, false, F8 ? 9 : 0> : (const void *)k_seg_strided<F, 0, false, F8 ? 9 : 0>; break;
                default: break;
            }
        }
        // transforms of three and more passes: their small tiles (2^7 / 2^8 rows) go two adjacent inner positions at a time --
        // tile rows of 128 contiguous bytes (k_seg_strided_wide: cfg 3 35.5 -> 33.9 ms; 4 and 8 positions measured slower)
        uint32_t ti = 0;
        if (F::BYTES == 8 && !packed && plan.n_pass >= 3 && a.logD >= 4 && a.logD <= 8 && ctx->tune.wide_ti != 1) {
            ti = ctx->tune.wide_ti ? ctx->tune.wide_ti : 2u;
            while (ti > 1 && (ti > a.I || ((size_t)(ti * SegCfg<F>::S + 2 + ti) << a.logD) * sizeof(T) > 80 * 1024)) ti >>= 1;
            if (ti < 2 || ((1u << a.logD) * ti * SegCfg<F>::S) / 16 < 64) ti = 0;
        }
        a.coset_inner = coset_inner(n_groups, ((uint64_t)run_cnt * a.O * a.I) / (ti ? ti : 1u));
        // first pass of a coset evaluation through k_seg_strided: output factors from a global table, the coset's h_c^i merged into
        // the input factors (GTAB1, seg_kernels.hpp).  D x I <= 2^20 entries (4 MiB for cfg 5's [9, 9] plan).  f128 only: there the
        // per-tile table arithmetic is two 78-instruction products per entry (cfg 5's strided pass 0.433 -> 0.423 ms, three
        // interleaved rounds); for f64 the saved 3 % of the pass's instructions are paid back in the prologue's extra load latency
        // (cfg 2: 1.374 against 1.365 ms over four rounds; WF_EXP_GTAB1_F64=1 to try again) -- profiles/r05_gtab1_ab.txt
        a.fout_tab = nullptr;
        if (!ti && d.rows_out && first && !packed && a.pre_on && !a.scale_on && d.logN <= 20 && !ctx->tune.no_gtab1 &&
            (F::BYTES == 16 || ctx->tune.gtab1_f64)) {
            rc = pass_factor_table<F>(ctx, d.logN, a.logD, (uint32_t)(d.logN - a.logD), inverse, &a.fout_tab);
            if (rc) return rc;
        }
        if (ti) {
            if constexpr (F::BYTES == 8) {
                const void *kw = ti == 8 ? (d.rows_out ? (const void *)k_seg_strided_wide<F, 1, 8> : (const void *)k_seg_strided_wide<F, 0, 8>)
                               : ti == 4 ? (d.rows_out ? (const void *)k_seg_strided_wide<F, 1, 4> : (const void *)k_seg_strided_wide<F, 0, 4>)
                                         : (d.rows_out ? (const void *)k_seg_strided_wide<F, 1, 2> : (const void *)k_seg_strided_wide<F, 0, 2>);
                size_t lds_w = ((size_t)(ti * SegCfg<F>::S + 2 + ti) << a.logD) * sizeof(T);
                // passes after the first: the output factors depend on (inner position, k) only -- a table in global memory
                // (<= 4 MiB, built once per context) instead of one rebuilt in LDS per tile: a work-group less LDS, a work-group more per CU
                const uint32_t logI_ = (uint32_t)(d.logN - done_bits - a.logD);
                a.fout_tab = nullptr;
                if (!first && !a.scale_on && !a.pre_on && a.logD + logI_ <= 19 && !ctx->tune.no_gtab) {
                    rc = pass_factor_table<F>(ctx, d.logN, a.logD, logI_, inverse, &a.fout_tab);
                    if (rc) return rc;
                    kw = ti == 8 ? (d.rows_out ? (const void *)k_seg_strided_wide<F, 1, 8, 1> : (const void *)k_seg_strided_wide<F, 0, 8, 1>)
                       : ti == 4 ? (d.rows_out ? (const void *)k_seg_strided_wide<F, 1, 4, 1> : (const void *)k_seg_strided_wide<F, 0, 4, 1>)
                                 : (d.rows_out ? (const void *)k_seg_strided_wide<F, 1, 2, 1> : (const void *)k_seg_strided_wide<F, 0, 2, 1>);
                    lds_w = ((size_t)(ti * SegCfg<F>::S + 1) << a.logD) * sizeof(T);
                }
                // the FIRST pass of a coset evaluation likewise (round 5): output factors w_N^(k i) from the global table ([I][D]: 32 MiB
                // for cfg 3's first pass, built once per context), the coset's h_c^i merged into the input factors ([TI][D] in LDS:
                // h_c^(d I + i)) -- three products per output-table entry and tile gone from a pass that sits at the vector ALU:
                // cfg 3 33.13 -> 32.81 ms, three interleaved rounds (profiles/r05_gtab1_ab.txt)
                if (first && d.rows_out && a.pre_on && !a.scale_on && ti == 2 && a.logD + logI_ <= 22 && !ctx->tune.no_gtab1_wide) {
                    rc = pass_factor_table<F>(ctx, d.logN, a.logD, logI_, inverse, &a.fout_tab);
                    if (rc) return rc;
                    kw = (const void *)k_seg_strided_wide<F, 1, 2, 2>;
                    lds_w = ((size_t)(ti * SegCfg<F>::S + 1 + ti) << a.logD) * sizeof(T);
                }
                // the evaluation passes of the three-pass plans on 2^7 / 2^8-row tiles: tile-size-specialised instantiations (same generic round loop)
                if (ti == 2 && d.rows_out && !ctx->tune.no_specialized) {
                    if (kw == (const void *)k_seg_strided_wide<F, 1, 2, 2>)
                        kw = a.logD == 8 ? (const void *)k_seg_strided_wide<F, 1, 2, 2, 8> : a.logD == 7 ? (const void *)k_seg_strided_wide<F, 1, 2, 2, 7> : kw;
                    else if (kw == (const void *)k_seg_strided_wide<F, 1, 2, 1>)
                        kw = a.logD == 8 ? (const void *)k_seg_strided_wide<F, 1, 2, 1, 8> : a.logD == 7 ? (const void *)k_seg_strided_wide<F, 1, 2, 1, 7> : kw;
                }
                const uint32_t threads_w = ((1u << a.logD) * ti * SegCfg<F>::S) / 16;
                if (lds_w > 64 * 1024) HIP_TRY(hipFuncSetAttribute(kw, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                void *kargs[] = {&a};
                HIP_TRY(hipLaunchKernel(kw, dim3((uint32_t)(grid / ti)), dim3(threads_w), kargs, lds_w, st));
            }
        } else if (kern) {
            void *kargs[] = {&a};
            HIP_TRY(hipLaunchKernel(kern, dim3((uint32_t)grid), dim3(threads), kargs, lds, st));
        } else if (d.rows_out && packed)
            hipLaunchKernelGGL((k_seg_strided<F, 1, true>), dim3((uint32_t)grid), dim3(threads), lds, st, a);
        else if (d.rows_out)
            hipLaunchKernelGGL((k_seg_strided<F, 1>), dim3((uint32_t)grid), dim3(threads), lds, st, a);
        else
            hipLaunchKernelGGL((k_seg_strided<F, 0>), dim3((uint32_t)grid), dim3(threads), lds, st, a);
        HIP_TRY(hipGetLastError());
        if (tailpack) {  // launch B: the tail segment, two cosets per tile row
            SegArgs<F> b = a;
            b.src = first ? d.in + (size_t)nf * N * S_ : work_tail;
            b.dst = work_tail;
            b.src_shared = first ? 1 : 0;
            b.n_seg = 1;
            b.seg_stride = 1;
            b.n_cosets = d.n_cosets / 2;
            b.cpr_log = 1;
            b.lg_log = S_ == 8 ? 2 : 1;
            b.base_cols = b.total_base_cols = tail_cols;
            const uint64_t grid_b = (uint64_t)(d.n_cosets / 2) * a.O * a.I;
            b.coset_inner = coset_inner(d.n_cosets / 2, a.O * a.I);
            hipLaunchKernelGGL((k_seg_strided<F, 1, true>), dim3((uint32_t)grid_b), dim3(threads), lds, st, b);
            HIP_TRY(hipGetLastError());
        }
        done_bits += a.logD;
    }
    a.n_seg = d.n_seg;
    a.seg_stride = d.n_seg;
    a.coset_inner = 0;
    a.fout_tab = nullptr;
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
        a.fin_tab = (single && d.rows_out && !packed) ? d.fin_tab : nullptr;
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
        const bool fuse_on = d.rows_out && d.leaves && !ctx->tune.no_fused_hash;
        const bool may_fuse = fuse_on && !packed;
        // rows of more than 16 segments (one BLAKE3 chunk) are fused chunk by chunk: the pass leaves chunk chaining values
        const bool chunked = d.n_seg > 16 && !ctx->tune.no_chunked;
        const uint32_t n_chunks = chunked ? (d.n_seg + 15) / 16 : 1;
        const uint64_t tickets = (uint64_t)d.n_cosets * a.O * n_chunks;
        const uint64_t launch_rows = (uint64_t)d.n_cosets << d.logN;
        // one segment of one trace: k_seg_last hashes its rows itself, and with tiles below 2^10 rows one work-group per
        // tile beats the ticket kernel (2^14..2^18 x 8: -3..-14 %, 2^22 x 8: -10 %; 2^20 x 8, 2^10-row tiles: +5 %)
        const bool one_seg = d.n_seg == 1 && d.total_base_cols == d.base_cols;
        const bool always = ctx->tune.persistent_always;  // (tests: the ticket kernel on every shape it can run)
        const bool persistent = may_fuse && !single && (d.n_seg <= 16 || chunked) && threads * 2 == (1u << a.logD) &&
                                (always || !one_seg || a.logD >= 10) &&
                                // several segments of one trace, rows of one chunk: below 2^20 LDE rows the separate
                                // row-hash kernel costs less than the ticket kernel's small tiles (2^14 x 16: -15 %)
                                (always || one_seg || chunked || d.total_base_cols != d.base_cols || launch_rows >= (1ull << 20)) &&
                                tickets % 8 == 0 && tickets < (1ull << 31) && launch_rows * n_chunks * 32 < (1ull << 40) &&
                                !ctx->tune.no_persistent;
        // coset-packed rows are hashed in the pass where the separate kernel is the slower one (measured): f128, four
        // lanes per coset, or rows gathered from several traces; one- and two-lane f64 rows keep k_hash_rows
        const bool fuse_packed = fuse_on && packed && (F::BYTES == 16 || a.lg_log >= 2 || d.total_base_cols != d.base_cols);
        const bool fuse = persistent || (may_fuse && d.n_seg == 1 && d.total_base_cols == d.base_cols) || fuse_packed;
        a.leaves = fuse ? (uint32_t *)d.leaves : nullptr;
        a.hash_epr = d.hash_epr;
        if (d.fused) *d.fused = fuse;
        prof_mark(ctx, st, tag_l);
        if (tailpack) {
            const bool small = threads <= 256;
            const void *kern = small ? (const void *)k_seg_last_hash_tp<F, true> : (const void *)k_seg_last_hash_tp<F, false>;
            if (!ctx->tune.no_specialized) {
                if (a.logD == 9 && small) kern = (const void *)k_seg_last_hash_tp<F, true, 9>;
                if (F::BYTES == 8 && a.logD == 10 && !small) kern = (const void *)k_seg_last_hash_tp<F, false, F::BYTES == 8 ? 10 : 0>;
            }
            a.src = d.work;
            a.src_tail = work_tail;
            a.tail_cols = tail_cols;
            a.store_cols = a.total_store_cols = d.base_cols;  // (the tail tile writes the rows' zero padding itself)
            a.tail_pad = 0;
            a.pad_traces = 0;
            a.leaves = (uint32_t *)d.leaves;
            a.hash_epr = d.hash_epr;
            if (d.fused) *d.fused = true;
            if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            const uint64_t tickets_tp = (uint64_t)(d.n_cosets / 2) * a.O;
            const uint64_t resident = (uint64_t)ctx->num_cus * std::max<size_t>(1, (160 * 1024) / lds);
            int rcq = ensure_tickets(ctx, st);
            if (rcq) return rcq;
            a.tile_counters = (uint32_t *)ctx->tickets.p;
            void *kargs[] = {&a};
            HIP_TRY(hipLaunchKernel(kern, dim3((uint32_t)std::min<uint64_t>(tickets_tp, resident)), dim3(threads), kargs, lds, st));
        } else if (persistent) {
            const bool multi = d.n_seg > 1 || d.total_base_cols != d.base_cols;
            const bool small = threads <= 256;  // the multi-segment variants without the 128-VGPR cap (2^22 x 64: last pass -3 %)
            const void *kern =
                chunked ? (a.pad_traces ? (small ? (const void *)k_seg_last_hash<F, true, true, true, true> : (const void *)k_seg_last_hash<F, true, true, true>)
                                        : (small ? (const void *)k_seg_last_hash<F, true, false, true, true> : (const void *)k_seg_last_hash<F, true, false, true>))
                : multi ? (a.pad_traces ? (small ? (const void *)k_seg_last_hash<F, true, true, false, true> : (const void *)k_seg_last_hash<F, true, true>)
                                        : (small ? (const void *)k_seg_last_hash<F, true, false, false, true> : (const void *)k_seg_last_hash<F, true, false>))
                        : (a.pad_traces ? (const void *)k_seg_last_hash<F, false, true> : (const void *)k_seg_last_hash<F, false, false>);
            // one segment of one trace in 2^10-row f64 tiles (the bench workload): the tile-size-specialised instantiation
            if (F::BYTES == 8 && !chunked && !multi && !a.pad_traces && a.logD == 10 && !ctx->tune.no_specialized)
                kern = (const void *)k_seg_last_hash<F, false, false, false, false, F::BYTES == 8 ? 10 : 0>;
            // several segments in 2^10-row tiles (2^20 x 64: last pass 3.68 -> 3.55 ms; the STARKPack shape of eight packed traces
            // of eight columns: 3.62 -> 3.52)
            if (F::BYTES == 8 && !chunked && multi && a.logD == 10 && !small && !ctx->tune.no_specialized)
                kern = a.pad_traces ? (const void *)k_seg_last_hash<F, true, true, false, false, F::BYTES == 8 ? 10 : 0>
                                    : (const void *)k_seg_last_hash<F, true, false, false, false, F::BYTES == 8 ? 10 : 0>;
            // rows longer than a BLAKE3 chunk (chunk by chunk inside the pass): 2^10-row tiles (2^20 x 200: last pass 14.39 -> 14.08 ms),
            // 2^9-row tiles (2^18 x 255: 3.65 -> 3.55)
            if (F::BYTES == 8 && chunked && !a.pad_traces && !ctx->tune.no_specialized) {
                if (a.logD == 10 && !small) kern = (const void *)k_seg_last_hash<F, true, false, true, false, F::BYTES == 8 ? 10 : 0>;
                if (a.logD == 9 && small) kern = (const void *)k_seg_last_hash<F, true, false, true, true, F::BYTES == 8 ? 9 : 0>;
            }
            // several segments in 2^7-row tiles (the last digit of the 2^22 plan: cfg 3's last pass 11.44 -> 11.12 ms)
            if (F::BYTES == 8 && !chunked && multi && !a.pad_traces && a.logD == 7 && small && !ctx->tune.no_specialized)
                kern = (const void *)k_seg_last_hash<F, true, false, false, true, F::BYTES == 8 ? 7 : 0>;
            // ... and in 2^9-row tiles (2^18 x 32: last pass 0.547 -> 0.521 ms)
            if (F::BYTES == 8 && !chunked && multi && !a.pad_traces && a.logD == 9 && small && !ctx->tune.no_specialized)
                kern = (const void *)k_seg_last_hash<F, true, false, false, true, F::BYTES == 8 ? 9 : 0>;
            // f128, 2^9-row tiles (cfg 5: last pass 0.513 -> 0.500 ms)
            if (F::BYTES == 16 && !chunked && multi && !a.pad_traces && a.logD == 9 && small && !ctx->tune.no_specialized)
                kern = (const void *)k_seg_last_hash<F, true, false, false, true, F::BYTES == 16 ? 9 : 0>;
            // ... and in 2^8-row tiles (2^17 x 32: last pass 0.278 -> 0.259 ms)
            if (F::BYTES == 8 && !chunked && multi && !a.pad_traces && a.logD == 8 && small && !ctx->tune.no_specialized)
                kern = (const void *)k_seg_last_hash<F, true, false, false, true, F::BYTES == 8 ? 8 : 0>;
            if (chunked) {
                int rcc = ensure(ctx, ctx->hash_tmp, (size_t)launch_rows * n_chunks * 32);
                if (rcc) return rcc;
                a.chunk_cvs = (uint32_t *)ctx->hash_tmp.p;
                a.n_chunks = n_chunks;
            }
            if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            const size_t lds_p = lds;  // (the two ticket words live in the unused last twiddle slot)
            const uint64_t resident = (uint64_t)ctx->num_cus * std::max<size_t>(1, (160 * 1024) / lds_p);
            int rcq = ensure_tickets(ctx, st);
            if (rcq) return rcq;
            a.tile_counters = (uint32_t *)ctx->tickets.p;
#if defined(WF_EXPERIMENTS) && defined(WF_EXP_STAMPS)
            a.stamps = exp_stamps_buffer();
#endif
            void *kargs[] = {&a};
            HIP_TRY(hipLaunchKernel(kern, dim3((uint32_t)std::min<uint64_t>(tickets, resident)), dim3(threads), kargs, lds_p, st));
            if (chunked) {
                HIP_TRY(hipGetLastError());
                launch_merge_chunks(st, ctx->hash_tmp.p, n_chunks, launch_rows, d.leaves, a.digest_words);
            }
        } else if (d.rows_out && packed)
            hipLaunchKernelGGL((k_seg_last<F, SEG_OUT_ROWS, true>), dim3((uint32_t)grid), dim3(threads), lds, st, a);
        else {
            // one work-group per tile: 2^10- and 2^9-row f64 tiles run the tile-size-specialised instantiations
            const void *kern = d.rows_out ? (const void *)k_seg_last<F, SEG_OUT_ROWS> : (const void *)k_seg_last<F, SEG_OUT_SEG>;
            constexpr bool F8 = F::BYTES == 8;
            if (F8 && threads * 2 == (1u << a.logD) && !ctx->tune.no_specialized) {
                if (a.logD == 10)
                    kern = d.rows_out ? (const void *)k_seg_last<F, SEG_OUT_ROWS, false, F8 ? 10 : 0> : (const void *)k_seg_last<F, SEG_OUT_SEG, false, F8 ? 10 : 0>;
                else if (a.logD == 9)
                    kern = d.rows_out ? (const void *)k_seg_last<F, SEG_OUT_ROWS, false, F8 ? 9 : 0> : (const void *)k_seg_last<F, SEG_OUT_SEG, false, F8 ? 9 : 0>;
            }
            if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            void *kargs[] = {&a};
            HIP_TRY(hipLaunchKernel(kern, dim3((uint32_t)grid), dim3(threads), kargs, lds, st));
        }
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
static void launch_merge_chunks(hipStream_t st, const void *cvs, uint32_t n_chunks, uint64_t n_rows, void *leaves, uint32_t dw) {
    // few, long rows: 16 lanes per row; otherwise one lane per row (measured: 8192 rows x 80 chunks 0.48 -> 0.37 ms for
    // chunks + merge, but 32768 x 20 and shorter rows are faster with a lane per row)
    if (n_chunks >= 32 && n_chunks <= 128 && n_rows <= 65536)
        hipLaunchKernelGGL(k_hash_merge_chunks_par, dim3((uint32_t)((n_rows + 15) / 16)), dim3(256), (size_t)16 * n_chunks * 32, st,
                           (const uint32_t *)cvs, n_chunks, n_rows, (uint32_t *)leaves, dw);
    else
        hipLaunchKernelGGL(k_hash_merge_chunks, dim3((uint32_t)((n_rows + 255) / 256)), dim3(256), 0, st, (const uint32_t *)cvs,
                           n_chunks, n_rows, (uint32_t *)leaves, dw);
}

template <class F>
static int run_hash_rows(wf_ctx *ctx, hipStream_t st, const void *lde, uint64_t trace_elems, uint64_t n_rows, uint32_t row_width,
                         uint32_t epr, uint32_t n_traces, void *leaves, uint32_t dw = 8) {
    HashArgs<F> h;
    h.lde = (const typename F::T *)lde;
    h.trace_elems = trace_elems;
    h.n_rows = n_rows;
    h.row_width = row_width;
    h.epr = epr;
    h.n_traces = n_traces;
    h.leaves = (uint32_t *)leaves;
    h.digest_words = dw;
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
        // 16-byte units that never straddle two traces' rows (always for f128; f64 with an even number of columns): the staged form
        // (round 5, kernels.hpp) -- a wave takes 16 rows x 4 chunks, quads of lanes load 64-byte pieces, one lane per (row, chunk)
        // hashes from a wave-private LDS region; every element is fetched once
        constexpr uint32_t EPL = 16 / F::BYTES;
        if (epr % EPL == 0 && !ctx->tune.no_staged_chunks) {
            const uint64_t waves = ((n_rows + 15) / 16) * ((chunks + 3) / 4);
            if ((waves + 3) / 4 > 0x7FFFFFFFull) return fail(WF_ERR_ARG, "rows too long for one launch");
            hipLaunchKernelGGL(k_hash_chunks_staged<F>, dim3((uint32_t)((waves + 3) / 4)), dim3(256), 0, st, h, (uint32_t)chunks,
                               (uint32_t *)ctx->hash_tmp.p);
        } else {
            // a lane walks its own row.  40 KiB of (unused) dynamic LDS per work-group = three work-groups, 12 waves per CU: the lanes
            // gather 16-byte elements a row apart, and more resident waves only evict each other's lines from the 32 KiB L1 (round 4:
            // 512 x 2^10 x 10 f128 0.387 -> 0.343 ms for the hashing launches, profiles/r04_attribution.txt)
            hipLaunchKernelGGL(k_hash_chunks<F>, dim3((uint32_t)g2), dim3(threads), 40 * 1024, st, h, (uint32_t)chunks,
                               (uint32_t *)ctx->hash_tmp.p);
        }
        HIP_TRY(hipGetLastError());
        launch_merge_chunks(st, ctx->hash_tmp.p, (uint32_t)chunks, n_rows, leaves, dw);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int DW>
static int run_merkle_dw(wf_ctx *ctx, hipStream_t st, const void *leaves, uint64_t n_leaves, void *nodes) {
    // (nodes[0] = Digest::default(), merkle/mod.rs:355, is written by the launch that produces the root)
    const uint32_t *children = (const uint32_t *)leaves;
    uint64_t n_children = n_leaves;
    while (n_children > 1) {
        const uint64_t n_par = n_children >> 1;
        const uint32_t threads = 256;
        const uint64_t grid = (n_par + threads - 1) / threads;
        // levels of >= 2^18 parents: two per launch; below that the LDS subtree kernel folds 9 levels per launch (one
        // launch less than switching at 2^16, 5 us at 2^23 leaves).  WF_EXP_MERKLE_L2_MIN: tuning switch
        const uint32_t l2_min = ctx->tune.merkle_l2_min;
        if (n_par >= ((uint64_t)1 << l2_min)) {  // two levels that still fill the chip: one lane per grandparent
            const uint64_t n_grand = n_par >> 1;
            const uint64_t blocks2 = std::min<uint64_t>((n_grand + threads - 1) / threads, (uint64_t)ctx->num_cus * 8);  // grid-stride
            hipLaunchKernelGGL(k_merkle_level2<DW>, dim3((uint32_t)blocks2), dim3(threads), 0, st,
                               children, (uint32_t *)nodes + n_par * 8, (uint32_t *)nodes + n_grand * 8, n_grand);
            HIP_TRY(hipGetLastError());
            n_children = n_grand;
        } else if (n_par >= (1u << 15) && l2_min == 16) {  // a level that still fills the chip: one lane per node
            hipLaunchKernelGGL(k_merkle_level<DW>, dim3((uint32_t)grid), dim3(threads), 0, st, children,
                               (uint32_t *)nodes + n_par * 8, n_par);
            HIP_TRY(hipGetLastError());
            n_children = n_par;
        } else {                    // the top of the tree: up to 9 levels per launch through LDS
            uint32_t total_levels = 0;
            for (uint64_t t = n_children; t > 1; t >>= 1) total_levels++;
            const uint32_t levels = std::min<uint32_t>(9, total_levels);
            hipLaunchKernelGGL(k_merkle_subtree<DW>, dim3((uint32_t)grid), dim3(threads), 0, st, children,
                               (uint32_t *)nodes, n_children, levels);
            HIP_TRY(hipGetLastError());
            n_children >>= levels;
        }
        children = (const uint32_t *)nodes + n_children * 8;  // that level lives at nodes[n .. 2n)
    }
    return 0;
}

// dw: digest words -- 8 = Blake3_256, 6 = Blake3_192 (48-byte merge inputs; the slots' last two words are zeros)
static int run_merkle(wf_ctx *ctx, hipStream_t st, const void *leaves, uint64_t n_leaves, void *nodes, uint32_t dw = 8) {
    return dw == 6 ? run_merkle_dw<6>(ctx, st, leaves, n_leaves, nodes) : run_merkle_dw<8>(ctx, st, leaves, n_leaves, nodes);
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
    const size_t work_vals = seg_plan<F>(p->log2_trace_len, b.n_seg, ctx->tune.max_digit, ctx->tune.full_tiles).n_pass > 1 ? seg_vals * n_cosets : 0;
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
    // single-pass plans, f128: the coset's input factors h_c^k from a [blowup][N] table (FTAB: a tile otherwise spends one 78-instruction
    // product per row on them -- the reference's example, 512 x 2^10 x 10: 3 % of its evaluation pass); f64 keeps the per-tile form
    // (its product is 17 instructions and the same trade measured flat for its strided pass: GTAB1)
    if (F::BYTES == 16 && !ctx->tune.no_ftab && seg_plan<F>(logR, b.n_seg, ctx->tune.max_digit, ctx->tune.full_tiles).n_pass == 1) {
        rc = coset_row_factors<F>(ctx, logR, logB, off, olo, ohi, &d.fin_tab);
        if (rc) return rc;
    }
    d.base_cols = base_cols;
    d.total_base_cols = b.total_base_cols;
    d.row_width = row_width;
    d.trace_lde_elems = Nrows * row_width;
    d.pad_in_kernel = pad_in_kernel;
    d.pad_traces = pad_traces;
    bool hashed = false;  // leaves produced by the last evaluation pass itself (one segment, one trace)
    d.leaves = d_leaves;
    d.hash_epr = b.total_base_cols;  // the combined row of all traces (= base_cols for one trace)
    d.digest_words = p->digest_bytes / 4;
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
            rc = run_hash_rows<F>(ctx, st, d_lde, Nrows * row_width, Nrows, (uint32_t)row_width, base_cols, p->n_traces, d_leaves, p->digest_bytes / 4);
            if (rc) return rc;
        }
        if (d_nodes) {
            prof_mark(ctx, st, "merkle");
            rc = run_merkle(ctx, st, d_leaves, Nrows, d_nodes, p->digest_bytes / 4);
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
static int trace_commit_pipelined_impl(wf_ctx *ctx, const wf_params *p, const void *const *cols_in, void *d_stage, void *d_polys,
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
        if (ctx->tune.fail_after_segment >= 0 && (int)g == ctx->tune.fail_after_segment)  // test hook (wf_tuning)
            return fail(WF_ERR_HIP, "injected failure after %u uploaded segment(s) (WF_EXP_FAIL_AFTER_SEGMENT)", g);
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

// A failure part-way leaves copies of the CALLER's columns queued on the copy stream and kernels queued into the handle's
// buffers: both streams are drained before the error is returned, so that neither the columns (pinned memory is read by
// the DMA engine after hipMemcpyAsync has returned) nor the buffers can go away under work in flight.
template <class F>
static int trace_commit_pipelined(wf_ctx *ctx, const wf_params *p, const void *const *cols_in, void *d_stage, void *d_polys,
                                  void *d_lde, void *d_leaves, void *d_nodes, hipStream_t st, void *const *polys_out) {
    const int rc = trace_commit_pipelined_impl<F>(ctx, p, cols_in, d_stage, d_polys, d_lde, d_leaves, d_nodes, st, polys_out);
    if (rc) {
        if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamSynchronize(st);
        (void)hipGetLastError();
    }
    return rc;
}

static bool pipelined_upload_ok(const wf_ctx *ctx, const wf_params *p, size_t colb) {
    if (ctx->tune.no_pipeline) return false;
    if (p->ext_degree != 1) return false;
    const uint32_t S = p->field == WF_FIELD_F64 ? SegCfg<F64>::S : SegCfg<F128>::S;
    const uint32_t n_seg = (p->n_cols * p->n_traces + S - 1) / S;
    if (n_seg < 2) return false;
    const int n_pass = p->field == WF_FIELD_F64 ? seg_plan<F64>(p->log2_trace_len, n_seg, ctx->tune.max_digit, ctx->tune.full_tiles).n_pass
                                                : seg_plan<F128>(p->log2_trace_len, n_seg, ctx->tune.max_digit, ctx->tune.full_tiles).n_pass;
    if (n_pass < 2) return false;
    return colb >= ctx->tune.pipeline_min_bytes;  // (below ~1 MiB per column the events cost more than they hide; tests lower it)
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

extern "C" {

int wf_plan_digits(uint32_t field, uint32_t log2_n, uint32_t n_segments, uint32_t digits_out[4]) {
    if (!digits_out) return fail(WF_ERR_ARG, "digits_out is null");
    if (field != WF_FIELD_F64 && field != WF_FIELD_F128) return fail(WF_ERR_FIELD, "unknown field id %u", field);
    if (log2_n < 1 || log2_n > 40) return fail(WF_ERR_TRACE_LENGTH, "transform size out of range");
    const Plan p = field == WF_FIELD_F64 ? seg_plan<F64>(log2_n, n_segments) : seg_plan<F128>(log2_n, n_segments);
    for (int i = 0; i < 4; i++) digits_out[i] = i < p.n_pass ? p.dig[i] : 0;
    return p.n_pass;
}


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
    int rc = run_merkle(ctx, st, d_leaves, n_leaves, d_nodes, ctx->digest_bytes / 4);
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
    const size_t colb = wf_column_bytes(p), ldeb = wf_lde_bytes(p);
    const size_t n_dig = (size_t)1 << (p->log2_trace_len + p->log2_blowup), digb = n_dig * 32;  // device side: 32-byte slots
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
    // host arrays of digests are Vec<ByteDigest<N>>: digest_bytes apart (24-byte digests are packed on the device, one
    // array after the other through the same staging buffer)
    if (leaves_out) {
        if ((rc = path_digests_to_host(ctx, st, ctx->io[3].p, leaves_out, n_dig, p->digest_bytes))) return rc;
        if (p->digest_bytes != 32) HIP_TRY(hipStreamSynchronize(st));
    }
    if (nodes_out && (rc = path_digests_to_host(ctx, st, ctx->io[4].p, nodes_out, n_dig, p->digest_bytes))) return rc;
    if (root_out) HIP_TRY(hipMemcpyAsync(root_out, (char *)ctx->io[4].p + 32, 32, hipMemcpyDeviceToHost, st));  // (zero-padded: Digest::as_bytes)
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
        rc = run_hash_rows<F64>(ctx, st, ctx->io[2].p, 0, n_rows, (uint32_t)row_elems, (uint32_t)row_elems, 1, ctx->io[3].p, ctx->digest_bytes / 4);
    else
        rc = run_hash_rows<F128>(ctx, st, ctx->io[2].p, 0, n_rows, (uint32_t)row_elems, (uint32_t)row_elems, 1, ctx->io[3].p, ctx->digest_bytes / 4);
    if (rc) return rc;
    if ((rc = path_digests_to_host(ctx, st, ctx->io[3].p, digests_out, n_rows, ctx->digest_bytes))) return rc;
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
    if ((rc = path_digests_from_host(ctx, st, leaves, ctx->io[3].p, n_leaves, ctx->digest_bytes))) return rc;
    rc = run_merkle(ctx, st, ctx->io[3].p, n_leaves, ctx->io[4].p, ctx->digest_bytes / 4);
    if (rc) return rc;
    if ((rc = path_digests_to_host(ctx, st, ctx->io[4].p, nodes_out, n_leaves, ctx->digest_bytes))) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

}  // extern "C"
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
    rc = comm_interleave(st, recv, d_leaves, R / W, W, per);
    if (rc) return rc;
    const uint64_t n_local = N / W;
    if (n_local >= 2) {
        rc = run_merkle(ctx, st, d_leaves, n_local, d_nodes, p->digest_bytes / 4);  // local layout: d_nodes[1] = this rank's sub-root
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
        rc = run_merkle(ctx, st, top + (size_t)W * 32, W, top, p->digest_bytes / 4);
        if (rc) return rc;
    }
    prof_mark(ctx, st, "between_calls");
    return 0;
}

extern "C" {

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

// ------------------------------------------------------------------------------------------------- entry points for the other units
int path_trace_commit(wf_ctx *ctx, const wf_params *p, const void *d_trace, void *d_polys, void *d_lde, void *d_leaves,
                      void *d_nodes, hipStream_t st, hipEvent_t input_read) {
    return p->field == WF_FIELD_F64 ? trace_commit_dev<F64>(ctx, p, d_trace, d_polys, d_lde, d_leaves, d_nodes, st, input_read)
                                    : trace_commit_dev<F128>(ctx, p, d_trace, d_polys, d_lde, d_leaves, d_nodes, st, input_read);
}

int path_trace_commit_pipelined(wf_ctx *ctx, const wf_params *p, const void *const *cols_in, void *d_stage, void *d_polys,
                                void *d_lde, void *d_leaves, void *d_nodes, hipStream_t st, void *const *polys_out) {
    return p->field == WF_FIELD_F64
               ? trace_commit_pipelined<F64>(ctx, p, cols_in, d_stage, d_polys, d_lde, d_leaves, d_nodes, st, polys_out)
               : trace_commit_pipelined<F128>(ctx, p, cols_in, d_stage, d_polys, d_lde, d_leaves, d_nodes, st, polys_out);
}

bool path_pipelined_upload_ok(const wf_ctx *ctx, const wf_params *p, size_t colb) { return pipelined_upload_ok(ctx, p, colb); }

int path_constraint_commit(wf_ctx *ctx, const wf_params *p, const void *d_polys, void *d_lde, void *d_leaves, void *d_nodes,
                           hipStream_t st, bool dense_rows) {
    return p->field == WF_FIELD_F64 ? constraint_commit_dev<F64>(ctx, p, d_polys, d_lde, d_leaves, d_nodes, st, dense_rows)
                                    : constraint_commit_dev<F128>(ctx, p, d_polys, d_lde, d_leaves, d_nodes, st, dense_rows);
}

bool path_dense_column_ok(const wf_params *p) {
    return p->field == WF_FIELD_F64 ? dense_column_ok<F64>(p) : dense_column_ok<F128>(p);
}
bool path_dense_matrix_ok(const wf_params *p) { return dense_matrix_ok(p); }

int path_hash_rows(wf_ctx *ctx, hipStream_t st, uint32_t field, const void *lde, uint64_t trace_elems, uint64_t n_rows,
                   uint32_t row_width, uint32_t epr, uint32_t n_traces, void *leaves, uint32_t digest_bytes) {
    return field == WF_FIELD_F64 ? run_hash_rows<F64>(ctx, st, lde, trace_elems, n_rows, row_width, epr, n_traces, leaves, digest_bytes / 4)
                                 : run_hash_rows<F128>(ctx, st, lde, trace_elems, n_rows, row_width, epr, n_traces, leaves, digest_bytes / 4);
}

int path_merkle(wf_ctx *ctx, hipStream_t st, const void *leaves, uint64_t n_leaves, void *nodes, uint32_t digest_bytes) {
    return run_merkle(ctx, st, leaves, n_leaves, nodes, digest_bytes / 4);
}

int path_digests_to_host(wf_ctx *ctx, hipStream_t st, const void *d_slots, void *host, size_t n, uint32_t digest_bytes) {
    if (digest_bytes == 32) {
        HIP_TRY(hipMemcpyAsync(host, d_slots, n * 32, hipMemcpyDeviceToHost, st));
        return 0;
    }
    int rc = ensure(ctx, ctx->pack_tmp, n * 24);
    if (rc) return rc;
    hipLaunchKernelGGL(k_digests_pack24, dim3((uint32_t)((3 * n + 255) / 256)), dim3(256), 0, st, (const uint2 *)d_slots, (uint2 *)ctx->pack_tmp.p, (uint64_t)n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host, ctx->pack_tmp.p, n * 24, hipMemcpyDeviceToHost, st));
    return 0;
}

int path_digests_from_host(wf_ctx *ctx, hipStream_t st, const void *host, void *d_slots, size_t n, uint32_t digest_bytes) {
    if (digest_bytes == 32) {
        HIP_TRY(hipMemcpyAsync(d_slots, host, n * 32, hipMemcpyHostToDevice, st));
        return 0;
    }
    int rc = ensure(ctx, ctx->pack_tmp, n * 24);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->pack_tmp.p, host, n * 24, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_digests_unpack24, dim3((uint32_t)((4 * n + 255) / 256)), dim3(256), 0, st, (const uint2 *)ctx->pack_tmp.p, (uint2 *)d_slots, (uint64_t)n);
    HIP_TRY(hipGetLastError());
    return 0;
}

int path_trace_commit_sharded(wf_comm *c, const wf_params *p, const void *d_trace, void *d_polys, void *d_lde_shard,
                              void *d_leaves, void *d_nodes, void *d_top, hipStream_t st) {
    return p->field == WF_FIELD_F64 ? trace_commit_sharded<F64>(c, p, d_trace, d_polys, d_lde_shard, d_leaves, d_nodes, d_top, st)
                                    : trace_commit_sharded<F128>(c, p, d_trace, d_polys, d_lde_shard, d_leaves, d_nodes, d_top, st);
}

#include "constraint_poly.hpp"
