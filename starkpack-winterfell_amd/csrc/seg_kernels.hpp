// "Segment" NTT kernels: the commitment path's transforms on a lane-interleaved layout.
//
// A segment is S = 64 / sizeof(element) base columns (8 for f64, 4 for f128) stored row by row:
//     seg[g][n][l]   g = segment, n = row (coefficient / evaluation index), l = lane
// i.e. one 64-byte row per index -- the device-side analogue of the reference's `Segment<B, 8>`
// (prover/src/matrix/segments.rs:35-41), chosen because the S lanes of a row share every twiddle, coset factor and
// index computation, and because every global access is a whole 64-byte row.  Base columns of all traces are packed
// back to back into segments (base column index B = trace * base_cols + column), only the last segment is padded.
//
// Transform structure (same as kernels.hpp): N = 2^L split into digit passes; a strided pass handles the rows
// (o, d, i), d = 0..D-1, of one inner position i; the last pass handles D contiguous rows and scatters to natural
// order.  Inside LDS (x[D][S]): radix-16 rounds with the 16 values of one lane in registers (f64), then radix-4 /
// radix-2 rounds on two lanes per thread (16-byte LDS accesses for f64); f128 uses the radix-4 / radix-2 rounds only.
// Work-groups have D/2 threads (one work item of the widest round each).  DESIGN.md §4 describes the kernels, docs/EXPERIMENTS.md the
// variants that were measured and dropped (their code is in the history, not here).  A -DWF_EXPERIMENTS build
// (scripts/exp_variants.sh) adds the time-attribution switches WF_EXP_SKIP_LOAD / _SKIP_NTT / _SKIP_STORE, the
// wrong-output store orders WF_EXP_LOCAL_STORE and the phase stamps WF_EXP_STAMPS; the product build has none of them.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "field.hpp"
#include "kernels.hpp"

#ifndef WF_EXPERIMENTS  // the diagnostic switches exist in experiment builds only
#undef WF_EXP_SKIP_LOAD
#undef WF_EXP_SKIP_NTT
#undef WF_EXP_SKIP_STORE
#undef WF_EXP_LOCAL_STORE
#undef WF_EXP_STAMPS
#endif

namespace wf {

template <class F>
struct SegCfg {
    static constexpr uint32_t S = 64 / F::BYTES;  // lanes per segment row
    static constexpr uint32_t HP = S / 2;         // lane pairs per row
    // radix-16 rounds keep 16 values (+ 8 constants) in registers: 2 VGPRs per value for f64; for f128 (4 per value)
    // that costs half the occupancy and runs slower than radix-4 rounds
    static constexpr bool RADIX16 = F::BYTES == 8;
    // f128: radix-8 rounds with the 8 values of one lane in registers (32 VGPRs, what a radix-4 round on lane pairs holds too):
    // three bits per LDS round trip instead of two -- a 2^9-row tile takes 3 rounds instead of 5, a 2^10-row tile 4 instead of 5.
    // The multiplication count is the same (no root of unity of this field is a shift: 0.5 products per element and bit either way).
    // Measured (profiles/r04_tail_pack.txt): cfg 5 1.21 -> 1.15 ms (every pass -5 %), f128 2^20 x 10 -1.6 %, the shapes with
    // 2^10-row tiles (rounds 8, 8, 8, 2 against five radix-4 / radix-2 rounds) within 1 % either way; 8, 8, 4, 4 no better.
    static constexpr bool RADIX8 = F::BYTES == 16;
    static constexpr uint32_t LOAD_BATCH = F::BYTES == 8 ? 8 : 4;  // 16- / 32-byte loads kept in flight per thread
};

template <class T>
struct alignas(16) Pair {
    T a, b;
};

// 16- or 32-byte store of a lane pair
template <class T>
__device__ __forceinline__ void store_pair(T *dst, const Pair<T> &v) {
    *reinterpret_cast<Pair<T> *>(dst) = v;
}

// Zero padding of an LDE row (segments.rs:65-72) from its first padding element `pz` (column `cols` of a row of
// `row_width` elements, row_width even): one single element if `cols` is odd, then aligned pairs.
template <class F>
__device__ __forceinline__ void store_row_padding(typename F::T *pz, uint32_t cols, uint32_t row_width) {
    uint32_t z = cols;
    if (z & 1) {
        *pz++ = F::zero();
        z++;
    }
    const Pair<typename F::T> zz{F::zero(), F::zero()};
    for (; z < row_width; z += 2, pz += 2) store_pair(pz, zz);
}

// A lane pair rebuilt from the 16-byte vector registers it was loaded into.  Tile data that waits in registers between
// its (early) global load and its use is kept in native vector registers: arrays of Pair<> in the same role stay in
// scratch memory in some instantiations (the PACKED ones), which doubles the tile's memory traffic.
template <class F>
__device__ __forceinline__ Pair<typename F::T> pair_from(const uint4 &qa, const uint4 &qb);
template <>
__device__ __forceinline__ Pair<uint64_t> pair_from<F64>(const uint4 &qa, const uint4 &) {
    Pair<uint64_t> v;
    v.a = ((uint64_t)qa.y << 32) | qa.x;
    v.b = ((uint64_t)qa.w << 32) | qa.z;
    return v;
}
template <>
__device__ __forceinline__ Pair<U128> pair_from<F128>(const uint4 &qa, const uint4 &qb) {
    Pair<U128> v;
    v.a = U128{((uint64_t)qa.y << 32) | qa.x, ((uint64_t)qa.w << 32) | qa.z};
    v.b = U128{((uint64_t)qb.y << 32) | qb.x, ((uint64_t)qb.w << 32) | qb.z};
    return v;
}

enum : int { SEG_OUT_SEG = 0, SEG_OUT_ROWS = 1 };

// Tile-size specialisation of the pass kernels: LOGD != 0 instantiates a kernel for tiles of exactly 2^LOGD rows run by
// 2^(LOGD-1) threads (what the launcher uses for that tile size).  Digit, round and work-item loops then have constant
// trip counts, LDS offsets become instruction immediates and the index arithmetic of every round folds away; LOGD == 0
// is the generic kernel (any tile size).  Same arithmetic, same outputs.
template <int LOGD>
__device__ __forceinline__ uint32_t tile_threads() {
    return LOGD ? (1u << (LOGD ? LOGD - 1 : 0)) : blockDim.x;
}
// (second argument: waves per SIMD the register allocation must allow -- two 512-thread work-groups per CU = 4)
#define WF_TILE_BOUNDS(LOGD, GENERIC) __launch_bounds__((LOGD) ? (1 << ((LOGD) ? (LOGD) - 1 : 0)) : (GENERIC), (LOGD) ? 4 : 1)

// Work-groups are dispatched to the 8 XCDs round-robin (blockIdx % 8), each XCD with its own L2.  This maps blockIdx
// to a logical index such that 8 consecutive logical indices run back to back on ONE XCD: the kernels make those the
// work-groups that write neighbouring 64-byte pieces of the same 512 bytes (adjacent inner positions of a strided pass,
// the cosets of one row group in the last pass), so that the pieces meet in that XCD's L2 and leave as whole lines.
__device__ __forceinline__ uint64_t xcd_group_index(uint64_t b, uint64_t total) {
    if ((total & 63) != 0) return b;
    const uint64_t xcd = b & 7, seq = b >> 3;
    return (((seq >> 3) << 3) + xcd) * 8 + (seq & 7);
}

// First strided pass of a coset evaluation: every coset reads the SAME polynomial tile.  With the coset as the outermost
// index the eight readers of a tile are an eighth of the launch apart and each fetches it again (cfg 3, PMC: 17.2 GB fetched
// for 2.1 GB of polynomials, and the pass 9.5 ms against 8.4 for the second strided pass that reads as much from its own
// buffer).  Here the logical index is (tile / 64, coset, tile % 64): an XCD takes logical indices in blocks of 8 out of every 64
// (xcd_group_index), so a block of 8 neighbouring tiles stays on one XCD and its cosets follow each other there -- the first
// reader brings the tile into that L2, the others find it.  lc = log2(cosets); tiles per coset is a multiple of 64.
__device__ __forceinline__ uint64_t coset_inner_split(uint64_t bid, uint32_t lc, uint32_t &c) {
    const uint64_t hi = bid >> 6;
    c = (uint32_t)hi & ((1u << lc) - 1);
    return ((hi >> lc) << 6) | (bid & 63);
}

// Tile indices are decoded with shifts where a factor is a power of two (I and O always are) and with 32-bit divisions
// otherwise (grids are below 2^31): a 64-bit division by a run-time value costs ~80 VALU instructions on this target,
// and the prologue of a tile is paid by every thread.
__device__ __forceinline__ uint32_t ilog2_pow2(uint64_t v) { return 63u - (uint32_t)__builtin_clzll(v); }
__host__ __device__ constexpr uint32_t ilog2_const(uint32_t v) { return v <= 1 ? 0 : 1 + ilog2_const(v >> 1); }

template <class F>
struct SegArgs {
    typedef typename F::T T;
    const T *src;
    T *dst;
    uint32_t logN, logD;
    uint64_t I, O;          // inner / outer row counts of the [O][D][I] view (last pass: I = 1)
    uint32_t n_seg;         // segments per coset
    uint32_t n_cosets;
    uint32_t src_shared;    // source indexed by segment only (first pass of an evaluation reads the polys)
    const T *fin_tab;       // k_seg_last, single-pass evaluation (FTAB): [blowup][N] input factors h_c^k (global, built once per context); nullptr: per tile
    const T *fout_tab;      // k_seg_strided_wide<.., GTAB> (later passes) and k_seg_strided (first pass of an evaluation, GTAB1): [I][D] output factors
                            // w_N^(k i N / (D I)) of this pass (global, built once per context); nullptr: rebuilt per tile
    uint32_t coset_inner;   // strided pass, src_shared: 1 + log2(n_cosets) -- the cosets of 64 neighbouring tiles follow each other on one
                            // XCD (coset_inner_split), so that the source tile they share is fetched once and found in that L2; 0: coset outermost
    Pow2L<F> tw;            // powers of the N-th root of this transform
    const T *digit_tw;      // [D] powers of the D-th root
    uint32_t pre_on;        // multiply input row n by h_c^n (first pass of a coset evaluation)
    Pow2L<F> pre;
    uint64_t pre_lo_stride, pre_hi_stride;
    uint32_t scale_on;      // multiply by `scale` (strided pass: folded into the twiddle table; last pass: at store)
    T scale;
    // last pass
    uint32_t n_prev;
    uint32_t prev_log[3];
    // SEG_OUT_ROWS
    uint32_t base_cols;        // base columns per trace
    uint32_t total_base_cols;  // over all traces
    // columns the last pass stores per trace / in total: base_cols / total_base_cols, or row_width for a single trace whose
    // segments cover the padded row exactly -- the zero padding lanes (segments.rs:65-72) are then written with the data
    // (whole 64-byte pieces) and the caller does not have to clear the matrix first
    uint32_t store_cols, total_store_cols;
    uint32_t pad_traces;  // the lane that stores a trace's last column also zeroes the rest of that (padded) row
    uint32_t tail_pad;  // zero elements the last segment writes after its own S lanes (0 or S: f128 rows of 8 elements)
    uint32_t seg_stride;       // strided pass: segments between consecutive cosets of src / dst (= n_seg unless a segment range runs)
    uint32_t coset0;           // first coset computed by this call (coset sharding across GPUs); 0 otherwise
    uint32_t rows_per_k;       // cosets held by the output matrix: row = k * rows_per_k + local coset (= blowup unless sharded)
    // coset-packed lanes (narrow matrices: total_base_cols <= S/2): a row holds 2^cpr_log cosets x 2^lg_log lanes;
    // lane L = (local coset L >> lg_log, column L & (2^lg_log - 1)); n_cosets then counts coset GROUPS
    uint32_t cpr_log, lg_log;
    uint64_t row_width;
    uint64_t trace_lde_elems;
    // fused leaf hashing (last pass, SEG_OUT_ROWS, one segment, one trace, not PACKED): the tile's rows are complete
    // rows of the matrix and at most 64 bytes long, so leaf k * rows_per_k + coset is hashed here, from LDS, and the LDE
    // is not read back by k_hash_rows (RowMatrix::commit_to_rows, row_matrix.rs:183-203)
    uint32_t *leaves;          // nullptr: no fused hashing
    uint32_t hash_epr;         // elements of a row that are hashed (elements_per_row)
    uint32_t digest_words;     // 8 (Blake3_256) or 6 (Blake3_192: the last two words of a leaf's 32-byte slot are zeros)
    uint32_t *chunk_cvs;       // k_seg_last_hash<.., CHUNKED>: [LDE row][n_chunks][8] chunk chaining values (rows > 1024 bytes)
    uint32_t n_chunks;         //   ceil(n_seg / 16): a BLAKE3 chunk is 16 blocks = 16 segments of a row
    uint32_t *tile_counters;   // k_seg_last_hash: 8 ticket + 8 exit counters, one per XCD, zero between launches (self-resetting)
    const T *src_tail;         // k_seg_last_hash_tp: the coset-packed tail segment's work buffer [coset pair][N][S]
    uint32_t tail_cols;        //   base columns in the tail segment (<= S / 2)
#ifdef WF_EXP_STAMPS
    unsigned long long *stamps;  // diagnostic build only: per work-group phase cycle sums of k_seg_last_hash (8 words each)
#endif
};

// radix-8 fields: the number of remaining bits that is NOT taken as a radix-8 round (seg_lds_ntt); the diagnostic build can put
// the round sequence of the first radix-8 version back (8, 8, 8, 2 for 2^10 rows) for a same-box comparison
#if defined(WF_EXPERIMENTS) && defined(WF_EXP_R8_NO_SPLIT)
constexpr uint32_t R8_SPLIT = 0;
#else
constexpr uint32_t R8_SPLIT = 4;
#endif
// radix-8 rounds of a 2^logD-row tile: all of them when 3 divides logD or leaves 2 bits (one radix-4 round follows); when it
// leaves 1 bit, one round fewer and two radix-4 rounds (8, 8, 4, 4 for 2^10 rows, not 8, 8, 8, 2: see seg_lds_ntt)
__host__ __device__ constexpr uint32_t radix8_rounds(uint32_t logD) {
    return (R8_SPLIT == 4 && logD >= 4 && logD % 3 == 1) ? logD / 3 - 1 : logD / 3;
}

// LDS position -> output index of seg_lds_ntt: radix-16 digits while >= 4 bits remain, then radix-4, then radix-2
template <class F>
__device__ __forceinline__ uint32_t seg_digit_reverse(uint32_t pos, uint32_t logD) {
    uint32_t k = 0, cur = logD, sh = 0;
    while (SegCfg<F>::RADIX16 && cur >= 4) {
        k |= ((pos >> (cur - 4)) & 15u) << sh;
        sh += 4;
        cur -= 4;
    }
    if (SegCfg<F>::RADIX8) {
        for (uint32_t r = radix8_rounds(logD); r > 0; r--) {
            k |= ((pos >> (cur - 3)) & 7u) << sh;
            sh += 3;
            cur -= 3;
        }
    }
    while (cur >= 2) {
        k |= ((pos >> (cur - 2)) & 3u) << sh;
        sh += 2;
        cur -= 2;
    }
    if (cur == 1) k |= (pos & 1u) << sh;
    return k;
}

__host__ __device__ constexpr uint32_t bitrev4(uint32_t v) {
    return ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3);
}

// 16-point DFT in registers (decimation in frequency, constants w[j] = w_16^j): v[bitrev4(k)] = sum_a x_a w_16^(a k).
// DIR = +1 / -1: the transform is known to be the forward / inverse one over Goldilocks, where w_16 = 2^12
// (f64/mod.rs:248-264: w_64 = 8), so w_16^j for j = 1, 2 is a left shift, for j = 6, 7 a negated right shift
// (2^72 = -2^-24, 2^84 = -2^-12), and the inverse constants are w_16^-j = -2^(96 - 12 j); the negation is folded into the
// preceding subtraction.  j = 3, 4, 5 (2^36, 2^48, 2^60) are word-aligned shifts (F64::mul_pow2, K > 32): 12 VALU against
// 17 for the table product, so a transform of known direction needs no radix-16 constants at all.
// DIR = 0: generic (constants from the table only).
template <class F, int DIR, int J>
__device__ __forceinline__ typename F::T radix16_twiddle(typename F::T u, typename F::T t, const typename F::T (&w)[8]) {
    if constexpr (DIR == 0 || F::FIELD_ID != 1) {
        return F::mul(F::sub(u, t), w[J]);
    } else if constexpr (DIR > 0) {
        if constexpr (J == 1) return F::template mul_pow2<12>(F::sub(u, t));
        if constexpr (J == 2) return F::template mul_pow2<24>(F::sub(u, t));
        if constexpr (J == 3) return F::template mul_pow2<36>(F::sub(u, t));
        if constexpr (J == 4) return F::template mul_pow2<48>(F::sub(u, t));
        if constexpr (J == 5) return F::template mul_pow2<60>(F::sub(u, t));
        if constexpr (J == 6) return F::template div_pow2<24>(F::sub(t, u));
        if constexpr (J == 7) return F::template div_pow2<12>(F::sub(t, u));
    } else {  // w_16^-J = -2^(96 - 12 J)
        if constexpr (J == 1) return F::template div_pow2<12>(F::sub(u, t));
        if constexpr (J == 2) return F::template div_pow2<24>(F::sub(u, t));
        if constexpr (J == 3) return F::template mul_pow2<60>(F::sub(t, u));
        if constexpr (J == 4) return F::template mul_pow2<48>(F::sub(t, u));
        if constexpr (J == 5) return F::template mul_pow2<36>(F::sub(t, u));
        if constexpr (J == 6) return F::template mul_pow2<24>(F::sub(t, u));
        if constexpr (J == 7) return F::template mul_pow2<12>(F::sub(t, u));
    }
    return F::zero();
}

// NEG: the difference comes out negated (t - u): how the last stage hands -X_k to a shift twiddle whose sign is not free
template <class F, int DIR, int S, int Q, int I, bool NEG = false>
__device__ __forceinline__ void radix16_bfly(typename F::T (&v)[16], const typename F::T (&w)[8]) {
    constexpr int half = 8 >> S;
    const typename F::T u = v[Q + I], t = v[Q + I + half];
    v[Q + I] = F::add(u, t);
    if constexpr (I == 0)
        v[Q + I + half] = NEG ? F::sub(t, u) : F::sub(u, t);
    else
        v[Q + I + half] = radix16_twiddle<F, DIR, (I << S)>(u, t, w);
}

template <class F, int DIR, int S, int Q, uint32_t NEGPOS = 0>
__device__ __forceinline__ void radix16_group(typename F::T (&v)[16], const typename F::T (&w)[8]) {
    constexpr int half = 8 >> S;
    radix16_bfly<F, DIR, S, Q, 0, S == 3 && ((NEGPOS >> (Q + 1)) & 1u)>(v, w);
    if constexpr (half > 1) radix16_bfly<F, DIR, S, Q, 1>(v, w);
    if constexpr (half > 2) {
        radix16_bfly<F, DIR, S, Q, 2>(v, w);
        radix16_bfly<F, DIR, S, Q, 3>(v, w);
    }
    if constexpr (half > 4) {
        radix16_bfly<F, DIR, S, Q, 4>(v, w);
        radix16_bfly<F, DIR, S, Q, 5>(v, w);
        radix16_bfly<F, DIR, S, Q, 6>(v, w);
        radix16_bfly<F, DIR, S, Q, 7>(v, w);
    }
}

// NEGPOS: bit q set = the value at v[q] (q odd: the differences of the last stage) comes out negated
template <class F, int DIR, uint32_t NEGPOS = 0>
__device__ __forceinline__ void radix16(typename F::T (&v)[16], const typename F::T (&w)[8]) {
    static_assert((NEGPOS & 0x5555u) == 0, "only the last stage's differences can be negated for free");
    radix16_group<F, DIR, 0, 0>(v, w);
    radix16_group<F, DIR, 1, 0>(v, w);
    radix16_group<F, DIR, 1, 8>(v, w);
    radix16_group<F, DIR, 2, 0>(v, w);
    radix16_group<F, DIR, 2, 4>(v, w);
    radix16_group<F, DIR, 2, 8>(v, w);
    radix16_group<F, DIR, 2, 12>(v, w);
    radix16_group<F, DIR, 3, 0, NEGPOS>(v, w);
    radix16_group<F, DIR, 3, 2, NEGPOS>(v, w);
    radix16_group<F, DIR, 3, 4, NEGPOS>(v, w);
    radix16_group<F, DIR, 3, 6, NEGPOS>(v, w);
    radix16_group<F, DIR, 3, 8, NEGPOS>(v, w);
    radix16_group<F, DIR, 3, 10, NEGPOS>(v, w);
    radix16_group<F, DIR, 3, 12, NEGPOS>(v, w);
    radix16_group<F, DIR, 3, 14, NEGPOS>(v, w);
}

// ---- twiddles that are powers of two (Goldilocks, transform of known direction) ------------------------------------
// After a radix-16 round with 2^6 (2^5) bits left, output k of the item at inner position jp is multiplied by
// w_64^(jp k) = 2^(3 jp k) (w_32^(jp k) = 2^(6 jp k)): w_64 = 8 (f64/mod.rs:248-264).  With jp uniform in a wave -- the
// rounds below deal their work items out that way for these tile sizes -- the exponents are compile-time constants of
// one of four (two) code paths chosen by a scalar branch: 15 shifts of 9-12 VALU instead of 15 general products of 17
// plus their table reads, and nothing at all for jp = 0.  STEP = 3 jp (6 jp); the inverse transform multiplies by
// 2^(192 - STEP k).  Exponents in (96, 128] are -2^K with K <= 32, whose shift ends in an addition: for those (all at
// k >= 8, the differences of the block's last stage: checked by radix16's static_assert) the block delivers -X_k.
template <int DIR, int STEP, int K>
__host__ __device__ constexpr int shift_tw_exp() {
    return DIR > 0 ? (STEP * K) % 192 : (192 - (STEP * K) % 192) % 192;
}
template <int DIR, int STEP, int K = 1>
__host__ __device__ constexpr uint32_t shift_tw_negpos() {
    if constexpr (K == 16) {
        return 0;
    } else {
        constexpr int E = shift_tw_exp<DIR, STEP, K>();
        return ((E > 96 && E <= 128) ? (1u << bitrev4(K)) : 0u) | shift_tw_negpos<DIR, STEP, K + 1>();
    }
}
template <class F, int DIR, int STEP, int K = 1>
__device__ __forceinline__ void shift_tw_apply(typename F::T (&v)[16]) {
    if constexpr (K < 16) {
        constexpr int E = shift_tw_exp<DIR, STEP, K>();
        constexpr int q = bitrev4(K);
        asm volatile("" : "+v"(v[q]));  // one shift after the other (measured the same as interleaved, one register fewer)
        if constexpr (E > 96 && E <= 128)
            v[q] = F::template mul_pow2<E - 96>(v[q]);  // v[q] = -X_k already
        else
            v[q] = F::template mul_pow2_192<E>(v[q]);
        asm volatile("" : "+v"(v[q]));
        shift_tw_apply<F, DIR, STEP, K + 1>(v);
    }
}
template <class F, int DIR, int STEP>
__device__ __forceinline__ void radix16_shift_tw(typename F::T (&v)[16], const typename F::T (&w)[8]) {
    radix16<F, DIR, shift_tw_negpos<DIR, STEP>()>(v, w);
    shift_tw_apply<F, DIR, STEP>(v);
}

// In-place transform of x[D][S] in LDS; twd[e] = w_D^e.  Rounds: radix-16 (one lane per work item, 16 values in
// registers) while >= 4 bits remain, then radix-4 / radix-2 (two lanes per item).  Natural order in,
// seg_digit_reverse order out.
// `first` != nullptr: the first radix-16 round takes its 16 inputs from there instead of LDS -- the pass kernels load
// them straight from global memory (work item wk = threadIdx.x: lane wk % S of rows a * D/16 + wk / S, a = 0..15;
// needs logD >= 4, RADIX16 and blockDim >= D/2), which saves the tile's trip through LDS before the first round.
// `nthr` = blockDim.x, passed in so that the tile-size-specialised kernels (LOGD != 0, below) make it a constant.
// One radix-16 round of seg_lds_ntt at `cur` remaining bits (no trailing barrier): work item wk = lane wk % S of the
// 16 rows base + a * 2^(cur-4), a = 0..15.  w16[j] = w_16^j.
// LANES: elements per tile row -- the S lanes of a segment row, or TI * S when a tile holds TI adjacent inner positions
// side by side (k_seg_strided_wide); the lanes of a row share every twiddle either way.
template <class F, int DIR, bool UNI = false, uint32_t LANES = SegCfg<F>::S>
__device__ __forceinline__ void seg_round16(typename F::T *x, const typename F::T *twd, const typename F::T (&w16)[8],
                                            uint32_t logD, uint32_t cur, uint32_t nthr, const typename F::T *first,
                                            uint32_t tid, bool use_first) {  // tid: threadIdx.x; first: read only if use_first
    typedef typename F::T T;
    constexpr uint32_t S = LANES;
    constexpr uint32_t s_shift = ilog2_const(S);
    const uint32_t D = 1u << logD;
    const uint32_t mlog = cur - 4, m = 1u << mlog;
    const uint32_t nwork = (D >> 4) * S;
    const uint32_t tshift = logD - cur;
    const uint32_t st = m * S;
    // Shift-twiddle rounds (radix16_shift_tw above): 2^10-row tiles at cur = 6 and 2^9-row tiles at cur = 5 -- the
    // sizes with at least 64 work items per inner position jp.  A wave (8 lanes x 8 items; nthr is a multiple of 64)
    // takes eight blocks p of ONE jp instead of two blocks of all four: its LDS accesses then sit 4 KiB apart (4-way
    // on the reads, 2-way on the writes -- under the VALU issue that bounds these kernels), its twiddle exponents are
    // compile-time constants.
    // UNI: only from seg_lds_fixed (logD and cur compile-time constants there; as a run-time branch of the generic round
    // loop the extra paths cost every kernel registers).
    constexpr bool SHIFT_TW = UNI && F::FIELD_ID == 1 && DIR != 0 && LANES == 8;
    const bool uni = SHIFT_TW && ((logD == 10 && cur == 6) || (logD == 9 && cur == 5));
    for (uint32_t wk = tid; wk < nwork; wk += nthr) {
        const uint32_t l = wk & (S - 1), u = wk >> s_shift;
        uint32_t jp = u & (m - 1), p = u >> mlog;
        if (uni) {
            const uint32_t wv = u >> 3;
            jp = wv & (m - 1);
            p = ((wv >> mlog) << 3) | (u & 7u);
        }
        const uint32_t row0 = (p << cur) + jp;
        uint32_t off[16];  // element offsets of the item's 16 rows row0 + a * m
        {
            const uint32_t base = row0 * S + l;
#pragma unroll
            for (int a = 0; a < 16; a++) off[a] = base + a * st;
        }
        T v[16];
        if (use_first) {  // uniform
#pragma unroll
            for (int a = 0; a < 16; a++) v[a] = first[a];
        } else {
#pragma unroll
            for (int a = 0; a < 16; a++) v[a] = x[off[a]];
        }
        bool done = false;
        if constexpr (SHIFT_TW) {
            if (uni) {  // uniform
                const uint32_t ju = __builtin_amdgcn_readfirstlane(jp) * (cur == 6 ? 1u : 2u);  // STEP / 3
                if (ju == 0)
                    radix16<F, DIR>(v, w16);
                else if (ju == 1)
                    radix16_shift_tw<F, DIR, 3>(v, w16);
                else if (ju == 2)
                    radix16_shift_tw<F, DIR, 6>(v, w16);
                else
                    radix16_shift_tw<F, DIR, 9>(v, w16);
                done = true;
            }
        }
        if (!done) {
            radix16<F, DIR>(v, w16);
            if (jp != 0) {
                const uint32_t e = jp << tshift;
#pragma unroll
                for (int k = 1; k < 16; k++) v[bitrev4(k)] = F::mul(v[bitrev4(k)], twd[e * k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 16; k++) x[off[k]] = v[bitrev4(k)];
    }
}

// One radix-8 round (fields without shift twiddles: f128) at `cur` remaining bits: work item wk = lane wk % LANES of the 8 rows
// row0 + a * 2^(cur-3); 8-point DIF in registers (w8[j] = w_8^j, j = 1..3), then the inter-round twiddles w_(2^cur)^(jp k).
__host__ __device__ constexpr uint32_t bitrev3(uint32_t v) { return ((v & 1) << 2) | (v & 2) | ((v & 4) >> 2); }
template <class F, uint32_t LANES = SegCfg<F>::S>
__device__ __forceinline__ void seg_round8(typename F::T *x, const typename F::T *twd, const typename F::T (&w8)[4], uint32_t logD,
                                           uint32_t cur, uint32_t nthr, uint32_t tid) {
    typedef typename F::T T;
    constexpr uint32_t l_shift = ilog2_const(LANES);
    const uint32_t D = 1u << logD;
    const uint32_t mlog = cur - 3, m = 1u << mlog;
    const uint32_t nwork = (D >> 3) * LANES;
    const uint32_t tshift = logD - cur;
    const uint32_t st = m * LANES;
    for (uint32_t wk = tid; wk < nwork; wk += nthr) {
        const uint32_t l = wk & (LANES - 1), u = wk >> l_shift;
        const uint32_t jp = u & (m - 1), p = u >> mlog;
        const uint32_t base = ((p << cur) + jp) * LANES + l;
        T v[8];
#pragma unroll
        for (int a = 0; a < 8; a++) v[a] = x[base + a * st];
        // stage 1: (v[i], v[i + 4]) -> (sum, difference * w_8^i)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const T a0 = v[i], a1 = v[i + 4];
            v[i] = F::add(a0, a1);
            v[i + 4] = i == 0 ? F::sub(a0, a1) : F::mul(F::sub(a0, a1), w8[i]);
        }
        // stage 2: within each half, (v[q + i], v[q + i + 2]) with w_4^i = w_8^(2 i)
#pragma unroll
        for (int q = 0; q < 8; q += 4) {
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const T a0 = v[q + i], a1 = v[q + i + 2];
                v[q + i] = F::add(a0, a1);
                v[q + i + 2] = i == 0 ? F::sub(a0, a1) : F::mul(F::sub(a0, a1), w8[2]);
            }
        }
        // stage 3
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            const T a0 = v[q], a1 = v[q + 1];
            v[q] = F::add(a0, a1);
            v[q + 1] = F::sub(a0, a1);
        }
        // v[bitrev3(k)] = X_k; inter-round twiddles w^(jp k)
        if (jp != 0) {
            const uint32_t e = jp << tshift;
#pragma unroll
            for (int k = 1; k < 8; k++) v[bitrev3(k)] = F::mul(v[bitrev3(k)], twd[e * k]);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) x[base + k * st] = v[bitrev3(k)];
    }
}

// (u - t) * w_4 of a transform of known direction over Goldilocks: w_4 = 2^48 forward, -2^48 inverse (2^96 = -1)
template <class F, int DIR>
__device__ __forceinline__ typename F::T mul_w4(typename F::T u, typename F::T t, typename F::T w4) {
    if constexpr (DIR == 0 || F::FIELD_ID != 1)
        return F::mul(F::sub(u, t), w4);
    else if constexpr (DIR > 0)
        return F::template mul_pow2<48>(F::sub(u, t));
    else
        return F::template mul_pow2<48>(F::sub(t, u));
}

// One radix-4 round (two lanes per work item, 16-byte LDS accesses for f64); w4 = w_4.
template <class F, int DIR = 0, bool UNI = false, uint32_t LANES = SegCfg<F>::S>
__device__ __forceinline__ void seg_round4(typename F::T *x, const typename F::T *twd, typename F::T w4, uint32_t logD,
                                           uint32_t cur, uint32_t nthr, uint32_t tid) {
    typedef typename F::T T;
    typedef Pair<T> P2;
    constexpr uint32_t S = LANES, HP = LANES / 2;
    constexpr uint32_t hp_shift = ilog2_const(HP);
    const uint32_t D = 1u << logD;
    const uint32_t mlog = cur - 2, m = 1u << mlog;
    const uint32_t nwork = (D >> 2) * HP;
    const uint32_t tshift = logD - cur;
    const uint32_t st = m * S;
    // cur == 3 (tiles of 2^7 and 2^11 rows): the twiddles are w_8^(jp k) = 2^(24 jp k), jp = 0, 1.  As in seg_round16 a wave
    // (4 lane pairs x 16 items) takes sixteen blocks of one jp, and the products become shifts (or nothing).
    // (HP == 8: the 16-lane rows of the wide strided kernel, 8 lane pairs x 8 items per wave -- also in the generic round loop, which is the
    // only form those tiles run)
    constexpr bool SHIFT_TW = F::FIELD_ID == 1 && DIR != 0 && ((UNI && HP == 4) || HP == 8);
    constexpr uint32_t IPW_LOG = HP == 8 ? 3 : 4;  // log2 of the work items of one wave: 64 / HP
    const bool uni = SHIFT_TW && cur == 3 && logD >= 7;
    for (uint32_t wk = tid; wk < nwork; wk += nthr) {
        const uint32_t lp = wk & (HP - 1), u = wk >> hp_shift;
        uint32_t jp = u & (m - 1), p = u >> mlog;
        if (uni) {
            const uint32_t wv = u >> IPW_LOG;
            jp = wv & 1u;
            p = ((wv >> 1) << IPW_LOG) | (u & ((1u << IPW_LOG) - 1));
        }
        const uint32_t row0 = (p << cur) + jp;
        // element offsets of rows row0 + k * m
        const uint32_t o0 = row0 * S + 2 * lp, o1 = o0 + st, o2 = o0 + 2 * st, o3 = o0 + 3 * st;
        P2 x0 = *reinterpret_cast<P2 *>(x + o0);
        P2 x1 = *reinterpret_cast<P2 *>(x + o1);
        P2 x2 = *reinterpret_cast<P2 *>(x + o2);
        P2 x3 = *reinterpret_cast<P2 *>(x + o3);
        P2 y0, y1, y2, y3;
        {
            T a = F::add(x0.a, x2.a), b = F::sub(x0.a, x2.a), c = F::add(x1.a, x3.a);
            T d = mul_w4<F, DIR>(x1.a, x3.a, w4);
            y0.a = F::add(a, c);
            y2.a = F::sub(a, c);
            y1.a = F::add(b, d);
            y3.a = F::sub(b, d);
        }
        {
            T a = F::add(x0.b, x2.b), b = F::sub(x0.b, x2.b), c = F::add(x1.b, x3.b);
            T d = mul_w4<F, DIR>(x1.b, x3.b, w4);
            y0.b = F::add(a, c);
            y2.b = F::sub(a, c);
            y1.b = F::add(b, d);
            y3.b = F::sub(b, d);
        }
        bool done = false;
        if constexpr (SHIFT_TW) {
            if (uni) {  // uniform
                if (__builtin_amdgcn_readfirstlane(jp) != 0) {
                    constexpr int E1 = shift_tw_exp<DIR, 24, 1>(), E2 = shift_tw_exp<DIR, 24, 2>(), E3 = shift_tw_exp<DIR, 24, 3>();
                    y1.a = F::template mul_pow2_192<E1>(y1.a);
                    y1.b = F::template mul_pow2_192<E1>(y1.b);
                    y2.a = F::template mul_pow2_192<E2>(y2.a);
                    y2.b = F::template mul_pow2_192<E2>(y2.b);
                    y3.a = F::template mul_pow2_192<E3>(y3.a);
                    y3.b = F::template mul_pow2_192<E3>(y3.b);
                }
                done = true;
            }
        }
        if (!done && jp != 0) {
            const uint32_t e = jp << tshift;
            const T t1 = twd[e], t2 = twd[2 * e], t3 = twd[3 * e];
            y1.a = F::mul(y1.a, t1);
            y1.b = F::mul(y1.b, t1);
            y2.a = F::mul(y2.a, t2);
            y2.b = F::mul(y2.b, t2);
            y3.a = F::mul(y3.a, t3);
            y3.b = F::mul(y3.b, t3);
        }
        *reinterpret_cast<P2 *>(x + o0) = y0;
        *reinterpret_cast<P2 *>(x + o1) = y1;
        *reinterpret_cast<P2 *>(x + o2) = y2;
        *reinterpret_cast<P2 *>(x + o3) = y3;
    }
}

template <class F, uint32_t LANES = SegCfg<F>::S>
__device__ __forceinline__ void seg_round2(typename F::T *x, uint32_t logD, uint32_t nthr, uint32_t tid) {
    typedef typename F::T T;
    typedef Pair<T> P2;
    constexpr uint32_t S = LANES, HP = LANES / 2;
    constexpr uint32_t hp_shift = ilog2_const(HP);
    const uint32_t nwork = (1u << (logD - 1)) * HP;
    for (uint32_t wk = tid; wk < nwork; wk += nthr) {
        const uint32_t lp = wk & (HP - 1), u = wk >> hp_shift;
        const uint32_t base = (u << 1) * S + 2 * lp, base1 = base + S;
        P2 x0 = *reinterpret_cast<P2 *>(x + base);
        P2 x1 = *reinterpret_cast<P2 *>(x + base1);
        P2 y0, y1;
        y0.a = F::add(x0.a, x1.a);
        y1.a = F::sub(x0.a, x1.a);
        y0.b = F::add(x0.b, x1.b);
        y1.b = F::sub(x0.b, x1.b);
        *reinterpret_cast<P2 *>(x + base) = y0;
        *reinterpret_cast<P2 *>(x + base1) = y1;
    }
}

__device__ __forceinline__ uint32_t opaque_tid() {
    // everything derived from the thread index is recomputed inside every iteration of the tile loop from this opaque
    // copy: otherwise the compiler hoists dozens of loop-invariant addresses out of the loop and holds them in VGPRs
    uint32_t t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// The rounds of a tile of 2^LOGD rows with every size a compile-time constant: what the Goldilocks tiles with power-of-two
// inter-round twiddles run (2^10 and 2^9 rows: after the second radix-16 round; 2^7 and 2^11: in the radix-4 round).
template <class F, int DIR, int LOGD, int CUR, uint32_t LANES = SegCfg<F>::S>
__device__ __forceinline__ void seg_lds_fixed(typename F::T *x, const typename F::T *twd, const typename F::T (&w16)[8],
                                              typename F::T w4, uint32_t nthr, const typename F::T *first, bool use_first,
                                              bool opaque) {
    // opaque (persistent kernels): the thread index is re-read per round, so that what a round derives from it is not hoisted
    // out of the kernel's tile loop and held in registers across it
    const uint32_t tid = opaque ? opaque_tid() : threadIdx.x;
    if constexpr (SegCfg<F>::RADIX16 && CUR >= 4) {
        seg_round16<F, DIR, true, LANES>(x, twd, w16, LOGD, CUR, nthr, first, tid, use_first && CUR == LOGD);
        __syncthreads();
        seg_lds_fixed<F, DIR, LOGD, CUR - 4, LANES>(x, twd, w16, w4, nthr, first, false, opaque);
    } else if constexpr (CUR >= 2) {
        seg_round4<F, DIR, true, LANES>(x, twd, w4, LOGD, CUR, nthr, tid);
        __syncthreads();
        seg_lds_fixed<F, DIR, LOGD, CUR - 2, LANES>(x, twd, w16, w4, nthr, first, false, opaque);
    } else if constexpr (CUR == 1) {
        seg_round2<F, LANES>(x, LOGD, nthr, tid);
        __syncthreads();
    }
}

// FIXMASK: bit L set = tiles of 2^L rows run the fixed-size round sequence (L = 7, 9, 10, 11: the sizes with shift twiddles
// between rounds); each kernel names the sizes it has the registers for.  `use_first`: whether `first` is to be used (a
// flag beside an always-valid pointer keeps the caller's register array out of scratch memory).
constexpr uint32_t FIX10 = 1u << 10, FIX7 = 1u << 7, FIX9 = 1u << 9, FIX11 = 1u << 11;
constexpr uint32_t FIX_BIG = FIX10 | FIX11;   // the fused last pass on tiles of 2^10 / 2^11 rows
constexpr uint32_t FIX_LAST = FIX10 | FIX9;   // the one-work-group-per-tile last pass
template <class F, int DIR = 0, uint32_t FIXMASK = FIX10, uint32_t LANES = SegCfg<F>::S>
__device__ __forceinline__ void seg_lds_ntt(typename F::T *x, const typename F::T *twd, uint32_t logD, uint32_t nthr,
                                            const typename F::T *first = nullptr, bool use_first = false,
                                            bool opaque = false) {
    typedef typename F::T T;
    const uint32_t D = 1u << logD;
    constexpr uint32_t fixmask = FIXMASK;
    if constexpr (fixmask != 0 && F::FIELD_ID == 1 && DIR != 0) {  // a transform of known direction reads no radix-16 / w_4 constants
        T w16[8];
#pragma unroll
        for (int j = 0; j < 8; j++) w16[j] = F::zero();
#define WF_FIXED_SIZE(L)                                                                                        \
    if constexpr ((fixmask & (1u << (L))) != 0) {                                                               \
        if (logD == (L)) { /* uniform */                                                                        \
            seg_lds_fixed<F, DIR, (L), (L), LANES>(x, twd, w16, F::zero(), nthr, first, use_first, opaque);       \
            return;                                                                                             \
        }                                                                                                       \
    }
        WF_FIXED_SIZE(10)
        WF_FIXED_SIZE(7)
        WF_FIXED_SIZE(9)
        WF_FIXED_SIZE(11)
        WF_FIXED_SIZE(8)
#undef WF_FIXED_SIZE
    }
    uint32_t cur = logD;
    if (SegCfg<F>::RADIX16 && logD >= 4) {
        T w16[8];
#pragma unroll
        for (int j = 0; j < 8; j++) w16[j] = twd[j * (D >> 4)];
        while (cur >= 4) {
            seg_round16<F, DIR, false, LANES>(x, twd, w16, logD, cur, nthr, first, threadIdx.x, use_first && cur == logD);
            cur -= 4;
            __syncthreads();
        }
    }
    if (SegCfg<F>::RADIX8 && logD >= 3) {
        T w8[4];
#pragma unroll
        for (int j = 0; j < 4; j++) w8[j] = twd[j * (D >> 3)];
        // radix-8 rounds, except that 4 remaining bits go as 4 + 4 points (not 8 + 2): a round's general products are its inner
        // constants plus the twiddles in front of the next round, and the LAST round has none of the latter -- 2^10 rows as
        // 8, 8, 4, 4 take 4.25 products per element against 4.5 for 8, 8, 8, 2 (2^7: 2.75 against 3), with as many LDS round trips
        for (uint32_t r = radix8_rounds(logD); r > 0; r--) {
            seg_round8<F, LANES>(x, twd, w8, logD, cur, nthr, threadIdx.x);
            cur -= 3;
            __syncthreads();
        }
    }
    T w4 = F::one();
    if (logD >= 2) w4 = twd[D >> 2];
    while (cur > 0) {
        if (cur >= 2) {
            seg_round4<F, DIR, false, LANES>(x, twd, w4, logD, cur, nthr, threadIdx.x);
            cur -= 2;
        } else {
            seg_round2<F, LANES>(x, logD, nthr, threadIdx.x);
            cur = 0;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Strided pass.  grid.x = n_cosets * n_seg * O * I ; work-group = (coset c, segment g, outer o, inner i)
// EVAL = 0: interpolation (inverse transform), 1: coset evaluation (forward transform); like k_seg_last<F, OUT> the two
// uses appear under different kernel names in profiles, and the direction selects the shift twiddles of radix16.
template <class F, int EVAL, bool PACKED = false, int LOGD = 0>
__global__ void WF_TILE_BOUNDS(LOGD, 1024) k_seg_strided(SegArgs<F> a) {
    typedef typename F::T T;
    if (LOGD) a.logD = LOGD;
    const uint32_t NT = tile_threads<LOGD>();
    typedef Pair<T> P2;
    constexpr uint32_t S = SegCfg<F>::S, HP = SegCfg<F>::HP;
    constexpr uint32_t hp_shift = HP == 4 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint32_t D = 1u << a.logD;
    T *x = reinterpret_cast<T *>(smem_raw);
    T *twd = x + (size_t)D * S;
    // f128 (ONE_TABLE): the three tables of the pass -- input factors while the tile is filled, digit twiddles during the
    // transform, output factors for the stores -- take turns in ONE region of D entries (two more barriers per tile; the
    // twiddles and the output factors wait in registers): a 2^10-row tile is 80 KiB instead of 96, two work-groups per CU,
    // a 2^9-row tile 40 KiB instead of 48, four instead of three.
    constexpr bool ONE_TABLE = F::BYTES == 16;
    T *aux = ONE_TABLE ? twd : twd + D;  // coset factors of the input rows, later the inter-pass twiddles of the output rows

    uint64_t bid = xcd_group_index(blockIdx.x, gridDim.x);  // neighbouring i on one XCD
    const uint32_t logI = ilog2_pow2(a.I), logO = ilog2_pow2(a.O);
    uint32_t c_in = 0;
    if (a.coset_inner) bid = coset_inner_split(bid, a.coset_inner - 1, c_in);
    const uint64_t i = bid & (a.I - 1);
    const uint64_t o = (bid >> logI) & (a.O - 1);
    const uint32_t rest = (uint32_t)(bid >> (logI + logO));
    const uint32_t c = a.coset_inner ? c_in : rest / a.n_seg, g = a.coset_inner ? rest : rest - c * a.n_seg;
    const uint64_t seg_elems = ((uint64_t)1 << a.logN) * S;
    // (seg_stride = segments of the whole matrix; n_seg < seg_stride when the launch covers a range of them)
    const T *src = a.src + (a.src_shared ? (uint64_t)g : (uint64_t)c * a.seg_stride + g) * seg_elems;
    T *dst = a.dst + ((uint64_t)c * a.seg_stride + g) * seg_elems;

    Pow2L<F> pre = a.pre;
    // PACKED: the two lanes of this thread may belong to different cosets; their tables and h_c^i factors
    Pow2L<F> pre_a = a.pre, pre_b = a.pre;
    const uint32_t lane_a = 2 * (threadIdx.x & (HP - 1)), lane_b = lane_a + 1;  // blockDim is a multiple of HP
    // a packed lane is live if its column exists and its local coset is one of the 2^cpr_log of this row
    const bool act_a = PACKED && (lane_a & ((1u << a.lg_log) - 1)) < a.total_base_cols && (lane_a >> a.lg_log) < (1u << a.cpr_log);
    const bool act_b = PACKED && (lane_b & ((1u << a.lg_log) - 1)) < a.total_base_cols && (lane_b >> a.lg_log) < (1u << a.cpr_log);
    if (PACKED && a.pre_on) {
        const uint32_t ca = a.coset0 + (c << a.cpr_log) + (act_a ? (lane_a >> a.lg_log) : 0);
        const uint32_t cb = a.coset0 + (c << a.cpr_log) + (act_b ? (lane_b >> a.lg_log) : 0);
        pre_a.lo += (uint64_t)ca * a.pre_lo_stride;
        pre_a.hi += (uint64_t)ca * a.pre_hi_stride;
        pre_b.lo += (uint64_t)cb * a.pre_lo_stride;
        pre_b.hi += (uint64_t)cb * a.pre_hi_stride;
    } else if (a.pre_on) {
        pre.lo += (uint64_t)(a.coset0 + c) * a.pre_lo_stride;
        pre.hi += (uint64_t)(a.coset0 + c) * a.pre_hi_stride;
    }
    const uint32_t nitems = D * HP;
    const uint64_t row0 = ((o << a.logD) << logI) + i;  // row index of d = 0; rows of this group are I apart
    // The first LOAD_BATCH row pieces of every thread (the whole tile up to D = 2^10) are requested before the tables
    // are built, so that their latency runs under the table arithmetic.  The rows are I apart: every load is its own
    // 64-byte gather, the load phase lives on memory-level parallelism.
    const uint32_t step = NT;
    const bool from_regs = !(PACKED && a.pre_on);
    // ---- prologue: every global read of the tile's setup is issued before the first one is used -- the operands of the
    // output factors start * (w_N^(i * N/(D*I)))^k (start = h_c^i for an evaluation, 1/n for an interpolation), the digit
    // twiddles, the operands of the input factors h_c^(d*I), then the tile's first LOAD_BATCH row pieces per thread (all
    // of them up to D = 2^10).  Branch-free with clamped indices (blockDim >= D/2: at most two table entries per
    // thread); the output factors wait in registers until the input factors in `aux` have been consumed.
    const uint32_t tw_shift = a.logN - a.logD - logI;
    const bool scale_in = !PACKED && a.pre_on;
    const Pow2L<F> pin = scale_in ? pre : a.tw;  // without input scaling the reads go to the root table and are dropped
    // GTAB1 (round 5; first pass of a coset evaluation, SegArgs::fout_tab set): the output factors w_N^(k i) come from a table in
    // global memory ([i][k], built once per context, D x I entries) and the coset's h_c^i moves from the output factors into the
    // INPUT factors -- h_c^(d I) * h_c^i = h_c^(d I + i), the coset factor of the row itself, at the same two table reads and one
    // product per entry -- so the per-tile arithmetic of the output table (two reads, two products per entry) is gone
    const bool gtab = !PACKED && a.fout_tab != nullptr;  // (uniform)
    uint32_t kq[2];
    T fo_a[2], fo_b[2], tw_q[2], fi_a[2], fi_b[2];
#pragma unroll
    for (uint32_t q = 0; q < 2; q++) {
        const uint32_t k = threadIdx.x + q * NT;
        kq[q] = k < D ? k : D - 1;
        if (gtab)
            fo_a[q] = fo_b[q] = a.fout_tab[(i << a.logD) + kq[q]];
        else
            a.tw.fetch(((uint64_t)kq[q] * i) << tw_shift, fo_a[q], fo_b[q]);
        tw_q[q] = a.digit_tw[kq[q]];
        pin.fetch(((uint64_t)kq[q] << logI) + (gtab ? i : 0), fi_a[q], fi_b[q]);
    }
    // `direct`: the 16 inputs of this thread's first radix-16 work item come straight from global memory into registers
    // (lane threadIdx % S of rows a * D/16 + threadIdx / S), the tile makes no trip through LDS before the first round
    // (not in the PACKED instantiations: there the 16-value array ends up in scratch memory)
    const bool direct = !PACKED && SegCfg<F>::RADIX16 && a.logD >= 4 && from_regs;
    const uint32_t m16 = D >> 4, l16 = threadIdx.x & (S - 1), j16 = threadIdx.x / S;
    const bool has16 = threadIdx.x < m16 * S;
    T vr[16];
    uint4 r0, r1, r2, r3, r4, r5, r6, r7;  // the non-direct route: lane pairs in vector registers (see pair_from)
    r0 = r1 = r2 = r3 = r4 = r5 = r6 = r7 = make_uint4(0, 0, 0, 0);
    if (direct) {
        if (has16) {
            // sixteen rows m16 * I apart: one address, then a running 64-bit add per row
            const T *pr = src + (row0 + ((uint64_t)j16 << logI)) * S + l16;
            const uint64_t rstep = ((uint64_t)m16 << logI) * S;
#pragma unroll
            for (uint32_t q = 0; q < 16; q++) {
#ifdef WF_EXP_SKIP_LOAD
                vr[q] = src[l16];
#else
                vr[q] = *pr;
#endif
                pr += rstep;
            }
        }
    } else if (from_regs) {
#ifdef WF_EXP_SKIP_LOAD
#define WF_ITEM_PTR(WK) reinterpret_cast<const uint4 *>(src + 2 * ((WK) & (HP - 1)))
#else
#define WF_ITEM_PTR(WK) \
    reinterpret_cast<const uint4 *>(src + (row0 + ((uint64_t)((WK) >> hp_shift) << logI)) * S + 2 * ((WK) & (HP - 1)))
#endif
#define WF_LOAD_ITEM(U, RA, RB)                            \
    do {                                                   \
        const uint32_t wk_ = threadIdx.x + (U) * step;     \
        if (wk_ < nitems) {                                \
            const uint4 *p_ = WF_ITEM_PTR(wk_);            \
            RA = p_[0];                                    \
            if (F::BYTES == 16) RB = p_[1];                \
        }                                                  \
    } while (0)
        if (F::BYTES == 8) {
            WF_LOAD_ITEM(0, r0, r0);
            WF_LOAD_ITEM(1, r1, r1);
            WF_LOAD_ITEM(2, r2, r2);
            WF_LOAD_ITEM(3, r3, r3);
            WF_LOAD_ITEM(4, r4, r4);
            WF_LOAD_ITEM(5, r5, r5);
            WF_LOAD_ITEM(6, r6, r6);
            WF_LOAD_ITEM(7, r7, r7);
        } else {
            WF_LOAD_ITEM(0, r0, r1);
            WF_LOAD_ITEM(1, r2, r3);
            WF_LOAD_ITEM(2, r4, r5);
            WF_LOAD_ITEM(3, r6, r7);
        }
#undef WF_LOAD_ITEM
#undef WF_ITEM_PTR
    }
    T fo[2];
    {
        T start = a.scale_on ? a.scale : F::one();
        if (scale_in && !gtab) start = pre.get(i);
        const bool trivial = !a.scale_on && !scale_in;
#pragma unroll
        for (uint32_t q = 0; q < 2; q++) {
            T f = fo_a[q];
            if (!gtab) {
                f = F::mul(fo_a[q], fo_b[q]);
                if (!trivial) f = F::mul(f, start);
            }
            fo[q] = f;
            if (threadIdx.x + q * NT < D) {
                if (!ONE_TABLE) twd[kq[q]] = tw_q[q];
                if (scale_in) aux[kq[q]] = F::mul(fi_a[q], fi_b[q]);  // h_c^(d*I); h_c^i goes into the output factors
            }
        }
    }
    // (ONE_TABLE: twiddles and output factors wait in vector registers for their turn in the table region)
    const uint4 twk0 = park(tw_q[0]), twk1 = park(tw_q[1]), fok0 = park(fo[0]), fok1 = park(fo[1]);
    __syncthreads();

    if (direct) {
        if (scale_in && has16) {
#pragma unroll
            for (uint32_t q = 0; q < 16; q++) vr[q] = F::mul(vr[q], aux[q * m16 + j16]);
        }
    } else if (PACKED && a.pre_on) {  // replicate the polynomial's lanes into every coset group of the row
        for (uint32_t wk = threadIdx.x; wk < nitems; wk += NT) {
            const uint32_t lp = wk & (HP - 1), d = wk >> hp_shift;
            P2 v;
            const uint32_t lgm = (1u << a.lg_log) - 1, cola = lane_a & lgm, colb = lane_b & lgm;
            const T *srow = src + (row0 + ((uint64_t)d << logI)) * S;
            // the two lanes of a pair share their coset (and so the factor) unless a coset is a single lane wide
            const T fa = pre_a.get((uint64_t)d << logI);
            const T fb = a.lg_log ? fa : pre_b.get((uint64_t)d << logI);
            v.a = act_a ? F::mul(srow[cola], fa) : F::zero();
            v.b = act_b ? F::mul(srow[colb], fb) : F::zero();
            *reinterpret_cast<P2 *>(x + d * S + 2 * lp) = v;
        }
    } else {
#define WF_FILL_ITEM(U, RA, RB)                                          \
    do {                                                                 \
        const uint32_t wk_ = threadIdx.x + (U) * step;                   \
        if (wk_ < nitems) {                                              \
            const uint32_t lp_ = wk_ & (HP - 1), d_ = wk_ >> hp_shift;   \
            P2 v_ = pair_from<F>(RA, RB);                                \
            if (a.pre_on) {                                              \
                const T f_ = aux[d_];                                    \
                v_.a = F::mul(v_.a, f_);                                 \
                v_.b = F::mul(v_.b, f_);                                 \
            }                                                            \
            *reinterpret_cast<P2 *>(x + d_ * S + 2 * lp_) = v_;          \
        }                                                                \
    } while (0)
        if (F::BYTES == 8) {
            WF_FILL_ITEM(0, r0, r0);
            WF_FILL_ITEM(1, r1, r1);
            WF_FILL_ITEM(2, r2, r2);
            WF_FILL_ITEM(3, r3, r3);
            WF_FILL_ITEM(4, r4, r4);
            WF_FILL_ITEM(5, r5, r5);
            WF_FILL_ITEM(6, r6, r6);
            WF_FILL_ITEM(7, r7, r7);
        } else {
            WF_FILL_ITEM(0, r0, r1);
            WF_FILL_ITEM(1, r2, r3);
            WF_FILL_ITEM(2, r4, r5);
            WF_FILL_ITEM(3, r6, r7);
        }
#undef WF_FILL_ITEM
    }
    if (!direct || scale_in) __syncthreads();  // LDS tile written / input factors consumed (uniform condition)
    if (ONE_TABLE) {  // the region now takes the digit twiddles
        if (threadIdx.x < D) twd[kq[0]] = unpark<F>(twk0);
        if (threadIdx.x + NT < D) twd[kq[1]] = unpark<F>(twk1);
        __syncthreads();
    } else {  // `aux` now takes the output factors (visible after the transform's barriers)
#pragma unroll
        for (uint32_t q = 0; q < 2; q++) {
            const uint32_t k = threadIdx.x + q * NT;
            if (k < D) aux[k] = fo[q];
        }
    }
#ifndef WF_EXP_SKIP_NTT  // tuning experiment: memory phases only (scripts/exp_variants.sh)
    // (the generic kernel runs the generic round loop: with the 16 direct values in registers it has no room for more)
    seg_lds_ntt<F, EVAL ? 1 : -1, (LOGD ? (1u << LOGD) : 0u)>(x, twd, a.logD, NT, direct ? vr : nullptr, direct);
#else
    if (direct && has16) {
#pragma unroll
        for (uint32_t q = 0; q < 16; q++) x[(q * m16 + j16) * S + l16] = vr[q];
    }
    __syncthreads();
#endif
    if (ONE_TABLE) {  // (the transform ended with a barrier: the twiddles are done with) the region takes the output factors
        if (threadIdx.x < D) aux[threadIdx.x] = unpark<F>(fok0);
        if (threadIdx.x + NT < D) aux[threadIdx.x + NT] = unpark<F>(fok1);
        __syncthreads();
    }

    T start_a = F::one(), start_b = F::one();
    if (PACKED && a.pre_on) {  // h_c^i of each lane's coset (the table above carries only the twiddle)
        start_a = pre_a.get(i);
        start_b = pre_b.get(i);
    }
    // this thread's row positions are pos0 + j * pstride (see k_seg_last): rev(pos) = rev(pos0) | rev(j * pstride)
    const uint32_t pstride = NT >> hp_shift, pos0 = threadIdx.x >> hp_shift;
    if (pos0 >= D) return;
    const uint32_t k0 = seg_digit_reverse<F>(pos0, a.logD);
    T *dst_lane = dst + row0 * S + lane_a;
    const uint32_t k_shift = logI + (S == 8 ? 3 : 2);  // rows k of the output are I apart: element offset k * I * S
    for (uint32_t pj = 0; pj < D; pj += pstride) {
        const uint32_t k = k0 | seg_digit_reverse<F>(pj, a.logD);
        P2 v = *reinterpret_cast<P2 *>(x + (pos0 + pj) * S + lane_a);
        const T f = aux[k];
        v.a = F::mul(v.a, f);
        v.b = F::mul(v.b, f);
        if (PACKED && a.pre_on) {
            v.a = F::mul(v.a, start_a);
            v.b = F::mul(v.b, start_b);
        }
#ifdef WF_EXP_SKIP_STORE
        if (*reinterpret_cast<const uint32_t *>(&v.a) == a.logN + 77777u)
#endif
        store_pair(dst_lane + ((uint64_t)k << k_shift), v);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Strided pass on WIDE tiles (f64, transforms of three and more passes: 2^22 rows and up).  The small digits of those plans
// (2^7 / 2^8 rows) make k_seg_strided's tiles 8-16 KiB of isolated 64-byte gathers by one or two waves, and the pass is bound
// by memory, not by the vector ALU (cfg 3: the pass takes the same 9.3 ms with its transform compiled out, 6.8 ms with its
// memory accesses compiled out; profiles/r04_attribution.txt).  Here one work-group takes TI ADJACENT inner positions
// i0 .. i0 + TI - 1 of (coset, segment, outer): a tile row is TI x 64 contiguous bytes -- whole 128-byte lines from TI = 2 --
// and the TI x S lanes of a row share the digit transform's twiddles (seg_lds_ntt with LANES = TI * S: the tile is a
// (2^7 rows) x (16 lanes) transform); only the output factor start_i * w_N^(k i N / (D I)) depends on the inner position:
// a table of TI x D entries, two per thread as in k_seg_strided.  Measured on cfg 3 (same box, interleaved): TI = 2 takes
// the two strided evaluation passes 9.28 -> 8.71 ms each and the commitment 35.5 -> 33.9 ms; TI = 4 and 8 (64 KiB tiles,
// two work-groups per CU) give 9.0-9.2 and 9.1 ms.
// grid.x = n_cosets * n_seg * O * (I / TI); blockDim = D * TI * S / 16 (one radix-16 work item per thread).
// GTAB: the output factors come from a table in global memory (SegArgs::fout_tab, [inner position][k]: passes after the first, whose
// factors w_N^(k i N / (D I)) depend on neither coset nor segment nor outer index -- D x I entries, 256 KiB for cfg 3's second pass, built
// once per context and served from L2) instead of being rebuilt in LDS by every tile.  What that buys is LDS: 17 KiB per work-group
// instead of 20, i.e. nine work-groups per CU instead of eight for the pass whose waves per CU are what it lives on.
// GTAB: 0 = factor tables rebuilt per tile; 1 = later passes (output factors from the global table, no input factors); 2 = FIRST pass of a coset
// evaluation with the output factors from the global table and the coset's h_c^i merged into the input factors ([TI][D] in LDS: h_c^(d I + i))
// LOGD != 0: the tile size as a compile-time constant of the SAME generic round loop (trip counts, strides and LDS offsets fold; round 5:
// cfg 3's two wide evaluation passes -1.2 % each) -- not the fixed-size round sequences of the narrow kernels, which measured 16 % slower here.
template <class F, int EVAL, int TI, int GTAB = 0, int LOGD = 0>
__global__ void __launch_bounds__(1024) k_seg_strided_wide(SegArgs<F> a) {
    if (LOGD) a.logD = LOGD;
    static_assert(SegCfg<F>::RADIX16 && TI >= 2 && (TI & (TI - 1)) == 0, "f64 tiles of 2, 4 or 8 inner positions");
    typedef typename F::T T;
    typedef Pair<T> P2;
    constexpr uint32_t S = SegCfg<F>::S, LANES = TI * S, HPW = LANES / 2;
    constexpr uint32_t LOGS = ilog2_const(S), LOGL = ilog2_const(LANES), LOGTI = ilog2_const(TI), LOGHPW = ilog2_const(HPW);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint32_t D = 1u << a.logD, NT = blockDim.x;
    T *x = reinterpret_cast<T *>(smem_raw);
    T *twd = x + (size_t)D * LANES;  // (stays in LDS also with GTAB: read from global memory inside the rounds the pass measured 3 % slower)
    T *fin = twd + D;    // h_c^(d I): factors of the input rows (first pass of a coset evaluation); GTAB == 2: [TI][D], h_c^(d I + i)
    T *fout = fin + D;   // [TI][D]: factors of the output rows (GTAB == 0)

    uint64_t bid = xcd_group_index(blockIdx.x, gridDim.x);  // 8 neighbouring tiles (8 * TI * 64 contiguous bytes per row) on one XCD
    uint32_t c_in = 0;
    if (a.coset_inner) bid = coset_inner_split(bid, a.coset_inner - 1, c_in);
    const uint32_t logI = ilog2_pow2(a.I), logO = ilog2_pow2(a.O), logIt = logI - LOGTI;
    const uint64_t i0 = (bid & ((a.I >> LOGTI) - 1)) << LOGTI;
    const uint64_t o = (bid >> logIt) & (a.O - 1);
    const uint32_t rest = (uint32_t)(bid >> (logIt + logO));
    const uint32_t c = a.coset_inner ? c_in : rest / a.n_seg, g = a.coset_inner ? rest : rest - c * a.n_seg;
    const uint64_t seg_elems = ((uint64_t)1 << a.logN) * S;
    const T *src = a.src + (a.src_shared ? (uint64_t)g : (uint64_t)c * a.seg_stride + g) * seg_elems;
    T *dst = a.dst + ((uint64_t)c * a.seg_stride + g) * seg_elems;
    Pow2L<F> pre = a.pre;
    const bool scale_in = a.pre_on != 0;
    if (scale_in) {
        pre.lo += (uint64_t)(a.coset0 + c) * a.pre_lo_stride;
        pre.hi += (uint64_t)(a.coset0 + c) * a.pre_hi_stride;
    }
    const uint64_t row0 = ((o << a.logD) << logI) + i0;  // row of d = 0, inner position i0; rows of the tile are I apart
    const uint32_t tw_shift = a.logN - a.logD - logI;
    const uint32_t tid = threadIdx.x;

    // prologue: every global read of the tile's setup is issued before the first one is used (as in k_seg_strided)
    const Pow2L<F> pst = scale_in ? pre : a.tw;  // (without input scaling these reads go to the root table and are dropped)
    T fo_a[2], fo_b[2], st_a[2], st_b[2];
    if constexpr (GTAB == 0) {
#pragma unroll
        for (uint32_t q = 0; q < 2; q++) {  // TI * D = 2 NT table entries: entry e = ti * D + k
            const uint32_t e = tid + q * NT, k = e & (D - 1), ti = e >> a.logD;
            const uint64_t i = i0 + ti;
            a.tw.fetch(((uint64_t)k * i) << tw_shift, fo_a[q], fo_b[q]);
            pst.fetch(i, st_a[q], st_b[q]);
        }
    } else if constexpr (GTAB == 2) {
#pragma unroll
        for (uint32_t q = 0; q < 2; q++) {  // input factors of row d at inner position i0 + ti: h_c^(d I + i0 + ti)
            const uint32_t e = tid + q * NT, k = e & (D - 1), ti = e >> a.logD;
            pst.fetch(((uint64_t)k << logI) + i0 + ti, st_a[q], st_b[q]);
        }
    }
    const bool has_d = tid < D;  // (NT = D * TI / 2 >= D)
    const uint32_t kd = has_d ? tid : 0;
    const T twv = a.digit_tw[kd];
    T fi_a, fi_b;
    if constexpr (GTAB == 0) pst.fetch((uint64_t)kd << logI, fi_a, fi_b);
    // the thread's radix-16 work item straight from global memory: lane tid % LANES of rows a * D/16 + tid / LANES
    const uint32_t m16 = D >> 4, l16 = tid & (LANES - 1), j16 = tid >> LOGL;
    T vr[16];
    {
        const T *pr = src + (row0 + ((uint64_t)j16 << logI)) * S + l16;
        const uint64_t rstep = ((uint64_t)m16 << logI) * S;
#pragma unroll
        for (uint32_t q = 0; q < 16; q++) {
#ifdef WF_EXP_SKIP_LOAD
            vr[q] = src[l16];
#else
            vr[q] = *pr;
#endif
            pr += rstep;
        }
    }
    if constexpr (GTAB == 2) {
#pragma unroll
        for (uint32_t q = 0; q < 2; q++) fin[tid + q * NT] = F::mul(st_a[q], st_b[q]);
    }
    if constexpr (GTAB == 0) {
#pragma unroll
        for (uint32_t q = 0; q < 2; q++) {
            T f = F::mul(fo_a[q], fo_b[q]);
            if (scale_in)
                f = F::mul(f, F::mul(st_a[q], st_b[q]));  // h_c^i rides on the output factors
            else if (a.scale_on)
                f = F::mul(f, a.scale);                    // 1/n of an interpolation
            fout[tid + q * NT] = f;
        }
    }
    if (has_d) {
        twd[kd] = twv;
        if constexpr (GTAB == 0)
            if (scale_in) fin[kd] = F::mul(fi_a, fi_b);
    }
    __syncthreads();
    if constexpr (GTAB == 0) {
        if (scale_in) {
#pragma unroll
            for (uint32_t q = 0; q < 16; q++) vr[q] = F::mul(vr[q], fin[q * m16 + j16]);
        }
    } else if constexpr (GTAB == 2) {
        const uint32_t tb = (l16 >> LOGS) << a.logD;  // this lane's inner position within the tile
#pragma unroll
        for (uint32_t q = 0; q < 16; q++) vr[q] = F::mul(vr[q], fin[tb + q * m16 + j16]);
    }
#ifndef WF_EXP_SKIP_NTT
    seg_lds_ntt<F, EVAL ? 1 : -1, 0u, LANES>(x, twd, a.logD, NT, vr, true);
#else
#pragma unroll
    for (uint32_t q = 0; q < 16; q++) x[(q * m16 + j16) * LANES + l16] = vr[q];
    __syncthreads();
#endif
    // store: this thread's lane pair of row positions pos0 + j * pstride (rev(pos) = rev(pos0) | rev(j * pstride))
    const uint32_t pstride = NT >> LOGHPW, pos0 = tid >> LOGHPW, prw = tid & (HPW - 1);
    const uint32_t k0 = seg_digit_reverse<F>(pos0, a.logD);
    const T *fo = GTAB ? a.fout_tab + ((i0 + ((2 * prw) >> LOGS)) << a.logD) : fout + (((2 * prw) >> LOGS) << a.logD);
    T *dst_lane = dst + row0 * S + 2 * prw;
    const uint32_t k_shift = logI + LOGS;  // output rows k are I apart
    if constexpr (GTAB != 0) {  // pstride = D / 8: eight rows per thread, their factors requested together
        uint32_t kk[8];
        T ff[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            kk[j] = k0 | seg_digit_reverse<F>(j * pstride, a.logD);
            ff[j] = fo[kk[j]];
        }
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            P2 v = *reinterpret_cast<P2 *>(x + (pos0 + j * pstride) * LANES + 2 * prw);
            v.a = F::mul(v.a, ff[j]);
            v.b = F::mul(v.b, ff[j]);
            store_pair(dst_lane + ((uint64_t)kk[j] << k_shift), v);
        }
        return;
    }
    for (uint32_t pj = 0; pj < D; pj += pstride) {
        const uint32_t k = k0 | seg_digit_reverse<F>(pj, a.logD);
        P2 v = *reinterpret_cast<P2 *>(x + (pos0 + pj) * LANES + 2 * prw);
        const T f = fo[k];
        v.a = F::mul(v.a, f);
        v.b = F::mul(v.b, f);
#ifdef WF_EXP_SKIP_STORE
        if (*reinterpret_cast<const uint32_t *>(&v.a) == a.logN + 77777u)
#endif
        store_pair(dst_lane + ((uint64_t)k << k_shift), v);
    }
}

// Row stores of f128 segments: a lane pair is 32 bytes and goes out as two 16-byte store instructions that each cover
// half of every pair; one element per thread makes each instruction write whole 64-byte runs (the S lanes of a row).
// `tail`: this segment also writes the S zero elements that pad the row (SegArgs::tail_pad, one trace); `pad`: the lane
// with a trace's last column also writes the zeros that pad that trace's row (SegArgs::pad_traces).
template <class F>
__device__ __forceinline__ void store_rows_by_element(const typename F::T *x, const SegArgs<F> &a, uint32_t g, uint32_t c,
                                                      uint64_t rev_o, uint32_t out_shift, uint32_t k_stride, uint32_t tid,
                                                      uint32_t n_threads, bool tail, bool pad) {
    using T = typename F::T;
    constexpr uint32_t S = SegCfg<F>::S;
    const uint32_t D = 1u << a.logD, e = tid & (S - 1), B = g * S + e;
    if (B >= a.total_store_cols) return;
    const uint32_t t = B / a.store_cols, col = B - t * a.store_cols;
    T *pa = a.dst + (uint64_t)t * a.trace_lde_elems + (uint64_t)c * a.row_width + col;
    const uint32_t n_pad = pad && col + 1 == a.store_cols ? (uint32_t)a.row_width - a.store_cols : 0;
    for (uint32_t pos = tid / S; pos < D; pos += n_threads / S) {
        const uint64_t k = rev_o + ((uint64_t)seg_digit_reverse<F>(pos, a.logD) << out_shift);
        const uint64_t off = (uint64_t)(uint32_t)k * k_stride;
        pa[off] = x[pos * S + e];
        if (tail) pa[off + S] = F::zero();
        for (uint32_t z = 1; z <= n_pad; z++) pa[off + z] = F::zero();
    }
}

// Row stores of the last evaluation pass for narrow side-by-side traces (STARKPack: several traces in the lanes of a
// segment, each with an LDE matrix of its own whose rows are padded to 8 elements).  A lane pair storing its own 16
// bytes would touch a different 64-byte row in every lane: four consecutive threads write one whole row instead -- the
// trace's columns that live in this segment and, from the segment holding its last column, the zero padding -- so a
// store instruction covers 16 complete rows.  `x` is the tile after its transform, [row position][S lanes].
template <class F>
__device__ __forceinline__ void store_rows_narrow(const typename F::T *x, const SegArgs<F> &a, uint32_t g, uint32_t c,
                                                  uint64_t rev_o, uint32_t out_shift, uint32_t k_stride, uint32_t tid,
                                                  uint32_t n_threads) {
    using T = typename F::T;
    constexpr uint32_t S = SegCfg<F>::S;
    const uint32_t D = 1u << a.logD, bc = a.store_cols;
    const uint32_t lo = g * S, hi = min(lo + S, a.total_store_cols);  // the lanes of this segment that hold columns
    // units (16 bytes = two f64 columns) per row: 4 for rows of 8 elements, 8 for rows of 16 (used for traces of 9 or 10
    // columns, where at least 3/8 of a row is padding: 8 x 10 columns at 2^20 12.5 -> 11.7 ms; with less padding, or rows
    // of 24, the lane pairs' own stores + one lane's zeros are faster)
    const uint32_t upr = (uint32_t)a.row_width >> 1;
    const uint32_t s2 = 2 * (tid % upr), psub = tid / upr, pstep = n_threads / upr;
    if (psub >= pstep) return;  // (threads beyond the last whole group of `upr`: only when upr does not divide n_threads)
    for (uint32_t t = lo / bc; t <= (hi - 1) / bc; t++) {
        const uint32_t b0 = t * bc + s2, b1 = b0 + 1, last = t * bc + bc - 1;
        const bool d0 = s2 < bc, d1 = s2 + 1 < bc;            // columns of the trace (else padding)
        const bool pad_here = last >= lo && last < hi;          // this segment writes the trace's padding
        const bool in0 = d0 ? (b0 >= lo && b0 < hi) : pad_here, in1 = d1 ? (b1 >= lo && b1 < hi) : pad_here;
        if (!in0 && !in1) continue;
        T *row = a.dst + (uint64_t)t * a.trace_lde_elems + (uint64_t)c * a.row_width + s2;
        for (uint32_t pos = psub; pos < D; pos += pstep) {
            const uint64_t k = rev_o + ((uint64_t)seg_digit_reverse<F>(pos, a.logD) << out_shift);
            const uint64_t off = (uint64_t)(uint32_t)k * k_stride;
            const T v0 = d0 && in0 ? x[pos * S + (b0 - lo)] : F::zero();
            const T v1 = d1 && in1 ? x[pos * S + (b1 - lo)] : F::zero();
            if (in0 && in1)
                store_pair(row + off, Pair<T>{v0, v1});
            else if (in0)
                row[off] = v0;
            else
                row[off + 1] = v1;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Last pass.  grid.x = n_cosets * n_seg * O ; work-group = (coset c, segment g, row block o) of D contiguous rows.
template <class F, int OUT, bool PACKED = false, int LOGD = 0>
__global__ void WF_TILE_BOUNDS(LOGD, 1024) k_seg_last(SegArgs<F> a) {
    typedef typename F::T T;
    const uint32_t logD_ = LOGD ? (uint32_t)LOGD : a.logD;  // (a local: see k_seg_last_hash)
    const uint32_t NT = tile_threads<LOGD>();
    typedef Pair<T> P2;
    constexpr uint32_t S = SegCfg<F>::S, HP = SegCfg<F>::HP;
    constexpr uint32_t hp_shift = HP == 4 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint32_t D = 1u << logD_;
    T *x = reinterpret_cast<T *>(smem_raw);
    T *twd = x + (size_t)D * S;
    T *aux = twd;  // input factors of a single-pass evaluation: same LDS as the twiddles, which are written after the fill

    // coset fastest: with SEG_OUT_ROWS the cosets of one row block write the 64-byte pieces of the same rows
    uint64_t bid = xcd_group_index(blockIdx.x, gridDim.x);
    const uint32_t logO = ilog2_pow2(a.O);
    const uint32_t b32 = (uint32_t)bid;  // grids are below 2^31
    const uint32_t q32 = b32 / a.n_cosets, c = b32 - q32 * a.n_cosets;
    const uint64_t o = q32 & (uint32_t)(a.O - 1);
    const uint32_t g = q32 >> logO;
    const uint64_t seg_elems = ((uint64_t)1 << a.logN) * S;
    const T *src = a.src + (a.src_shared ? (uint64_t)g : (uint64_t)c * a.n_seg + g) * seg_elems + (o << logD_) * S;

    // natural-order contribution of the earlier digits: o = (k1, k2, ..), k1 most significant -> k1 + N1*k2 + ..
    uint64_t rev_o = 0;
    {
        uint32_t bits = 0;
        for (uint32_t q = 0; q < a.n_prev; q++) bits += a.prev_log[q];
        uint32_t hi = bits, sh = 0;
        for (uint32_t q = 0; q < a.n_prev; q++) {
            hi -= a.prev_log[q];
            rev_o |= ((o >> hi) & (((uint64_t)1 << a.prev_log[q]) - 1)) << sh;
            sh += a.prev_log[q];
        }
    }

    Pow2L<F> pre = a.pre, pre_a = a.pre, pre_b = a.pre;
    const uint32_t lane_a = 2 * (threadIdx.x & (HP - 1)), lane_b = lane_a + 1;
    const bool act_a = PACKED && (lane_a & ((1u << a.lg_log) - 1)) < a.total_base_cols && (lane_a >> a.lg_log) < (1u << a.cpr_log);
    const bool act_b = PACKED && (lane_b & ((1u << a.lg_log) - 1)) < a.total_base_cols && (lane_b >> a.lg_log) < (1u << a.cpr_log);
    if (PACKED && a.pre_on) {
        const uint32_t ca = a.coset0 + (c << a.cpr_log) + (act_a ? (lane_a >> a.lg_log) : 0);
        const uint32_t cb = a.coset0 + (c << a.cpr_log) + (act_b ? (lane_b >> a.lg_log) : 0);
        pre_a.lo += (uint64_t)ca * a.pre_lo_stride;
        pre_a.hi += (uint64_t)ca * a.pre_hi_stride;
        pre_b.lo += (uint64_t)cb * a.pre_lo_stride;
        pre_b.hi += (uint64_t)cb * a.pre_hi_stride;
    } else if (a.pre_on) {
        pre.lo += (uint64_t)(a.coset0 + c) * a.pre_lo_stride;
        pre.hi += (uint64_t)(a.coset0 + c) * a.pre_hi_stride;
    }
    const uint32_t nitems = D * HP;
    // prologue as in k_seg_strided: the digit twiddles, the operands of the input factors (single-pass evaluation only)
    // and the tile's rows are all requested before the first of them is used
    const uint32_t step = NT;
    const bool from_regs = !(PACKED && a.pre_on);
    const bool scale_in = !PACKED && a.pre_on;
    const Pow2L<F> pin = scale_in ? pre : a.tw;
    uint32_t kq[2];
    T tw_q[2], fi_a[2], fi_b[2];
#pragma unroll
    for (uint32_t q = 0; q < 2; q++) {
        const uint32_t k = threadIdx.x + q * NT;  // blockDim >= D / 2
        kq[q] = k < D ? k : D - 1;
        tw_q[q] = a.digit_tw[kq[q]];
        if (!PACKED && a.fin_tab)  // FTAB (round 5): the coset's factors h_c^k of a single-pass evaluation from a [coset][N] table in global memory
            fi_a[q] = fi_b[q] = a.fin_tab[((uint64_t)(a.coset0 + c) << a.logN) + kq[q]];
        else
            pin.fetch(kq[q], fi_a[q], fi_b[q]);  // single-pass evaluation: row index = coefficient index
    }
    const uint4 twk0 = park(tw_q[0]), twk1 = park(tw_q[1]);  // (wait in vector registers: as an array they sit in scratch for f128)
    // (no direct first round here, unlike k_seg_strided: the tile is one contiguous 64 KiB run, which 16-byte-per-lane
    // loads staged through LDS stream faster than sixteen 8-byte loads per thread -- measured 0.61 -> 0.69 ms with it)
    // blockDim >= D/2: a tile is at most LOAD_BATCH lane pairs per thread, eight 16-byte registers either way
    uint4 r0, r1, r2, r3, r4, r5, r6, r7;
    r0 = r1 = r2 = r3 = r4 = r5 = r6 = r7 = make_uint4(0, 0, 0, 0);
#ifdef WF_EXP_SKIP_LOAD
#define WF_ITEM_PTR(WK) reinterpret_cast<const uint4 *>(a.src + 2 * (uint64_t)((WK) & 63))
#else
#define WF_ITEM_PTR(WK) reinterpret_cast<const uint4 *>(src + 2 * (uint64_t)(WK))
#endif
#define WF_LOAD_ITEM(U, RA, RB)                            \
    do {                                                   \
        const uint32_t wk_ = threadIdx.x + (U) * step;     \
        if (wk_ < nitems) {                                \
            const uint4 *p_ = WF_ITEM_PTR(wk_);            \
            RA = p_[0];                                    \
            if (F::BYTES == 16) RB = p_[1];                \
        }                                                  \
    } while (0)
    if (from_regs) {
        if (F::BYTES == 8) {
            WF_LOAD_ITEM(0, r0, r0);
            WF_LOAD_ITEM(1, r1, r1);
            WF_LOAD_ITEM(2, r2, r2);
            WF_LOAD_ITEM(3, r3, r3);
            WF_LOAD_ITEM(4, r4, r4);
            WF_LOAD_ITEM(5, r5, r5);
            WF_LOAD_ITEM(6, r6, r6);
            WF_LOAD_ITEM(7, r7, r7);
        } else {
            WF_LOAD_ITEM(0, r0, r1);
            WF_LOAD_ITEM(1, r2, r3);
            WF_LOAD_ITEM(2, r4, r5);
            WF_LOAD_ITEM(3, r6, r7);
        }
    }
#undef WF_LOAD_ITEM
#undef WF_ITEM_PTR
#pragma unroll
    for (uint32_t q = 0; q < 2; q++) {
        if (threadIdx.x + q * NT < D) {
            if (scale_in)
                aux[kq[q]] = (!PACKED && a.fin_tab) ? fi_a[q] : F::mul(fi_a[q], fi_b[q]);  // (in the twiddles' place until the tile is filled)
            else
                twd[kq[q]] = unpark<F>(q == 0 ? twk0 : twk1);
        }
    }
    __syncthreads();
    if (PACKED && a.pre_on) {  // single pass: replicate the polynomial row into the coset groups, scaled per lane
        for (uint32_t wk = threadIdx.x; wk < nitems; wk += NT) {
            P2 v;
            const uint32_t d = wk >> hp_shift;
            const uint32_t lgm = (1u << a.lg_log) - 1, cola = lane_a & lgm, colb = lane_b & lgm;
            const T *srow = src + (uint64_t)d * S;
            const T fa = pre_a.get(d);
            const T fb = a.lg_log ? fa : pre_b.get(d);  // lanes of a pair share their coset unless it is one lane wide
            v.a = act_a ? F::mul(srow[cola], fa) : F::zero();
            v.b = act_b ? F::mul(srow[colb], fb) : F::zero();
            *reinterpret_cast<P2 *>(x + 2 * wk) = v;
        }
    } else {
#define WF_FILL_ITEM(U, RA, RB)                                \
    do {                                                       \
        const uint32_t wk_ = threadIdx.x + (U) * step;         \
        if (wk_ < nitems) {                                    \
            P2 v_ = pair_from<F>(RA, RB);                      \
            if (a.pre_on) {                                    \
                const T f_ = aux[wk_ >> hp_shift];             \
                v_.a = F::mul(v_.a, f_);                       \
                v_.b = F::mul(v_.b, f_);                       \
            }                                                  \
            *reinterpret_cast<P2 *>(x + 2 * wk_) = v_;         \
        }                                                      \
    } while (0)
        if (F::BYTES == 8) {
            WF_FILL_ITEM(0, r0, r0);
            WF_FILL_ITEM(1, r1, r1);
            WF_FILL_ITEM(2, r2, r2);
            WF_FILL_ITEM(3, r3, r3);
            WF_FILL_ITEM(4, r4, r4);
            WF_FILL_ITEM(5, r5, r5);
            WF_FILL_ITEM(6, r6, r6);
            WF_FILL_ITEM(7, r7, r7);
        } else {
            WF_FILL_ITEM(0, r0, r1);
            WF_FILL_ITEM(1, r2, r3);
            WF_FILL_ITEM(2, r4, r5);
            WF_FILL_ITEM(3, r6, r7);
        }
#undef WF_FILL_ITEM
    }
    if (scale_in) {  // the input factors have been consumed: the digit twiddles take their place
        __syncthreads();
        if (threadIdx.x < D) twd[kq[0]] = unpark<F>(twk0);
        if (threadIdx.x + NT < D) twd[kq[1]] = unpark<F>(twk1);
    }
    __syncthreads();
#ifndef WF_EXP_SKIP_NTT
    seg_lds_ntt<F, OUT == SEG_OUT_ROWS ? 1 : -1, (LOGD ? (1u << LOGD) : FIX_LAST)>(x, twd, logD_, NT);
#endif

    // Store.  Work item = (row position pos, lane pair); this thread's positions are pos0 + j * pstride, j = 0, 1, ..
    // (pstride is a power of two > pos0), so its output indices are rev(pos0) | rev(j * pstride): the per-thread
    // part is computed once, the per-iteration part is wave-uniform (scalar ALU), and everything that depends only on
    // the lane (column, trace, destination base) is hoisted out of the loop.
    const uint32_t out_shift = a.logN - logD_;
    const uint32_t pstride = NT >> hp_shift, pos0 = threadIdx.x >> hp_shift;
    const uint32_t k0 = seg_digit_reverse<F>(pos0 < D ? pos0 : 0, logD_);
    const bool has_rows = pos0 < D;  // more threads than work items in tiny transforms
    if (OUT == SEG_OUT_SEG) {
        if (!has_rows) return;
        T *dst = a.dst + ((uint64_t)c * a.n_seg + g) * seg_elems + lane_a;
        for (uint32_t pj = 0; pj < D; pj += pstride) {
            const uint64_t k = rev_o + ((uint64_t)(k0 | seg_digit_reverse<F>(pj, logD_)) << out_shift);
            P2 v = *reinterpret_cast<P2 *>(x + (pos0 + pj) * S + lane_a);
            if (a.scale_on) {
                v.a = F::mul(v.a, a.scale);
                v.b = F::mul(v.b, a.scale);
            }
            store_pair(dst + k * S, v);
        }
    } else {
        // destination of each lane for k = 0: dst + trace * trace_elems + (coset) * row_width + column
        const uint32_t k_stride = a.rows_per_k * (uint32_t)a.row_width;  // elements between consecutive k (< 2^19)
        T *pa = nullptr, *pb = nullptr, *pz = nullptr, *pz2 = nullptr;  // pz*: first padding element after a trace's last column
        bool pair_store = false;
        if (PACKED) {
            const uint32_t lgm = (1u << a.lg_log) - 1;
            if (act_a) {
                const uint32_t col = lane_a & lgm, t0 = col / a.base_cols;
                pa = a.dst + (uint64_t)t0 * a.trace_lde_elems +
                     (((uint64_t)c << a.cpr_log) + (lane_a >> a.lg_log)) * a.row_width + (col - t0 * a.base_cols);
            }
            if (act_b) {
                const uint32_t col = lane_b & lgm, t1 = col / a.base_cols;
                pb = a.dst + (uint64_t)t1 * a.trace_lde_elems +
                     (((uint64_t)c << a.cpr_log) + (lane_b >> a.lg_log)) * a.row_width + (col - t1 * a.base_cols);
            }
            // both lanes in the same coset and trace, neighbouring columns at an even offset: one 16-byte store
            pair_store = pa && pb && pb == pa + 1 && ((pa - a.dst) & 1) == 0;
        } else {
            const uint32_t B = g * S + lane_a;  // global base column of lane a
            if (B < a.total_store_cols) {
                const uint32_t t0 = B / a.store_cols, c0 = B - t0 * a.store_cols;
                pa = a.dst + (uint64_t)t0 * a.trace_lde_elems + (uint64_t)c * a.row_width + c0;
                pair_store = c0 + 1 < a.store_cols && (c0 & 1) == 0;  // both lanes in one trace, 16-byte aligned
                if (a.pad_traces && c0 + 1 == a.store_cols) pz = pa + 1;
            }
            if (B + 1 < a.total_store_cols) {
                const uint32_t t1 = (B + 1) / a.store_cols, c1 = (B + 1) - t1 * a.store_cols;
                pb = a.dst + (uint64_t)t1 * a.trace_lde_elems + (uint64_t)c * a.row_width + c1;
                if (a.pad_traces && c1 + 1 == a.store_cols) pz2 = pb + 1;
            }
        }
        if (!has_rows) pa = pb = pz = pz2 = nullptr;
        if (!PACKED && a.pad_traces && (a.row_width == 8 || (F::BYTES == 8 && a.row_width == 16 && a.store_cols <= 10))) {
            store_rows_narrow<F>(x, a, g, c, rev_o, out_shift, k_stride, threadIdx.x, NT);
            pa = pb = nullptr;
        } else if (!PACKED && F::BYTES == 16) {
            store_rows_by_element<F>(x, a, g, c, rev_o, out_shift, k_stride, threadIdx.x, NT,
                                     a.tail_pad && g + 1 == a.n_seg, a.pad_traces);
            pa = pb = nullptr;
        }
        if (PACKED && a.store_cols != a.base_cols) {
            // One trace, padded rows of 8 elements: whole rows leave this pass, one thread per 16-byte unit (four per f64
            // row, eight per f128 row) -- for f64, slot s of the row of (position pos, local coset j) is lanes j * 2^lg + 2 s, + 1 of the tile row while
            // 2 s < 2^lg and zeros after that (the lanes past the last column are zero in the tile).  The four local
            // rows of a position are adjacent in the LDE: 16 consecutive threads write 256 contiguous bytes.  (A lane
            // pair writing its own slot and its row's zeros touched 64 different rows per store instruction.)
            const uint32_t lg = a.lg_log, gl = 1u << lg, cpr = a.cpr_log;
            constexpr uint32_t EPU = 16 / F::BYTES;                   // elements per 16-byte unit: 2 (f64) or 1 (f128)
            const uint32_t upr = (uint32_t)a.row_width / EPU;          // units per row: 4 (f64) or 8 (f128) -- a power of two
            const uint32_t n_items = (D << cpr) * upr;
            for (uint32_t idx = threadIdx.x; idx < n_items; idx += NT) {
                const uint32_t u = idx & (upr - 1), rj = idx / upr, j = rj & ((1u << cpr) - 1), pos = rj >> cpr;
                const uint64_t k = rev_o + ((uint64_t)seg_digit_reverse<F>(pos, logD_) << out_shift);
                T *row = a.dst + (uint64_t)(uint32_t)k * k_stride + (((uint64_t)c << cpr) + j) * a.row_width + EPU * u;
                const T *xs = x + pos * S + (j << lg) + EPU * u;
                if (EPU == 2) {
                    P2 v{F::zero(), F::zero()};
                    if (2 * u < gl) v.a = xs[0];
                    if (2 * u + 1 < gl) v.b = xs[1];
                    store_pair(row, v);
                } else {
                    row[0] = u < gl ? xs[0] : F::zero();
                }
            }
            pa = pb = nullptr;
        }
        const bool tail = !PACKED && F::BYTES == 16 && a.tail_pad && g + 1 == a.n_seg;  // (single trace: every lane has its pa)
        for (uint32_t pj = 0; (pa || pb) && pj < D; pj += pstride) {
            const uint64_t k = rev_o + ((uint64_t)(k0 | seg_digit_reverse<F>(pj, logD_)) << out_shift);
            const P2 v = *reinterpret_cast<P2 *>(x + (pos0 + pj) * S + lane_a);
            const uint64_t off = (uint64_t)(uint32_t)k * k_stride;  // k < 2^32 rows: one 32 x 32 -> 64 multiply
            if (tail) store_pair(pa + off + S, P2{F::zero(), F::zero()});
            if (pz) store_row_padding<F>(pz + off, a.store_cols, (uint32_t)a.row_width);
            if (pz2) store_row_padding<F>(pz2 + off, a.store_cols, (uint32_t)a.row_width);
            if (pair_store) {
#ifdef WF_EXP_SKIP_STORE
                if (*reinterpret_cast<const uint32_t *>(&v.a) == a.logN + 77777u)
#endif
                store_pair(pa + off, v);
            } else {
                if (pa) pa[off] = v.a;
                if (pb) pb[off] = v.b;
            }
        }
    }

    // Fused leaf hashing, after the row stores have been issued (they drain while the lanes hash): one lane per row
    // position (blockDim >= D/2: at most two rows per thread); the S lanes of a row are its whole 64 bytes (unused lanes
    // are zero), canonical bytes as in hash_elements (blake/mod.rs:46-59).
    if (OUT == SEG_OUT_ROWS && PACKED && a.leaves) {
        // coset-packed rows: the 2^lg lanes of local coset j are the message of LDE row (c << cpr) + j (its unused lanes,
        // like the rest of the 64-byte block, are zero)
        constexpr uint32_t WPE = F::BYTES / 4;
        const uint32_t lg = a.lg_log, ncos = 1u << a.cpr_log;
        for (uint32_t pos = threadIdx.x; pos < D; pos += NT) {
            const uint64_t k = rev_o + ((uint64_t)seg_digit_reverse<F>(pos, logD_) << (a.logN - logD_));
            uint32_t *leaf = a.leaves + ((uint64_t)(uint32_t)k * a.rows_per_k + ((uint64_t)c << a.cpr_log)) * 8;
#pragma unroll 1
            for (uint32_t j = 0; j < ncos; j++) {
                const T *e = x + (size_t)pos * S + (j << lg);
                uint32_t m[16], out[8];
#pragma unroll
                for (uint32_t w = 0; w < 16; w++) m[w] = 0;
                elem_words<F>(e[0], &m[0]);
                if (lg >= 1) elem_words<F>(e[1], &m[WPE]);
                if (S >= 8 && lg >= 2) {  // (f128 rows pack at most two lanes per coset)
                    elem_words<F>(e[2], &m[(2 * WPE) & 15]);
                    elem_words<F>(e[3], &m[(3 * WPE) & 15]);
                }
                b3::set_iv(out);
                b3::compress(out, m, 0, 0, a.hash_epr * F::BYTES, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT);
                uint4 *dl = reinterpret_cast<uint4 *>(leaf + j * 8);
                dl[0] = make_uint4(out[0], out[1], out[2], out[3]);
                dl[1] = digest_hi(out[4], out[5], out[6], out[7], a.digest_words);
            }
        }
    }
    if (OUT == SEG_OUT_ROWS && !PACKED && a.leaves) {
        constexpr uint32_t WPE = F::BYTES / 4;
        for (uint32_t pos = threadIdx.x; pos < D; pos += NT) {
            T ev[S];
            uint4 *evq = reinterpret_cast<uint4 *>(ev);
            const uint4 *q = reinterpret_cast<const uint4 *>(x + (size_t)pos * S);
#pragma unroll
            for (uint32_t w = 0; w < 4; w++) evq[w] = q[w];
            uint32_t m[16], out[8];
#pragma unroll
            for (uint32_t e = 0; e < S; e++) elem_words<F>(ev[e], &m[e * WPE]);
            b3::set_iv(out);
            b3::compress(out, m, 0, 0, a.hash_epr * F::BYTES, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT);
            const uint64_t k = rev_o + ((uint64_t)seg_digit_reverse<F>(pos, logD_) << (a.logN - logD_));
            uint4 *dl = reinterpret_cast<uint4 *>(a.leaves + ((uint64_t)(uint32_t)k * a.rows_per_k + c) * 8);
            dl[0] = make_uint4(out[0], out[1], out[2], out[3]);
            dl[1] = digest_hi(out[4], out[5], out[6], out[7], a.digest_words);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Last pass of a multi-pass evaluation with fused leaf hashing, as a PERSISTENT kernel, for matrices whose combined row
// (all base columns of all traces, RowMatrix::commit_to_comb_rows, row_matrix.rs:204-238) is at most one BLAKE3 chunk:
// n_seg <= 16 segments.  Segment g of a row IS the g-th 64-byte block of the hashed message (base columns are packed
// into segments in concatenation order), so a work-group that walks over the n_seg tiles of one (coset, row block)
// keeps the chaining values of its rows in registers and compresses one block per tile; the LDE is never read back.
// The grid is one resident set of work-groups; each takes (coset, row block) tickets from a per-XCD counter until they
// run out (every work-group reaches that exit); the rows of its next tile are requested right after the row stores of
// the current one and arrive while the lanes hash -- the only stretch of the tile loop that has 32 VGPRs to spare (the
// transform itself needs ~80 of the 128 that two resident work-groups per CU allow).  Same tiles, arithmetic and
// outputs as k_seg_last<F, ROWS> + k_hash_rows; a static tile walk (t += gridDim) measured 10 % slower than the
// one-tile-per-work-group kernel, the dynamic one 5 % faster.

// MULTI = false: one segment of one trace (the bench workload) -- no chaining values to carry, one lane pair mapping.
// CHUNKED (with MULTI): rows longer than one BLAKE3 chunk -- a ticket is (coset, row block, chunk): the work-group walks the
// 16 segments of that chunk and writes the rows' chunk chaining values; k_hash_merge_chunks folds them into the leaves.
// SMALL: tiles of at most 2^9 rows (<= 256 threads): compiled without the 128-VGPR cap that 1024-thread work-groups impose
// (the multi-segment variants spill a few registers under it)
template <class F, bool MULTI, bool PADT = false, bool CHUNKED = false, bool SMALL = false, int LOGD = 0>  // PADT: SegArgs::pad_traces (kept out of the other instantiations' registers)
// (minimum waves per SIMD: 4 for the tile-size-specialised forms and for the small-tile chunk-by-chunk form of padded packed traces,
// whose 130 registers otherwise cost it a wave: 13 traces x 20 columns at 2^18 last pass 4.28 -> 4.09 ms; the same cap on the
// one-chunk small-tile padded form spills and loses 2-7 %: profiles/r04_attribution.txt)
__global__ void __launch_bounds__((LOGD) ? (1 << ((LOGD) ? (LOGD) - 1 : 0)) : (SMALL ? 256 : 1024), ((LOGD) || (SMALL && PADT && CHUNKED)) ? 4 : 1)
k_seg_last_hash(SegArgs<F> a) {
    typedef typename F::T T;
    const uint32_t logD_ = LOGD ? (uint32_t)LOGD : a.logD;  // (a local, not a.logD = LOGD: a modified copy of the arguments would live in scratch, where decode() indexes prev_log[])
    const uint32_t NT = tile_threads<LOGD>();
    typedef Pair<T> P2;
    constexpr uint32_t S = SegCfg<F>::S, HP = SegCfg<F>::HP;
    constexpr uint32_t hp_shift = HP == 4 ? 2 : 1;
    constexpr uint32_t WPE = F::BYTES / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint32_t D = 1u << logD_;
    T *x = reinterpret_cast<T *>(smem_raw);
    T *twd = x + (size_t)D * S;
    const uint32_t n_chunks = CHUNKED ? a.n_chunks : 1;
    const uint64_t total = (uint64_t)a.n_cosets * a.O * n_chunks;  // tickets: (coset, row block[, chunk])
    const uint64_t seg_elems = ((uint64_t)1 << a.logN) * S;
    const uint32_t step = NT;
    const uint32_t out_shift = a.logN - logD_;
    const uint32_t k_stride = a.rows_per_k * (uint32_t)a.row_width;  // < 2^19
    const uint32_t hash_bytes = a.hash_epr * F::BYTES;             // <= 1024 (one chunk) unless CHUNKED

    // (twd[D - 1] is never read: the largest exponent a round uses is (D / 16 - 1) * 15, or (D / 4 - 1) * 3 without
    // radix-16 rounds -- its 8 / 16 bytes hold the two ticket words, so that a 2^10-row f128 tile (64 KiB + 16 KiB of
    // twiddles) stays at exactly half a CU's LDS: two work-groups per CU)
    for (uint32_t e = threadIdx.x; e < D - 1; e += step) twd[e] = a.digit_tw[e];

    auto decode = [&](uint64_t t, uint32_t &c, uint64_t &o, uint64_t &rev_o, uint32_t &ch) {
        uint64_t bid = xcd_group_index(t, total);  // coset fastest, 8 consecutive tickets on one XCD
        const uint32_t b32 = (uint32_t)bid, q32 = b32 / a.n_cosets;  // grids are below 2^31
        c = b32 - q32 * a.n_cosets;
        o = q32;
        ch = 0;
        if (CHUNKED) {  // chunk next-fastest: the chunks of one row block follow each other on an XCD
            const uint32_t q2 = q32 / n_chunks;
            ch = q32 - q2 * n_chunks;
            o = q2;
        }
        rev_o = 0;
        uint32_t bits = 0;
        for (uint32_t q = 0; q < a.n_prev; q++) bits += a.prev_log[q];
        uint32_t hi = bits, sh = 0;
        for (uint32_t q = 0; q < a.n_prev; q++) {
            hi -= a.prev_log[q];
            rev_o |= ((o >> hi) & (((uint64_t)1 << a.prev_log[q]) - 1)) << sh;
            sh += a.prev_log[q];
        }
    };
    auto tile_src = [&](uint32_t c, uint64_t o, uint32_t g) -> const T * {
        return a.src + ((uint64_t)c * a.n_seg + g) * seg_elems + (o << logD_) * S;
    };

    // Tickets are handed out dynamically, per XCD (work-groups are dispatched to XCD blockIdx % 8): ticket index =
    // 8 * n + xcd, so that the order within an XCD -- 8 consecutive tickets = the cosets of one row block -- is kept and
    // a slow work-group does not hold back a fixed share of the work.  total is a multiple of 8.
    uint32_t *ticket_sh = reinterpret_cast<uint32_t *>(twd + (D - 1));  // two words in the unused last twiddle slot
    const uint32_t xcd = blockIdx.x & 7;
    const uint64_t per_xcd = total >> 3;
    auto next_ticket = [&](uint32_t slot) -> uint64_t {  // uniform result; includes a barrier
        if (threadIdx.x == 0) ticket_sh[slot] = atomicAdd(a.tile_counters + xcd, 1u);
        __syncthreads();
        return ticket_sh[slot];
    };
    uint64_t ticket = next_ticket(0);
    // Counters reset themselves: every work-group takes exactly one ticket past the end, then signs off on the exit
    // counter of its XCD (words 8..15); the last one to sign off zeroes both for the next launch.  No memset between
    // launches -- a captured graph replays correctly (a 32-byte memset node did not: its replays kept stale counters).
    auto sign_off = [&]() {
        if (threadIdx.x == 0) {
            // (no fence: this thread's past-the-end ticket atomic has returned -- the branch here depends on its value --
            // before the exit atomic is issued; a device-scope release fence would write back the XCD's L2 per work-group)
            const uint32_t mine = (gridDim.x + 7 - xcd) >> 3;  // work-groups with blockIdx % 8 == xcd
            if (atomicAdd(a.tile_counters + 8 + xcd, 1u) == mine - 1) {
                atomicExch(a.tile_counters + xcd, 0u);
                atomicExch(a.tile_counters + 8 + xcd, 0u);
            }
        }
    };
    if (ticket >= per_xcd) {
        sign_off();
        return;
    }
    uint32_t c, g = 0, ch = 0;  // g stays 0 without MULTI, ch without CHUNKED
    uint64_t o, rev_o;
    decode(ticket * 8 + xcd, c, o, rev_o, ch);
    g = 16 * ch;
    const T *src = tile_src(c, o, g);
    // blockDim == D / 2 (the launcher guarantees it): the tile is one contiguous run of 4 * D 16-byte chunks (a row is 64
    // bytes for either field), eight per thread, copied to the same offsets of `x`.  Native vector registers rather
    // than arrays or structs for everything carried around the tile loop: those would live in scratch memory.
    uint4 q0, q1, q2, q3, q4, q5, q6, q7;
    uint4 cva0 = make_uint4(0, 0, 0, 0), cva1 = cva0, cvb0 = cva0, cvb1 = cva0;  // chaining values of this lane's two rows
#define WF_TILE_LOAD(SRC, TID)                                      \
    do {                                                            \
        const uint4 *s_ = reinterpret_cast<const uint4 *>(SRC) + (TID); \
        q0 = s_[0];                                                 \
        q1 = s_[step];                                              \
        q2 = s_[2 * step];                                          \
        q3 = s_[3 * step];                                          \
        q4 = s_[4 * step];                                          \
        q5 = s_[5 * step];                                          \
        q6 = s_[6 * step];                                          \
        q7 = s_[7 * step];                                          \
    } while (0)
    WF_TILE_LOAD(src, threadIdx.x);
#ifdef WF_EXP_STAMPS
    unsigned long long st_acc[7] = {0, 0, 0, 0, 0, 0, 0}, st_t0, st_t1;
#define WF_STAMP(i)                                   \
    do {                                              \
        st_t1 = __builtin_amdgcn_s_memtime();         \
        st_acc[i] += st_t1 - st_t0;                   \
        st_t0 = st_t1;                                \
    } while (0)
    st_t0 = __builtin_amdgcn_s_memtime();
#else
#define WF_STAMP(i)
#endif

    while (true) {
        {
            uint4 *d_ = reinterpret_cast<uint4 *>(x);
            const uint32_t t_ = opaque_tid();
#define WF_TILE_PUT(J, Q) d_[t_ + (J) * step] = Q
            WF_TILE_PUT(0, q0);
            WF_TILE_PUT(1, q1);
            WF_TILE_PUT(2, q2);
            WF_TILE_PUT(3, q3);
            WF_TILE_PUT(4, q4);
            WF_TILE_PUT(5, q5);
            WF_TILE_PUT(6, q6);
            WF_TILE_PUT(7, q7);
#undef WF_TILE_PUT
            // the rows are in LDS: end the registers' live ranges here.  They are refilled only `if (more)` below, so left alone
            // their old contents count as live through the transform and the hashing of the last tile's iteration -- 32 VGPRs
            // that the transform needs (a constant costs nothing until the iteration that leaves the loop)
            q0 = q1 = q2 = q3 = q4 = q5 = q6 = q7 = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        WF_STAMP(0);  // tile in LDS (waits for the prefetched rows)
        // (SMALL instantiations -- tiles of at most 2^9 rows -- have no 128-register cap: the 2^7- and 2^9-row sequences too)
        seg_lds_ntt<F, 1, (LOGD ? (1u << LOGD) : SMALL ? (FIX7 | FIX9) : FIX_BIG)>(x, twd, logD_, NT, nullptr, false, true);
        WF_STAMP(1);  // transform

        // row stores: lane pair (2l, 2l+1) of row position pos -> its place in LDE row k * rows_per_k + c of its trace
        {
            const uint32_t tid = opaque_tid();
            const uint32_t pstride = step >> hp_shift, pos0 = tid >> hp_shift, lane_a = 2 * (tid & (HP - 1));
            const uint32_t B = g * S + lane_a;  // global base column of lane a
            T *pa = nullptr, *pb = nullptr, *pz = nullptr, *pz2 = nullptr;  // pz*: first padding element after a trace's last column
            bool pair = false;
            if (F::BYTES == 16 && !(PADT && a.row_width == 8)) {  // (f128 rows of 8: thread quads below)
                store_rows_by_element<F>(x, a, g, c, rev_o, out_shift, k_stride, tid, NT, a.tail_pad && g + 1 == a.n_seg, PADT);
            } else if (!MULTI) {
                if (pos0 < D && lane_a < a.store_cols) {
                    pa = a.dst + (uint64_t)c * a.row_width + lane_a;
                    pair = lane_a + 1 < a.store_cols;
                    if (PADT && lane_a + 2 >= a.store_cols) pz = pa + (a.store_cols - lane_a);
                }
            } else if (PADT && (a.row_width == 8 || (F::BYTES == 8 && a.row_width == 16 && a.store_cols <= 10))) {
                store_rows_narrow<F>(x, a, g, c, rev_o, out_shift, k_stride, tid, NT);
            } else if (pos0 < D && B < a.total_store_cols) {
                const uint32_t t0 = B / a.store_cols, c0 = B - t0 * a.store_cols;
                pa = a.dst + (uint64_t)t0 * a.trace_lde_elems + (uint64_t)c * a.row_width + c0;
                pair = c0 + 1 < a.store_cols && (c0 & 1) == 0;  // both lanes in one trace, 16-byte aligned
                if (PADT && c0 + 1 == a.store_cols) pz = pa + 1;
                if (B + 1 < a.total_store_cols) {
                    const uint32_t t1 = (B + 1) / a.store_cols, c1 = (B + 1) - t1 * a.store_cols;
                    pb = a.dst + (uint64_t)t1 * a.trace_lde_elems + (uint64_t)c * a.row_width + c1;
                    if (PADT && c1 + 1 == a.store_cols) pz2 = pb + 1;
                }
            }
            if (pa) {
                const uint32_t k0 = seg_digit_reverse<F>(pos0, logD_);
                const bool tail = F::BYTES == 16 && a.tail_pad && g + 1 == a.n_seg;  // (single trace: every lane has its pa)
                for (uint32_t pj = 0; pj < D; pj += pstride) {
#if defined(WF_EXP_LOCAL_STORE) && WF_EXP_LOCAL_STORE == 1  // diagnostic (wrong output): rows of a tile next to each other, cosets interleaved
                    const uint64_t k = (o << logD_) + pos0 + pj;
#elif defined(WF_EXP_LOCAL_STORE)  // diagnostic (wrong output): the tile written as one contiguous 64 KiB run
                    const uint64_t k = 0;
#else
                    const uint64_t k = rev_o + ((uint64_t)(k0 | seg_digit_reverse<F>(pj, logD_)) << out_shift);
#endif
                    const P2 v = *reinterpret_cast<P2 *>(x + (pos0 + pj) * S + lane_a);
#if defined(WF_EXP_LOCAL_STORE) && WF_EXP_LOCAL_STORE == 2
                    const uint64_t off = (((uint64_t)c * a.O + o) << logD_) * S + (uint64_t)(pos0 + pj) * S - (uint64_t)c * a.row_width;
#else
                    const uint64_t off = (uint64_t)(uint32_t)k * k_stride;
#endif
                    if (pair) {
                        store_pair(pa + off, v);
                    } else {
                        pa[off] = v.a;
                        if (pb) pb[off] = v.b;
                    }
                    if (tail) store_pair(pa + off + S, P2{F::zero(), F::zero()});
                    if (PADT && pz) store_row_padding<F>(pz + off, a.store_cols, (uint32_t)a.row_width);
                    if (PADT && MULTI && pz2) store_row_padding<F>(pz2 + off, a.store_cols, (uint32_t)a.row_width);
                }
            }
        }

        WF_STAMP(2);  // row stores issued
        // the next tile -- the next segment of this row block, or the first one of a new ticket -- starts its way into
        // registers; it lands while the leaves are hashed
        bool more = true;
        uint32_t cn = c, gn = g + 1, chn = ch;
        uint64_t on = o, rev_on = rev_o;
        const uint32_t g_end = CHUNKED ? min(16 * ch + 16, a.n_seg) : a.n_seg;  // one past the last segment of this ticket
        if (!MULTI || gn == g_end) {
            ticket = next_ticket(1);
            more = ticket < per_xcd;
            if (more) decode(ticket * 8 + xcd, cn, on, rev_on, chn);
            gn = 16 * chn;
        }
        if (more) {
            src = tile_src(cn, on, gn);
            WF_TILE_LOAD(src, opaque_tid());
        }

        WF_STAMP(3);  // ticket + next tile requested
        // block g of the rows' messages: one lane per row position, two rows per lane (tid and tid + D/2); the S lanes
        // of a tile row are its 64 message bytes (lanes past the last column are zero), canonical as in hash_elements
        {
            const uint32_t tid = opaque_tid();
            const bool last = !MULTI || g + 1 == g_end, first = g == 16 * ch;
            const uint32_t flags = (first ? (uint32_t)b3::CHUNK_START : 0u) |
                                   (last ? (uint32_t)(CHUNKED ? b3::CHUNK_END : b3::CHUNK_END | b3::ROOT) : 0u);
            const uint32_t blen = min(64u, hash_bytes - 64u * g);
#pragma unroll 1  // one compression in flight: interleaving the two rows measured slower (and doubles the code)
            for (uint32_t r = 0; r < 2; r++) {
                const uint32_t pos = tid + r * step;
                T ev[S];
                uint4 *evq = reinterpret_cast<uint4 *>(ev);
                const uint4 *q = reinterpret_cast<const uint4 *>(x + (size_t)pos * S);
#pragma unroll
                for (uint32_t w = 0; w < 4; w++) evq[w] = q[w];
                uint32_t m[16], cv[8];
#pragma unroll
                for (uint32_t e = 0; e < S; e++) elem_words<F>(ev[e], &m[e * WPE]);
                if (!MULTI || first) {
                    b3::set_iv(cv);
                } else {
                    const uint4 lo = r == 0 ? cva0 : cvb0, hi = r == 0 ? cva1 : cvb1;
                    cv[0] = lo.x; cv[1] = lo.y; cv[2] = lo.z; cv[3] = lo.w;
                    cv[4] = hi.x; cv[5] = hi.y; cv[6] = hi.z; cv[7] = hi.w;
                }
                b3::compress(cv, m, ch, 0, blen, flags);  // counter = chunk index
                const uint4 lo = make_uint4(cv[0], cv[1], cv[2], cv[3]), hi = make_uint4(cv[4], cv[5], cv[6], cv[7]);
                if (last) {
#if defined(WF_EXP_LOCAL_STORE) && WF_EXP_LOCAL_STORE == 1
                    const uint64_t k = (o << logD_) + pos;
                    const uint64_t row = (uint64_t)(uint32_t)k * a.rows_per_k + c;
#elif defined(WF_EXP_LOCAL_STORE)
                    const uint64_t row = (((uint64_t)c * a.O + o) << logD_) + pos;
#else
                    const uint64_t k = rev_o + ((uint64_t)seg_digit_reverse<F>(pos, logD_) << out_shift);
                    const uint64_t row = (uint64_t)(uint32_t)k * a.rows_per_k + c;
#endif
                    uint4 *dl = reinterpret_cast<uint4 *>(CHUNKED ? a.chunk_cvs + (row * n_chunks + ch) * 8 : a.leaves + row * 8);
                    dl[0] = lo;
                    dl[1] = CHUNKED ? hi : digest_hi(hi.x, hi.y, hi.z, hi.w, a.digest_words);  // (chunk chaining values stay whole)
                } else if (r == 0) {
                    cva0 = lo;
                    cva1 = hi;
                } else {
                    cvb0 = lo;
                    cvb1 = hi;
                }
            }
        }
        WF_STAMP(4);  // hashing
#ifdef WF_EXP_STAMPS
        st_acc[6]++;
#endif
        if (!more) {
            sign_off();
            break;
        }
        __syncthreads();  // x is rewritten by the next tile
        WF_STAMP(5);  // end-of-tile barrier
        c = cn;
        g = MULTI ? gn : 0;
        ch = chn;
        o = on;
        rev_o = rev_on;
    }
#ifdef WF_EXP_STAMPS
    if (threadIdx.x == 0 && a.stamps)
        for (int i = 0; i < 7; i++) atomicAdd(a.stamps + (size_t)blockIdx.x * 8 + i, st_acc[i]);
#endif
}

#undef WF_TILE_LOAD
#undef WF_STAMP

// ---------------------------------------------------------------------------------------------------------------
// The fused persistent last pass for ONE trace whose last segment is at most half full (f128: 9, 10, 13, 14 .. columns; f64:
// 9 .. 12, 17 .. 20 ..): the tail segment's lanes would be transformed half empty in every coset.  Instead the tail is
// COSET-PACKED -- a tile row of the tail holds the tail columns of two cosets, lane = (local coset j, column) with S / 2 lanes
// per coset, as the PACKED kernels lay narrow matrices out -- and a ticket is (coset pair, row block): the work-group walks the
// full segments of coset 2 cg, then those of coset 2 cg + 1 (a BLAKE3 block each, chaining values in registers; the first
// coset's wait in their rows' leaf slots, written and read back by the same thread), then ONE tail tile that finishes the rows
// of both cosets.  10 f128 columns: 2 * 2 + 1 = 5 tiles per coset pair instead of 6; 10 f64 columns: 3 instead of 4.
// Same arithmetic and outputs as k_seg_last_hash<F, true>: row (k, c) = segments 0 .. nf - 1 of coset c, then the tail's lanes
// of local coset c & 1, then zeros up to the row width (the tail tile writes those too).
template <class F, bool SMALL = false, int LOGD = 0>
__global__ void WF_TILE_BOUNDS(LOGD, SMALL ? 256 : 1024) k_seg_last_hash_tp(SegArgs<F> a) {
    typedef typename F::T T;
    typedef Pair<T> P2;
    const uint32_t logD_ = LOGD ? (uint32_t)LOGD : a.logD;
    const uint32_t NT = tile_threads<LOGD>();
    constexpr uint32_t S = SegCfg<F>::S, HP = SegCfg<F>::HP, LGT = ilog2_const(S / 2);  // S / 2 lanes per coset in a tail row
    constexpr uint32_t hp_shift = HP == 4 ? 2 : 1;
    constexpr uint32_t WPE = F::BYTES / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint32_t D = 1u << logD_;
    T *x = reinterpret_cast<T *>(smem_raw);
    T *twd = x + (size_t)D * S;
    const uint32_t nf = a.n_seg - 1;                       // full segments
    const uint32_t n_pairs = a.n_cosets >> 1;
    const uint64_t total = (uint64_t)n_pairs * a.O;        // tickets: (coset pair, row block); a multiple of 8
    const uint64_t seg_elems = ((uint64_t)1 << a.logN) * S;
    const uint32_t step = NT;
    const uint32_t out_shift = a.logN - logD_;
    const uint32_t k_stride = a.rows_per_k * (uint32_t)a.row_width;
    const uint32_t hash_bytes = a.hash_epr * F::BYTES;
    const uint32_t n_tiles = 2 * nf + 1;                   // per ticket

    for (uint32_t e = threadIdx.x; e < D - 1; e += step) twd[e] = a.digit_tw[e];  // (twd[D - 1] holds the ticket words: k_seg_last_hash)

    auto decode = [&](uint64_t t, uint32_t &cg, uint64_t &o, uint64_t &rev_o) {
        const uint64_t bid = xcd_group_index(t, total);  // coset pair fastest, 8 consecutive tickets on one XCD
        const uint32_t b32 = (uint32_t)bid, q32 = b32 / n_pairs;
        cg = b32 - q32 * n_pairs;
        o = q32;
        rev_o = 0;
        uint32_t bits = 0;
        for (uint32_t q = 0; q < a.n_prev; q++) bits += a.prev_log[q];
        uint32_t hi = bits, sh = 0;
        for (uint32_t q = 0; q < a.n_prev; q++) {
            hi -= a.prev_log[q];
            rev_o |= ((o >> hi) & (((uint64_t)1 << a.prev_log[q]) - 1)) << sh;
            sh += a.prev_log[q];
        }
    };
    // tile s of ticket (cg, o): s < 2 nf: segment s % nf of coset 2 cg + s / nf; s == 2 nf: the tail tile of the pair
    auto tile_src = [&](uint32_t cg, uint64_t o, uint32_t s) -> const T * {
        if (s == 2 * nf) return a.src_tail + (uint64_t)cg * seg_elems + (o << logD_) * S;
        const uint32_t j = s >= nf ? 1u : 0u, g = s - j * nf;
        return a.src + ((uint64_t)(2 * cg + j) * nf + g) * seg_elems + (o << logD_) * S;
    };

    uint32_t *ticket_sh = reinterpret_cast<uint32_t *>(twd + (D - 1));
    const uint32_t xcd = blockIdx.x & 7;
    const uint64_t per_xcd = total >> 3;
    auto next_ticket = [&](uint32_t slot) -> uint64_t {
        if (threadIdx.x == 0) ticket_sh[slot] = atomicAdd(a.tile_counters + xcd, 1u);
        __syncthreads();
        return ticket_sh[slot];
    };
    auto sign_off = [&]() {
        if (threadIdx.x == 0) {
            const uint32_t mine = (gridDim.x + 7 - xcd) >> 3;
            if (atomicAdd(a.tile_counters + 8 + xcd, 1u) == mine - 1) {
                atomicExch(a.tile_counters + xcd, 0u);
                atomicExch(a.tile_counters + 8 + xcd, 0u);
            }
        }
    };
    uint64_t ticket = next_ticket(0);
    if (ticket >= per_xcd) {
        sign_off();
        return;
    }
    uint32_t cg, s = 0;
    uint64_t o, rev_o;
    decode(ticket * 8 + xcd, cg, o, rev_o);
    uint4 q0, q1, q2, q3, q4, q5, q6, q7;
    uint4 cva0 = make_uint4(0, 0, 0, 0), cva1 = cva0, cvb0 = cva0, cvb1 = cva0;  // chaining values of this lane's two rows
#define WF_TILE_LOAD(SRC, TID)                                          \
    do {                                                                \
        const uint4 *s_ = reinterpret_cast<const uint4 *>(SRC) + (TID); \
        q0 = s_[0];                                                     \
        q1 = s_[step];                                                  \
        q2 = s_[2 * step];                                              \
        q3 = s_[3 * step];                                              \
        q4 = s_[4 * step];                                              \
        q5 = s_[5 * step];                                              \
        q6 = s_[6 * step];                                              \
        q7 = s_[7 * step];                                              \
    } while (0)
    WF_TILE_LOAD(tile_src(cg, o, 0), threadIdx.x);

    while (true) {
        {
            uint4 *d_ = reinterpret_cast<uint4 *>(x);
            const uint32_t t_ = opaque_tid();
            d_[t_] = q0;
            d_[t_ + step] = q1;
            d_[t_ + 2 * step] = q2;
            d_[t_ + 3 * step] = q3;
            d_[t_ + 4 * step] = q4;
            d_[t_ + 5 * step] = q5;
            d_[t_ + 6 * step] = q6;
            d_[t_ + 7 * step] = q7;
            q0 = q1 = q2 = q3 = q4 = q5 = q6 = q7 = make_uint4(0, 0, 0, 0);  // (ends their live ranges: see k_seg_last_hash)
        }
        __syncthreads();
        seg_lds_ntt<F, 1, (LOGD ? (1u << LOGD) : SMALL ? (FIX7 | FIX9) : FIX_BIG)>(x, twd, logD_, NT, nullptr, false, true);

        const bool tail = s == 2 * nf;
        const uint32_t j = s >= nf ? 1u : 0u, g = tail ? nf : s - j * nf, c = 2 * cg + j;  // (tail: j, c of the SECOND coset; unused)
        uint4 pka0 = make_uint4(0, 0, 0, 0), pka1 = pka0, pkb0 = pka0, pkb1 = pka0;  // the first coset's chaining values, on their way back (tail tile)
        // ---- row stores
        if (!tail) {
            const uint32_t tid = opaque_tid();
            if (F::BYTES == 16) {  // one element per thread: whole 64-byte runs per store instruction
                store_rows_by_element<F>(x, a, g, c, rev_o, out_shift, k_stride, tid, NT, false, false);
            } else {
                const uint32_t pstride = step >> hp_shift, pos0 = tid >> hp_shift, lane_a = 2 * (tid & (HP - 1));
                T *pa = a.dst + (uint64_t)c * a.row_width + g * S + lane_a;
                const uint32_t k0 = seg_digit_reverse<F>(pos0, logD_);
                for (uint32_t pj = 0; pj < D; pj += pstride) {
                    const uint64_t k = rev_o + ((uint64_t)(k0 | seg_digit_reverse<F>(pj, logD_)) << out_shift);
                    store_pair(pa + (uint64_t)(uint32_t)k * k_stride, *reinterpret_cast<P2 *>(x + (pos0 + pj) * S + lane_a));
                }
            }
        } else {
            // the rest of both cosets' rows from column nf * S on: the tail columns, then the zero padding -- one thread per
            // 16-byte unit, the two local rows of a position next to each other.  The remainder of a row is 4 or 8 units (rows are
            // multiples of 8 elements, full segments of S): this thread keeps its unit and local coset and walks over positions
            // pos0 + it * pstride, whose digit reversal splits as in the loops above.
            const uint32_t tid = opaque_tid();
            constexpr uint32_t EPU = 16 / F::BYTES;
            const uint32_t rest = (uint32_t)a.row_width - nf * S, lupr = ilog2_pow2(rest / EPU);
            const uint32_t u = tid & ((1u << lupr) - 1), jj = (tid >> lupr) & 1u, pos0 = tid >> (lupr + 1), pstride = NT >> (lupr + 1);
            // the first coset's chaining values come back from their leaf slots while the rows are stored
            {
                const uint64_t ka = rev_o + ((uint64_t)seg_digit_reverse<F>(tid, logD_) << out_shift);
                const uint64_t kb = rev_o + ((uint64_t)seg_digit_reverse<F>(tid + step, logD_) << out_shift);
                const uint4 *la = reinterpret_cast<const uint4 *>(a.leaves + ((uint64_t)(uint32_t)ka * a.rows_per_k + 2 * cg) * 8);
                const uint4 *lb = reinterpret_cast<const uint4 *>(a.leaves + ((uint64_t)(uint32_t)kb * a.rows_per_k + 2 * cg) * 8);
                pka0 = la[0];
                pka1 = la[1];
                pkb0 = lb[0];
                pkb1 = lb[1];
            }
            const uint32_t k0 = seg_digit_reverse<F>(pos0, logD_);
            T *row0p = a.dst + (uint64_t)(2 * cg + jj) * a.row_width + nf * S + EPU * u;
            const T *xs0 = x + (jj << LGT);
            const bool live_a = EPU * u < a.tail_cols, live_b = EPU * u + 1 < a.tail_cols;
            for (uint32_t pj = 0; pj < D; pj += pstride) {
                const uint64_t k = rev_o + ((uint64_t)(k0 | seg_digit_reverse<F>(pj, logD_)) << out_shift);
                T *row = row0p + (uint64_t)(uint32_t)k * k_stride;
                const T *xs = xs0 + (pos0 + pj) * S;
                if (EPU == 2) {
                    P2 v{F::zero(), F::zero()};
                    if (live_a) v.a = xs[2 * u];
                    if (live_b) v.b = xs[2 * u + 1];
                    store_pair(row, v);
                } else {
                    row[0] = live_a ? xs[u] : F::zero();
                }
            }
        }
        // ---- the next tile starts its way into registers
        bool more = true;
        uint32_t cgn = cg, sn = s + 1;
        uint64_t on = o, rev_on = rev_o;
        if (sn == n_tiles) {
            ticket = next_ticket(1);
            more = ticket < per_xcd;
            if (more) decode(ticket * 8 + xcd, cgn, on, rev_on);
            sn = 0;
        }
        if (more) WF_TILE_LOAD(tile_src(cgn, on, sn), opaque_tid());
        // ---- hashing: block g of the rows of coset c (full tile), or the last block of the rows of both cosets (tail tile)
        {
            const uint32_t tid = opaque_tid();
#pragma unroll 1
            for (uint32_t r = 0; r < 2; r++) {
                const uint32_t pos = tid + r * step;
                const uint64_t k = rev_o + ((uint64_t)seg_digit_reverse<F>(pos, logD_) << out_shift);
                if (!tail) {
                    T ev[S];
                    uint4 *evq = reinterpret_cast<uint4 *>(ev);
                    const uint4 *q = reinterpret_cast<const uint4 *>(x + (size_t)pos * S);
#pragma unroll
                    for (uint32_t w = 0; w < 4; w++) evq[w] = q[w];
                    uint32_t m[16], cv[8];
#pragma unroll
                    for (uint32_t e = 0; e < S; e++) elem_words<F>(ev[e], &m[e * WPE]);
                    if (g == 0) {
                        b3::set_iv(cv);
                    } else {
                        const uint4 lo = r == 0 ? cva0 : cvb0, hi = r == 0 ? cva1 : cvb1;
                        cv[0] = lo.x; cv[1] = lo.y; cv[2] = lo.z; cv[3] = lo.w;
                        cv[4] = hi.x; cv[5] = hi.y; cv[6] = hi.z; cv[7] = hi.w;
                    }
                    b3::compress(cv, m, 0, 0, 64, g == 0 ? (uint32_t)b3::CHUNK_START : 0u);
                    const uint4 lo = make_uint4(cv[0], cv[1], cv[2], cv[3]), hi = make_uint4(cv[4], cv[5], cv[6], cv[7]);
                    if (j == 0 && g + 1 == nf) {
                        // the first coset's chaining values wait in their rows' leaf slots (this thread reads them back at the tail tile)
                        uint4 *dl = reinterpret_cast<uint4 *>(a.leaves + ((uint64_t)(uint32_t)k * a.rows_per_k + c) * 8);
                        dl[0] = lo;
                        dl[1] = hi;
                    } else if (r == 0) {
                        cva0 = lo;
                        cva1 = hi;
                    } else {
                        cvb0 = lo;
                        cvb1 = hi;
                    }
                } else {
                    const uint32_t blen = hash_bytes - 64u * nf;
#pragma unroll 1
                    for (uint32_t jj = 0; jj < 2; jj++) {
                        uint4 *dl = reinterpret_cast<uint4 *>(a.leaves + ((uint64_t)(uint32_t)k * a.rows_per_k + 2 * cg + jj) * 8);
                        uint32_t m[16], cv[8];
#pragma unroll
                        for (uint32_t w = 0; w < 16; w++) m[w] = 0;
                        const T *e = x + (size_t)pos * S + (jj << LGT);
#pragma unroll
                        for (uint32_t q = 0; q < S / 2; q++) elem_words<F>(e[q], &m[q * WPE]);  // (lanes past the last column are zero)
                        uint4 lo, hi;
                        if (jj == 0) {
                            lo = r == 0 ? pka0 : pkb0;
                            hi = r == 0 ? pka1 : pkb1;
                        } else {
                            lo = r == 0 ? cva0 : cvb0;
                            hi = r == 0 ? cva1 : cvb1;
                        }
                        cv[0] = lo.x; cv[1] = lo.y; cv[2] = lo.z; cv[3] = lo.w;
                        cv[4] = hi.x; cv[5] = hi.y; cv[6] = hi.z; cv[7] = hi.w;
                        b3::compress(cv, m, 0, 0, blen, (uint32_t)(b3::CHUNK_END | b3::ROOT));
                        dl[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
                        dl[1] = digest_hi(cv[4], cv[5], cv[6], cv[7], a.digest_words);
                    }
                }
            }
        }
        if (!more) {
            sign_off();
            break;
        }
        __syncthreads();  // x is rewritten by the next tile
        cg = cgn;
        s = sn;
        o = on;
        rev_o = rev_on;
    }
}
#undef WF_TILE_LOAD

// ---------------------------------------------------------------------------------------------------------------
// Layout changes between the caller's columns ([col][row][ext coordinate]) and segments.
template <class F>
struct XposeArgs {
    typedef typename F::T T;
    const T *src;
    T *dst;
    uint64_t R;               // rows
    uint32_t W;               // coordinates per column element
    uint32_t total_base_cols;
    uint32_t seg0;            // first segment of the launch (a rank's share of a segment-sharded interpolation)
};

// Both directions stage XPOSE_TILES tiles of (256 / S rows) x (S lanes) through LDS so that the column side is
// accessed in runs of 256 B (f64) / 1 KiB (f128) per column and the segment side in whole 64-byte rows.
// grid.x = n_seg * ceil(R / (XPOSE_TILES * 256 / S))
constexpr uint32_t XPOSE_TILES = 4;

template <class F>
__global__ void __launch_bounds__(256) k_cols_to_seg(XposeArgs<F> a) {
    typedef typename F::T T;
    constexpr uint32_t S = SegCfg<F>::S, RPT = 256 / S, RPB = RPT * XPOSE_TILES;
    __shared__ T tile[XPOSE_TILES][RPT][S + 1];
    const uint64_t blocks_per_seg = (a.R + RPB - 1) / RPB;
    const uint32_t g = a.seg0 + (uint32_t)(blockIdx.x / blocks_per_seg);
    const uint64_t r0 = (blockIdx.x % blocks_per_seg) * RPB;
    {
        const uint32_t rl = threadIdx.x % RPT, l = threadIdx.x / RPT;
        const uint32_t B = g * S + l;
        const bool live = B < a.total_base_cols;
        const uint32_t col = B / a.W, w = B - col * a.W;
#pragma unroll
        for (uint32_t j = 0; j < XPOSE_TILES; j++) {
            const uint64_t r = r0 + j * RPT + rl;
            T v = F::zero();
            if (live && r < a.R) v = a.src[((uint64_t)col * a.R + r) * a.W + w];
            tile[j][rl][l] = v;
        }
    }
    __syncthreads();
    {
        const uint32_t rl = threadIdx.x / S, l = threadIdx.x % S;
#pragma unroll
        for (uint32_t j = 0; j < XPOSE_TILES; j++) {
            const uint64_t r = r0 + j * RPT + rl;
            if (r < a.R) a.dst[((uint64_t)g * a.R + r) * S + l] = tile[j][rl][l];
        }
    }
}

template <class F>
__global__ void __launch_bounds__(256) k_seg_to_cols(XposeArgs<F> a) {
    typedef typename F::T T;
    constexpr uint32_t S = SegCfg<F>::S, RPT = 256 / S, RPB = RPT * XPOSE_TILES;
    __shared__ T tile[XPOSE_TILES][RPT][S + 1];
    const uint64_t blocks_per_seg = (a.R + RPB - 1) / RPB;
    const uint32_t g = a.seg0 + (uint32_t)(blockIdx.x / blocks_per_seg);
    const uint64_t r0 = (blockIdx.x % blocks_per_seg) * RPB;
    {
        const uint32_t rl = threadIdx.x / S, l = threadIdx.x % S;
#pragma unroll
        for (uint32_t j = 0; j < XPOSE_TILES; j++) {
            const uint64_t r = r0 + j * RPT + rl;
            if (r < a.R) tile[j][rl][l] = a.src[((uint64_t)g * a.R + r) * S + l];
        }
    }
    __syncthreads();
    {
        const uint32_t rl = threadIdx.x % RPT, l = threadIdx.x / RPT;
        const uint32_t B = g * S + l;
        if (B >= a.total_base_cols) return;
        const uint32_t col = B / a.W, w = B - col * a.W;
#pragma unroll
        for (uint32_t j = 0; j < XPOSE_TILES; j++) {
            const uint64_t r = r0 + j * RPT + rl;
            if (r < a.R) a.dst[((uint64_t)col * a.R + r) * a.W + w] = tile[j][rl][l];
        }
    }
}

}  // namespace wf
