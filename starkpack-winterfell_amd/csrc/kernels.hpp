// Two-level power tables, leaf hashing and Merkle kernels, query gathers (gfx950).  The transforms are in seg_kernels.hpp
// (segment layout: the commitment path) and col_kernels.hpp (one column: the stand-alone math::fft entry points).
// See DESIGN.md.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "blake3_dev.hpp"
#include "field.hpp"

namespace wf {

// Two-level table of powers of one base g: g^e = lo[e & mask] * hi[e >> s].
template <class F>
struct Pow2L {
    const typename F::T *lo;
    const typename F::T *hi;
    uint32_t s;
    uint32_t mask;
    __device__ __forceinline__ typename F::T get(uint64_t e) const {
        typename F::T a = lo[e & mask];
        uint64_t h = e >> s;
        if (h) a = F::mul(a, hi[h]);
        return a;
    }
    // branch-free form for prologues that want all their table reads in flight at once: g^e = a * b (hi[0] = 1)
    __device__ __forceinline__ void fetch(uint64_t e, typename F::T &a, typename F::T &b) const {
        a = lo[e & mask];
        b = hi[e >> s];
    }
};

enum : uint32_t { SCALE_NONE = 0, SCALE_CONST = 1, SCALE_SERIES = 2 };

// Zero fill (n 16-byte words) -- a kernel rather than hipMemsetAsync: memset nodes of a captured graph were seen to
// replay wrongly on this ROCm (tests/test_gpu_graph.py), kernels replay as launched.
static __global__ void __launch_bounds__(256) k_zero16(uint4 *__restrict__ dst, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = make_uint4(0, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// Leaf hashing: leaf j = BLAKE3(canonical LE bytes of row j of trace 0 || row j of trace 1 || ..)
// (RowMatrix::commit_to_comb_rows, prover/src/matrix/row_matrix.rs:204-238; Blake3_256::hash_elements,
// crypto/src/hash/blake/mod.rs:46-59).  One lane per row.
template <class F>
struct HashArgs {
    const typename F::T *lde;   // [n_traces] matrices
    uint64_t trace_elems;       // elements per matrix
    uint64_t n_rows;
    uint32_t row_width;         // elements per stored row
    uint32_t epr;               // elements of a row that are hashed (elements_per_row)
    uint32_t n_traces;
    uint32_t *leaves;           // n_rows * 8 words (32-byte slots)
    uint32_t digest_words;      // 8 (Blake3_256) or 6 (Blake3_192: words 6, 7 of a leaf's slot are written as zeros)
};

// the second half of a digest slot: words 4..7, the last two zero for a 24-byte digest
__device__ __forceinline__ uint4 digest_hi(uint32_t w4, uint32_t w5, uint32_t w6, uint32_t w7, uint32_t digest_words) {
    return digest_words == 6 ? make_uint4(w4, w5, 0u, 0u) : make_uint4(w4, w5, w6, w7);
}

template <class F>
__device__ __forceinline__ void elem_words(typename F::T v, uint32_t *w);
template <>
__device__ __forceinline__ void elem_words<F64>(uint64_t v, uint32_t *w) {
    const uint64_t c = F64::to_canonical(v);  // f64/mod.rs:605-610: canonical as_int(), little endian
    w[0] = (uint32_t)c;
    w[1] = (uint32_t)(c >> 32);
}
template <>
__device__ __forceinline__ void elem_words<F128>(U128 v, uint32_t *w) {
    w[0] = (uint32_t)v.lo;
    w[1] = (uint32_t)(v.lo >> 32);
    w[2] = (uint32_t)v.hi;
    w[3] = (uint32_t)(v.hi >> 32);
}

// One element parked in a native vector register (the same lesson: arrays of U128 that wait through a kernel phase end up
// in scratch memory) and back.
__device__ __forceinline__ uint4 park(uint64_t v) { return make_uint4((uint32_t)v, (uint32_t)(v >> 32), 0u, 0u); }
__device__ __forceinline__ uint4 park(const U128 &v) {
    return make_uint4((uint32_t)v.lo, (uint32_t)(v.lo >> 32), (uint32_t)v.hi, (uint32_t)(v.hi >> 32));
}
template <class F>
__device__ __forceinline__ typename F::T unpark(const uint4 &q);
template <>
__device__ __forceinline__ uint64_t unpark<F64>(const uint4 &q) {
    return ((uint64_t)q.y << 32) | q.x;
}
template <>
__device__ __forceinline__ U128 unpark<F128>(const uint4 &q) {
    return U128{((uint64_t)q.y << 32) | q.x, ((uint64_t)q.w << 32) | q.z};
}

// Compile-time loop: f(std::integral_constant<int, I>) for I = B .. E - 1.  Where `#pragma unroll` is a request the optimizer may
// decline (it does for the larger instantiations of k_fri_drp: their register arrays then turn into scratch memory), this
// cannot fail.
template <int B, int E, class Fn>
__device__ __forceinline__ void static_for(Fn &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// The smallest native register type that holds one element: arrays of it stay in registers where arrays of U128 (or of
// structs of them) are left in scratch memory by the compiler.
template <class F> struct ElemReg;
template <> struct ElemReg<F64> {
    typedef uint2 type;
    static __device__ __forceinline__ uint2 put(uint64_t v) { return make_uint2((uint32_t)v, (uint32_t)(v >> 32)); }
    static __device__ __forceinline__ uint64_t get(const uint2 &q) { return ((uint64_t)q.y << 32) | q.x; }
};
template <> struct ElemReg<F128> {
    typedef uint4 type;
    static __device__ __forceinline__ uint4 put(const U128 &v) { return park(v); }
    static __device__ __forceinline__ U128 get(const uint4 &q) { return unpark<F128>(q); }
};

// Rows of at most one BLAKE3 chunk (1024 bytes); longer rows use k_hash_chunks + k_hash_merge_chunks.
template <class F>
__global__ void __launch_bounds__(256) k_hash_rows(HashArgs<F> a) {
    typedef typename F::T T;
    constexpr uint32_t EPB = 64 / F::BYTES;  // elements per 64-byte block
    constexpr uint32_t WPE = F::BYTES / 4;   // words per element
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.n_rows) return;
    const uint64_t total_elems = (uint64_t)a.n_traces * a.epr;
    const uint64_t len = total_elems * F::BYTES;
    uint32_t out[8];
    if (a.n_traces == 1) {
        const T *row = a.lde + j * a.row_width;
        const uint32_t epr = a.epr;
        auto load = [&](uint64_t bi, uint32_t(&m)[16]) {
            const uint32_t e0 = (uint32_t)bi * EPB;
            // whole 64-byte block inside the stored row (rows are 64-byte aligned, lanes past `epr` are zero by the
            // padding rule, exactly the zero padding BLAKE3 wants): four 16-byte loads
            if (e0 + EPB <= a.row_width) {
                const uint4 *q = reinterpret_cast<const uint4 *>(row + e0);
                T ev[EPB];
                uint4 *dstv = reinterpret_cast<uint4 *>(ev);
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) dstv[i] = q[i];
#pragma unroll
                for (uint32_t i = 0; i < EPB; i++) elem_words<F>(ev[i], &m[i * WPE]);
                return;
            }
#pragma unroll
            for (uint32_t i = 0; i < EPB; i++) {
                if (e0 + i < epr) {
                    elem_words<F>(row[e0 + i], &m[i * WPE]);
                } else {
#pragma unroll
                    for (uint32_t q = 0; q < WPE; q++) m[i * WPE + q] = 0;
                }
            }
        };
        b3::hash_stream(len, load, out);
    } else {
        // rows of several traces back to back; blocks are requested in increasing order, so the (trace, column)
        // position is carried along instead of being recomputed with divisions
        uint32_t t = 0, col = 0;
        const T *rowp = a.lde + j * a.row_width;
        const bool whole_blocks = a.epr % EPB == 0;  // every 64-byte block lies inside one trace's row
        auto load = [&](uint64_t, uint32_t(&m)[16]) {
            if (whole_blocks && t < a.n_traces) {
                const uint4 *q = reinterpret_cast<const uint4 *>(rowp + col);
                T ev[EPB];
                uint4 *dstv = reinterpret_cast<uint4 *>(ev);
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) dstv[i] = q[i];
#pragma unroll
                for (uint32_t i = 0; i < EPB; i++) elem_words<F>(ev[i], &m[i * WPE]);
                col += EPB;
                if (col == a.epr) {
                    col = 0;
                    t++;
                    rowp += a.trace_elems;
                }
                return;
            }
#pragma unroll
            for (uint32_t i = 0; i < EPB; i++) {
                if (t < a.n_traces) {
                    elem_words<F>(rowp[col], &m[i * WPE]);
                    if (++col == a.epr) {
                        col = 0;
                        t++;
                        rowp += a.trace_elems;
                    }
                } else {
#pragma unroll
                    for (uint32_t q = 0; q < WPE; q++) m[i * WPE + q] = 0;
                }
            }
        };
        b3::hash_stream(len, load, out);
    }
    uint4 *dst = reinterpret_cast<uint4 *>(a.leaves + j * 8);
    dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
    dst[1] = digest_hi(out[4], out[5], out[6], out[7], a.digest_words);
}

// Rows longer than one BLAKE3 chunk (1024 bytes: packed traces, wide traces): one lane per (row, chunk) computes the
// chunk chaining values, a second kernel folds each row's chaining values into the leaf.  Keeps the chip busy when
// there are few, long rows (e.g. 512 packed traces of 2^10 steps: 8192 rows of 80 chunks).
template <class F>
__global__ void __launch_bounds__(256) k_hash_chunks(HashArgs<F> a, uint32_t chunks_per_row, uint32_t *cvs) {
    typedef typename F::T T;
    constexpr uint32_t EPB = 64 / F::BYTES, WPE = F::BYTES / 4, EPC = 1024 / F::BYTES;
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_rows * chunks_per_row) return;
    // consecutive lanes take consecutive rows of the same chunk index: coalescing across rows like k_hash_rows
    const uint64_t c = g / a.n_rows, j = g - c * a.n_rows;
    const uint64_t total_elems = (uint64_t)a.n_traces * a.epr;
    const uint64_t e0 = c * EPC;
    const uint32_t celems = (uint32_t)(total_elems - e0 < EPC ? total_elems - e0 : EPC);
    uint32_t t = (uint32_t)(e0 / a.epr), col = (uint32_t)(e0 - (uint64_t)t * a.epr), left = celems;
    const T *rowp = a.lde + (uint64_t)t * a.trace_elems + j * a.row_width;
    auto load = [&](uint32_t, uint32_t(&m)[16]) {
#pragma unroll
        for (uint32_t i = 0; i < EPB; i++) {
            if (left) {
                elem_words<F>(rowp[col], &m[i * WPE]);
                left--;
                if (++col == a.epr) {
                    col = 0;
                    rowp += a.trace_elems;
                }
            } else {
#pragma unroll
                for (uint32_t q = 0; q < WPE; q++) m[i * WPE + q] = 0;
            }
        }
    };
    uint32_t cv[8];
    b3::chunk_cv(c, celems * F::BYTES, load, cv);
    uint4 *dst = reinterpret_cast<uint4 *>(cvs + (j * chunks_per_row + c) * 8);
    dst[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    dst[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}

// The same chunk chaining values with every global access a whole 64-byte piece (round 5).  In k_hash_chunks a lane walks its own
// row: consecutive lanes are a matrix row apart (256 bytes in the reference's example: 512 traces x 10 f128 columns, rows padded
// to 16 elements), so each load instruction touches 64 lines for 16 bytes each and the kernel lives on the L1 keeping those lines
// until the lane comes back for the next element -- which it does not with more than a few waves per CU (measured: 1.63 GB
// fetched for 0.67 GB hashed).  Here a wave owns 16 consecutive rows x 4 chunks (item n: row n & 15, chunk slot n >> 4) and moves
// one 64-byte block per item and step: FOUR lanes load the block of one item (16 bytes each: a 64-byte piece per quad, sixteen
// quads = sixteen consecutive rows of one trace's matrix, i.e. one 4 KiB window per load instruction), the pieces cross a
// wave-private 4 KiB LDS region (units XORed with (n >> 2) & 3: one lane per item reading 64 bytes at a 64-byte stride would
// otherwise hit four banks sixteen times over), and every lane compresses the block of its own item.  Each element is fetched
// once, whatever the L1 does.  Elements are loaded in 16-byte units: epr must be a multiple of 16 / sizeof(element) (always true
// for f128; f64 matrices with an odd number of columns keep k_hash_chunks).
// grid.x * 4 waves >= ceil(n_rows / 16) * ceil(chunks_per_row / 4); consecutive waves take consecutive row groups.
template <class F>
__global__ void __launch_bounds__(256) k_hash_chunks_staged(HashArgs<F> a, uint32_t chunks_per_row, uint32_t *cvs) {
    typedef typename F::T T;
    constexpr uint32_t EPB = 64 / F::BYTES, WPE = F::BYTES / 4, EPC = 1024 / F::BYTES, EPL = 16 / F::BYTES;
    __shared__ uint4 stage[4][256];  // 4 KiB per wave
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint4 *sh = stage[wave];
    const uint64_t row_groups = (a.n_rows + 15) >> 4;
    const uint64_t gw = (uint64_t)blockIdx.x * 4 + wave;
    const uint64_t rg = gw % row_groups, cg = gw / row_groups;
    if (cg * 4 >= chunks_per_row) return;  // (whole wave: no work-group barrier below)
    const uint64_t total_elems = (uint64_t)a.n_traces * a.epr;

    // loader role: quad q = row within the group, unit i of the item's block; one stream per chunk slot k
    const uint32_t q = lane >> 2, i = lane & 3u;
    const uint64_t lrow = rg * 16 + q;
    const T *lp[4];
    uint32_t lcol[4];
    uint32_t lleft[4];  // elements from this lane's next unit to the end of its chunk (0: nothing more to load; at most a chunk's 64 / 128)
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint64_t c = cg * 4 + k;
        const uint64_t e0 = c * EPC + (uint64_t)i * EPL;  // first element of this lane's unit in block 0
        const bool live = lrow < a.n_rows && c < chunks_per_row && e0 < total_elems;
        const uint64_t cend = (c + 1) * EPC < total_elems ? (c + 1) * EPC : total_elems;
        const uint32_t t = live ? (uint32_t)(e0 / a.epr) : 0u;
        lcol[k] = live ? (uint32_t)(e0 - (uint64_t)t * a.epr) : 0u;
        lp[k] = a.lde + (uint64_t)t * a.trace_elems + (live ? lrow : 0) * a.row_width;
        lleft[k] = live ? (uint32_t)(cend - e0) : 0u;
    }
    auto fetch = [&](uint32_t k) -> uint4 {  // this lane's 16 bytes of the current block of slot k, then on to the next block
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (lleft[k]) {
            v = *reinterpret_cast<const uint4 *>(lp[k] + lcol[k]);
            if (EPL == 2 && lleft[k] == 1) v.z = v.w = 0u;  // (f64: the chunk ends on the first element of the unit)
        }
        lleft[k] = lleft[k] > EPB ? lleft[k] - EPB : 0;
        lcol[k] += EPB;
        while (lcol[k] >= a.epr) {  // into the next trace's matrix (same row)
            lcol[k] -= a.epr;
            lp[k] += a.trace_elems;
        }
        return v;
    };
    // hasher role: item = lane
    const uint64_t hrow = rg * 16 + (lane & 15u), hc = cg * 4 + (lane >> 4);
    const bool hlive = hrow < a.n_rows && hc < chunks_per_row;
    const uint64_t he0 = hc * EPC;
    const uint32_t celems = hlive ? (uint32_t)(total_elems - he0 < EPC ? total_elems - he0 : EPC) : 0u;
    const uint32_t clen = celems * F::BYTES, nblocks = (clen + 63) >> 6;
    uint32_t wave_blocks = nblocks;  // the wave steps until its longest item is done
#pragma unroll
    for (uint32_t o = 32; o > 0; o >>= 1) wave_blocks = max(wave_blocks, (uint32_t)__shfl_xor((int)wave_blocks, (int)o));
    const uint32_t swz = (lane >> 2) & 3u;  // of the item this lane hashes
    uint32_t cv[8];
    b3::set_iv(cv);
    uint4 r0 = fetch(0), r1 = fetch(1), r2 = fetch(2), r3 = fetch(3);
    for (uint32_t b = 0; b < wave_blocks; b++) {
        // pieces of item n = q + 16 k: unit i goes to n * 4 + (i ^ ((n >> 2) & 3)); (q + 16 k) >> 2 = (q >> 2) + 4 k: same low bits for every k
        {
            const uint32_t u = i ^ ((q >> 2) & 3u);
            sh[(q + 0) * 4 + u] = r0;
            sh[(q + 16) * 4 + u] = r1;
            sh[(q + 32) * 4 + u] = r2;
            sh[(q + 48) * 4 + u] = r3;
        }
        if (b + 1 < wave_blocks) {  // the next block is on its way while this one is compressed
            r0 = fetch(0);
            r1 = fetch(1);
            r2 = fetch(2);
            r3 = fetch(3);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        T ev[EPB];
        uint4 *evq = reinterpret_cast<uint4 *>(ev);
#pragma unroll
        for (uint32_t w = 0; w < 4; w++) evq[w] = sh[lane * 4 + (w ^ swz)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // (the region is rewritten by the next step)
        if (b < nblocks) {
            uint32_t m[16];
            if (celems - b * EPB >= EPB) {  // a whole block (every block but the last one of a row's last chunk): no per-word selects
#pragma unroll
                for (uint32_t e = 0; e < EPB; e++) elem_words<F>(ev[e], &m[e * WPE]);
            } else {
#pragma unroll
                for (uint32_t e = 0; e < EPB; e++) {
                    if (b * EPB + e < celems) {
                        elem_words<F>(ev[e], &m[e * WPE]);
                    } else {
#pragma unroll
                        for (uint32_t w = 0; w < WPE; w++) m[e * WPE + w] = 0;
                    }
                }
            }
            const uint32_t blen = clen - b * 64 < 64 ? clen - b * 64 : 64;
            const uint32_t flags = (b == 0 ? (uint32_t)b3::CHUNK_START : 0u) | (b == nblocks - 1 ? (uint32_t)b3::CHUNK_END : 0u);
            b3::compress(cv, m, (uint32_t)hc, (uint32_t)(hc >> 32), blen, flags);
        }
    }
    if (hlive) {
        uint4 *dst = reinterpret_cast<uint4 *>(cvs + (hrow * chunks_per_row + hc) * 8);
        dst[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
        dst[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
    }
}

static __global__ void __launch_bounds__(256) k_hash_merge_chunks(const uint32_t *__restrict__ cvs, uint32_t chunks_per_row,
                                                           uint64_t n_rows, uint32_t *__restrict__ leaves, uint32_t digest_words) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_rows) return;
    uint32_t out[8];
    b3::merge_chunk_cvs(cvs + j * chunks_per_row * 8, chunks_per_row, out);
    uint4 *dst = reinterpret_cast<uint4 *>(leaves + j * 8);
    dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
    dst[1] = digest_hi(out[4], out[5], out[6], out[7], digest_words);
}

// The same fold with 16 lanes per row (16 rows per work-group): a row's chaining values sit in LDS and are merged level
// by level in place -- adjacent pairs, an odd last one carried up, which is BLAKE3's left-full tree.  One lane per row
// (above) leaves the chip idle when there are few long rows: 8192 rows of 80 chunks are 79 dependent compressions on 128
// waves.  Dynamic LDS: 16 * n * 32 bytes (the launcher uses this kernel for n <= 128).
static __global__ void __launch_bounds__(256) k_hash_merge_chunks_par(const uint32_t *__restrict__ cvs, uint32_t n, uint64_t n_rows,
                                                               uint32_t *__restrict__ leaves, uint32_t digest_words) {
    extern __shared__ __attribute__((aligned(16))) unsigned char merge_smem[];
    const uint32_t lane = threadIdx.x & 15, r = threadIdx.x >> 4;
    const uint64_t row = (uint64_t)blockIdx.x * 16 + r;
    const bool live = row < n_rows;
    uint4 *my = reinterpret_cast<uint4 *>(merge_smem) + (size_t)r * n * 2;  // chaining value k = my[2k], my[2k + 1]
    if (live) {
        const uint4 *src = reinterpret_cast<const uint4 *>(cvs + row * n * 8);
        for (uint32_t i = lane; i < 2 * n; i += 16) my[i] = src[i];
    }
    __syncthreads();
    for (uint32_t cnt = n; cnt > 1;) {  // cnt is the same for every row: the barriers below are uniform
        const uint32_t pairs = cnt >> 1, odd = cnt & 1;
        for (uint32_t i0 = 0; i0 < pairs; i0 += 16) {
            const uint32_t i = i0 + lane;
            const bool act = live && i < pairs;
            uint32_t m[16], cv[8];
            if (act) {
                const uint4 q0 = my[4 * i], q1 = my[4 * i + 1], q2 = my[4 * i + 2], q3 = my[4 * i + 3];
                m[0] = q0.x; m[1] = q0.y; m[2] = q0.z; m[3] = q0.w;
                m[4] = q1.x; m[5] = q1.y; m[6] = q1.z; m[7] = q1.w;
                m[8] = q2.x; m[9] = q2.y; m[10] = q2.z; m[11] = q2.w;
                m[12] = q3.x; m[13] = q3.y; m[14] = q3.z; m[15] = q3.w;
                b3::set_iv(cv);
                b3::compress(cv, m, 0, 0, 64, b3::PARENT | (cnt == 2 ? (uint32_t)b3::ROOT : 0u));
            }
            __syncthreads();  // a round's outputs (slots i0..i0+15) overlap only its own inputs (2 i0..2 i0+31), and only for i0 = 0
            if (act) {
                my[2 * i] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
                my[2 * i + 1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
            }
        }
        if (odd && live && lane == 0) {  // the unpaired last value moves up unchanged (every pair of this level has been read)
            const uint4 lo = my[2 * (cnt - 1)], hi = my[2 * (cnt - 1) + 1];
            my[2 * pairs] = lo;
            my[2 * pairs + 1] = hi;
        }
        __syncthreads();
        cnt = pairs + odd;
    }
    if (live && lane == 0) {
        uint4 *dst = reinterpret_cast<uint4 *>(leaves + row * 8);
        dst[0] = my[0];
        dst[1] = digest_hi(my[1].x, my[1].y, my[1].z, my[1].w, digest_words);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Merkle levels (build_merkle_nodes, crypto/src/merkle/mod.rs:350-374): each work-group folds 2*blockDim children
// through up to `levels` levels, keeping the intermediate digests in LDS and writing every level to `nodes`.
// children: digests of the level below (n_children of them); the parents level has n_children/2 nodes stored at
// nodes[n_children/2 .. n_children).
template <int DW>
__global__ void __launch_bounds__(256) k_merkle_subtree(const uint32_t *__restrict__ children,
                                                        uint32_t *__restrict__ nodes, uint64_t n_children,
                                                        uint32_t levels) {
    __shared__ uint32_t sh[256 * 8];
    const uint32_t tid = threadIdx.x;
    uint64_t n_par = n_children >> 1;                        // nodes in the first produced level
    uint64_t first = (uint64_t)blockIdx.x * blockDim.x;      // this group's slice of that level
    uint32_t width = (uint32_t)min((uint64_t)blockDim.x, n_par - first);
    uint32_t m[16], cv[8];
    if (tid < width) {
        const uint4 *src = reinterpret_cast<const uint4 *>(children + (first + tid) * 16);
        uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
        m[0] = q0.x; m[1] = q0.y; m[2] = q0.z; m[3] = q0.w;
        m[4] = q1.x; m[5] = q1.y; m[6] = q1.z; m[7] = q1.w;
        m[8] = q2.x; m[9] = q2.y; m[10] = q2.z; m[11] = q2.w;
        m[12] = q3.x; m[13] = q3.y; m[14] = q3.z; m[15] = q3.w;
        b3::merge_slots<DW>(m, cv);
        uint4 *dst = reinterpret_cast<uint4 *>(nodes + (n_par + first + tid) * 8);
        dst[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
        dst[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
#pragma unroll
        for (int i = 0; i < 8; i++) sh[tid * 8 + i] = cv[i];
    }
    for (uint32_t lv = 1; lv < levels; lv++) {
        __syncthreads();
        n_par >>= 1;
        first >>= 1;
        width >>= 1;
        if (width <= 64) {
            // narrow levels: four lanes per node (b3::merge_quad) -- a level is one compression's latency, and a lane of a
            // quad issues a third of the instructions (a 9-level launch 14 -> 9 us)
            const uint32_t node = tid >> 2, q = tid & 3;
            const bool act = node < width;  // whole quads
            uint32_t lo = 0, hi = 0;
            if (act) b3::merge_quad<DW>(sh + node * 16, q, lo, hi);
            __syncthreads();
            if (act) {
                uint32_t *dst = nodes + (n_par + first + node) * 8;
                dst[q] = lo;
                dst[4 + q] = hi;
                sh[node * 8 + q] = lo;
                sh[node * 8 + 4 + q] = hi;
            }
            continue;
        }
        const bool act = tid < width;
        if (act) {
#pragma unroll
            for (int i = 0; i < 16; i++) m[i] = sh[tid * 16 + i];
            b3::merge_slots<DW>(m, cv);
        }
        __syncthreads();
        if (act) {
            uint4 *dst = reinterpret_cast<uint4 *>(nodes + (n_par + first + tid) * 8);
            dst[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
            dst[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
#pragma unroll
            for (int i = 0; i < 8; i++) sh[tid * 8 + i] = cv[i];
        }
    }
    // the launch that produces the root also writes nodes[0] = Digest::default() (merkle/mod.rs:355) -- by a kernel rather
    // than a memset so that a captured graph of the commitment replays it
    if (blockIdx.x == 0 && tid == 0 && (n_children >> levels) == 1) {
        uint4 *dst = reinterpret_cast<uint4 *>(nodes);
        dst[0] = make_uint4(0, 0, 0, 0);
        dst[1] = make_uint4(0, 0, 0, 0);
    }
}

// One Merkle level per launch (used while a level still fills the chip): parents[i] = merge(children[2i], children[2i+1]).
template <int DW>
__global__ void __launch_bounds__(256) k_merkle_level(const uint32_t *__restrict__ children,
                                                      uint32_t *__restrict__ parents, uint64_t n_parents) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parents) return;
    const uint4 *src = reinterpret_cast<const uint4 *>(children + i * 16);
    uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
    uint32_t m[16], cv[8];
    m[0] = q0.x; m[1] = q0.y; m[2] = q0.z; m[3] = q0.w;
    m[4] = q1.x; m[5] = q1.y; m[6] = q1.z; m[7] = q1.w;
    m[8] = q2.x; m[9] = q2.y; m[10] = q2.z; m[11] = q2.w;
    m[12] = q3.x; m[13] = q3.y; m[14] = q3.z; m[15] = q3.w;
    b3::merge_slots<DW>(m, cv);
    uint4 *dst = reinterpret_cast<uint4 *>(parents + i * 8);
    dst[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    dst[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}

// 24-byte digests travel between host arrays (ByteDigest<24>, 24 bytes apart: the reference's Vec<Digest>) and the device's
// 32-byte slots: one 8-byte word per thread.  pack: slots -> 24-byte entries; unpack: the reverse, words 6, 7 zeroed.
static __global__ void __launch_bounds__(256) k_digests_pack24(const uint2 *__restrict__ slots, uint2 *__restrict__ packed, uint64_t n) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= 3 * n) return;
    const uint64_t i = g / 3, w = g - 3 * i;
    packed[g] = slots[4 * i + w];
}
static __global__ void __launch_bounds__(256) k_digests_unpack24(const uint2 *__restrict__ packed, uint2 *__restrict__ slots, uint64_t n) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= 4 * n) return;
    const uint64_t i = g >> 2, w = g & 3;
    slots[g] = w < 3 ? packed[3 * i + w] : make_uint2(0u, 0u);
}

// Query service: gather rows of all traces at the queried positions.  grid = (n positions, n traces)
template <class F>
__global__ void __launch_bounds__(256) k_gather_rows(const typename F::T *__restrict__ lde, uint64_t trace_elems,
                                                     uint32_t row_width, uint32_t epr,
                                                     const uint64_t *__restrict__ positions,
                                                     typename F::T *__restrict__ out) {
    const uint32_t i = blockIdx.x, t = blockIdx.y, n_traces = gridDim.y;
    const typename F::T *row = lde + (uint64_t)t * trace_elems + positions[i] * row_width;
    typename F::T *dst = out + ((uint64_t)i * n_traces + t) * epr;
    for (uint32_t e = threadIdx.x; e < epr; e += blockDim.x) dst[e] = row[e];
}

// gather 32-byte digests: src id = index < n_leaves ? leaves[index] : nodes[index - n_leaves]
static __global__ void __launch_bounds__(256) k_gather_digests(const uint4 *__restrict__ leaves, const uint4 *__restrict__ nodes,
                                                        uint64_t n_leaves, const uint64_t *__restrict__ ids,
                                                        uint32_t n, uint4 *__restrict__ out) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 2 * n) return;
    const uint64_t id = ids[g >> 1];
    const uint4 *src = id < n_leaves ? leaves + 2 * id : nodes + 2 * (id - n_leaves);
    out[g] = src[g & 1];
}

// Two Merkle levels per launch: lane i reads four children (128 contiguous bytes), writes parents 2i, 2i+1 and
// grandparent i.  Halves the launches and the re-reads of the level-per-launch form for the wide levels.
// (six waves per SIMD instead of the seven its 72 VGPRs allow: 0.251 -> 0.240 ms for the 2^23-leaf tree, same box, same code)
template <int DW>
__attribute__((amdgpu_waves_per_eu(6, 6)))
__global__ void __launch_bounds__(256) k_merkle_level2(const uint32_t *__restrict__ children,
                                                       uint32_t *__restrict__ parents,
                                                       uint32_t *__restrict__ grandparents, uint64_t n_grand) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_grand; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 *src = reinterpret_cast<const uint4 *>(children + i * 32);
    uint32_t m[16], cv[8], g[16];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint4 q0 = src[4 * h], q1 = src[4 * h + 1], q2 = src[4 * h + 2], q3 = src[4 * h + 3];
        m[0] = q0.x; m[1] = q0.y; m[2] = q0.z; m[3] = q0.w;
        m[4] = q1.x; m[5] = q1.y; m[6] = q1.z; m[7] = q1.w;
        m[8] = q2.x; m[9] = q2.y; m[10] = q2.z; m[11] = q2.w;
        m[12] = q3.x; m[13] = q3.y; m[14] = q3.z; m[15] = q3.w;
        b3::merge_slots<DW>(m, cv);
        uint4 *dst = reinterpret_cast<uint4 *>(parents + (2 * i + h) * 8);
        dst[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
        dst[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
#pragma unroll
        for (int k = 0; k < 8; k++) g[8 * h + k] = cv[k];
    }
    b3::merge_slots<DW>(g, cv);
    uint4 *dst = reinterpret_cast<uint4 *>(grandparents + i * 8);
    dst[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    dst[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
    }
}

// Hash contiguous rows of `row_elems` elements (wf_hash_rows building block) is k_hash_rows with n_traces = 1.

}  // namespace wf
