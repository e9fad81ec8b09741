// Column-layout NTT kernels: the stand-alone math::fft entry points on ONE column of E (evaluate_poly, interpolate_poly,
// interpolate_poly_with_offset: math/src/fft/mod.rs:85,171,274,362), the offset interpolation of the combined constraint
// column (prover/src/constraints/evaluation_table.rs:180-181) and the FRI remainder.  The commitment path itself runs the
// segment-layout kernels of seg_kernels.hpp; a single column has no neighbouring columns to share twiddles with, so here
// the S lanes of an LDS tile row are ADJACENT INNER POSITIONS of the one column (times its W extension coordinates):
//     tile row d, lane v = t * W + w   <->   element (o, d, i0 + t), coordinate w        t < Tl = S / W
// -- contiguous in memory, one 64-byte run per row (48 bytes for W = 3: two lanes of the row stay zero).  The lanes of a
// row share the digit transform's twiddles exactly as the columns of a segment do, so the in-LDS transform IS
// seg_lds_ntt (radix-16 rounds in registers, shift twiddles over Goldilocks, the fixed-size round sequences); what
// differs from a segment pass is the inter-pass twiddle, which depends on the inner position and is looked up per
// element (two-level table: one product to form it, one to apply it).
//
// NTT of size N = 2^L in 1..4 digit passes, index maps as in seg_kernels.hpp: input n = (n1, .., nP), n1 most
// significant; output k = k1 + N1 k2 + ..; nothing is bit-reverse permuted in memory.
#pragma once

#include "seg_kernels.hpp"

namespace wf {

template <class F>
struct ColArgs {
    typedef typename F::T T;
    const T *src;
    T *dst;
    uint32_t logN, logD;
    uint32_t Tl;         // adjacent inner positions (strided pass) / adjacent k1 rows (last pass) per tile: S / W, or 1
    uint64_t I, O;       // inner / outer counts of the [O][D][I] view (last pass: I = 1)
    uint32_t n_prev;     // last pass: earlier digits, most significant first
    uint32_t prev_log[3];
    uint64_t col_elems;  // elements per column (= N); the grid covers a batch of columns
    Pow2L<F> tw;         // powers of the N-th root of this transform (forward or inverse)
    const T *digit_tw;   // [D] powers of the D-th root
    uint32_t scale_mode; // last pass: SCALE_NONE / SCALE_CONST (x scale) / SCALE_SERIES (output k x out_pow^k)
    T scale;
    Pow2L<F> out_pow;    // SCALE_SERIES: lo table pre-multiplied by 1/n
};

// Strided pass: view [O][D][I] of a column (I contiguous); one work-group transforms the D axis for Tl adjacent inner
// positions and multiplies by the inter-pass twiddle w_(D I)^(k i).  grid.x = batch * O * (I / Tl); blockDim = D / 2.
template <class F, int W, int DIR>
__global__ void __launch_bounds__(1024) k_col_strided(ColArgs<F> a) {
    typedef typename F::T T;
    constexpr uint32_t S = SegCfg<F>::S, LS = S == 8 ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint32_t D = 1u << a.logD, V = a.Tl * W;
    T *x = reinterpret_cast<T *>(smem_raw);
    T *twd = x + (size_t)D * S;

    const uint32_t logI = ilog2_pow2(a.I), logO = ilog2_pow2(a.O), logT = ilog2_pow2(a.Tl);
    const uint64_t bid = blockIdx.x;
    const uint64_t i0 = (bid & ((a.I >> logT) - 1)) << logT;
    const uint64_t o = (bid >> (logI - logT)) & (a.O - 1);
    const uint64_t b = bid >> (logI - logT + logO);
    const T *src = a.src + b * a.col_elems * W;
    T *dst = a.dst + b * a.col_elems * W;

    for (uint32_t e = threadIdx.x; e < D; e += blockDim.x) twd[e] = a.digit_tw[e];
    const uint32_t total = D * S;
    for (uint32_t wk = threadIdx.x; wk < total; wk += blockDim.x) {
        const uint32_t d = wk >> LS, v = wk & (S - 1);
        x[wk] = v < V ? src[((((o << a.logD) + d) << logI) + i0) * W + v] : F::zero();
    }
    __syncthreads();
    seg_lds_ntt<F, DIR, FIX9 | FIX10 | FIX11>(x, twd, a.logD, blockDim.x);
    // store position pos as output digit k, times w_N^(k i N / (D I))
    const uint32_t tw_shift = a.logN - a.logD - logI;
    for (uint32_t wk = threadIdx.x; wk < total; wk += blockDim.x) {
        const uint32_t pos = wk >> LS, v = wk & (S - 1);
        if (v >= V) continue;
        const uint32_t k = seg_digit_reverse<F>(pos, a.logD);
        const uint64_t i = i0 + v / W;
        const uint64_t e = ((uint64_t)k * i) << tw_shift;
        T val = x[wk];
        if (e) val = F::mul(val, a.tw.get(e));
        dst[((((o << a.logD) + k) << logI) + i0) * W + v] = val;
    }
}

// Last pass: view [O][D] (D contiguous); Tl adjacent values of the most significant earlier digit k1 per work-group, so
// that the natural-order outputs form runs of Tl elements.  grid.x = batch * (O / Tl); blockDim = D / 2.
template <class F, int W, int DIR>
__global__ void __launch_bounds__(1024) k_col_last(ColArgs<F> a) {
    typedef typename F::T T;
    constexpr uint32_t S = SegCfg<F>::S, LS = S == 8 ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint32_t D = 1u << a.logD, V = a.Tl * W;
    T *x = reinterpret_cast<T *>(smem_raw);
    T *twd = x + (size_t)D * S;

    const uint32_t log1 = a.n_prev ? a.prev_log[0] : 0, logT = ilog2_pow2(a.Tl), logO = ilog2_pow2(a.O);
    const uint32_t logOlo = logO - log1;  // bits of the remaining earlier digits
    const uint64_t bid = blockIdx.x;
    const uint64_t k1_0 = (bid & (((uint64_t)1 << (log1 - logT)) - 1)) << logT;
    const uint64_t o_rest = (bid >> (log1 - logT)) & (((uint64_t)1 << logOlo) - 1);
    const uint64_t b = bid >> (logO - logT);
    // natural output index contributed by the earlier digits other than k1: o_rest = (k2, k3, ..), k2 most significant
    uint64_t rev_rest = 0;
    {
        uint32_t sh_out = log1, hi = logOlo;
        for (uint32_t q = 1; q < a.n_prev; q++) {
            hi -= a.prev_log[q];
            rev_rest |= ((o_rest >> hi) & (((uint64_t)1 << a.prev_log[q]) - 1)) << sh_out;
            sh_out += a.prev_log[q];
        }
    }
    const T *src = a.src + b * a.col_elems * W;
    T *dst = a.dst + b * a.col_elems * W;

    for (uint32_t e = threadIdx.x; e < D; e += blockDim.x) twd[e] = a.digit_tw[e];
    // line t of the tile = the D * W contiguous values of outer index (k1_0 + t, o_rest); consecutive threads read
    // consecutive values of one line
    const uint32_t line = D * W;
    for (uint32_t t = 0; t < a.Tl; t++) {
        const T *ln = src + ((((k1_0 + t) << logOlo) + o_rest) << a.logD) * W;
        for (uint32_t g = threadIdx.x; g < line; g += blockDim.x) {
            const uint32_t d = g / W, w = g - d * W;
            x[d * S + t * W + w] = ln[g];
        }
    }
    if (V < S) {  // dead lanes (W = 3, or a transform of a single pass: Tl = 1)
        for (uint32_t wk = threadIdx.x; wk < D * S; wk += blockDim.x)
            if ((wk & (S - 1)) >= V) x[wk] = F::zero();
    }
    __syncthreads();
    seg_lds_ntt<F, DIR, FIX9 | FIX10 | FIX11>(x, twd, a.logD, blockDim.x);
    const uint32_t out_shift = a.logN - a.logD, total = D * S;
    for (uint32_t wk = threadIdx.x; wk < total; wk += blockDim.x) {
        const uint32_t pos = wk >> LS, v = wk & (S - 1);
        if (v >= V) continue;
        const uint32_t t = v / W, w = v - t * W;
        const uint64_t k = (k1_0 + t) + rev_rest + ((uint64_t)seg_digit_reverse<F>(pos, a.logD) << out_shift);
        T val = x[wk];
        if (a.scale_mode == SCALE_CONST)
            val = F::mul(val, a.scale);
        else if (a.scale_mode == SCALE_SERIES)
            val = F::mul(val, a.out_pow.get(k));
        dst[k * W + w] = val;
    }
}

}  // namespace wf
