// FRI layer kernels: the data-parallel pieces of FriProver::build_layer
// (/root/reference/fri/src/prover/mod.rs:191-216): transpose_slice (utils/core/src/lib.rs:206-227) and the
// degree-respecting projection apply_drp (fri/src/folding/mod.rs:85-117).  Row hashing (hash_values,
// fri/src/utils.rs:41-50) and the Merkle tree reuse k_hash_rows / k_merkle_* of kernels.hpp.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "field.hpp"
#include "kernels.hpp"

namespace wf {

// Extension-field product.  f64: x^2 = x - 2 (f64/mod.rs:401-417), x^3 = x + 1 (:446-472); f128: x^2 = x + 1
// (f128/mod.rs:273-279).  Any correct formula gives the reference's values (unique reduced representatives).
template <class F, int W>
struct Ext {
    typedef typename F::T T;
    T c[W];
};

template <class F, int WE>
__device__ __forceinline__ Ext<F, WE> ext_zero() {
    Ext<F, WE> r;
#pragma unroll
    for (int w = 0; w < WE; w++) r.c[w] = F::zero();
    return r;
}
template <class F, int WE>
__device__ __forceinline__ Ext<F, WE> ext_add(Ext<F, WE> a, const Ext<F, WE> &b) {
#pragma unroll
    for (int w = 0; w < WE; w++) a.c[w] = F::add(a.c[w], b.c[w]);
    return a;
}
template <class F, int WE>
__device__ __forceinline__ Ext<F, WE> ext_load(const typename F::T *p) {
    Ext<F, WE> r;
#pragma unroll
    for (int w = 0; w < WE; w++) r.c[w] = p[w];
    return r;
}
template <class F, int WE>
__device__ __forceinline__ void ext_store(typename F::T *p, const Ext<F, WE> &v) {
#pragma unroll
    for (int w = 0; w < WE; w++) p[w] = v.c[w];
}

template <class F, int W>
__host__ __device__ __forceinline__ Ext<F, W> ext_mul(const Ext<F, W> &a, const Ext<F, W> &b) {
    typedef typename F::T T;
    Ext<F, W> r;
    if constexpr (W == 1) {
        r.c[0] = F::mul(a.c[0], b.c[0]);
    } else if constexpr (W == 2) {
        const T a0b0 = F::mul(a.c[0], b.c[0]), a1b1 = F::mul(a.c[1], b.c[1]);
        const T cross = F::sub(F::sub(F::mul(F::add(a.c[0], a.c[1]), F::add(b.c[0], b.c[1])), a0b0), a1b1);  // a0b1+a1b0
        if constexpr (F::FIELD_ID == 1) {  // phi^2 = phi - 2
            r.c[0] = F::sub(a0b0, F::add(a1b1, a1b1));
            r.c[1] = F::add(cross, a1b1);
        } else {  // phi^2 = phi + 1
            r.c[0] = F::add(a0b0, a1b1);
            r.c[1] = F::add(cross, a1b1);
        }
    } else {  // cubic over f64: phi^3 = phi + 1, phi^4 = phi^2 + phi
        T d[5];
        d[0] = F::mul(a.c[0], b.c[0]);
        d[1] = F::add(F::mul(a.c[0], b.c[1]), F::mul(a.c[1], b.c[0]));
        d[2] = F::add(F::add(F::mul(a.c[0], b.c[2]), F::mul(a.c[1], b.c[1])), F::mul(a.c[2], b.c[0]));
        d[3] = F::add(F::mul(a.c[1], b.c[2]), F::mul(a.c[2], b.c[1]));
        d[4] = F::mul(a.c[2], b.c[2]);
        r.c[0] = F::add(d[0], d[3]);
        r.c[1] = F::add(F::add(d[1], d[3]), d[4]);
        r.c[2] = F::add(d[2], d[4]);
    }
    return r;
}

// transpose_slice: out[i][j] = src[i + j * rows]; one thread per element, j fastest: the N threads of a row write its
// N * W elements back to back (stores fully coalesced) and each of the N source streams is read in runs of 64 / N
// elements per wave (with i fastest every 16-byte store went to a different row: 1.4 TB/s)
template <class F>
__global__ void __launch_bounds__(256) k_fri_transpose(const typename F::T *__restrict__ src,
                                                       typename F::T *__restrict__ dst, uint64_t rows, uint32_t N,
                                                       uint32_t W) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * N) return;
    const uint64_t i = idx / N, j = idx - i * N;
    for (uint32_t w = 0; w < W; w++) dst[idx * W + w] = src[(i + j * rows) * W + w];
}

template <class F>
struct DrpArgs {
    typedef typename F::T T;
    const T *values;   // rows x N elements of E
    T *out;            // rows elements of E
    uint64_t rows;
    Pow2L<F> ginv;     // powers of g^-1, g = root of unity of order rows * N
    const T *tw;       // [N] powers of the inverse N-th root
    T sinv;            // domain_offset^-1
    T ninv;            // 1 / N
    T alpha[3];
};

__host__ __device__ constexpr int drp_bitrev(int k, int bits) {
    int r = 0;
    for (int b = 0; b < bits; b++) r |= ((k >> b) & 1) << (bits - 1 - b);
    return r;
}

// apply_drp: per row, interpolate the N values (inverse DFT, coefficient k scaled by (1/N) * (s^-1 g^-i)^k) and
// evaluate the resulting polynomial at alpha (Horner in E).
template <class F, int W, int N>
__global__ void __launch_bounds__(256) k_fri_drp(DrpArgs<F> a) {
    typedef typename F::T T;
    typedef Ext<F, W> E;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.rows) return;
    // the N values of the row in native vector registers (ElemReg): as an array of E they sit in scratch memory for f128 and
    // for the cubic extension (400-784 bytes per thread)
    typedef ElemReg<F> R;
    typename R::type v[N * W];
#pragma unroll
    for (int j = 0; j < N; j++)
#pragma unroll
        for (int w = 0; w < W; w++) v[j * W + w] = R::put(a.values[(i * N + j) * W + w]);
    // radix-2 decimation in frequency with the inverse root: natural in, bit-reversed out (static_for: every index a constant)
    constexpr int LOGN = N == 2 ? 1 : (N == 4 ? 2 : (N == 8 ? 3 : 4));
    static_for<0, LOGN>([&](auto sc) {
        constexpr int s = decltype(sc)::value, half = (N / 2) >> s;
        static_for<0, N / 2>([&](auto bc) {  // butterfly b of the stage: block q, position k
            constexpr int b = decltype(bc)::value, q = (b / half) * 2 * half, k = b % half;
            const T t = a.tw[(k << s) & (N - 1)];
            static_for<0, W>([&](auto wc) {
                constexpr int w = decltype(wc)::value;
                const T u = R::get(v[(q + k) * W + w]), x = R::get(v[(q + k + half) * W + w]);
                v[(q + k) * W + w] = R::put(F::add(u, x));
                T d = F::sub(u, x);
                if (k != 0) d = F::mul(d, t);
                v[(q + k + half) * W + w] = R::put(d);
            });
        });
    });
    // coefficient k sits at bit-reversed position; scale by (1/N) * inv_offset^k and fold with alpha from the top
    const T inv_off = F::mul(a.sinv, a.ginv.get(i));
    typename R::type scale[N];
    scale[0] = R::put(a.ninv);
    static_for<1, N>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        scale[k] = R::put(F::mul(R::get(scale[k - 1]), inv_off));
    });
    E alpha;
#pragma unroll
    for (int w = 0; w < W; w++) alpha.c[w] = a.alpha[w];
    E acc;
#pragma unroll
    for (int w = 0; w < W; w++) acc.c[w] = F::zero();
    static_for<0, N>([&](auto rc) {
        constexpr int k = N - 1 - decltype(rc)::value;
        constexpr int pos = drp_bitrev(k, LOGN);
        acc = ext_mul<F, W>(acc, alpha);
#pragma unroll
        for (int w = 0; w < W; w++) acc.c[w] = F::add(acc.c[w], F::mul(R::get(v[pos * W + w]), R::get(scale[k])));
    });
#pragma unroll
    for (int w = 0; w < W; w++) a.out[i * W + w] = acc.c[w];
}

// Out-of-domain evaluation: ColMatrix::evaluate_columns_at (prover/src/matrix/col_matrix.rs:249-254),
// i.e. polynom::eval of every column at one point z of an extension field (TracePolyTable::get_ood_frame,
// prover/src/trace/poly_table.rs:60-73).  P(z) = sum_k c_k z^k in two launches: work-group (block b, column) takes the
// EVAL_BLOCK coefficients from b * EVAL_BLOCK on -- lane t those at t, t + 256, .. (coalesced), Horner in z^256 -- folds
// its 256 lane values sum_t z^t v_t pairwise through LDS and scales by z^(b * EVAL_BLOCK); k_eval_columns_sum adds the
// block values of a column.  (One work-group per column with a contiguous chunk per lane: 1.35 ms for 8 columns of 2^20.)
constexpr uint32_t EVAL_BLOCK = 4096, EVAL_POINTS = 2;

// powers of a point prepared by the host: pw[s] = z^(2^s) for the exponents the two kernels use
//   s = 0..7   pairwise fold of the 256 lane values of a block          (k_eval_columns_at)
//   s = 8      Horner step of a lane: z^256
//   s = 12..19 pairwise fold of 256 lane values over BLOCKS: Z^(2^(s-12)), Z = z^EVAL_BLOCK   (k_eval_columns_sum)
//   s = 20     Horner step of a lane over blocks: Z^256
constexpr int EVAL_POWERS = 21;

template <class F>
struct EvalAtArgs {
    typedef typename F::T T;
    const T *polys;   // [n_cols] columns of n elements of WC coordinates
    T *partial;       // [points][n_cols][n_blocks] elements of WZ coordinates: sum_k c_(base + k) z^k of every block
    T *out;           // [points][n_cols] elements of WZ coordinates
    uint64_t n;
    uint32_t n_blocks;  // ceil(n / EVAL_BLOCK)
    uint32_t n_cols;
    T pw[EVAL_POINTS][EVAL_POWERS][3];  // up to two points per launch (an out-of-domain frame is z and z g)
};

template <class F, int WZ>
__device__ __forceinline__ Ext<F, WZ> eval_power(const EvalAtArgs<F> &a, uint32_t pt, uint32_t s) {
    Ext<F, WZ> r;
#pragma unroll
    for (int w = 0; w < WZ; w++) r.c[w] = a.pw[pt][s][w];
    return r;
}

// sum_t y^t v_t over the 256 lanes: v_t <- v_2t + y^(2^q) v_2t+1, multipliers a.pw[pt][s0 + q]; result in sh[0]
template <class F, int WZ>
__device__ __forceinline__ void eval_fold(const EvalAtArgs<F> &a, uint32_t pt, uint32_t s0, Ext<F, WZ> *sh, uint32_t t) {
    typedef Ext<F, WZ> E;
    uint32_t q = 0;
#pragma unroll 1
    for (uint32_t width = 128; width >= 1; width >>= 1, q++) {
        E v;
        if (t < width) {
            const E lo = sh[2 * t], hi = sh[2 * t + 1];
            v = ext_mul<F, WZ>(hi, eval_power<F, WZ>(a, pt, s0 + q));
#pragma unroll
            for (int w = 0; w < WZ; w++) v.c[w] = F::add(v.c[w], lo.c[w]);
        }
        __syncthreads();
        if (t < width) sh[t] = v;
        __syncthreads();
    }
}

template <class F, int WC, int WZ>
__global__ void __launch_bounds__(256) k_eval_columns_at(EvalAtArgs<F> a) {
    typedef typename F::T T;
    typedef Ext<F, WZ> E;
    __shared__ __attribute__((aligned(16))) unsigned char sh_raw[256 * sizeof(E)];
    E *sh = reinterpret_cast<E *>(sh_raw);
    const uint32_t blk = blockIdx.x, col = blockIdx.y, pt = blockIdx.z, t = threadIdx.x;
    const T *poly = a.polys + (uint64_t)col * a.n * WC;
    const uint64_t base = (uint64_t)blk * EVAL_BLOCK;
    // every coefficient of this lane is requested before the first one is used (the Horner chain below is serial)
    // (compile-time loops and native register arrays: the f128 instantiations kept `cf` in scratch memory otherwise -- the
    // optimizer declines to unroll a Horner loop of fifteen extension products)
    constexpr int STEPS = EVAL_BLOCK / 256;
    typedef ElemReg<F> R;
    typename R::type cf[STEPS * WC];
    static_for<0, STEPS>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const uint64_t k = base + t + 256u * (uint32_t)j;
#pragma unroll
        for (int w = 0; w < WC; w++) cf[j * WC + w] = R::put(k < a.n ? poly[k * WC + w] : F::zero());
    });
    const E zp = eval_power<F, WZ>(a, pt, 8);  // z^256
    E acc;
#pragma unroll
    for (int w = 0; w < WZ; w++) acc.c[w] = w < WC ? R::get(cf[(STEPS - 1) * WC + (w < WC ? w : 0)]) : F::zero();
    static_for<0, STEPS - 1>([&](auto rc) {
        constexpr int j = STEPS - 2 - decltype(rc)::value;
        acc = ext_mul<F, WZ>(acc, zp);
#pragma unroll
        for (int w = 0; w < WC; w++) acc.c[w] = F::add(acc.c[w], R::get(cf[j * WC + w]));
    });
    sh[t] = acc;
    __syncthreads();
    eval_fold<F, WZ>(a, pt, 0, sh, t);
    if (t == 0) {
        const E r = sh[0];
#pragma unroll
        for (int w = 0; w < WZ; w++) a.partial[(((uint64_t)pt * a.n_cols + col) * a.n_blocks + blk) * WZ + w] = r.c[w];
    }
}

// P(z) = sum_b partial_b Z^b, Z = z^EVAL_BLOCK: the same two-level Horner over the blocks of one column and point
template <class F, int WZ>
__global__ void __launch_bounds__(256) k_eval_columns_sum(EvalAtArgs<F> a) {
    typedef Ext<F, WZ> E;
    __shared__ __attribute__((aligned(16))) unsigned char sh_raw[256 * sizeof(E)];
    E *sh = reinterpret_cast<E *>(sh_raw);
    const uint32_t pt = blockIdx.y, col = pt * a.n_cols + blockIdx.x, t = threadIdx.x;  // (point, column)
    const E step = eval_power<F, WZ>(a, pt, 20);  // Z^256
    E acc;
#pragma unroll
    for (int w = 0; w < WZ; w++) acc.c[w] = F::zero();
    const uint32_t rounds = (a.n_blocks + 255) / 256;
#pragma unroll 1
    for (int j = (int)rounds - 1; j >= 0; j--) {  // lane t: blocks t, t + 256, ..
        const uint32_t blk = t + 256u * (uint32_t)j;
        if (j != (int)rounds - 1) acc = ext_mul<F, WZ>(acc, step);
        if (blk < a.n_blocks) {
#pragma unroll
            for (int w = 0; w < WZ; w++) acc.c[w] = F::add(acc.c[w], a.partial[((uint64_t)col * a.n_blocks + blk) * WZ + w]);
        }
    }
    sh[t] = acc;
    __syncthreads();
    eval_fold<F, WZ>(a, pt, 12, sh, t);
    if (t == 0) {
        const E r = sh[0];
#pragma unroll
        for (int w = 0; w < WZ; w++) a.out[(uint64_t)col * WZ + w] = r.c[w];
    }
}

}  // namespace wf
