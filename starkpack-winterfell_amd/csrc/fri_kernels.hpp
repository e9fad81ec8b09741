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

template <class F, int W>
__device__ __forceinline__ Ext<F, W> ext_mul(const Ext<F, W> &a, const Ext<F, W> &b) {
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

// apply_drp: per row, interpolate the N values (inverse DFT, coefficient k scaled by (1/N) * (s^-1 g^-i)^k) and
// evaluate the resulting polynomial at alpha (Horner in E).
template <class F, int W, int N>
__global__ void __launch_bounds__(256) k_fri_drp(DrpArgs<F> a) {
    typedef typename F::T T;
    typedef Ext<F, W> E;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.rows) return;
    E v[N];
#pragma unroll
    for (int j = 0; j < N; j++)
#pragma unroll
        for (int w = 0; w < W; w++) v[j].c[w] = a.values[(i * N + j) * W + w];
    // radix-2 decimation in frequency with the inverse root: natural in, bit-reversed out
    constexpr int LOGN = N == 2 ? 1 : (N == 4 ? 2 : (N == 8 ? 3 : 4));
#pragma unroll
    for (int s = 0; s < LOGN; s++) {
        const int half = (N / 2) >> s;
#pragma unroll
        for (int q = 0; q < N; q += 2 * half) {
#pragma unroll
            for (int k = 0; k < half; k++) {
                const T t = a.tw[(k << s) & (N - 1)];
#pragma unroll
                for (int w = 0; w < W; w++) {
                    const T u = v[q + k].c[w], x = v[q + k + half].c[w];
                    v[q + k].c[w] = F::add(u, x);
                    T d = F::sub(u, x);
                    if (k != 0) d = F::mul(d, t);
                    v[q + k + half].c[w] = d;
                }
            }
        }
    }
    // coefficient k sits at bit-reversed position; scale by (1/N) * inv_offset^k and fold with alpha from the top
    const T inv_off = F::mul(a.sinv, a.ginv.get(i));
    T scale[N];
    scale[0] = a.ninv;
#pragma unroll
    for (int k = 1; k < N; k++) scale[k] = F::mul(scale[k - 1], inv_off);
    E alpha;
#pragma unroll
    for (int w = 0; w < W; w++) alpha.c[w] = a.alpha[w];
    E acc;
#pragma unroll
    for (int w = 0; w < W; w++) acc.c[w] = F::zero();
#pragma unroll
    for (int k = N - 1; k >= 0; k--) {
        int pos = 0;
#pragma unroll
        for (int b = 0; b < LOGN; b++) pos |= ((k >> b) & 1) << (LOGN - 1 - b);
        acc = ext_mul<F, W>(acc, alpha);
#pragma unroll
        for (int w = 0; w < W; w++) acc.c[w] = F::add(acc.c[w], F::mul(v[pos].c[w], scale[k]));
    }
#pragma unroll
    for (int w = 0; w < W; w++) a.out[i * W + w] = acc.c[w];
}

// Out-of-domain evaluation: ColMatrix::evaluate_columns_at (prover/src/matrix/col_matrix.rs:249-254),
// i.e. polynom::eval of every column at one point z of an extension field (TracePolyTable::get_ood_frame,
// prover/src/trace/poly_table.rs:60-73).  One work-group per column: every lane runs Horner over a contiguous chunk,
// lane 0 then folds the partial values with z^chunk.
template <class F>
struct EvalAtArgs {
    typedef typename F::T T;
    const T *polys;   // [n_cols] columns of n elements of WC coordinates
    T *out;           // [n_cols] elements of WZ coordinates
    uint64_t n;
    T z[3];
};

template <class F, int WC, int WZ>
__global__ void __launch_bounds__(256) k_eval_columns_at(EvalAtArgs<F> a) {
    typedef typename F::T T;
    typedef Ext<F, WZ> E;
    __shared__ __attribute__((aligned(16))) unsigned char sh_raw[256 * sizeof(E)];
    E *partial = reinterpret_cast<E *>(sh_raw);
    const T *poly = a.polys + (uint64_t)blockIdx.x * a.n * WC;
    const uint32_t nt = (uint32_t)(a.n < 256 ? a.n : 256);
    const uint64_t chunk = a.n / nt;  // n and nt are powers of two
    E z;
#pragma unroll
    for (int w = 0; w < WZ; w++) z.c[w] = a.z[w];
    if (threadIdx.x < nt) {
        const uint64_t k0 = (uint64_t)threadIdx.x * chunk;
        E acc;
#pragma unroll
        for (int w = 0; w < WZ; w++) acc.c[w] = F::zero();
        for (uint64_t k = k0 + chunk; k-- > k0;) {
            acc = ext_mul<F, WZ>(acc, z);
#pragma unroll
            for (int w = 0; w < WC; w++) acc.c[w] = F::add(acc.c[w], poly[k * WC + w]);
        }
        partial[threadIdx.x] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        E zc = z;  // z^chunk by squaring (chunk is a power of two)
        for (uint64_t c = 1; c < chunk; c <<= 1) zc = ext_mul<F, WZ>(zc, zc);
        E acc = partial[nt - 1];
        for (int t = (int)nt - 2; t >= 0; t--) {
            acc = ext_mul<F, WZ>(acc, zc);
#pragma unroll
            for (int w = 0; w < WZ; w++) acc.c[w] = F::add(acc.c[w], partial[t].c[w]);
        }
#pragma unroll
        for (int w = 0; w < WZ; w++) a.out[(uint64_t)blockIdx.x * WZ + w] = acc.c[w];
    }
}

}  // namespace wf
