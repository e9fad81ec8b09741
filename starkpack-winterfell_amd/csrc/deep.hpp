// DEEP composition polynomial on the polynomials that resident commitments keep in HBM -- the caller that sits between
// the out-of-domain evaluation and the DEEP LDE + FRI (§8f-2, §8f-1):
// DeepCompositionPoly::add_trace_polys + add_composition_poly (/root/reference/prover/src/composer/mod.rs:62-193).
//
// What the reference computes, as one formula.  With A(x) = sum_i cc_i T_i(x) over every column of every packed trace
// and C(x) = sum_i cc'_i H_i(x) over the constraint composition columns, each accumulator gets its out-of-domain value
// subtracted from coefficient 0 (acc_trace_poly :227-236, `poly[0] -= value_at_z` :182) and is divided by (x - z) or
// (x - z g) with polynom::syn_div_in_place (math/src/polynom/mod.rs:535-542), which drops the remainder:
//     q_i = sum_{j > i} p_j b^(j - i - 1).
// No q_i reads p_0, so the subtracted values only ever change the dropped remainder, and since the quotient is linear
//     D_i = sum_{j > i} (A_j + C_j) z^(j - i - 1)  +  sum_{j > i} A_j (z g)^(j - i - 1),      D_{n-1} = 0.
// Field arithmetic is exact, so any order of evaluation gives the reference's coefficients bit for bit; the parity tests
// compare with a CPU restatement that follows the reference's order literally, subtraction included.
//
// Kernels: k_deep_combine (the two linear combinations, one thread per coefficient index, column chunks in grid.y so that
// thousands of short columns still fill the chip), k_deep_reduce (chunks summed), then the two suffix recurrences as a
// blocked scan: k_deep_scan<0> (per-block Horner sums), k_deep_carry (carries into the blocks), k_deep_scan<1> (the
// recurrence inside a block, both quotients added).  HBM-bound on the columns: one read of every coefficient.
#pragma once

namespace wf {

#ifndef WF_DEEP_RUN_LOG
#define WF_DEEP_RUN_LOG 2
#endif
constexpr uint32_t DEEP_RUN_LOG = WF_DEEP_RUN_LOG, DEEP_RUN = 1u << DEEP_RUN_LOG, DEEP_BLOCK = 256 * DEEP_RUN;

// The column table, sorted by the host (a sum does not care about the order): base-field trace columns, then trace
// columns over E, then the constraint composition columns (over E); coeffs[] follows the same order.
struct DeepTable {
    const void *const *ptrs;  // [n_cols]
    uint32_t n_base;          // [0, n_base): one coordinate per coefficient, accumulate into A
    uint32_t n_trace;         // [n_base, n_trace): E coordinates, accumulate into A; [n_trace, n_cols): into C
    uint32_t n_cols;
};

// Two consecutive coefficient indices per thread (16-byte loads of base-field columns), columns four at a time so that
// their loads are in flight together.  partial[a][chunk][k] = sum over the chunk's columns of cc_col * col[k]
// (math::mul_acc, utils/mod.rs:143-153: c.mul_base(b) for base columns, the full product for columns over E).
template <class F, int WE>
__global__ void __launch_bounds__(256) k_deep_combine(DeepTable tab, const typename F::T *__restrict__ coeffs,
                                                      uint32_t cols_per_chunk, uint32_t n_chunks, uint64_t n,
                                                      typename F::T *__restrict__ partial) {
    typedef typename F::T T;
    typedef Ext<F, WE> E;
    const uint64_t k = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 2;  // n is even
    const uint32_t chunk = blockIdx.y;
    if (k >= n) return;
    const uint32_t c0 = chunk * cols_per_chunk, c1 = min(tab.n_cols, c0 + cols_per_chunk);
    E acc[2][2];  // [A / C][k, k + 1]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[a][j] = ext_zero<F, WE>();
    // base-field columns (all of them when the composition is over the base field itself)
    const uint32_t b1 = WE == 1 ? c1 : min(c1, tab.n_base);
#pragma unroll 4
    for (uint32_t c = c0; c < b1; c++) {
        const Pair<T> v = *reinterpret_cast<const Pair<T> *>(reinterpret_cast<const T *>(tab.ptrs[c]) + k);
        const E cc = ext_load<F, WE>(coeffs + (size_t)c * WE);
        E t0, t1;
#pragma unroll
        for (int w = 0; w < WE; w++) {
            t0.c[w] = F::mul(cc.c[w], v.a);
            t1.c[w] = F::mul(cc.c[w], v.b);
        }
        if (WE == 1 && c >= tab.n_trace) {
            acc[1][0] = ext_add<F, WE>(acc[1][0], t0);
            acc[1][1] = ext_add<F, WE>(acc[1][1], t1);
        } else {
            acc[0][0] = ext_add<F, WE>(acc[0][0], t0);
            acc[0][1] = ext_add<F, WE>(acc[0][1], t1);
        }
    }
    if (WE > 1) {
#pragma unroll 2
        for (uint32_t c = max(c0, tab.n_base); c < c1; c++) {
            const T *p = reinterpret_cast<const T *>(tab.ptrs[c]) + k * WE;
            const E v0 = ext_load<F, WE>(p), v1 = ext_load<F, WE>(p + WE);
            const E cc = ext_load<F, WE>(coeffs + (size_t)c * WE);
            const E t0 = ext_mul<F, WE>(cc, v0), t1 = ext_mul<F, WE>(cc, v1);
            if (c >= tab.n_trace) {
                acc[1][0] = ext_add<F, WE>(acc[1][0], t0);
                acc[1][1] = ext_add<F, WE>(acc[1][1], t1);
            } else {
                acc[0][0] = ext_add<F, WE>(acc[0][0], t0);
                acc[0][1] = ext_add<F, WE>(acc[0][1], t1);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 2; a++) {
        T *dst = partial + (((size_t)a * n_chunks + chunk) * n + k) * WE;
        ext_store<F, WE>(dst, acc[a][0]);
        ext_store<F, WE>(dst + WE, acc[a][1]);
    }
}

// partial[a][0][k] = sum over the chunks; a block = 32 consecutive k x 8 lanes that each sum every eighth chunk
template <class F, int WE>
__global__ void __launch_bounds__(256) k_deep_reduce(typename F::T *__restrict__ partial, uint32_t n_chunks, uint64_t n) {
    typedef Ext<F, WE> E;
    __shared__ __attribute__((aligned(16))) unsigned char sh_raw[256 * sizeof(E)];
    E *sh = reinterpret_cast<E *>(sh_raw);
    const uint32_t t = threadIdx.x, kk = t & 31, cl = t >> 5, a = blockIdx.y;
    const uint64_t k = (uint64_t)blockIdx.x * 32 + kk;
    E s = ext_zero<F, WE>();
    if (k < n) {
#pragma unroll 4
        for (uint32_t c = cl; c < n_chunks; c += 8) s = ext_add<F, WE>(s, ext_load<F, WE>(partial + (((size_t)a * n_chunks + c) * n + k) * WE));
    }
    sh[t] = s;
    __syncthreads();
    if (cl == 0 && k < n) {
#pragma unroll
        for (uint32_t j = 1; j < 8; j++) s = ext_add<F, WE>(s, sh[j * 32 + kk]);
        ext_store<F, WE>(partial + (((size_t)a * n_chunks) * n + k) * WE, s);
    }
}

template <class F>
struct DeepScanArgs {
    typedef typename F::T T;
    const T *acc_a, *acc_c;  // [n] elements of E: A and C
    T *sums;                 // [2][nb] elements of E: S_b = sum_{j in block b} P_j y^(j - block start)
    T *carry;                // [2][nb]: R_b = sum_{j >= end of block b} P_j y^(j - end)
    T *out;                  // [n] elements of E
    uint64_t n;
    uint32_t nb;             // ceil(n / DEEP_BLOCK)
    uint32_t L;              // blocks per thread of k_deep_carry: ceil(nb / 256)
    // stream 0 scans A + C with y = z, stream 1 scans A with y = z g; powers prepared by the host:
    T y[2][3];
    T run_pow[2][8][3];      // y^(DEEP_RUN 2^d): the multipliers of the thread-level scan in k_deep_scan
    T block_pow[2][3];       // Y = y^DEEP_BLOCK
    T carry_pow[2][8][3];    // (Y^L)^(2^d): the same for k_deep_carry
};

template <class F, int WE>
__device__ __forceinline__ Ext<F, WE> ext_from(const typename F::T (&v)[3]) {
    Ext<F, WE> r;
#pragma unroll
    for (int w = 0; w < WE; w++) r.c[w] = v[w];
    return r;
}

// I_t = sum_{u >= t} v_u m^(u - t) over the 256 threads of the block; pw[d] = m^(2^d).  sh: 2 x 256 elements (the
// two halves alternate, so one barrier per step is enough).
template <class F, int WE>
__device__ __forceinline__ Ext<F, WE> deep_suffix_scan(Ext<F, WE> v, const typename F::T (&pw)[8][3], Ext<F, WE> *sh, uint32_t t) {
#pragma unroll 1
    for (uint32_t q = 0; q < 8; q++) {
        Ext<F, WE> *buf = sh + (q & 1) * 256;
        const uint32_t d = 1u << q;
        buf[t] = v;
        __syncthreads();
        if (t + d < 256) v = ext_add<F, WE>(v, ext_mul<F, WE>(buf[t + d], ext_from<F, WE>(pw[q])));
    }
    __syncthreads();  // the caller may reuse sh
    return v;
}

// MODE 0: grid (nb, 2): block sums of one stream.  MODE 1: grid (nb): out_i = q_i(stream 0) + q_i(stream 1).
template <class F, int WE, int MODE>
__global__ void __launch_bounds__(256) k_deep_scan(DeepScanArgs<F> a) {
    typedef Ext<F, WE> E;
    __shared__ __attribute__((aligned(16))) unsigned char sh_raw[2 * 256 * sizeof(E)];
    E *sh = reinterpret_cast<E *>(sh_raw);
    const uint32_t t = threadIdx.x, b = blockIdx.x;
    const uint64_t k0 = (uint64_t)b * DEEP_BLOCK + (uint64_t)t * DEEP_RUN;
    E outv[DEEP_RUN];
#pragma unroll
    for (uint32_t j = 0; j < DEEP_RUN; j++) outv[j] = ext_zero<F, WE>();
    const uint32_t s_begin = MODE == 0 ? blockIdx.y : 0, s_end = MODE == 0 ? blockIdx.y + 1 : 2;
    E pa[DEEP_RUN];  // A_j of this thread's run (both streams read it)
#pragma unroll
    for (uint32_t j = 0; j < DEEP_RUN; j++) pa[j] = k0 + j < a.n ? ext_load<F, WE>(a.acc_a + (k0 + j) * WE) : ext_zero<F, WE>();
#pragma unroll 1
    for (uint32_t s = s_begin; s < s_end; s++) {
        const E y = ext_from<F, WE>(a.y[s]);
        E p[DEEP_RUN];
#pragma unroll
        for (uint32_t j = 0; j < DEEP_RUN; j++) {
            p[j] = pa[j];
            if (s == 0 && k0 + j < a.n) p[j] = ext_add<F, WE>(p[j], ext_load<F, WE>(a.acc_c + (k0 + j) * WE));
        }
        E sr = p[DEEP_RUN - 1];  // this thread's run: sum_j p_j y^j
#pragma unroll
        for (int j = (int)DEEP_RUN - 2; j >= 0; j--) sr = ext_add<F, WE>(ext_mul<F, WE>(sr, y), p[j]);
        E R = ext_zero<F, WE>();
        if (MODE == 1) {
            R = ext_load<F, WE>(a.carry + ((size_t)s * a.nb + b) * WE);
            // the carry into the block rides on the last run: s'_255 = s_255 + y^RUN R
            if (t == 255) sr = ext_add<F, WE>(sr, ext_mul<F, WE>(ext_from<F, WE>(a.run_pow[s][0]), R));
        }
        const E I = deep_suffix_scan<F, WE>(sr, a.run_pow[s], sh, t);
        if (MODE == 0) {
            if (t == 0) ext_store<F, WE>(a.sums + ((size_t)s * a.nb + b) * WE, I);
        } else {
            sh[t] = I;
            __syncthreads();
            E c = t < 255 ? sh[t + 1] : R;  // q of the run's last index
            __syncthreads();
#pragma unroll
            for (int j = (int)DEEP_RUN - 1; j >= 0; j--) {
                outv[j] = ext_add<F, WE>(outv[j], c);            // q_i = c ...
                c = ext_add<F, WE>(ext_mul<F, WE>(c, y), p[j]);  // ... q_(i-1) = p_i + y q_i   (syn_div_in_place)
            }
        }
    }
    if (MODE == 1) {
#pragma unroll
        for (uint32_t j = 0; j < DEEP_RUN; j++)
            if (k0 + j < a.n) ext_store<F, WE>(a.out + (k0 + j) * WE, outv[j]);
    }
}

// carry[s][b] = sum_{b' > b} sums[s][b'] Y^(b' - b - 1), Y = y^DEEP_BLOCK; grid (2), thread t owns L consecutive blocks
template <class F, int WE>
__global__ void __launch_bounds__(256) k_deep_carry(DeepScanArgs<F> a) {
    typedef typename F::T T;
    typedef Ext<F, WE> E;
    __shared__ __attribute__((aligned(16))) unsigned char sh_raw[2 * 256 * sizeof(E)];
    E *sh = reinterpret_cast<E *>(sh_raw);
    const uint32_t t = threadIdx.x, s = blockIdx.x, L = a.L;
    const E Y = ext_from<F, WE>(a.block_pow[s]);
    const T *S = a.sums + (size_t)s * a.nb * WE;
    T *R = a.carry + (size_t)s * a.nb * WE;
    E sr = ext_zero<F, WE>();
#pragma unroll 1
    for (int j = (int)L - 1; j >= 0; j--) {
        const uint64_t blk = (uint64_t)t * L + (uint32_t)j;
        sr = ext_mul<F, WE>(sr, Y);
        if (blk < a.nb) sr = ext_add<F, WE>(sr, ext_load<F, WE>(S + blk * WE));
    }
    const E I = deep_suffix_scan<F, WE>(sr, a.carry_pow[s], sh, t);
    sh[t] = I;
    __syncthreads();
    E c = t < 255 ? sh[t + 1] : ext_zero<F, WE>();
#pragma unroll 1
    for (int j = (int)L - 1; j >= 0; j--) {
        const uint64_t blk = (uint64_t)t * L + (uint32_t)j;
        if (blk < a.nb) {
            ext_store<F, WE>(R + blk * WE, c);
            c = ext_add<F, WE>(ext_mul<F, WE>(c, Y), ext_load<F, WE>(S + blk * WE));
        } else {
            c = ext_mul<F, WE>(c, Y);
        }
    }
}

// --------------------------------------------------------------------------------------------------------- host side
struct DeepColumn {  // one column as the entry point collects it
    const void *ptr;
    uint32_t wc;   // coordinates per coefficient
    uint32_t acc;  // 0: trace column (A), 1: constraint composition column (C)
    size_t coeff;  // index of its coefficient in the caller's order (trace coefficients, then constraint coefficients)
};

template <class F, int WE>
static Ext<F, WE> ext_pow2k(Ext<F, WE> v, uint32_t k) {  // v^(2^k)
    for (uint32_t i = 0; i < k; i++) v = ext_mul<F, WE>(v, v);
    return v;
}

template <class F, int WE>
static int deep_launch(wf_ctx *ctx, hipStream_t st, const DeepTable &tab, const typename F::T *d_coeffs, uint64_t n,
                       const typename F::T *z, typename F::T *d_out) {
    typedef typename F::T T;
    typedef Ext<F, WE> E;
    const uint32_t kblocks = (uint32_t)((n / 2 + 255) / 256), n_cols = tab.n_cols;
    uint32_t n_chunks = 1;
    if (kblocks < 1024) {  // short polynomials: split the columns over grid.y until ~2048 work-groups exist
        n_chunks = (2048 + kblocks - 1) / kblocks;
        const uint32_t most = (n_cols + 7) / 8;  // at least eight columns per chunk
        if (n_chunks > most) n_chunks = most;
        if (n_chunks < 1) n_chunks = 1;
    }
    const uint32_t cpc = (n_cols + n_chunks - 1) / n_chunks;
    n_chunks = (n_cols + cpc - 1) / cpc;
    const size_t eb = (size_t)WE * sizeof(T);
    int rc;
    if ((rc = ensure(ctx, ctx->scratch, (size_t)2 * n_chunks * n * eb))) return rc;
    DeepScanArgs<F> a;
    memset(&a, 0, sizeof(a));
    a.n = n;
    a.nb = (uint32_t)((n + DEEP_BLOCK - 1) / DEEP_BLOCK);
    a.L = (a.nb + 255) / 256;
    if ((rc = ensure(ctx, ctx->io[4], (size_t)4 * a.nb * eb))) return rc;
    T *partial = (T *)ctx->scratch.p;
    a.acc_a = partial;
    a.acc_c = partial + (size_t)n_chunks * n * WE;
    a.sums = (T *)ctx->io[4].p;
    a.carry = a.sums + (size_t)2 * a.nb * WE;
    a.out = d_out;
    // y[0] = z, y[1] = z * g with g the generator of the trace domain (composer/mod.rs:77-79), and their powers
    uint32_t logn = 0;
    while (((uint64_t)1 << logn) < n) logn++;
    const T g = f_root_of_unity<F>(logn);
    for (int s = 0; s < 2; s++) {
        E y;
        for (int w = 0; w < WE; w++) y.c[w] = s == 0 ? z[w] : F::mul(z[w], g);
        E rp = ext_pow2k<F, WE>(y, DEEP_RUN_LOG);          // y^DEEP_RUN
        const E Y = ext_pow2k<F, WE>(y, 8 + DEEP_RUN_LOG);  // y^DEEP_BLOCK
        E cp;                                      // Y^L by square and multiply
        for (int w = 0; w < WE; w++) cp.c[w] = w == 0 ? F::one() : F::zero();
        E base = Y;
        for (uint32_t e = a.L; e; e >>= 1) {
            if (e & 1) cp = ext_mul<F, WE>(cp, base);
            base = ext_mul<F, WE>(base, base);
        }
        for (int w = 0; w < WE; w++) {
            a.y[s][w] = y.c[w];
            a.block_pow[s][w] = Y.c[w];
        }
        for (int d = 0; d < 8; d++) {
            for (int w = 0; w < WE; w++) {
                a.run_pow[s][d][w] = rp.c[w];
                a.carry_pow[s][d][w] = cp.c[w];
            }
            rp = ext_mul<F, WE>(rp, rp);
            cp = ext_mul<F, WE>(cp, cp);
        }
    }
    prof_mark(ctx, st, "deep.combine");
    hipLaunchKernelGGL((k_deep_combine<F, WE>), dim3(kblocks, n_chunks), dim3(256), 0, st, tab, d_coeffs, cpc, n_chunks, n, partial);
    HIP_TRY(hipGetLastError());
    if (n_chunks > 1) {
        prof_mark(ctx, st, "deep.reduce_chunks");
        hipLaunchKernelGGL((k_deep_reduce<F, WE>), dim3((uint32_t)((n + 31) / 32), 2), dim3(256), 0, st, partial, n_chunks, n);
        HIP_TRY(hipGetLastError());
    }
    prof_mark(ctx, st, "deep.block_sums");
    hipLaunchKernelGGL((k_deep_scan<F, WE, 0>), dim3(a.nb, 2), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    prof_mark(ctx, st, "deep.carry");
    hipLaunchKernelGGL((k_deep_carry<F, WE>), dim3(2), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    prof_mark(ctx, st, "deep.scan");
    hipLaunchKernelGGL((k_deep_scan<F, WE, 1>), dim3(a.nb), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    prof_mark(ctx, st, "between_calls");
    return 0;
}

// `staging` (host, owned by the caller until the stream has been synchronised): the sorted pointer table followed by
// the coefficients in the same order
template <class F>
static int deep_compose_dev(wf_ctx *ctx, hipStream_t st, const std::vector<DeepColumn> &cols, const void *trace_coeffs,
                            const void *constraint_coeffs, size_t n_trace_cols, uint32_t ext, uint64_t n, const void *z_host,
                            void *d_out, std::vector<unsigned char> &staging) {
    typedef typename F::T T;
    const size_t n_cols = cols.size(), cb = (size_t)ext * sizeof(T);
    const T *z = (const T *)z_host;
    bool z_zero = true;
    for (uint32_t w = 0; w < ext; w++) {
        if (!F::is_valid(z[w])) return fail(WF_ERR_ARG, "z is not a valid field element");
        const T zero = F::zero();
        if (memcmp(&z[w], &zero, sizeof(T)) != 0) z_zero = false;
    }
    if (z_zero) return fail(WF_ERR_ARG, "z is zero (syn_div_in_place: \"constant cannot be zero\")");  // polynom/mod.rs:529
    // order: base-field trace columns, trace columns over E, constraint columns (stable within each class)
    std::vector<const DeepColumn *> order;
    order.reserve(n_cols);
    DeepTable tab;
    for (const DeepColumn &c : cols)
        if (c.acc == 0 && (c.wc == 1 && ext > 1)) order.push_back(&c);
    if (ext == 1)
        for (const DeepColumn &c : cols)
            if (c.acc == 0) order.push_back(&c);
    tab.n_base = (uint32_t)order.size();
    if (ext > 1)
        for (const DeepColumn &c : cols)
            if (c.acc == 0 && c.wc != 1) order.push_back(&c);
    tab.n_trace = (uint32_t)order.size();
    for (const DeepColumn &c : cols)
        if (c.acc == 1) order.push_back(&c);
    tab.n_cols = (uint32_t)order.size();
    const size_t table_bytes = (n_cols * sizeof(void *) + 15) & ~(size_t)15;
    staging.resize(table_bytes + n_cols * cb);
    const void **ptrs = reinterpret_cast<const void **>(staging.data());
    unsigned char *cc = staging.data() + table_bytes;
    for (size_t i = 0; i < n_cols; i++) {
        const DeepColumn &c = *order[i];
        ptrs[i] = c.ptr;
        const unsigned char *src = c.acc == 0 ? (const unsigned char *)trace_coeffs + c.coeff * cb
                                              : (const unsigned char *)constraint_coeffs + (c.coeff - n_trace_cols) * cb;
        memcpy(cc + i * cb, src, cb);
        const T *v = reinterpret_cast<const T *>(cc + i * cb);
        for (uint32_t w = 0; w < ext; w++)
            if (!F::is_valid(v[w])) return fail(WF_ERR_ARG, "coefficient %zu is not a valid field element", c.coeff);
    }
    int rc = ensure(ctx, ctx->io[3], staging.size());
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->io[3].p, staging.data(), staging.size(), hipMemcpyHostToDevice, st));
    tab.ptrs = (const void *const *)ctx->io[3].p;
    const T *d_cc = (const T *)((char *)ctx->io[3].p + table_bytes);
    switch (ext) {
        case 1: return deep_launch<F, 1>(ctx, st, tab, d_cc, n, z, (T *)d_out);
        case 2: return deep_launch<F, 2>(ctx, st, tab, d_cc, n, z, (T *)d_out);
        case 3:
            if constexpr (F::FIELD_ID == 1) return deep_launch<F, 3>(ctx, st, tab, d_cc, n, z, (T *)d_out);
            return fail(WF_ERR_EXTENSION, "f128 has no cubic extension");
        default: return fail(WF_ERR_EXTENSION, "unsupported extension degree %u", ext);
    }
}

}  // namespace wf

using namespace wf;

extern "C" {

int wf_deep_compose(wf_ctx *ctx, const wf_commitment *const *trace_commitments, size_t n_trace_commitments,
                    const wf_commitment *constraint_commitment, const void *z, uint32_t ext_degree, const void *trace_coeffs,
                    const void *constraint_coeffs, void *poly_out, wf_fri_prover *fri, size_t lde_blowup) {
    if (!ctx) return fail(WF_ERR_ARG, "ctx is null");
    if (!trace_commitments || n_trace_commitments == 0) return fail(WF_ERR_ARG, "no trace commitments");
    if (!z || !trace_coeffs) return fail(WF_ERR_ARG, "null argument");
    if (!poly_out && !fri) return fail(WF_ERR_ARG, "neither poly_out nor a FRI prover to take the polynomial");
    if (constraint_commitment && !constraint_coeffs) return fail(WF_ERR_ARG, "constraint_coeffs is null");
    const wf_commitment *first = trace_commitments[0];
    if (!first) return fail(WF_ERR_ARG, "trace commitment 0 is null");
    const uint32_t field = first->p.field, logn = first->p.log2_trace_len;
    if (ext_degree < 1 || ext_degree > 3 || (field == WF_FIELD_F128 && ext_degree == 3))
        return fail(WF_ERR_EXTENSION, "unsupported extension degree %u", ext_degree);
    std::vector<DeepColumn> cols;
    const size_t eb = wf_elem_bytes(field), n = (size_t)1 << logn;
    auto add = [&](const wf_commitment *c, uint32_t acc, const char *what, size_t idx) -> int {
        if (!c) return fail(WF_ERR_ARG, "%s commitment %zu is null", what, idx);
        if (c->ctx != ctx) return fail(WF_ERR_ARG, "%s commitment %zu belongs to another context", what, idx);
        if (!c->polys) return fail(WF_ERR_ARG, "%s commitment %zu holds no polynomials", what, idx);
        if (c->p.field != field) return fail(WF_ERR_FIELD, "%s commitment %zu is over another field", what, idx);
        if (c->p.log2_trace_len != logn)
            return fail(WF_ERR_TRACE_LENGTH, "%s commitment %zu has polynomials of 2^%u coefficients, not 2^%u", what, idx,
                        c->p.log2_trace_len, logn);
        const uint32_t wc = c->p.ext_degree;
        if (acc == 0 ? (wc != 1 && wc != ext_degree) : wc != ext_degree)
            return fail(WF_ERR_EXTENSION, "%s commitment %zu holds columns of extension degree %u; the composition is over degree %u",
                        what, idx, wc, ext_degree);
        const size_t colb = n * wc * eb, nc = (size_t)c->p.n_cols * c->p.n_traces;
        for (size_t i = 0; i < nc; i++) cols.push_back(DeepColumn{(const char *)c->polys + i * colb, wc, acc, cols.size()});
        return 0;
    };
    int rc;
    for (size_t i = 0; i < n_trace_commitments; i++)
        if ((rc = add(trace_commitments[i], 0, "trace", i))) return rc;
    const size_t n_trace_cols = cols.size();
    if (constraint_commitment && (rc = add(constraint_commitment, 1, "constraint", 0))) return rc;
    if (cols.size() > 0x7FFFFFFFull) return fail(WF_ERR_ARG, "too many columns");
    if (fri) {
        if (fri->ctx != ctx) return fail(WF_ERR_ARG, "the FRI prover belongs to another context");
        if (fri->field != field || fri->ext != ext_degree)
            return fail(WF_ERR_EXTENSION, "the FRI prover works over another field or extension degree");
    }
    HIP_TRY(hipSetDevice(ctx->device));
    WF_ENTER(ctx, ctx->stream);
    hipStream_t st = ctx->stream;
    const size_t poly_bytes = n * ext_degree * eb;
    if ((rc = ensure(ctx, ctx->io[0], poly_bytes))) return rc;
    std::vector<unsigned char> staging;  // read by a queued copy: lives until the stream has been synchronised
    rc = field == WF_FIELD_F64
             ? deep_compose_dev<F64>(ctx, st, cols, trace_coeffs, constraint_coeffs, n_trace_cols, ext_degree, n, z, ctx->io[0].p, staging)
             : deep_compose_dev<F128>(ctx, st, cols, trace_coeffs, constraint_coeffs, n_trace_cols, ext_degree, n, z, ctx->io[0].p, staging);
    if (rc) {
        (void)hipStreamSynchronize(st);
        return rc;
    }
    if (poly_out) HIP_TRY(hipMemcpyAsync(poly_out, ctx->io[0].p, poly_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (fri) return fri_begin_poly_impl(fri, ctx->io[0].p, true, n, lde_blowup);
    return 0;
}

}  // extern "C"
