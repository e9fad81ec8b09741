// Twiddle / coset / series tables of a context (host side; templates over the field, no kernels): included by the units
// that launch transforms (path.hip, fri.hip).  The tables live in wf_ctx::tables and are built once per shape.
#pragma once

#include "wf_internal.hpp"

#include "kernels.hpp"

using namespace wf;

// One table to the device: allocated through dev_malloc (which gives the context's parked buffers back to the driver and retries
// when the device is full), filled synchronously, and freed again if the copy fails -- a cached table is either complete or absent.
static int table_upload(wf_ctx *ctx, void **dst, const void *src, size_t bytes) {
    *dst = nullptr;
    hipError_t e = dev_malloc(ctx, dst, bytes);
    if (e != hipSuccess) return fail(WF_ERR_HIP, "table allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(*dst);
        *dst = nullptr;
        return fail(WF_ERR_HIP, "table upload failed: %s", hipGetErrorString(e));
    }
    return 0;
}

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
    int rc = table_upload(ctx, &ts.lo, lo.data(), lo.size() * sizeof(T));
    if (rc) return rc;
    if ((rc = table_upload(ctx, &ts.hi, hi.data(), hi.size() * sizeof(T)))) {
        (void)hipFree(ts.lo);
        ts.lo = nullptr;
        return rc;
    }
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

// Input factors of a SINGLE-pass coset evaluation (k_seg_last, FTAB): entry [c][k] = h_c^k, h_c = offset * g^c, k < 2^logR -- what a tile
// otherwise rebuilds from the two-level coset tables (two reads and one product per row and tile: 78 vector instructions for f128)
template <class F>
static int coset_row_factors(wf_ctx *ctx, uint32_t logR, uint32_t logB, typename F::T offset, uint64_t off_lo, uint64_t off_hi,
                             const typename F::T **out) {
    auto key = std::make_tuple((int)F::FIELD_ID, (int)logR, 8, (int)logB, off_lo, off_hi);
    auto it = ctx->tables.find(key);
    if (it == ctx->tables.end()) {
        typedef typename F::T T;
        const T g = f_root_of_unity<F>(logR + logB);
        const size_t N = (size_t)1 << logR, B = (size_t)1 << logB;
        std::vector<T> tab(N * B);
        T h = offset;
        for (size_t c = 0; c < B; c++) {
            T acc = F::one();
            for (size_t k = 0; k < N; k++) {
                tab[c * N + k] = acc;
                acc = F::mul(acc, h);
            }
            h = F::mul(h, g);
        }
        TableSet ts;
        const int rcu = table_upload(ctx, &ts.lo, tab.data(), tab.size() * sizeof(T));
        if (rcu) return rcu;
        it = ctx->tables.emplace(key, ts).first;
    }
    *out = (const typename F::T *)it->second.lo;
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
        int rcu = table_upload(ctx, &ts.lo, lo.data(), nlo * sizeof(T));
        if (rcu) return rcu;
        if ((rcu = table_upload(ctx, &ts.hi, hi.data(), nhi * sizeof(T)))) {
            (void)hipFree(ts.lo);
            return rcu;
        }
        ts.s = s;
        ts.mask = (uint32_t)(nlo - 1);
        it = ctx->tables.emplace(key, ts).first;
    }
    *out = &it->second;
    return 0;
}

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
        const int rcu = table_upload(ctx, &ts.lo, tab.data(), tab.size() * sizeof(T));
        if (rcu) return rcu;
        it = ctx->tables.emplace(key, ts).first;
    }
    *out = (const typename F::T *)it->second.lo;
    return 0;
}

// Output factors of a strided pass after the first (k_seg_strided_wide<.., GTAB>): entry [i][k] = w^((k i) << shift), w the 2^logN-th
// root of the transform (or its inverse), k < 2^logD, i < 2^logI, shift = logN - logD - logI
template <class F>
static int pass_factor_table(wf_ctx *ctx, uint32_t logN, uint32_t logD, uint32_t logI, bool inverse, const typename F::T **out) {
    auto key = std::make_tuple((int)F::FIELD_ID, (int)logN, inverse ? 7 : 6, (int)(logD | (logI << 8)), (uint64_t)0, (uint64_t)0);
    auto it = ctx->tables.find(key);
    if (it == ctx->tables.end()) {
        typedef typename F::T T;
        T w = f_root_of_unity<F>(logN ? logN : 1);
        if (inverse) w = f_inv<F>(w);
        const uint64_t D = (uint64_t)1 << logD, I = (uint64_t)1 << logI;
        const T base = f_pow<F>(w, (u128)1 << (logN - logD - logI));  // w_(D I)
        std::vector<T> tab(D * I);
        T step = F::one();  // base^i
        for (uint64_t i = 0; i < I; i++) {
            T acc = F::one();
            for (uint64_t k = 0; k < D; k++) {
                tab[i * D + k] = acc;
                acc = F::mul(acc, step);
            }
            step = F::mul(step, base);
        }
        TableSet ts;
        const int rcu = table_upload(ctx, &ts.lo, tab.data(), tab.size() * sizeof(T));
        if (rcu) return rcu;
        it = ctx->tables.emplace(key, ts).first;
    }
    *out = (const typename F::T *)it->second.lo;
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

