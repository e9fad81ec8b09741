// EXPERIMENT, NOT PART OF THE PRODUCT BUILD (round 5; docs/EXPERIMENTS.md "register-resident radix-32 tiles").  Bit-exact (the cfg 2
// oracle comparison and the parity suites passed with it), measured SLOWER than the LDS-resident pair it would replace: strided pass
// 358.7 -> 375.1 us with 3.8 % FEWER vector instructions, fused last pass 541 -> 587 us with 9 % more (profiles/r05_r32.txt) -- at two
// waves per SIMD a wave's own issue rate (one instruction per ~4-5 cycles, ~20 % of them hazard s_nops behind carry chains) no
// longer covers the SIMD.  Commit af968a6 has it wired into path.hip (WF_EXP_NO_R32 selected the old kernels); to try it again:
// put this file back into csrc/, #include it after seg_kernels.hpp and restore that commit's two launcher hunks.
//
// Register-resident 2^10-row f64 tiles: two radix-32 rounds, LDS as the exchange buffer only (round 5).
//
// The segment kernels of seg_kernels.hpp keep a tile in LDS and walk it round by round (radix 16, 16, 4 for 2^10 rows: the tile
// goes through LDS three times, with a barrier and a set of LDS addresses per round).  Over Goldilocks w_32 = 2^6
// (f64/mod.rs:248-264: w_64 = 8), so a 32-point transform needs no general product at all, and 1024 = 32 x 32: a work-group of
// 256 threads holds the whole tile in registers -- thread (d0, l) the 32 values of lane l of rows a * 32 + d0 -- runs a radix-32
// round, multiplies by the ONE layer of general twiddles w_1024^(d0 k1), hands the values over through LDS once and runs the
// second radix-32 round on (k1, l).  Per element: 10 butterfly levels as before, 30 shift twiddles per 32 values and round instead
// of 24 + the shift layer between the second and third round (9.4), one general twiddle, ONE exchange instead of three LDS round
// trips.  Two work-groups of 256 threads per CU (two waves per SIMD, up to 256 VGPRs each): what hides latencies here is the 32
// independent values per thread, not the number of waves.
//
// Data flow of the strided pass:   global (8-byte gathers, rows I apart) -> registers -> round 1 -> LDS -> round 2 -> output factors
//                                  -> global, straight from registers (8 lanes = one 64-byte row piece per 8 threads)
// of the fused last pass:          global -> registers (prefetched a tile ahead) -> round 1 -> LDS -> round 2 -> LDS in place (rows
//                                  k = k1 + 32 k0 as 64-byte runs) -> 16-byte row stores + one lane per row hashing, as k_seg_last_hash
// Same arithmetic and outputs as k_seg_strided<F64, EVAL, false, 10> / k_seg_last_hash<F64, false, .., 10> (field arithmetic is
// exact: every order of the same butterflies gives the same bits); WF_EXP_NO_R32 selects those.
#pragma once

#include "seg_kernels.hpp"

namespace wf {

__host__ __device__ constexpr uint32_t bitrev5(uint32_t v) {
    return ((v & 1) << 4) | ((v & 2) << 2) | (v & 4) | ((v & 8) >> 2) | ((v & 16) >> 4);
}

// (u - t) * w_32^J of a transform of known direction: w_32 = 2^6, w_32^-J = 2^(192 - 6 J) (2 has order 192, 2^96 = -1).
// Exponents in (96, 128] are -2^K with K <= 32, whose shift ends in an addition: the sign goes into the subtraction instead.
template <int DIR, int J>
__device__ __forceinline__ uint64_t r32_twiddle(uint64_t u, uint64_t t) {
    constexpr int E = DIR > 0 ? (6 * J) % 192 : (192 - (6 * J) % 192) % 192;
    if constexpr (E == 0)
        return F64::sub(u, t);
    else if constexpr (E > 96 && E <= 128)
        return F64::template mul_pow2<E - 96>(F64::sub(t, u));
    else
        return F64::template mul_pow2_192<E>(F64::sub(u, t));
}

// 32-point DFT in registers, decimation in frequency: v[bitrev5(k)] = sum_a x_a w_32^(a k).  80 butterflies, 49 shift twiddles.
template <int DIR>
__device__ __forceinline__ void radix32(uint64_t (&v)[32]) {
    static_for<0, 5>([&](auto s_) {
        constexpr int S = decltype(s_)::value;
        constexpr int half = 16 >> S;
        static_for<0, (1 << S)>([&](auto g_) {
            constexpr int Q = decltype(g_)::value * 2 * half;
            static_for<0, half>([&](auto i_) {
                constexpr int I = decltype(i_)::value;
                const uint64_t u = v[Q + I], t = v[Q + I + half];
                v[Q + I] = F64::add(u, t);
                v[Q + I + half] = r32_twiddle<DIR, (I << S)>(u, t);
            });
        });
    });
}

constexpr uint32_t R32_LOGD = 10, R32_D = 1024, R32_NT = 256;

// Exchange buffer y[k1][d0][l] (elements): the d0 slot is XORed with k1 & 7 so that the eight k1 a wave reads (rows 2 KiB apart)
// fall into different banks; writes (fixed k1, eight consecutive d0) stay one contiguous 512-byte run.
__device__ __forceinline__ uint32_t r32_y_off(uint32_t k1, uint32_t d0, uint32_t l) { return (k1 << 8) + (((d0 ^ (k1 & 7u)) << 3) | l); }

// ---------------------------------------------------------------------------------------------------------------
// Strided pass on 2^10-row f64 tiles.  grid.x = n_cosets * n_seg * O * I as k_seg_strided; blockDim = 256.
// LDS: exchange buffer 64 KiB + digit twiddles 8 KiB + one factor table 8 KiB (input factors, then output factors) = 80 KiB.
template <int EVAL>
__global__ void __launch_bounds__(256, 2) k_seg_strided_r32(SegArgs<F64> a) {
    typedef F64 F;
    typedef uint64_t T;
    constexpr uint32_t S = 8, D = R32_D, NT = R32_NT;
    constexpr int DIR = EVAL ? 1 : -1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *x = reinterpret_cast<T *>(smem_raw);
    T *twd = x + (size_t)D * S;
    T *aux = twd + D;

    uint64_t bid = xcd_group_index(blockIdx.x, gridDim.x);  // neighbouring i on one XCD
    const uint32_t logI = ilog2_pow2(a.I), logO = ilog2_pow2(a.O);
    uint32_t c_in = 0;
    if (a.coset_inner) bid = coset_inner_split(bid, a.coset_inner - 1, c_in);
    const uint64_t i = bid & (a.I - 1);
    const uint64_t o = (bid >> logI) & (a.O - 1);
    const uint32_t rest = (uint32_t)(bid >> (logI + logO));
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
    const uint64_t row0 = ((o << R32_LOGD) << logI) + i;  // row of d = 0; the tile's rows are I apart
    const uint32_t tw_shift = a.logN - R32_LOGD - logI;
    const uint32_t tid = threadIdx.x, l = tid & 7u, d0 = tid >> 3;

    // prologue: every global read of the tile's setup is issued before the first one is used (as in k_seg_strided); four table
    // entries per thread
    const Pow2L<F> pin = scale_in ? pre : a.tw;  // (without input scaling these reads go to the root table and are dropped)
    T fo_a[4], fo_b[4], tw_q[4], fi_a[4], fi_b[4];
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) {
        const uint32_t k = tid + q * NT;
        a.tw.fetch(((uint64_t)k * i) << tw_shift, fo_a[q], fo_b[q]);
        tw_q[q] = a.digit_tw[k];
        pin.fetch((uint64_t)k << logI, fi_a[q], fi_b[q]);
    }
    // the thread's 32 values: lane l of rows a * 32 + d0
    T v[32];
    {
        const T *pr = src + (row0 + ((uint64_t)d0 << logI)) * S + l;
        const uint64_t rstep = ((uint64_t)32 << logI) * S;
#pragma unroll
        for (uint32_t q = 0; q < 32; q++) {
            v[q] = *pr;
            pr += rstep;
        }
    }
    T fo[4];
    {
        T start = a.scale_on ? a.scale : F::one();
        if (scale_in) start = pre.get(i);
        const bool trivial = !a.scale_on && !scale_in;
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            T f = F::mul(fo_a[q], fo_b[q]);
            if (!trivial) f = F::mul(f, start);
            fo[q] = f;
            twd[tid + q * NT] = tw_q[q];
            if (scale_in) aux[tid + q * NT] = F::mul(fi_a[q], fi_b[q]);  // h_c^(d I); h_c^i rides on the output factors
        }
    }
    __syncthreads();
    if (scale_in) {
#pragma unroll
        for (uint32_t q = 0; q < 32; q++) v[q] = F::mul(v[q], aux[q * 32 + d0]);
        __syncthreads();  // input factors consumed: the table takes the output factors
    }
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) aux[tid + q * NT] = fo[q];

    radix32<DIR>(v);
    // the one layer of general twiddles, w_1024^(d0 k1) (twd[0] = 1: no branch for d0 = 0), and the hand-over
    static_for<0, 32>([&](auto k_) {
        constexpr uint32_t k1 = decltype(k_)::value;
        T y = v[bitrev5(k1)];
        if constexpr (k1 != 0) y = F::mul(y, twd[d0 * k1]);
        x[r32_y_off(k1, d0, l)] = y;
    });
    __syncthreads();
    const uint32_t k1 = d0;  // second round: thread (k1, l)
    static_for<0, 32>([&](auto d_) {
        constexpr uint32_t dd = decltype(d_)::value;
        v[dd] = x[r32_y_off(k1, dd, l)];
    });
    radix32<DIR>(v);
    // output row k = k1 + 32 k0 (I rows apart), times its factor; eight threads write one 64-byte row piece
    {
        T *pw = dst + row0 * S + l + ((uint64_t)k1 << (logI + 3));
        const uint64_t wstep = (uint64_t)32 << (logI + 3);
        static_for<0, 32>([&](auto k_) {
            constexpr uint32_t k0 = decltype(k_)::value;
            *pw = F::mul(v[bitrev5(k0)], aux[k1 + 32 * k0]);
            pw += wstep;
        });
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused, persistent last pass on 2^10-row f64 tiles: one segment of one trace (k_seg_last_hash<F64, false, false, false, false, 10>
// is the LDS-resident form; tickets, decode and sign-off are the same).  blockDim = 256; LDS: 64 KiB + 8 KiB of digit twiddles.
// After the second round the tile goes back to LDS IN PLACE -- thread (k1, l) overwrites exactly the 64-byte-row slots it read --
// as rows: row k = k1 + 32 k0 is the 64-byte run at k1 * 2 KiB + (k0 ^ (k1 & 7)) * 64 B, its four 16-byte units XORed with
// (k0 >> 3) & 3 (one lane per row reading 64 bytes at a 64-byte lane stride would hit four banks sixteen times over).  The units
// move between the eight lanes of a row only, i.e. inside one wave, whose LDS accesses execute in order: no barrier between the
// reads of the exchange and these writes.
__device__ __forceinline__ uint32_t r32_z_off(uint32_t k, uint32_t unit) {  // 16-byte unit `unit` of row k (element offset)
    const uint32_t k1 = k & 31u, k0 = k >> 5;
    return (k1 << 8) + ((k0 ^ (k1 & 7u)) << 3) + ((unit ^ ((k0 >> 3) & 3u)) << 1);
}

__global__ void __launch_bounds__(256, 2) k_seg_last_hash_r32(SegArgs<F64> a) {
    typedef F64 F;
    typedef uint64_t T;
    typedef Pair<T> P2;
    constexpr uint32_t S = 8, D = R32_D, NT = R32_NT, logD_ = R32_LOGD;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *x = reinterpret_cast<T *>(smem_raw);
    T *twd = x + (size_t)D * S;
    const uint64_t total = (uint64_t)a.n_cosets * a.O;  // tickets: (coset, row block)
    const uint64_t seg_elems = ((uint64_t)1 << a.logN) * S;
    const uint32_t out_shift = a.logN - logD_;
    const uint32_t k_stride = a.rows_per_k * (uint32_t)a.row_width;
    const uint32_t hash_bytes = a.hash_epr * F::BYTES;

    for (uint32_t e = threadIdx.x; e < D - 1; e += NT) twd[e] = a.digit_tw[e];  // (twd[D - 1] is never read: the ticket words live there)

    auto decode = [&](uint64_t t, uint32_t &c, uint64_t &o, uint64_t &rev_o) {
        const uint64_t bid = xcd_group_index(t, total);  // coset fastest, 8 consecutive tickets on one XCD
        const uint32_t b32 = (uint32_t)bid, q32 = b32 / a.n_cosets;
        c = b32 - q32 * a.n_cosets;
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
    auto tile_src = [&](uint32_t c, uint64_t o) -> const T * { return a.src + (uint64_t)c * a.n_seg * seg_elems + (o << logD_) * S; };

    uint32_t *ticket_sh = reinterpret_cast<uint32_t *>(twd + (D - 1));
    const uint32_t xcd = blockIdx.x & 7;
    const uint64_t per_xcd = total >> 3;
    auto sign_off = [&]() {  // (k_seg_last_hash: the counters reset themselves)
        if (threadIdx.x == 0) {
            const uint32_t mine = (gridDim.x + 7 - xcd) >> 3;
            if (atomicAdd(a.tile_counters + 8 + xcd, 1u) == mine - 1) {
                atomicExch(a.tile_counters + xcd, 0u);
                atomicExch(a.tile_counters + 8 + xcd, 0u);
            }
        }
    };
    if (threadIdx.x == 0) ticket_sh[0] = atomicAdd(a.tile_counters + xcd, 1u);
    __syncthreads();
    uint64_t ticket = ticket_sh[0];
    if (ticket >= per_xcd) {
        sign_off();
        return;
    }
    uint32_t c;
    uint64_t o, rev_o;
    decode(ticket * 8 + xcd, c, o, rev_o);

    // a tile is one contiguous 64 KiB run; lane l of row a * 32 + d0 is element a * 256 + tid
    T nv[32];
    {
        const T *s_ = tile_src(c, o) + threadIdx.x;
#pragma unroll
        for (uint32_t q = 0; q < 32; q++) nv[q] = s_[q * NT];
    }
    bool first_tile = true;
    while (true) {
        T v[32];
#pragma unroll
        for (uint32_t q = 0; q < 32; q++) {
            v[q] = nv[q];
            nv[q] = 0;  // (ends the live range: refilled only `if (more)`, see k_seg_last_hash)
        }
        // thread 0 draws the next ticket now; the work-group learns it at the exchange barrier
        uint32_t my_ticket = 0;
        if (threadIdx.x == 0) my_ticket = atomicAdd(a.tile_counters + xcd, 1u);
        radix32<1>(v);
        {
            const uint32_t tid = opaque_tid(), l = tid & 7u, d0 = tid >> 3;
            if (!first_tile) __syncthreads();  // the previous tile's rows have been stored and hashed: the buffer is free
            static_for<0, 32>([&](auto k_) {
                constexpr uint32_t k1 = decltype(k_)::value;
                T y = v[bitrev5(k1)];
                if constexpr (k1 != 0) y = F::mul(y, twd[d0 * k1]);
                x[r32_y_off(k1, d0, l)] = y;
            });
            if (tid == 0) ticket_sh[1] = my_ticket;
        }
        first_tile = false;
        __syncthreads();
        // the next tile starts its way into registers: it has the second round, the row stores and the hashing to arrive
        ticket = ticket_sh[1];
        const bool more = ticket < per_xcd;
        uint32_t cn = c;
        uint64_t on = o, rev_on = rev_o;
        {
            const uint32_t tid = opaque_tid(), l = tid & 7u, k1 = tid >> 3;
            static_for<0, 32>([&](auto d_) {
                constexpr uint32_t dd = decltype(d_)::value;
                v[dd] = x[r32_y_off(k1, dd, l)];
            });
            if (more) {
                decode(ticket * 8 + xcd, cn, on, rev_on);
                const T *s_ = tile_src(cn, on) + tid;
#pragma unroll
                for (uint32_t q = 0; q < 32; q++) nv[q] = s_[q * NT];
            }
            radix32<1>(v);
            // rows back to LDS, in place (same wave: see above)
            static_for<0, 32>([&](auto k_) {
                constexpr uint32_t k0 = decltype(k_)::value;
                x[r32_z_off(k1 + 32 * k0, l >> 1) + (l & 1u)] = v[bitrev5(k0)];
            });
        }
        __syncthreads();
        // row stores: 16-byte unit `w` of rows pos0 + 64 j -> LDE row k * rows_per_k + c (four threads = one 64-byte row)
        {
            const uint32_t tid = opaque_tid(), pos0 = tid >> 2, w = tid & 3u, lane_a = 2 * w;
            if (lane_a < a.store_cols) {
                T *pa = a.dst + (uint64_t)c * a.row_width + lane_a;
                const bool pair = lane_a + 1 < a.store_cols;
#pragma unroll 4
                for (uint32_t j = 0; j < 16; j++) {
                    const uint32_t k = pos0 + 64 * j;
                    const P2 val = *reinterpret_cast<const P2 *>(x + r32_z_off(k, w));
                    const uint64_t off = (uint64_t)(uint32_t)(rev_o + ((uint64_t)k << out_shift)) * k_stride;
                    if (pair)
                        store_pair(pa + off, val);
                    else
                        pa[off] = val.a;
                }
            }
        }
        // leaves: one lane per row, four rows per lane (k0 = tid & 31, k1 = tid / 32 + 8 r)
        {
            const uint32_t tid = opaque_tid();
#pragma unroll 1
            for (uint32_t r = 0; r < 4; r++) {
                const uint32_t k = ((tid >> 5) + 8 * r) + 32 * (tid & 31u);
                T ev[S];
                uint4 *evq = reinterpret_cast<uint4 *>(ev);
#pragma unroll
                for (uint32_t w = 0; w < 4; w++) evq[w] = *reinterpret_cast<const uint4 *>(x + r32_z_off(k, w));
                uint32_t m[16], cv[8];
#pragma unroll
                for (uint32_t e = 0; e < S; e++) elem_words<F>(ev[e], &m[e * 2]);
                b3::set_iv(cv);
                b3::compress(cv, m, 0, 0, hash_bytes, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT);
                const uint64_t row = (uint64_t)(uint32_t)(rev_o + ((uint64_t)k << out_shift)) * a.rows_per_k + c;
                uint4 *dl = reinterpret_cast<uint4 *>(a.leaves + row * 8);
                dl[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
                dl[1] = digest_hi(cv[4], cv[5], cv[6], cv[7], a.digest_words);
            }
        }
        if (!more) {
            sign_off();
            break;
        }
        c = cn;
        o = on;
        rev_o = rev_on;
    }
}

}  // namespace wf
