// Base-field arithmetic shared by host (table construction) and device (kernels).
//
// F64  : Goldilocks p = 2^64 - 2^32 + 1, values are Montgomery residues x*2^64 mod p in [0,p), i.e. exactly the
//        in-memory form of the reference's f64::BaseElement (math/src/field/f64/mod.rs:48-53), so buffers can be
//        handed to / taken from a winter-prover process without conversion.
// F128 : p = 2^128 - 45*2^40 + 1, canonical integers in [0,p) (math/src/field/f128/mod.rs:35, IS_CANONICAL :73).
//
// Every operation returns the unique fully reduced representative, which is what makes results bit-identical to
// the reference whatever the evaluation order.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define WF_HD __host__ __device__ __forceinline__

namespace wf {

typedef unsigned __int128 u128;

// ------------------------------------------------------------------------------------------------ F64
struct F64 {
    typedef uint64_t T;
    static constexpr uint64_t P = 0xFFFFFFFF00000001ull;
    static constexpr uint64_t R2 = 0xFFFFFFFE00000001ull;  // 2^128 mod p
    static constexpr uint32_t TWO_ADICITY = 32;
    static constexpr uint64_t TWO_ADIC_ROOT = 7277203076849721926ull;  // canonical; order 2^32
    static constexpr int BYTES = 8;
    static constexpr int FIELD_ID = 1;

    static WF_HD T zero() { return 0; }
    static WF_HD T one() { return 0xFFFFFFFFull; }  // 2^64 mod p

    // x * 2^-64 mod p for x < p * 2^64 (math/src/field/f64/mod.rs:651-661 computes the same value)
    static WF_HD uint64_t mont_reduce(uint64_t xl, uint64_t xh) {
#if defined(__HIP_DEVICE_COMPILE__)
        // explicit 32-bit carry chains: 8 VALU (the 64-bit formulation below compiles to ~16 on gfx950)
        const uint32_t l0 = (uint32_t)xl, l1 = (uint32_t)(xl >> 32), h0 = (uint32_t)xh, h1 = (uint32_t)(xh >> 32);
        uint32_t ah, t, bl, rl, u, rh, r2l;
        const uint32_t e = __builtin_add_overflow(l1, l0, &ah);  // a = xl + (xl << 32) = (ah : l0), carry e
        const uint32_t b1 = __builtin_sub_overflow(l0, ah, &t);  // b = a - (a >> 32) - e
        const uint32_t b2 = __builtin_sub_overflow(t, e, &bl);
        const uint32_t bh = ah - (b1 | b2);
        const uint32_t c1 = __builtin_sub_overflow(h0, bl, &rl);  // r = xh - b
        const uint32_t c2 = __builtin_sub_overflow(h1, bh, &u);
        const uint32_t c3 = __builtin_sub_overflow(u, c1, &rh);
        const uint32_t m = 0u - (c2 | c3);                        // borrowed: r += p, i.e. r -= 2^32 - 1
        const uint32_t d1 = __builtin_sub_overflow(rl, m, &r2l);
        return ((uint64_t)(rh - d1) << 32) | r2l;
#else
        uint64_t a = xl + (xl << 32);
        uint64_t e = a < xl;
        uint64_t b = a - (a >> 32) - e;
        uint64_t r = xh - b;
        return xh < b ? r - 0xFFFFFFFFull : r;
#endif
    }
    static WF_HD T mul(T a, T b) {
#if defined(__HIP_DEVICE_COMPILE__)
        // 32-bit schoolbook: four v_mad_u64_u32 share the partial products (gfx950 has no 64-bit multiplier)
        const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
        const uint64_t p00 = (uint64_t)a0 * b0;
        const uint64_t p01 = (uint64_t)a0 * b1 + (p00 >> 32);
        const uint64_t p10 = (uint64_t)a1 * b0 + (uint32_t)p01;
        const uint64_t hi = (uint64_t)a1 * b1 + (p01 >> 32) + (p10 >> 32);
        const uint64_t lo = (p10 << 32) | (uint32_t)p00;
#else
        u128 x = (u128)a * (u128)b;
        uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
#endif
        return mont_reduce(lo, hi);
    }
    // a + b: the sum wraps at 2^64 or lands in [p, 2^64) -> add 2^32 - 1 (= -p mod 2^64); both cases end in [0, p)
    static WF_HD T add(T a, T b) {
        uint64_t r = a + b;
        const bool fix = (r < a) | (r >= P);
        return r + (fix ? 0xFFFFFFFFull : 0ull);
    }
    static WF_HD T sub(T a, T b) {
#if defined(__HIP_DEVICE_COMPILE__)
        // explicit 32-bit borrow chain (5 VALU): r = a - b, then r -= (2^32 - 1) if it borrowed
        const uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
        uint32_t rl, t, rh;
        const uint32_t b1 = __builtin_sub_overflow(al, bl, &rl);
        const uint32_t b2 = __builtin_sub_overflow(ah, bh, &t);
        const uint32_t b3 = __builtin_sub_overflow(t, b1, &rh);
        const uint32_t m = 0u - (b2 | b3);
        uint32_t r2l;
        const uint32_t c1 = __builtin_sub_overflow(rl, m, &r2l);
        return ((uint64_t)(rh - c1) << 32) | r2l;
#else
        uint64_t r = a - b;
        return a < b ? r - 0xFFFFFFFFull : r;
#endif
    }
    // x * 2^K (0 < K < 64) and x * 2^-K (0 < K <= 32) without a general product (2 is a 192nd root of unity: 2^64 = 2^32 - 1,
    // 2^96 = -1).  x * 2^K = lo + hi * 2^64 = lo + hi * (2^32 - 1); x * 2^-K = (x >> K) - (x mod 2^K) * 2^(96 - K)
    // = q - m * (2^32 - 1) with m = (x mod 2^K) << (32 - K).  9 VALU each against 17 for mul; Montgomery form is
    // preserved because the factor is a plain integer.
    // 32 < K < 64 (K = 32 + M): x 2^M = y2:y1:y0 in 32-bit words (y2 < 2^M), and with 2^64 = 2^32 - 1, 2^96 = -1
    //     x 2^K = y0 2^32 + y1 2^64 + y2 2^96 = (y0 + y1) 2^32 - (y1 + y2).
    // y0 + y1 = ah + k 2^32: k 2^64 = k (2^32 - 1) makes the minuend A = ah:(k ? ffffffff : 0); k = 1 needs
    // ah = y0 + y1 - 2^32 <= (2^32 - 2^M) + (2^32 - 1) - 2^32 < ffffffff, and k = 0 gives ah:0 <= ffffffff00000000, so A < p
    // either way; the subtrahend y1 + y2 < 2^33 < p: one canonical subtraction finishes.  12 VALU, 7 of them single-word.
    template <int K>
    static WF_HD T mul_pow2(T x) {
        static_assert(K > 0 && K < 64, "shift out of range");
        if constexpr (K <= 32) {
            const uint64_t lo = x << K;
            const uint64_t hi = x >> (64 - K);
            return add(lo, (hi << 32) - hi);  // add() reduces any 64-bit lo correctly here: see the bound in DESIGN.md §4
        } else {
            constexpr int M = K - 32;
            const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32);
            const uint32_t y0 = x0 << M, y1 = (x1 << M) | (x0 >> (32 - M)), y2 = x1 >> (32 - M);
            uint32_t ah, bl;
            const uint32_t k = __builtin_add_overflow(y0, y1, &ah);
            const uint32_t kb = __builtin_add_overflow(y1, y2, &bl);
            return sub(((uint64_t)ah << 32) | (0u - k), ((uint64_t)kb << 32) | bl);
        }
    }
    template <int K>
    static WF_HD T div_pow2(T x) {
        static_assert(K > 0 && K <= 32, "shift out of range");
        const uint64_t q = x >> K;
        const uint64_t m = (uint64_t)((uint32_t)x << (32 - K));
        return sub(q, (m << 32) - m);
    }
    // The negated forms whose sign is free: the positive forms above end in one canonical subtraction of two values < p
    // (K > 32: A - B, division: q - m (2^32 - 1)), so swapping its operands negates the result.
    template <int K>
    static WF_HD T neg_mul_pow2(T x) {
        static_assert(K > 32 && K < 64, "shift out of range");
        constexpr int M = K - 32;
        const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32);
        const uint32_t y0 = x0 << M, y1 = (x1 << M) | (x0 >> (32 - M)), y2 = x1 >> (32 - M);
        uint32_t ah, bl;
        const uint32_t k = __builtin_add_overflow(y0, y1, &ah);
        const uint32_t kb = __builtin_add_overflow(y1, y2, &bl);
        return sub(((uint64_t)kb << 32) | bl, ((uint64_t)ah << 32) | (0u - k));
    }
    template <int K>
    static WF_HD T neg_div_pow2(T x) {
        static_assert(K > 0 && K <= 32, "shift out of range");
        const uint64_t q = x >> K;
        const uint64_t m = (uint64_t)((uint32_t)x << (32 - K));
        return sub((m << 32) - m, q);
    }
    // x * 2^E for any exponent of the group generated by 2 (order 192: 2^96 = -1).  [0, 64): the shift; [64, 96): 2^E =
    // -2^-(96 - E); (128, 160): -2^(E - 96) with E - 96 > 32; [160, 192): 2^-(192 - E) -- all at the price of the positive
    // form.  Only [96, 128] (-2^K with K <= 32, whose positive form ends in an addition) pays a separate negation here:
    // callers that can hand in -x instead (a preceding subtraction with swapped operands) use mul_pow2<E - 96> themselves.
    template <int E>
    static WF_HD T mul_pow2_192(T x) {
        static_assert(E >= 0 && E < 192, "exponent out of range");
        if constexpr (E == 0) return x;
        else if constexpr (E < 64) return mul_pow2<E>(x);
        else if constexpr (E < 96) return neg_div_pow2<96 - E>(x);
        else if constexpr (E == 96) return sub(0, x);
        else if constexpr (E <= 128) return sub(0, mul_pow2<E - 96>(x));
        else if constexpr (E < 160) return neg_mul_pow2<E - 96>(x);
        else return div_pow2<192 - E>(x);
    }
    static WF_HD T from_canonical(uint64_t v) { return mul(v % P, R2); }
    static WF_HD uint64_t to_canonical(T x) { return mont_reduce(x, 0); }
    static WF_HD T from_u128_canonical(u128 v) { return from_canonical((uint64_t)(v % (u128)P)); }
    static WF_HD bool is_valid(T x) { return x < P; }
};

// ------------------------------------------------------------------------------------------------ F128
struct U128 {
    uint64_t lo, hi;
};

struct F128 {
    typedef U128 T;
    static constexpr uint32_t TWO_ADICITY = 40;
    static constexpr int BYTES = 16;
    static constexpr int FIELD_ID = 2;
    // p = 2^128 - C, C = 45*2^40 - 1
    static constexpr uint64_t P_LO = 0xFFFFD30000000001ull, P_HI = 0xFFFFFFFFFFFFFFFFull;
    static constexpr uint64_t C_LO = 0x00002CFFFFFFFFFFull;  // 45*2^40 - 1 (46 bits)

    static WF_HD u128 P() { return ((u128)P_HI << 64) | (u128)P_LO; }
    static WF_HD u128 w(T a) { return ((u128)a.hi << 64) | (u128)a.lo; }
    static WF_HD T n(u128 v) { return T{(uint64_t)v, (uint64_t)(v >> 64)}; }

    static WF_HD T zero() { return T{0, 0}; }
    static WF_HD T one() { return T{1, 0}; }
    // ---- device arithmetic on explicit 32-bit limbs (gfx950 has no 64-bit multiplier or 128-bit carry chain; the
    // unsigned __int128 formulation below compiles to 163 / 16 / 22 VALU for mul / add / sub, this one to 78 / 14 / 13)
    static constexpr uint32_t C0 = 0xFFFFFFFFu, C1 = 0x2CFFu;  // limbs of C
    static constexpr uint32_t K = 45u << 8;                   // C + 1 = K * 2^32
    static WF_HD void limbs(T x, uint32_t (&l)[4]) {
        l[0] = (uint32_t)x.lo;
        l[1] = (uint32_t)(x.lo >> 32);
        l[2] = (uint32_t)x.hi;
        l[3] = (uint32_t)(x.hi >> 32);
    }
    static WF_HD T unlimbs(const uint32_t (&l)[4]) {
        return T{((uint64_t)l[1] << 32) | l[0], ((uint64_t)l[3] << 32) | l[2]};
    }
    static WF_HD uint32_t adc(uint32_t a, uint32_t b, uint32_t cin, uint32_t &cout) {
        uint32_t s;
        const uint32_t c1 = __builtin_add_overflow(a, b, &s);
        const uint32_t c2 = __builtin_add_overflow(s, cin, &s);
        cout = c1 | c2;
        return s;
    }
    static WF_HD uint32_t sbb(uint32_t a, uint32_t b, uint32_t bin, uint32_t &bout) {
        uint32_t s;
        const uint32_t c1 = __builtin_sub_overflow(a, b, &s);
        const uint32_t c2 = __builtin_sub_overflow(s, bin, &s);
        bout = c1 | c2;
        return s;
    }
    static WF_HD uint64_t mad32(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }
    static WF_HD uint64_t pair32(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
    // y = z + C mod 2^128, returns the carry out (set  <=>  z >= p)
    static WF_HD uint32_t add_c(const uint32_t (&z)[4], uint32_t (&y)[4]) {
        uint32_t c = __builtin_add_overflow(z[0], C0, &y[0]);
        y[1] = adc(z[1], C1, c, c);
        c = __builtin_add_overflow(z[2], c, &y[2]);
        c = __builtin_add_overflow(z[3], c, &y[3]);
        return c;
    }
    static WF_HD T add_limbs(T x, T y) {
        uint32_t a[4], b[4], s[4], t[4], r[4], c;
        limbs(x, a);
        limbs(y, b);
        c = __builtin_add_overflow(a[0], b[0], &s[0]);
        s[1] = adc(a[1], b[1], c, c);
        s[2] = adc(a[2], b[2], c, c);
        s[3] = adc(a[3], b[3], c, c);
        const uint32_t c2 = add_c(s, t);
        const bool take = (c | c2) != 0;  // wrapped at 2^128 or landed in [p, 2^128): subtract p = add C
#pragma unroll
        for (int i = 0; i < 4; i++) r[i] = take ? t[i] : s[i];
        return unlimbs(r);
    }
    static WF_HD T sub_limbs(T x, T y) {
        uint32_t a[4], b[4], d[4], r[4], bo, q;
        limbs(x, a);
        limbs(y, b);
        bo = __builtin_sub_overflow(a[0], b[0], &d[0]);
        d[1] = sbb(a[1], b[1], bo, bo);
        d[2] = sbb(a[2], b[2], bo, bo);
        d[3] = sbb(a[3], b[3], bo, bo);
        const uint32_t m = 0u - bo;  // borrowed: add p = subtract C (mod 2^128)
        q = __builtin_sub_overflow(d[0], m, &r[0]);
        r[1] = sbb(d[1], C1 & m, q, q);
        q = __builtin_sub_overflow(d[2], q, &r[2]);
        r[3] = d[3] - q;
        return unlimbs(r);
    }
    static WF_HD T mul_limbs(T x, T y) {
        uint32_t a[4], b[4], r[8];
        limbs(x, a);
        limbs(y, b);
        // 256-bit product, row by row; every mad's addend stays below 2^33, so a 32x32+64 mad cannot overflow
        {
            uint64_t p[4];
#pragma unroll
            for (int j = 0; j < 4; j++) p[j] = mad32(a[j], b[0], 0);
            uint32_t c;
            r[0] = (uint32_t)p[0];
            c = __builtin_add_overflow((uint32_t)(p[0] >> 32), (uint32_t)p[1], &r[1]);
            r[2] = adc((uint32_t)(p[1] >> 32), (uint32_t)p[2], c, c);
            r[3] = adc((uint32_t)(p[2] >> 32), (uint32_t)p[3], c, c);
            r[4] = (uint32_t)(p[3] >> 32) + c;
        }
#pragma unroll
        for (int i = 1; i < 4; i++) {
            uint32_t carry = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint64_t t;
                if (j == 0) {
                    t = mad32(a[j], b[i], pair32(r[i + j], 0));
                } else {
                    uint32_t lo;
                    const uint32_t cc = __builtin_add_overflow(r[i + j], carry, &lo);
                    t = mad32(a[j], b[i], pair32(lo, cc));
                }
                r[i + j] = (uint32_t)t;
                carry = (uint32_t)(t >> 32);
            }
            r[i + 4] = carry;
        }
        // first fold: L + H*C = L + ((H*K) << 32) - H with H = r[4..7]; the result v has 6 limbs, v[5] < 2^15
        uint32_t hk[5], v[6];
        {
            uint64_t t = mad32(r[4], K, 0);
            hk[0] = (uint32_t)t;
            t = mad32(r[5], K, pair32((uint32_t)(t >> 32), 0));
            hk[1] = (uint32_t)t;
            t = mad32(r[6], K, pair32((uint32_t)(t >> 32), 0));
            hk[2] = (uint32_t)t;
            t = mad32(r[7], K, pair32((uint32_t)(t >> 32), 0));
            hk[3] = (uint32_t)t;
            hk[4] = (uint32_t)(t >> 32);
            uint32_t c, w1, w2, w3, w4, w5, bo;
            c = __builtin_add_overflow(r[1], hk[0], &w1);
            w2 = adc(r[2], hk[1], c, c);
            w3 = adc(r[3], hk[2], c, c);
            c = __builtin_add_overflow(hk[3], c, &w4);
            w5 = hk[4] + c;
            bo = __builtin_sub_overflow(r[0], r[4], &v[0]);
            v[1] = sbb(w1, r[5], bo, bo);
            v[2] = sbb(w2, r[6], bo, bo);
            v[3] = sbb(w3, r[7], bo, bo);
            bo = __builtin_sub_overflow(w4, bo, &v[4]);
            v[5] = w5 - bo;
        }
        // second fold: t = v[5]:v[4] < 2^47, z = v[0..3] + ((t*K) << 32) - t with t*K < 2^61
        uint32_t z[4], net;
        {
            const uint64_t t0 = mad32(v[4], K, 0);
            const uint32_t tk0 = (uint32_t)t0;
            const uint32_t tk1 = (uint32_t)(t0 >> 32) + v[5] * K;
            uint32_t c, w1, w2, w3, bo;
            c = __builtin_add_overflow(v[1], tk0, &w1);
            w2 = adc(v[2], tk1, c, c);
            c = __builtin_add_overflow(v[3], c, &w3);
            bo = __builtin_sub_overflow(v[0], v[4], &z[0]);
            z[1] = sbb(w1, v[5], bo, bo);
            bo = __builtin_sub_overflow(w2, bo, &z[2]);
            bo = __builtin_sub_overflow(w3, bo, &z[3]);
            net = c ^ bo;  // the integer is non-negative: (carry, borrow) is (0,0), (1,0) or (1,1)
        }
        // net set: the value is z + 2^128 = z + C (mod p), already below p; otherwise subtract p once if z >= p
        uint32_t t[4], out[4];
        const uint32_t cy = add_c(z, t);
        const bool take = (net | cy) != 0;
#pragma unroll
        for (int i = 0; i < 4; i++) out[i] = take ? t[i] : z[i];
        return unlimbs(out);
    }

    static WF_HD T add(T a, T b) {
#if defined(__HIP_DEVICE_COMPILE__)
        return add_limbs(a, b);
#else
        u128 x = w(a), z = P() - w(b);
        return n(x < z ? x + w(b) : x - z);
#endif
    }
    static WF_HD T sub(T a, T b) {
#if defined(__HIP_DEVICE_COMPILE__)
        return sub_limbs(a, b);
#else
        u128 x = w(a), y = w(b);
        return n(x < y ? P() - y + x : x - y);
#endif
    }
    // 64x64 -> 128
    static WF_HD void mul64(uint64_t a, uint64_t b, uint64_t &lo, uint64_t &hi) {
#if defined(__HIP_DEVICE_COMPILE__)
        lo = a * b;
        hi = __umul64hi(a, b);
#else
        u128 x = (u128)a * (u128)b;
        lo = (uint64_t)x;
        hi = (uint64_t)(x >> 64);
#endif
    }
    // (a * b) mod p : 256-bit schoolbook product folded twice with 2^128 = C (mod p)
    static WF_HD T mul(T a, T b) {
#if defined(__HIP_DEVICE_COMPILE__)
        return mul_limbs(a, b);
#else
        return mul_wide(a, b);
#endif
    }
    static WF_HD T mul_wide(T a, T b) {
        uint64_t p0l, p0h, p1l, p1h, p2l, p2h, p3l, p3h;
        mul64(a.lo, b.lo, p0l, p0h);
        mul64(a.lo, b.hi, p1l, p1h);
        mul64(a.hi, b.lo, p2l, p2h);
        mul64(a.hi, b.hi, p3l, p3h);
        // r0..r3 : 64-bit limbs of the product
        uint64_t r0 = p0l;
        u128 acc = (u128)p0h + p1l + p2l;
        uint64_t r1 = (uint64_t)acc;
        acc = (acc >> 64) + p1h + p2h + p3l;
        uint64_t r2 = (uint64_t)acc;
        uint64_t r3 = (uint64_t)(acc >> 64) + p3h;
        // fold: H = r3:r2 (128 bits), L = r1:r0 ; x = L + H*C, H*C < 2^174
        uint64_t h0l, h0h, h1l, h1h;
        mul64(r2, C_LO, h0l, h0h);
        mul64(r3, C_LO, h1l, h1h);
        u128 t = (u128)r0 + h0l;
        uint64_t s0 = (uint64_t)t;
        t = (t >> 64) + r1 + h0h + h1l;
        uint64_t s1 = (uint64_t)t;
        uint64_t s2 = (uint64_t)(t >> 64) + h1h;  // < 2^47
        // second fold: s2 * C < 2^93
        uint64_t g_l, g_h;
        mul64(s2, C_LO, g_l, g_h);
        u128 low = ((u128)s1 << 64) | s0;
        u128 g = ((u128)g_h << 64) | g_l;
        u128 z = low + g;
        bool carry = z < low;  // value = z + carry*2^128 = z + carry*C (mod p)
        if (carry) z += (u128)C_LO;  // cannot carry again: z < 2^93 here
        if (z >= P()) z -= P();
        return n(z);
    }
    static WF_HD T from_u128_canonical(u128 v) { return n(v % P()); }
    static WF_HD bool is_valid(T x) { return w(x) < P(); }
};

// ------------------------------------------------------------------------------------------------ generic helpers
template <class F>
WF_HD typename F::T f_pow(typename F::T b, u128 e) {
    typename F::T r = F::one();
    while (e) {
        if (e & 1) r = F::mul(r, b);
        b = F::mul(b, b);
        e >>= 1;
    }
    return r;
}

template <class F>
struct FieldInfo;
template <>
struct FieldInfo<F64> {
    static u128 modulus() { return (u128)F64::P; }
    static F64::T two_adic_root() { return F64::from_canonical(F64::TWO_ADIC_ROOT); }
};
template <>
struct FieldInfo<F128> {
    static u128 modulus() { return F128::P(); }
    // 23953097886125630542083529559205016746 (order 2^40), math/src/field/f128/mod.rs:38
    static F128::T two_adic_root() { return F128::T{0x86B8723E1920F4AAull, 0x120532E7B364080Aull}; }
};

template <class F>
inline typename F::T f_inv(typename F::T x) {
    return f_pow<F>(x, FieldInfo<F>::modulus() - 2);
}
// root of unity of order 2^n (math/src/field/traits.rs:254-263)
template <class F>
inline typename F::T f_root_of_unity(uint32_t n) {
    return f_pow<F>(FieldInfo<F>::two_adic_root(), (u128)1 << (F::TWO_ADICITY - n));
}

}  // namespace wf
