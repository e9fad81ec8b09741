/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement of the reference's 128-bit base field
 *   p = 2^128 - 45 * 2^40 + 1, elements are canonical integers in [0,p) (u128, little endian).
 * Follows /root/reference/math/src/field/f128/mod.rs:
 *   M, G                       :35-38
 *   add / sub / mul            :418-475
 *   mul_128x64 .. add64_with_carry :577-620
 *   GENERATOR=3, TWO_ADICITY=40, TWO_ADIC_ROOT_OF_UNITY=G :165-174
 */
#ifndef ORACLE_FIELD_F128_H
#define ORACLE_FIELD_F128_H

#include <stdint.h>

typedef unsigned __int128 f128e;

#define F128_M ((((f128e)0xFFFFFFFFFFFFFFFFULL) << 64) + (f128e)0xFFFFD30000000001ULL) /* 2^128 - 45*2^40 + 1 */
#define F128_GENERATOR_INT ((f128e)3)
#define F128_TWO_ADICITY 40u
/* G = 23953097886125630542083529559205016746 */
#define F128_TWO_ADIC_ROOT ((((f128e)0x120532E7B364080AULL) << 64) | (f128e)0x86B8723E1920F4AAULL)

/* f128/mod.rs:418-425 */
static inline f128e f128_add(f128e a, f128e b) {
    f128e z = F128_M - b;
    return a < z ? F128_M - z + a : a - z;
}

/* f128/mod.rs:428-434 */
static inline f128e f128_sub(f128e a, f128e b) { return a < b ? F128_M - b + a : a - b; }

/* helpers, f128/mod.rs:577-620 */
static inline void f128_mul_128x64(f128e a, uint64_t b, uint64_t *z0, uint64_t *z1, uint64_t *z2) {
    f128e z_lo = (f128e)(uint64_t)a * (f128e)b;
    f128e z_hi = (a >> 64) * (f128e)b;
    z_hi = z_hi + (z_lo >> 64);
    *z0 = (uint64_t)z_lo;
    *z1 = (uint64_t)z_hi;
    *z2 = (uint64_t)(z_hi >> 64);
}

static inline void f128_sub_192(uint64_t a0, uint64_t a1, uint64_t a2, uint64_t b0, uint64_t b1, uint64_t b2,
                                uint64_t *r0, uint64_t *r1, uint64_t *r2) {
    f128e z0 = (f128e)a0 - (f128e)b0;
    f128e z1 = (f128e)a1 - ((f128e)b1 + (z0 >> 127));
    f128e z2 = (f128e)a2 - ((f128e)b2 + (z1 >> 127));
    *r0 = (uint64_t)z0;
    *r1 = (uint64_t)z1;
    *r2 = (uint64_t)z2;
}

/* x - (x >> 128) * m   (mul_reduce + mul_by_modulus) */
static inline void f128_mul_reduce(uint64_t *z0, uint64_t *z1, uint64_t *z2) {
    uint64_t a = *z2;
    f128e a_lo = (f128e)a * F128_M; /* wrapping */
    uint64_t a_hi = a == 0 ? 0 : a - 1;
    f128_sub_192(*z0, *z1, *z2, (uint64_t)a_lo, (uint64_t)(a_lo >> 64), a_hi, z0, z1, z2);
}

static inline void f128_sub_modulus(uint64_t *lo, uint64_t *hi) {
    f128e z = (f128e)0 - F128_M;
    z += (f128e)*lo;
    z += ((f128e)*hi) << 64;
    *lo = (uint64_t)z;
    *hi = (uint64_t)(z >> 64);
}

/* f128/mod.rs:437-475 */
static inline f128e f128_mul(f128e a, f128e b) {
    uint64_t x0, x1, x2;
    f128_mul_128x64(a, (uint64_t)(b >> 64), &x0, &x1, &x2);
    f128_mul_reduce(&x0, &x1, &x2);
    if (x2 == 1) f128_sub_modulus(&x0, &x1);

    uint64_t y0, y1, y2;
    f128_mul_128x64(a, (uint64_t)b, &y0, &y1, &y2);

    f128e t = (f128e)y1 + (f128e)x0;
    y1 = (uint64_t)t;
    t = (f128e)y2 + (f128e)x1 + (t >> 64);
    y2 = (uint64_t)t;
    uint64_t y3 = (uint64_t)(t >> 64);
    if (y3 == 1) f128_sub_modulus(&y1, &y2);

    uint64_t z0 = y0, z1 = y1, z2 = y2;
    f128_mul_reduce(&z0, &z1, &z2);
    if (z2 == 1 || (z1 == (uint64_t)(F128_M >> 64) && z0 >= (uint64_t)F128_M)) f128_sub_modulus(&z0, &z1);
    return (((f128e)z1) << 64) + (f128e)z0;
}

static inline f128e f128_exp(f128e base, f128e power) {
    f128e r = 1, b = base;
    while (power) {
        if (power & 1) r = f128_mul(r, b);
        b = f128_mul(b, b);
        power >>= 1;
    }
    return r;
}

static inline f128e f128_inv(f128e x) { return f128_exp(x, F128_M - 2); }

static inline f128e f128_root_of_unity(uint32_t n) {
    return f128_exp(F128_TWO_ADIC_ROOT, ((f128e)1) << (F128_TWO_ADICITY - n));
}

#endif
