/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement of the reference's 64-bit "Goldilocks" base field
 *   p = 2^64 - 2^32 + 1, elements kept in Montgomery form (x * 2^64 mod p, fully reduced).
 * Follows /root/reference/math/src/field/f64/mod.rs:
 *   M, R2                      :37-40
 *   new()                      :57-59
 *   as_int()                   :275-282
 *   add / sub / mul            :314-366
 *   mont_red_cst()             :651-661
 *   GENERATOR=7, TWO_ADICITY=32, TWO_ADIC_ROOT_OF_UNITY :248-264
 *   get_root_of_unity          math/src/field/traits.rs:254-263
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 */
#ifndef ORACLE_FIELD_F64_H
#define ORACLE_FIELD_F64_H

#include <stdint.h>

typedef unsigned __int128 orc_u128;

#define F64_M 0xFFFFFFFF00000001ULL
#define F64_R2 0xFFFFFFFE00000001ULL
#define F64_GENERATOR_INT 7ULL
#define F64_TWO_ADICITY 32u
#define F64_TWO_ADIC_ROOT_INT 7277203076849721926ULL

/* f64/mod.rs:651-661 */
static inline uint64_t f64_mont_red_cst(orc_u128 x) {
    uint64_t xl = (uint64_t)x;
    uint64_t xh = (uint64_t)(x >> 64);
    uint64_t a = xl + (xl << 32);
    uint64_t e = a < xl; /* carry out of xl + (xl << 32) */
    uint64_t b = a - (a >> 32) - e;
    uint64_t r = xh - b;
    uint64_t c = xh < b; /* borrow */
    return r - (uint64_t)(0u - (uint32_t)c);
}

/* f64/mod.rs:57-59 : canonical integer -> Montgomery */
static inline uint64_t f64_new(uint64_t v) { return f64_mont_red_cst((orc_u128)v * (orc_u128)F64_R2); }

/* f64/mod.rs:275-282 : Montgomery -> canonical integer in [0,p) */
static inline uint64_t f64_as_int(uint64_t x) {
    uint64_t a = x + (x << 32);
    uint64_t e = a < x;
    uint64_t b = a - (a >> 32) - e;
    uint64_t r = 0 - b;
    uint64_t c = 0 < b;
    return r - (uint64_t)(0u - (uint32_t)c);
}

/* f64/mod.rs:314-325 */
static inline uint64_t f64_add(uint64_t a, uint64_t b) {
    uint64_t t = F64_M - b;
    uint64_t x1 = a - t;
    uint64_t c1 = a < t;
    return x1 - (uint64_t)(0u - (uint32_t)c1);
}

/* f64/mod.rs:334-344 */
static inline uint64_t f64_sub(uint64_t a, uint64_t b) {
    uint64_t x1 = a - b;
    uint64_t c1 = a < b;
    return x1 - (uint64_t)(0u - (uint32_t)c1);
}

/* f64/mod.rs:353-360 */
static inline uint64_t f64_mul(uint64_t a, uint64_t b) { return f64_mont_red_cst((orc_u128)a * (orc_u128)b); }

static inline uint64_t f64_exp(uint64_t base, uint64_t power) {
    uint64_t r = f64_new(1), b = base;
    while (power) {
        if (power & 1) r = f64_mul(r, b);
        b = f64_mul(b, b);
        power >>= 1;
    }
    return r;
}

/* inverse of 0 is 0 (f64/mod.rs inv contract) */
static inline uint64_t f64_inv(uint64_t x) { return f64_exp(x, F64_M - 2); }

/* traits.rs:254-263 */
static inline uint64_t f64_root_of_unity(uint32_t n) {
    return f64_exp(f64_new(F64_TWO_ADIC_ROOT_INT), 1ULL << (F64_TWO_ADICITY - n));
}

#endif
