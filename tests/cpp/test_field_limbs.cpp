// CPU check of the 32-bit-limb f128 arithmetic that the device code uses (csrc/field.hpp: add_limbs / sub_limbs /
// mul_limbs) against the wide unsigned __int128 formulation and against a bit-serial double-and-add product.
// Edge patterns: values next to p, next to 0, all-ones / all-zero limbs, single bits.
#include <stdio.h>
#include <stdlib.h>

#include "../../starkpack-winterfell_amd/csrc/field.hpp"

typedef unsigned __int128 u128;
using wf::F128;
using wf::U128;

static u128 P;
static u128 addmod(u128 a, u128 b) {
    u128 s = a + b;
    if (s < a || s >= P) s -= P;
    return s;
}
static u128 mulmod(u128 a, u128 b) {
    u128 r = 0;
    for (int i = 127; i >= 0; i--) {
        r = addmod(r, r);
        if ((b >> i) & 1) r = addmod(r, a);
    }
    return r;
}
static uint64_t rs = 88172645463325252ull;
static uint64_t rnd() {
    rs ^= rs << 13;
    rs ^= rs >> 7;
    rs ^= rs << 17;
    return rs;
}
static u128 pattern(long mode) {
    u128 x = ((u128)rnd() << 64) | rnd();
    switch (mode % 6) {
        case 0: return x % P;
        case 1: return P - 1 - (rnd() % 1000);
        case 2: return rnd() % 1000;
        case 3: return ((x % P) | (((u128)0xFFFFFFFFull) << (32 * (rnd() % 3)))) % P;
        case 4: return ((P - 1) & ~(((u128)0xFFFFFFFFull) << (32 * (rnd() % 4)))) % P;
        default: return ((u128)1 << (rnd() % 128)) % P;
    }
}
static U128 n(u128 v) { return U128{(uint64_t)v, (uint64_t)(v >> 64)}; }
static u128 w(U128 v) { return ((u128)v.hi << 64) | v.lo; }

// x * 2^E mod p by repeated doubling on 128-bit integers (2^192 = 1: any E in [0, 192))
static uint64_t shl_mod(uint64_t x, int e) {
    u128 v = x;
    const u128 P64 = wf::F64::P;
    for (int i = 0; i < e; i++) v = (v << 1) % P64;
    return (uint64_t)v;
}
template <int E>
static long chk_pow2_192(uint64_t x) {
    long bad = 0;
    if constexpr (E < 192) {
        const uint64_t want = shl_mod(x, E);
        if (wf::F64::mul_pow2_192<E>(x) != want) bad++;
        // the form the transform rounds use for (96, 128]: the caller hands in -x, the shift is E - 96
        if constexpr (E > 96 && E <= 128) {
            const uint64_t nx = x ? wf::F64::P - x : 0;
            if (wf::F64::mul_pow2<E - 96>(nx) != want) bad++;
        }
        bad += chk_pow2_192<E + 1>(x);
    }
    return bad;
}

int main(int argc, char **argv) {
    const long iters = argc > 1 ? atol(argv[1]) : 400000;
    P = F128::P();
    long bad = 0;
    for (long it = 0; it < iters; it++) {
        u128 a = pattern(it), b = pattern(it / 6);
        if (it == 0) a = b = P - 1;
        if (it == 1) { a = 0; b = P - 1; }
        const u128 m = w(F128::mul_limbs(n(a), n(b)));
        if (m != w(F128::mul_wide(n(a), n(b))) || m != mulmod(a, b)) bad++;
        if (w(F128::add_limbs(n(a), n(b))) != addmod(a, b)) bad++;
        const u128 d = a >= b ? a - b : a + (P - b);
        if (w(F128::sub_limbs(n(a), n(b))) != d) bad++;
    }
    // Goldilocks shift twiddles (F64::mul_pow2 / div_pow2) against the general product by the same constant
    {
        using wf::F64;
        const u128 P64 = F64::P;
        auto chk = [&](uint64_t x) {
            const uint64_t e12 = (uint64_t)(((u128)x << 12) % P64), e24 = (uint64_t)(((u128)x << 24) % P64);
            const uint64_t e32 = (uint64_t)(((u128)x << 32) % P64);
            if (F64::mul_pow2<12>(x) != e12 || F64::mul_pow2<24>(x) != e24 || F64::mul_pow2<32>(x) != e32) bad++;
            // the word-aligned middle range (K > 32: 2^36, 2^48 = w_4, 2^60 of the radix-16 block, and the range's ends)
            if (F64::mul_pow2<33>(x) != (uint64_t)((((u128)x << 33) % P64)) || F64::mul_pow2<36>(x) != (uint64_t)(((u128)x << 36) % P64) ||
                F64::mul_pow2<48>(x) != (uint64_t)(((u128)x << 48) % P64) || F64::mul_pow2<60>(x) != (uint64_t)(((u128)x << 60) % P64) ||
                F64::mul_pow2<63>(x) != (uint64_t)(((u128)x << 63) % P64))
                bad++;
            const uint64_t d12 = F64::div_pow2<12>(x), d24 = F64::div_pow2<24>(x), d32 = F64::div_pow2<32>(x);
            if ((uint64_t)(((u128)d12 << 12) % P64) != x || d12 >= F64::P) bad++;
            if ((uint64_t)(((u128)d24 << 24) % P64) != x || d24 >= F64::P) bad++;
            if ((uint64_t)(((u128)d32 << 32) % P64) != x || d32 >= F64::P) bad++;
        };
        const uint64_t edge[] = {0, 1, 2, 0xFFF, 0x1000, 0xFFFFFF, 0x1000000, 0xFFFFFFFFull, 0x100000000ull,
                                 0xFFFFFFFF00000000ull, 0xFFFFFFFEFFFFFFFFull, 0xFFFFFFFF00000000ull - 1, F64::P - 1,
                                 F64::P - 2, F64::P - 0x1000, 0xFFFFF00000000000ull, 0x000FFFFFFFFFFFFFull,
                                 0xFFF0000000000001ull % F64::P, 0x8000000000000000ull, 0x7FFFFFFFFFFFFFFFull,
                                 0x0000FFFFFFFFFFFFull, 0x0000FFFF0000FFFFull, 0xFFFF0000FFFFFFFFull % F64::P, 0x00000000FFFF0000ull,
                                 0x0FFFFFFFFFFFFFFFull, 0xFFFFFFFFull << 4, 0xFFFFFFFFull << 16, 0xFFFFFFFFull << 28};
        for (uint64_t e : edge) chk(e % F64::P);
        for (long it = 0; it < iters; it++) chk(rnd() % F64::P);
        // every exponent of the order-192 group (the wave-uniform twiddles w_64^(j k) = 2^(3 j k), w_32^k = 2^(6 k), w_8^k = 2^(24 k)
        // and their inverses are among them)
        for (uint64_t e : edge) bad += chk_pow2_192<0>(e % F64::P);
        for (long it = 0; it < iters / 100 + 1000; it++) bad += chk_pow2_192<0>(rnd() % F64::P);
    }
    printf("checked %ld triples, bad=%ld\n", iters, bad);
    if (bad == 0) printf("ALL OK\n");
    return bad != 0;
}
