// Bring-up experiment (not part of the product): Goldilocks butterflies with the carry chains of W independent values issued in
// LOCKSTEP -- every step of the 64-bit modular arithmetic (add_co, addc, subb, cndmask ..) is written for W values, with a scheduling
// barrier behind it, so that the VALU that reads a carry is W - 1 instructions behind the VALU that wrote it and the hazard recogniser
// has no s_nop to insert (gfx950: two wait states between a VALU that writes VCC / an SGPR and the VALU that reads it).
// Against the plain butterflies of scripts/occupancy_rate.hip (17-23 % of the issued instructions are s_nop).
//   hipcc -O3 --offload-arch=gfx950 -I starkpack-winterfell_amd/csrc scripts/lockstep_bfly.hip -o build/lockstep_bfly
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "field.hpp"

using namespace wf;

#define STEP_BARRIER() __builtin_amdgcn_sched_barrier(0)

template <int W>
struct Lock {
    // r = a + b mod p, W values in lockstep (F64::add: r = a + b; fix if it wrapped or landed in [p, 2^64))
    static __device__ __forceinline__ void add(const uint64_t (&a)[W], const uint64_t (&b)[W], uint64_t (&r)[W]) {
        uint32_t lo[W], hi[W], c[W], c2[W];
#pragma unroll
        for (int i = 0; i < W; i++) c[i] = __builtin_add_overflow((uint32_t)a[i], (uint32_t)b[i], &lo[i]);
        STEP_BARRIER();
#pragma unroll
        for (int i = 0; i < W; i++) {
            uint32_t s;
            const uint32_t k1 = __builtin_add_overflow((uint32_t)(a[i] >> 32), (uint32_t)(b[i] >> 32), &s);
            const uint32_t k2 = __builtin_add_overflow(s, c[i], &hi[i]);
            c2[i] = k1 | k2;
        }
        STEP_BARRIER();
        // wrapped, or >= p = ffffffff00000001: hi == ffffffff and lo >= 1
        uint32_t m[W];
#pragma unroll
        for (int i = 0; i < W; i++) m[i] = (c2[i] | ((hi[i] == 0xFFFFFFFFu) & (lo[i] != 0u))) ? 0xFFFFFFFFu : 0u;
        STEP_BARRIER();
        // + (2^32 - 1) = - p mod 2^64:  lo += ffffffff (carry), hi += carry
#pragma unroll
        for (int i = 0; i < W; i++) c[i] = __builtin_add_overflow(lo[i], m[i], &lo[i]);
        STEP_BARRIER();
#pragma unroll
        for (int i = 0; i < W; i++) r[i] = ((uint64_t)(hi[i] + c[i]) << 32) | lo[i];
        STEP_BARRIER();
    }
    static __device__ __forceinline__ void sub(const uint64_t (&a)[W], const uint64_t (&b)[W], uint64_t (&r)[W]) {
        uint32_t rl[W], rh[W], b1[W], bw[W];
#pragma unroll
        for (int i = 0; i < W; i++) b1[i] = __builtin_sub_overflow((uint32_t)a[i], (uint32_t)b[i], &rl[i]);
        STEP_BARRIER();
#pragma unroll
        for (int i = 0; i < W; i++) {
            uint32_t t;
            const uint32_t k1 = __builtin_sub_overflow((uint32_t)(a[i] >> 32), (uint32_t)(b[i] >> 32), &t);
            const uint32_t k2 = __builtin_sub_overflow(t, b1[i], &rh[i]);
            bw[i] = 0u - (k1 | k2);
        }
        STEP_BARRIER();
#pragma unroll
        for (int i = 0; i < W; i++) b1[i] = __builtin_sub_overflow(rl[i], bw[i], &rl[i]);
        STEP_BARRIER();
#pragma unroll
        for (int i = 0; i < W; i++) r[i] = ((uint64_t)(rh[i] - b1[i]) << 32) | rl[i];
        STEP_BARRIER();
    }
    static __device__ __forceinline__ void mul(const uint64_t (&a)[W], uint64_t b, uint64_t (&r)[W]) {
        const uint32_t b0 = (uint32_t)b, bh = (uint32_t)(b >> 32);
        uint64_t p00[W], p01[W], p10[W], hi[W];
#pragma unroll
        for (int i = 0; i < W; i++) p00[i] = (uint64_t)(uint32_t)a[i] * b0;
        STEP_BARRIER();
#pragma unroll
        for (int i = 0; i < W; i++) p01[i] = (uint64_t)(uint32_t)a[i] * bh + (p00[i] >> 32);
        STEP_BARRIER();
#pragma unroll
        for (int i = 0; i < W; i++) p10[i] = (uint64_t)(uint32_t)(a[i] >> 32) * b0 + (uint32_t)p01[i];
        STEP_BARRIER();
#pragma unroll
        for (int i = 0; i < W; i++) hi[i] = (uint64_t)(uint32_t)(a[i] >> 32) * bh + (p01[i] >> 32) + (p10[i] >> 32);
        STEP_BARRIER();
        // mont_reduce(lo, hi), field.hpp, step by step
        uint32_t l0[W], l1[W], ah[W], e[W], t[W], bl[W], bhh[W], k1[W], rl[W], u[W], rh[W], c1[W], c2[W];
#pragma unroll
        for (int i = 0; i < W; i++) {
            l0[i] = (uint32_t)p00[i];
            l1[i] = (uint32_t)p10[i];
            e[i] = __builtin_add_overflow(l1[i], l0[i], &ah[i]);
        }
        STEP_BARRIER();
#pragma unroll
        for (int i = 0; i < W; i++) k1[i] = __builtin_sub_overflow(l0[i], ah[i], &t[i]);
        STEP_BARRIER();
#pragma unroll
        for (int i = 0; i < W; i++) {
            const uint32_t k2 = __builtin_sub_overflow(t[i], e[i], &bl[i]);
            bhh[i] = ah[i] - (k1[i] | k2);
        }
        STEP_BARRIER();
#pragma unroll
        for (int i = 0; i < W; i++) c1[i] = __builtin_sub_overflow((uint32_t)hi[i], bl[i], &rl[i]);
        STEP_BARRIER();
#pragma unroll
        for (int i = 0; i < W; i++) {
            const uint32_t c2a = __builtin_sub_overflow((uint32_t)(hi[i] >> 32), bhh[i], &u[i]);
            const uint32_t c3 = __builtin_sub_overflow(u[i], c1[i], &rh[i]);
            c2[i] = 0u - (c2a | c3);
        }
        STEP_BARRIER();
#pragma unroll
        for (int i = 0; i < W; i++) c1[i] = __builtin_sub_overflow(rl[i], c2[i], &rl[i]);
        STEP_BARRIER();
#pragma unroll
        for (int i = 0; i < W; i++) r[i] = ((uint64_t)(rh[i] - c1[i]) << 32) | rl[i];
        STEP_BARRIER();
    }
};

template <int W, bool LOCK>
__global__ void __launch_bounds__(256) k_bfly(uint64_t *io, int iters) {
    extern __shared__ unsigned char pad[];
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t a[W], b[W];
#pragma unroll
    for (int q = 0; q < W; q++) {
        a[q] = (io[tid] + 977 * q) % F64::P;
        b[q] = (io[tid] * 3 + 7 + 131 * q) % F64::P;
    }
    uint64_t w = 0x0123456789ABCDEFull % F64::P;
    for (int i = 0; i < iters; i++) {
        if constexpr (LOCK) {
            uint64_t t[W], s[W], d[W];
            Lock<W>::mul(b, w, t);
            Lock<W>::sub(a, t, d);
            Lock<W>::add(a, t, s);
#pragma unroll
            for (int q = 0; q < W; q++) {
                a[q] = s[q];
                b[q] = d[q];
            }
        } else {
#pragma unroll
            for (int q = 0; q < W; q++) {
                const uint64_t t = F64::mul(b[q], w);
                b[q] = F64::sub(a[q], t);
                a[q] = F64::add(a[q], t);
            }
        }
        w += 2;
    }
    if (iters < 0) pad[threadIdx.x] = 1;
    uint64_t x = 0;
#pragma unroll
    for (int q = 0; q < W; q++) x ^= a[q] ^ b[q];
    io[tid] = x;
}

template <class T>
static double run(void (*kern)(T *, int), void *io, int cus, int k, int iters) {
    const size_t lds = (size_t)(160 * 1024 / k) & ~(size_t)255;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(cus * k), dim3(256), lds, 0, (T *)io, iters);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(cus * k), dim3(256), lds, 0, (T *)io, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    uint64_t *io;
    const size_t n = (size_t)cus * 8 * 256;
    (void)hipMalloc(&io, n * 8);
    // same inputs for both forms: the outputs must agree
    uint64_t *h = (uint64_t *)malloc(n * 8), *h1 = (uint64_t *)malloc(n * 8), *h2 = (uint64_t *)malloc(n * 8);
    for (size_t i = 0; i < n; i++) h[i] = 0x9E3779B97F4A7C15ull * (i + 1);
    (void)hipMemcpy(io, h, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_bfly<4, false>), dim3(cus * 8), dim3(256), 0, 0, io, 50);
    (void)hipMemcpy(h1, io, n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(io, h, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_bfly<4, true>), dim3(cus * 8), dim3(256), 0, 0, io, 50);
    (void)hipMemcpy(h2, io, n * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) bad += h1[i] != h2[i];
    printf("# lockstep form against the plain one on %zu lanes x 4 butterflies x 50 iterations: %zu mismatches\n", n, bad);
    printf("# waves/SIMD | plain W=4: ms, G bfly/s | lockstep W=4 | plain W=8 | lockstep W=8\n");
    for (int k : {2, 4, 8}) {
        const int it = 1500;
        const double a = run(k_bfly<4, false>, io, cus, k, it), b = run(k_bfly<4, true>, io, cus, k, it);
        const double c = run(k_bfly<8, false>, io, cus, k, it / 2), d = run(k_bfly<8, true>, io, cus, k, it / 2);
        const double lanes = (double)cus * k * 256;
        printf("%d | %7.3f ms %8.1f G/s | %7.3f ms %8.1f G/s | %7.3f ms %8.1f G/s | %7.3f ms %8.1f G/s\n", k, a, lanes * 4 * it / (a * 1e-3) / 1e9, b,
               lanes * 4 * it / (b * 1e-3) / 1e9, c, lanes * 8 * (it / 2) / (c * 1e-3) / 1e9, d, lanes * 8 * (it / 2) / (d * 1e-3) / 1e9);
    }
    return bad != 0;
}
