// Bring-up microbenchmarks (not part of the product): integer-VALU cost of the field operations and of one
// BLAKE3 compression on gfx950, and the cost of the LDE's scattered 64-byte row writes.
//   hipcc -O3 --offload-arch=gfx950 -I starkpack-winterfell_amd/csrc scripts/microbench.hip -o /tmp/microbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#include "blake3_dev.hpp"
#include "field.hpp"

using namespace wf;

#define CHECK(x)                                                              \
    do {                                                                      \
        hipError_t e = (x);                                                   \
        if (e != hipSuccess) {                                                \
            printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__);   \
            return 1;                                                         \
        }                                                                     \
    } while (0)

// ---- variants of the Goldilocks Montgomery product -------------------------------------------------------------
__device__ __forceinline__ uint64_t mul_v0(uint64_t a, uint64_t b) { return F64::mul(a, b); }

// 32-bit limb schoolbook with explicit mad_u64_u32 shapes, then the same reduction
__device__ __forceinline__ uint64_t mul_v1(uint64_t a, uint64_t b) {
    uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    uint64_t p00 = (uint64_t)a0 * b0;
    uint64_t p01 = (uint64_t)a0 * b1 + (p00 >> 32);
    uint64_t p10 = (uint64_t)a1 * b0 + (uint32_t)p01;
    uint64_t p11 = (uint64_t)a1 * b1 + (p01 >> 32) + (p10 >> 32);
    uint64_t lo = (p10 << 32) | (uint32_t)p00;
    return F64::mont_reduce(lo, p11);
}

// reduction written on 32-bit halves: x = xl + 2^64 xh ; result = xh - (xl + (xl<<32) - ((xl + (xl<<32))>>32) - carry)
__device__ __forceinline__ uint64_t mul_v2(uint64_t a, uint64_t b) {
    uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    uint64_t p00 = (uint64_t)a0 * b0;
    uint64_t p01 = (uint64_t)a0 * b1 + (p00 >> 32);
    uint64_t p10 = (uint64_t)a1 * b0 + (uint32_t)p01;
    uint64_t xh = (uint64_t)a1 * b1 + (p01 >> 32) + (p10 >> 32);
    uint32_t l0 = (uint32_t)p00, l1 = (uint32_t)p10;  // xl = l1:l0
    // a = xl + (xl << 32) = (l1 + l0) : l0  with carry e out of the high word
    uint32_t ah = l1 + l0;
    uint32_t e = ah < l1;
    // b = a - (a >> 32) - e = (ah : l0) - ah - e
    uint64_t av = ((uint64_t)ah << 32) | l0;
    uint64_t bb = av - ah - e;
    uint64_t r = xh - bb;
    return xh < bb ? r - 0xFFFFFFFFull : r;
}

template <int VAR>
__device__ __forceinline__ uint64_t mulv(uint64_t a, uint64_t b) {
    if (VAR == 0) return mul_v0(a, b);
    if (VAR == 1) return mul_v1(a, b);
    if (VAR == 3) return F64::mul_pow2<12>(a);   // shift twiddles of the radix-16 block
    if (VAR == 4) return F64::div_pow2<24>(a);
    return mul_v2(a, b);
}

template <int VAR>
__global__ void k_mul(uint64_t *io, int iters) {
    uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t a = io[tid] | 1, b = a * 3 + 7, c = a ^ 0x1234567, d = a + 99;
    uint64_t w = 0x0123456789ABCDEFull % F64::P;
    for (int i = 0; i < iters; i++) {
        a = mulv<VAR>(a, w);
        b = mulv<VAR>(b, w);
        c = mulv<VAR>(c, w);
        d = mulv<VAR>(d, w);
        w += 2;
    }
    io[tid] = a ^ b ^ c ^ d;
}

__global__ void k_addsub(uint64_t *io, int iters) {
    uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t a = io[tid] % F64::P, b = (a * 3 + 7) % F64::P, c = (a ^ 0x1234567) % F64::P, d = (a + 99) % F64::P;
    for (int i = 0; i < iters; i++) {
        uint64_t t = F64::add(a, b);
        b = F64::sub(a, b);
        a = t;
        t = F64::add(c, d);
        d = F64::sub(c, d);
        c = t;
    }
    io[tid] = a ^ b ^ c ^ d;
}

// butterfly: (a, b) -> (a + w b, a - w b)
template <int VAR>
__global__ void k_bfly(uint64_t *io, int iters) {
    uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t a = io[tid] % F64::P, b = (a * 3 + 7) % F64::P, c = (a ^ 0x1234567) % F64::P, d = (a + 99) % F64::P;
    uint64_t w = 0x0123456789ABCDEFull % F64::P;
    for (int i = 0; i < iters; i++) {
        uint64_t t = mulv<VAR>(b, w);
        b = F64::sub(a, t);
        a = F64::add(a, t);
        t = mulv<VAR>(d, w);
        d = F64::sub(c, t);
        c = F64::add(c, t);
        w += 2;
    }
    io[tid] = a ^ b ^ c ^ d;
}

__global__ void k_blake(uint32_t *io, int iters) {
    uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint32_t m[16], cv[8];
    for (int i = 0; i < 16; i++) m[i] = io[tid] + i * 0x9E3779B9u;
    b3::set_iv(cv);
    for (int i = 0; i < iters; i++) {
        b3::compress(cv, m, 0, 0, 64, 11);
        m[i & 15] ^= cv[0];
    }
    io[tid] = cv[0] ^ cv[7];
}

// scattered 64-byte chunks: chunk q of n goes to position perm(q); stride pattern of the LDE last pass
__global__ void k_write(uint4 *dst, uint64_t n_chunks, uint64_t stride_chunks, int mode) {
    // each lane writes 16 B; 4 lanes = one 64-byte chunk
    uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t q = g >> 2, part = g & 3;
    if (q >= n_chunks) return;
    uint64_t pos;
    if (mode == 0) {
        pos = q;  // contiguous
    } else {
        // q = k2 * groups + k1 ... consecutive chunks of one work-group are `stride_chunks` apart
        uint64_t per = n_chunks / stride_chunks;  // chunks per "row"
        uint64_t r = q / per, c = q % per;
        pos = c * stride_chunks + r;
    }
    dst[pos * 4 + part] = make_uint4((uint32_t)q, (uint32_t)part, 3, 4);
}

template <class K>
static float timeit(K launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const int blocks = 256 * 16, threads = 256, iters = 2000;
    const uint64_t nthreads = (uint64_t)blocks * threads;
    uint64_t *io;
    CHECK(hipMalloc(&io, nthreads * 8));
    std::vector<uint64_t> h(nthreads);
    for (uint64_t i = 0; i < nthreads; i++) h[i] = i * 0x9E3779B97F4A7C15ull;
    CHECK(hipMemcpy(io, h.data(), nthreads * 8, hipMemcpyHostToDevice));

    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_mul<0>, dim3(blocks), dim3(threads), 0, 0, io, iters); }, 5);
    printf("modmul v0 (umul64hi)      : %8.1f Gmul/s\n", nthreads * 4.0 * iters / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_mul<1>, dim3(blocks), dim3(threads), 0, 0, io, iters); }, 5);
    printf("modmul v1 (mad_u64_u32)   : %8.1f Gmul/s\n", nthreads * 4.0 * iters / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_mul<2>, dim3(blocks), dim3(threads), 0, 0, io, iters); }, 5);
    printf("modmul v2 (32-bit reduce) : %8.1f Gmul/s\n", nthreads * 4.0 * iters / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_addsub, dim3(blocks), dim3(threads), 0, 0, io, iters); }, 5);
    printf("add+sub pairs             : %8.1f Gpair/s\n", nthreads * 2.0 * iters / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_bfly<0>, dim3(blocks), dim3(threads), 0, 0, io, iters); }, 5);
    printf("butterfly v0              : %8.1f Gbfly/s\n", nthreads * 2.0 * iters / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_bfly<1>, dim3(blocks), dim3(threads), 0, 0, io, iters); }, 5);
    printf("butterfly v1              : %8.1f Gbfly/s\n", nthreads * 2.0 * iters / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_bfly<2>, dim3(blocks), dim3(threads), 0, 0, io, iters); }, 5);
    printf("butterfly v2              : %8.1f Gbfly/s\n", nthreads * 2.0 * iters / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_bfly<3>, dim3(blocks), dim3(threads), 0, 0, io, iters); }, 5);
    printf("butterfly, w = 2^12 shift : %8.1f Gbfly/s\n", nthreads * 2.0 * iters / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_bfly<4>, dim3(blocks), dim3(threads), 0, 0, io, iters); }, 5);
    printf("butterfly, w = 2^-24 shift: %8.1f Gbfly/s\n", nthreads * 2.0 * iters / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_blake, dim3(blocks), dim3(threads), 0, 0, (uint32_t *)io, 200); }, 5);
    printf("blake3 compress           : %8.2f Gcompress/s\n", nthreads * 200.0 / ms / 1e6);

    // write patterns: 512 MiB of 64-byte chunks
    const uint64_t n_chunks = (512ull << 20) / 64;
    uint4 *dst;
    CHECK(hipMalloc(&dst, n_chunks * 64));
    const uint64_t g = n_chunks * 4;
    for (uint64_t stride : {0ull, 8ull, 64ull, 1024ull, 8192ull}) {
        int mode = stride ? 1 : 0;
        ms = timeit([&] { hipLaunchKernelGGL(k_write, dim3((uint32_t)(g / 256)), dim3(256), 0, 0, dst, n_chunks,
                                             stride ? stride : 1, mode); }, 5);
        printf("write 512 MiB, 64-B chunks, consecutive chunks %6llu apart: %7.3f ms = %6.2f TB/s\n",
               (unsigned long long)stride, ms, 512.0 / 1024 / 1024 / ms * 1e3 * 1.048576);
    }
    return 0;
}
