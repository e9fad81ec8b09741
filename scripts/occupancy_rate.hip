// Bring-up microbenchmark (not part of the product): the issue rate of the path's own instruction sequences against the number of
// waves per SIMD -- Goldilocks butterflies with a general twiddle (27 vector instructions, carry chains and v_mad_u64_u32: ~20 % of the
// issued instructions are the hazard s_nops gfx950 needs between a VALU that writes VCC / an SGPR and the VALU that reads it) and
// BLAKE3 compressions (no carries).  Register-only loops, every wave runs the whole loop; occupancy is set by the grid (k work-groups of
// 256 threads per CU = k waves per SIMD) and held there by dynamic LDS (160 KiB / k per work-group).
//   hipcc -O3 --offload-arch=gfx950 -I starkpack-winterfell_amd/csrc scripts/occupancy_rate.hip -o /tmp/occupancy_rate && /tmp/occupancy_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "blake3_dev.hpp"
#include "field.hpp"

using namespace wf;

__global__ void __launch_bounds__(256) k_bfly(uint64_t *io, int iters) {
    extern __shared__ unsigned char pad[];
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t a = io[tid] % F64::P, b = (a * 3 + 7) % F64::P, c = (a ^ 0x1234567) % F64::P, d = (a + 99) % F64::P;
    uint64_t w = 0x0123456789ABCDEFull % F64::P;
    for (int i = 0; i < iters; i++) {
        uint64_t t = F64::mul(b, w);
        b = F64::sub(a, t);
        a = F64::add(a, t);
        t = F64::mul(d, w);
        d = F64::sub(c, t);
        c = F64::add(c, t);
        w += 2;
    }
    if (iters < 0) pad[threadIdx.x] = 1;  // (keeps the dynamic LDS allocation alive)
    io[tid] = a ^ b ^ c ^ d;
}

// the same butterflies on 16 independent values per thread (the instruction-level parallelism a register-resident radix-16 / 32 round has)
__global__ void __launch_bounds__(256) k_bfly_ilp(uint64_t *io, int iters) {
    extern __shared__ unsigned char pad[];
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t v[16];
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = (io[tid] + 977 * q) % F64::P;
    uint64_t w = 0x0123456789ABCDEFull % F64::P;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint64_t t = F64::mul(v[q + 8], w);
            v[q + 8] = F64::sub(v[q], t);
            v[q] = F64::add(v[q], t);
        }
        w += 2;
    }
    if (iters < 0) pad[threadIdx.x] = 1;
    uint64_t x = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) x ^= v[q];
    io[tid] = x;
}

__global__ void __launch_bounds__(256) k_blake(uint32_t *io, int iters) {
    extern __shared__ unsigned char pad[];
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint32_t m[16], cv[8];
    for (int i = 0; i < 16; i++) m[i] = io[tid] + i * 0x9E3779B9u;
    b3::set_iv(cv);
    for (int i = 0; i < iters; i++) {
        b3::compress(cv, m, 0, 0, 64, 11);
        m[i & 15] ^= cv[0];
    }
    if (iters < 0) pad[threadIdx.x] = 1;
    io[tid] = cv[0] ^ cv[7];
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
    void *io;
    (void)hipMalloc(&io, (size_t)cus * 8 * 256 * 16);
    (void)hipMemset(io, 0x5A, (size_t)cus * 8 * 256 * 16);
    printf("# %d CUs; k work-groups of 256 threads per CU = k waves per SIMD; rates per SIMD (1024 SIMDs)\n", cus);
    printf("# waves/SIMD | butterflies (2 chains/lane): ms, G bfly/s, cycles@2.4GHz per bfly-wave | butterflies (8 chains/lane) | BLAKE3: ms, G compress/s\n");
    for (int k : {1, 2, 3, 4, 5, 6, 8}) {
        const int it = 3000, itb = 300;
        const double a = run(k_bfly, io, cus, k, it), b = run(k_bfly_ilp, io, cus, k, it / 4), c = run(k_blake, io, cus, k, itb);
        const double lanes = (double)cus * k * 256;
        const double ra = lanes * 2 * it / (a * 1e-3), rb = lanes * 8 * (it / 4) / (b * 1e-3), rc = lanes * itb / (c * 1e-3);
        // wave-butterflies per second per SIMD -> cycles per wave-butterfly at 2.4 GHz
        const double ca = 2.4e9 / (ra / 64 / (cus * 4)), cb = 2.4e9 / (rb / 64 / (cus * 4));
        printf("%d | %7.3f ms %8.1f G/s %6.1f cyc | %7.3f ms %8.1f G/s %6.1f cyc | %7.3f ms %6.2f G/s\n", k, a, ra / 1e9, ca, b, rb / 1e9, cb, c, rc / 1e9);
    }
    return 0;
}
