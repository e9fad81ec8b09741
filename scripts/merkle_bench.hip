// Bring-up microbenchmark (not part of the product): variants of the two-level Merkle kernel on 2^23 leaves.
//   hipcc -O3 --offload-arch=gfx950 -I starkpack-winterfell_amd/csrc scripts/merkle_bench.hip -o /tmp/merkle_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#include "blake3_dev.hpp"
#include "field.hpp"
#include "kernels.hpp"

using namespace wf;

#define CHECK(x)                                                            \
    do {                                                                    \
        hipError_t e = (x);                                                 \
        if (e != hipSuccess) {                                              \
            printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

__device__ __forceinline__ void unpack(const uint4 (&q)[4], uint32_t (&m)[16]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        m[4 * j] = q[j].x;
        m[4 * j + 1] = q[j].y;
        m[4 * j + 2] = q[j].z;
        m[4 * j + 3] = q[j].w;
    }
}

// C: all eight loads first
template <bool STRIDE>
__global__ void __launch_bounds__(256) k_l2_upfront(const uint32_t *__restrict__ children, uint32_t *__restrict__ parents,
                                                    uint32_t *__restrict__ grandparents, uint64_t n_grand) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_grand; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 *src = reinterpret_cast<const uint4 *>(children + i * 32);
        uint4 qa[4], qb[4];
#pragma unroll
        for (int j = 0; j < 4; j++) qa[j] = src[j];
#pragma unroll
        for (int j = 0; j < 4; j++) qb[j] = src[4 + j];
        __builtin_amdgcn_sched_barrier(0);
        uint32_t m[16], cv[8], g[16];
        unpack(qa, m);
        b3::merge(m, cv);
        uint4 *dst = reinterpret_cast<uint4 *>(parents + (2 * i) * 8);
        dst[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
        dst[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
#pragma unroll
        for (int k = 0; k < 8; k++) g[k] = cv[k];
        unpack(qb, m);
        b3::merge(m, cv);
        dst[2] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
        dst[3] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
#pragma unroll
        for (int k = 0; k < 8; k++) g[8 + k] = cv[k];
        b3::merge(g, cv);
        uint4 *dg = reinterpret_cast<uint4 *>(grandparents + i * 8);
        dg[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
        dg[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
        if (!STRIDE) break;
    }
}

// D: prefetch the next item while hashing the current one
__global__ void __launch_bounds__(256) k_l2_prefetch(const uint32_t *__restrict__ children, uint32_t *__restrict__ parents,
                                                     uint32_t *__restrict__ grandparents, uint64_t n_grand) {
    const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_grand) return;
    uint4 qa[4], qb[4];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(children + i * 32);
#pragma unroll
        for (int j = 0; j < 4; j++) qa[j] = src[j];
#pragma unroll
        for (int j = 0; j < 4; j++) qb[j] = src[4 + j];
    }
    while (true) {
        uint32_t m[16], cv[8], g[16];
        uint32_t m2[16];
        unpack(qa, m);
        unpack(qb, m2);
        const uint64_t nx = i + step;
        const bool more = nx < n_grand;
        if (more) {
            const uint4 *src = reinterpret_cast<const uint4 *>(children + nx * 32);
#pragma unroll
            for (int j = 0; j < 4; j++) qa[j] = src[j];
#pragma unroll
            for (int j = 0; j < 4; j++) qb[j] = src[4 + j];
        }
        __builtin_amdgcn_sched_barrier(0);
        b3::merge(m, cv);
        uint4 *dst = reinterpret_cast<uint4 *>(parents + (2 * i) * 8);
        dst[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
        dst[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
#pragma unroll
        for (int k = 0; k < 8; k++) g[k] = cv[k];
        b3::merge(m2, cv);
        dst[2] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
        dst[3] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
#pragma unroll
        for (int k = 0; k < 8; k++) g[8 + k] = cv[k];
        b3::merge(g, cv);
        uint4 *dg = reinterpret_cast<uint4 *>(grandparents + i * 8);
        dg[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
        dg[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
        if (!more) break;
        i = nx;
    }
}

// E: same arithmetic, inputs synthesised in registers, one 32-byte store per item (compute bound reference)
__global__ void __launch_bounds__(256) k_l2_compute_only(uint32_t *__restrict__ grandparents, uint64_t n_grand) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_grand; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t m[16], cv[8], g[16];
#pragma unroll
        for (int k = 0; k < 16; k++) m[k] = (uint32_t)i * 0x9E3779B9u + k;
        b3::merge(m, cv);
#pragma unroll
        for (int k = 0; k < 8; k++) g[k] = cv[k];
#pragma unroll
        for (int k = 0; k < 16; k++) m[k] ^= 0x55555555u;
        b3::merge(m, cv);
#pragma unroll
        for (int k = 0; k < 8; k++) g[8 + k] = cv[k];
        b3::merge(g, cv);
        uint4 *dg = reinterpret_cast<uint4 *>(grandparents + i * 8);
        dg[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
        dg[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
    }
}

// three levels per launch: lane i reads eight children (256 contiguous bytes) in two halves
__global__ void __launch_bounds__(256) k_l3(const uint32_t *__restrict__ children, uint32_t *__restrict__ l1,
                                            uint32_t *__restrict__ l2, uint32_t *__restrict__ l3, uint64_t n3) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t top[16];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const uint4 *src = reinterpret_cast<const uint4 *>(children + (2 * i + half) * 32);
            uint32_t m[16], cv[8], g[16];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                uint4 q[4];
#pragma unroll
                for (int j = 0; j < 4; j++) q[j] = src[4 * h + j];
                unpack(q, m);
                b3::merge(m, cv);
                uint4 *dst = reinterpret_cast<uint4 *>(l1 + (4 * i + 2 * half + h) * 8);
                dst[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
                dst[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
#pragma unroll
                for (int k = 0; k < 8; k++) g[8 * h + k] = cv[k];
            }
            b3::merge(g, cv);
            uint4 *dg = reinterpret_cast<uint4 *>(l2 + (2 * i + half) * 8);
            dg[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
            dg[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
#pragma unroll
            for (int k = 0; k < 8; k++) top[8 * half + k] = cv[k];
        }
        uint32_t cv[8];
        b3::merge(top, cv);
        uint4 *dt = reinterpret_cast<uint4 *>(l3 + i * 8);
        dt[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
        dt[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
    }
}

template <class K>
static float timeit(K launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
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
    const uint64_t n_leaves = 1ull << 23, n_par = n_leaves / 2, n_grand = n_par / 2;
    uint32_t *leaves, *nodes, *nodes2;
    CHECK(hipMalloc(&leaves, n_leaves * 32));
    CHECK(hipMalloc(&nodes, n_leaves * 32));
    CHECK(hipMalloc(&nodes2, n_leaves * 32));
    std::vector<uint32_t> h(n_leaves * 8);
    for (uint64_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 0x9E3779B97F4A7C15ull >> 20);
    CHECK(hipMemcpy(leaves, h.data(), n_leaves * 32, hipMemcpyHostToDevice));
    const double comp = 3.0 * n_grand;
    auto report = [&](const char *name, float ms) { printf("%-44s %7.3f ms  %6.2f Gcompress/s\n", name, ms, comp / ms / 1e6); };
    auto check = [&](const char *name) {
        // compare parents + grandparents against the product kernel's output in `nodes`
        std::vector<uint32_t> a(n_par * 8 + n_grand * 8), b(a.size());
        hipMemcpy(a.data(), nodes + n_grand * 8, a.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), nodes2 + n_grand * 8, b.size() * 4, hipMemcpyDeviceToHost);
        if (a != b) printf("   !! %s differs from the product kernel\n", name);
    };
    for (uint32_t blocks : {1024u, 2048u, 4096u, 8192u}) {
        char nm[96];
        float ms = timeit([&] { hipLaunchKernelGGL(k_merkle_level2, dim3(blocks), dim3(256), 0, 0, leaves, nodes + n_par * 8, nodes + n_grand * 8, n_grand); }, 10);
        snprintf(nm, sizeof nm, "product k_merkle_level2, %u blocks", blocks);
        report(nm, ms);
        ms = timeit([&] { hipLaunchKernelGGL(k_l2_upfront<true>, dim3(blocks), dim3(256), 0, 0, leaves, nodes2 + n_par * 8, nodes2 + n_grand * 8, n_grand); }, 10);
        snprintf(nm, sizeof nm, "loads up front, %u blocks", blocks);
        report(nm, ms);
        check(nm);
        ms = timeit([&] { hipLaunchKernelGGL(k_l2_prefetch, dim3(blocks), dim3(256), 0, 0, leaves, nodes2 + n_par * 8, nodes2 + n_grand * 8, n_grand); }, 10);
        snprintf(nm, sizeof nm, "prefetch next item, %u blocks", blocks);
        report(nm, ms);
        check(nm);
        ms = timeit([&] { hipLaunchKernelGGL(k_l2_compute_only, dim3(blocks), dim3(256), 0, 0, nodes2 + n_grand * 8, n_grand); }, 10);
        snprintf(nm, sizeof nm, "compute only, %u blocks", blocks);
        report(nm, ms);
    }
    for (uint32_t blocks : {1024u, 2048u, 4096u}) {
        const uint64_t n3 = n_grand / 2;
        float ms = timeit([&] { hipLaunchKernelGGL(k_l3, dim3(blocks), dim3(256), 0, 0, leaves, nodes2 + n_par * 8, nodes2 + n_grand * 8, nodes2 + n3 * 8, n3); }, 10);
        printf("three levels per launch, %u blocks             %7.3f ms  %6.2f Gcompress/s (7 per lane)\n", blocks, ms, 7.0 * n3 / ms / 1e6);
        // reference: two launches of the product kernel covering the same three levels
        float ms2 = timeit([&] {
            hipLaunchKernelGGL(k_merkle_level2, dim3(2048), dim3(256), 0, 0, leaves, nodes + n_par * 8, nodes + n_grand * 8, n_grand);
            hipLaunchKernelGGL(k_merkle_level, dim3((uint32_t)(n3 / 256)), dim3(256), 0, 0, nodes + n_grand * 8, nodes + n3 * 8, n3); }, 10);
        printf("   product: level2 + level for the same levels  %7.3f ms\n", ms2);
    }
    {
        float ms = timeit([&] { hipLaunchKernelGGL(k_l2_upfront<false>, dim3((uint32_t)(n_grand / 256)), dim3(256), 0, 0, leaves, nodes2 + n_par * 8, nodes2 + n_grand * 8, n_grand); }, 10);
        report("loads up front, one item per thread", ms);
        check("one item per thread");
        ms = timeit([&] { hipLaunchKernelGGL(k_merkle_level, dim3((uint32_t)(n_par / 256)), dim3(256), 0, 0, leaves, nodes2 + n_par * 8, n_par); }, 10);
        printf("%-44s %7.3f ms  %6.2f Gcompress/s\n", "k_merkle_level (one level, 2^22 parents)", ms, n_par / ms / 1e6);
    }
    return 0;
}
