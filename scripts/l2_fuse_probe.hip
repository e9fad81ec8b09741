// Probe (not part of the product): can two consecutive strided passes of a three-pass transform be FUSED THROUGH AN XCD's L2?
// Geometry of cfg 3 with digits [7, 7, 8]: a column of N = 2^22 rows of 64 bytes; row n = n1 * 2^15 + n2 * 2^8 + n3.
// Pass 1 transforms over n1 (tile (n2, n3): 2^7 rows 2^15 apart), pass 2 over n2 (tile (k1, n3): 2^7 rows 2^8 apart).  For a
// fixed n3 the 2^14 rows {(a, b, n3)} form a SUPER-TILE of 1 MiB (TI = 1) or 2 MiB (TI = 2 adjacent n3: 128-byte rows) that
// pass 1 writes and pass 2 reads: 128 tiles each.  This program moves the same bytes with no arithmetic
//   (a) as two separate launches over the whole buffer (what the product does), and
//   (b) as ONE persistent launch: per-XCD ticket queues ordered p1(0) p1(1) p2(0) p1(2) p2(1) ..., a completion counter per
//       super-tile, phase-2 tickets waiting for their super-tile's 128 phase-1 tiles (same XCD: same L2),
// and reports the times.  If (b) is not clearly faster than (a), the fused kernel is not worth building.
//   hipcc -O3 --offload-arch=gfx950 scripts/l2_fuse_probe.hip -o /tmp/l2probe && /tmp/l2probe [cols] [ti]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                            \
    do {                                                                    \
        hipError_t e = (x);                                                 \
        if (e != hipSuccess) {                                              \
            printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

constexpr uint32_t LOG1 = 7, LOG2 = 7, LOG3 = 8;  // digits; N = 2^22 rows
constexpr uint64_t ROWQ = 4;                      // 16-byte quarters per 64-byte row

// one tile: 128 rows of TI * 64 bytes, `stride` rows apart, starting at row `row0` of column `col`: load everything, add one,
// store in place.  blockDim = 64 * TI (TI * 4 quarters per row, 16 rows per step... 8 uint4 per thread)
template <int TI>
__device__ __forceinline__ void move_tile(uint4 *buf, uint64_t col, uint64_t row0, uint64_t stride) {
    constexpr uint32_t QPR = TI * ROWQ;           // quarters per tile row
    const uint32_t t = threadIdx.x, q = t % QPR, r0 = t / QPR, rstep = (64 * TI) / QPR;  // rows per step = 16
    uint4 v[8];
    uint4 *base = buf + ((col << 22) + row0) * ROWQ + q;
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = base[(uint64_t)(r0 + i * rstep) * stride * ROWQ];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        v[i].x += 1;
        base[(uint64_t)(r0 + i * rstep) * stride * ROWQ] = v[i];
    }
}

// (a) separate passes: grid = cols * 2^15 / TI tiles
__device__ __forceinline__ uint64_t xcd_group_index(uint64_t b) {  // 8 consecutive logical tiles on one XCD (as the product's kernels)
    const uint64_t xcd = b & 7, seq = b >> 3;
    return (((seq >> 3) << 3) + xcd) * 8 + (seq & 7);
}

template <int TI, int PASS>
__global__ void __launch_bounds__(64 * TI) k_pass(uint4 *buf) {
    const uint64_t b = xcd_group_index(blockIdx.x);
    const uint64_t n3t = b % ((1u << LOG3) / TI), rest = b / ((1u << LOG3) / TI);
    const uint64_t m = rest % 128, col = rest / 128;  // m = n2 (pass 1) or k1 (pass 2)
    if (PASS == 1)
        move_tile<TI>(buf, col, (m << LOG3) + n3t * TI, (uint64_t)1 << (LOG2 + LOG3));
    else
        move_tile<TI>(buf, col, (m << (LOG2 + LOG3)) + n3t * TI, (uint64_t)1 << LOG3);
}

// (b) fused: persistent work-groups, per-XCD tickets.  Super-tiles of XCD x: s = x + 8 * j, j = 0 .. per_xcd - 1; s -> (col, n3t).
// Ticket blocks of 128: block 0 = p1(0), block 1 = p1(1), then block 2 i = p2(i - 1), block 2 i + 1 = p1(i + 1) ...
template <int TI, int BATCH>  // BATCH tiles per ticket
__global__ void __launch_bounds__(64 * TI) k_fused(uint4 *buf, uint32_t *tickets, uint32_t *done, uint32_t per_xcd, uint32_t n_super) {
    __shared__ uint32_t tk;
    const uint32_t xcd = blockIdx.x & 7;
    const uint32_t n_blocks = 2 * per_xcd;  // ticket blocks of this XCD
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) tk = atomicAdd(tickets + xcd, 1u);
        __syncthreads();
        const uint32_t t = tk * BATCH;
        const uint32_t blk = t >> 7, m0 = t & 127;
        if (blk >= n_blocks) break;
        uint32_t phase, j;
        if (blk == 0) { phase = 1; j = 0; }
        else if (blk == n_blocks - 1) { phase = 2; j = per_xcd - 1; }
        else if (blk & 1) { phase = 1; j = (blk + 1) >> 1; }
        else { phase = 2; j = (blk >> 1) - 1; }
        const uint32_t s = xcd + 8 * j;
        if (s >= n_super) continue;
        const uint64_t n3t = s % ((1u << LOG3) / TI), col = s / ((1u << LOG3) / TI);
        if (phase == 1) {
            for (uint32_t m = m0; m < m0 + BATCH; m++)
                move_tile<TI>(buf, col, ((uint64_t)m << LOG3) + n3t * TI, (uint64_t)1 << (LOG2 + LOG3));
            __builtin_amdgcn_s_waitcnt(0);  // this wave's stores have reached L2
            __syncthreads();
            if (threadIdx.x == 0) atomicAdd(done + s, (uint32_t)BATCH);
        } else {
            if (threadIdx.x == 0) {
                uint32_t spins = 0;
                while (__hip_atomic_load(done + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 128u) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > (1u << 26)) break;
                }
            }
            __syncthreads();
            asm volatile("buffer_inv sc0" ::: "memory");  // this CU's vector L1 may hold stale lines of the super-tile (gfx94x+: BUFFER_INV, group scope)
            for (uint32_t m = m0; m < m0 + BATCH; m++)
                move_tile<TI>(buf, col, ((uint64_t)m << (LOG2 + LOG3)) + n3t * TI, (uint64_t)1 << LOG3);
        }
    }
}

template <int TI>
int run(uint32_t cols) {
    const size_t bytes = (size_t)cols << 28;  // 2^22 rows * 64 bytes per column
    uint4 *buf;
    uint32_t *ctr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMemset(buf, 1, bytes));
    const uint32_t n_super = cols * ((1u << LOG3) / TI), per_xcd = (n_super + 7) / 8;
    CHECK(hipMalloc(&ctr, (16 + n_super) * 4));
    hipEvent_t e0, e1, e2;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventCreate(&e2));
    const uint32_t tiles = cols * 128 * ((1u << LOG3) / TI);
    float best_sep = 1e9f, best_p1 = 1e9f, best_fused = 1e9f;
    int dev_cus = 256;
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_pass<TI, 1>), dim3(tiles), dim3(64 * TI), 0, 0, buf);
        CHECK(hipEventRecord(e1));
        hipLaunchKernelGGL((k_pass<TI, 2>), dim3(tiles), dim3(64 * TI), 0, 0, buf);
        CHECK(hipEventRecord(e2));
        CHECK(hipEventSynchronize(e2));
        float a, b;
        CHECK(hipEventElapsedTime(&a, e0, e1));
        CHECK(hipEventElapsedTime(&b, e0, e2));
        if (b < best_sep) { best_sep = b; best_p1 = a; }
    }
    for (uint32_t cfg : {0x104u, 0x108u, 0x404u, 0x408u, 0x410u, 0x804u, 0x808u, 0x810u, 0x1008u, 0x1010u}) {
        const uint32_t wg_per_cu = cfg & 0xFF, batch = cfg >> 8;
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipMemset(ctr, 0, (16 + n_super) * 4));
            CHECK(hipEventRecord(e0));
            const dim3 g(dev_cus * wg_per_cu), bl(64 * TI);
            if (batch == 1) hipLaunchKernelGGL((k_fused<TI, 1>), g, bl, 0, 0, buf, ctr, ctr + 16, per_xcd, n_super);
            if (batch == 4) hipLaunchKernelGGL((k_fused<TI, 4>), g, bl, 0, 0, buf, ctr, ctr + 16, per_xcd, n_super);
            if (batch == 8) hipLaunchKernelGGL((k_fused<TI, 8>), g, bl, 0, 0, buf, ctr, ctr + 16, per_xcd, n_super);
            if (batch == 16) hipLaunchKernelGGL((k_fused<TI, 16>), g, bl, 0, 0, buf, ctr, ctr + 16, per_xcd, n_super);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float a;
            CHECK(hipEventElapsedTime(&a, e0, e1));
            if (a < best) best = a;
        }
        printf("  fused, %2u tiles per ticket, %2u work-groups per CU: %.3f ms = %.2f TB/s of read + write\n", batch, wg_per_cu, best, 4.0 * bytes / best / 1e9);
        if (best < best_fused) best_fused = best;
    }
    printf("TI = %d, %u columns (%zu MiB): separate passes %.3f ms (pass 1 %.3f) = %.2f TB/s of read + write;  fused best %.3f ms  (x %.2f)\n",
           TI, cols, bytes >> 20, best_sep, best_p1, 4.0 * bytes / best_sep / 1e9, best_fused, best_sep / best_fused);
    CHECK(hipFree(buf));
    CHECK(hipFree(ctr));
    return 0;
}

int main(int argc, char **argv) {
    const uint32_t cols = argc > 1 ? (uint32_t)atoi(argv[1]) : 16;
    const int ti = argc > 2 ? atoi(argv[2]) : 0;
    if (ti == 0 || ti == 1)
        if (run<1>(cols)) return 1;
    if (ti == 0 || ti == 2)
        if (run<2>(cols)) return 1;
    return 0;
}
