// libwf_yardstick.so -- measurement helper of bench.py, NOT part of the product ABI (nothing in include/wf_lde.h, nothing
// in libwf_lde.so uses it): the integer-VALU rates of the path's own instruction sequences in isolation, measured in the
// run that quotes them.  Same device functions as the kernels (field.hpp, blake3_dev.hpp), register-only loops, one
// resident set of waves (8 per SIMD), four independent chains per lane:
//   Goldilocks butterflies (a, b) -> (a + w b, a - w b) with a general twiddle  : the unit of every NTT pass over f64
//   f128 butterflies                                                            : the same over the 128-bit field
//   BLAKE3 compressions of one 64-byte block                                    : leaves and Merkle nodes
// One loop per call, so that the caller can sample the driver's clock reading while exactly that loop runs (s_memtime is
// of no use for this: its counters are per XCD and not aligned with each other, and a single wave's span says nothing
// because the arbiter favours old waves -- measured, round 3).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "blake3_dev.hpp"
#include "field.hpp"

using namespace wf;

namespace {

__global__ void __launch_bounds__(256) y_bfly64(uint64_t *io, int iters) {
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
    io[tid] = a ^ b ^ c ^ d;
}

__global__ void __launch_bounds__(256) y_bfly128(U128 *io, int iters) {
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    U128 a = io[tid], b = U128{a.lo * 3 + 7, a.hi >> 1}, w = U128{0x0123456789ABCDEFull, 0x0FEDCBA987654321ull};
    a.hi >>= 1;  // < 2^127 < p
    for (int i = 0; i < iters; i++) {
        const U128 t = F128::mul(b, w);
        b = F128::sub(a, t);
        a = F128::add(a, t);
        w.lo += 2;
    }
    io[tid] = U128{a.lo ^ b.lo, a.hi ^ b.hi};
}

__global__ void __launch_bounds__(256) y_blake(uint32_t *io, int iters) {
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint32_t m[16], cv[8];
    for (int i = 0; i < 16; i++) m[i] = io[tid] + i * 0x9E3779B9u;
    b3::set_iv(cv);
    for (int i = 0; i < iters; i++) {
        b3::compress(cv, m, 0, 0, 64, 11);
        m[i & 15] ^= cv[0];
    }
    io[tid] = cv[0] ^ cv[7];
}

template <class Launch>
int time_loop(Launch launch, hipStream_t st, double units, double out[2]) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1;
    launch();  // warm-up: code object, clocks
    launch();
    if (hipStreamSynchronize(st) != hipSuccess) return -2;
    double best = 1e30;
    for (int rep = 0; rep < 5; rep++) {
        (void)hipEventRecord(e0, st);
        launch();
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) return -3;
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    out[0] = units / (best * 1e-3);
    out[1] = best;
    return 0;
}

}  // namespace

// which: 0 = Goldilocks butterflies, 1 = f128 butterflies, 2 = BLAKE3 compressions.  out[0] = units per second (best of
// five launches after two warm-ups), out[1] = that launch's duration in ms.  Returns 0, or a negative number when the
// device cannot be used.
extern "C" int wf_yardstick_run(int device, int which, double out[2]) {
    if (which < 0 || which > 2 || !out) return -9;
    if (hipSetDevice(device) != hipSuccess) return -10;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) return -11;
    const int threads = 256, blocks = cus * 8;  // 8 waves per SIMD: one resident set, every wave runs the whole loop
    const size_t n = (size_t)blocks * threads;
    void *io = nullptr;
    hipStream_t st = nullptr;
    if (hipMalloc(&io, n * 16) != hipSuccess) return -12;
    if (hipMemset(io, 0x5A, n * 16) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
        (void)hipFree(io);
        return -13;
    }
    const int it64 = 4000, it128 = 1500, itb = 400;
    int rc;
    if (which == 0)
        rc = time_loop([&] { hipLaunchKernelGGL(y_bfly64, dim3(blocks), dim3(threads), 0, st, (uint64_t *)io, it64); }, st, (double)n * 2.0 * it64, out);
    else if (which == 1)
        rc = time_loop([&] { hipLaunchKernelGGL(y_bfly128, dim3(blocks), dim3(threads), 0, st, (U128 *)io, it128); }, st, (double)n * it128, out);
    else
        rc = time_loop([&] { hipLaunchKernelGGL(y_blake, dim3(blocks), dim3(threads), 0, st, (uint32_t *)io, itb); }, st, (double)n * itb, out);
    (void)hipStreamDestroy(st);
    (void)hipFree(io);
    return rc;
}
