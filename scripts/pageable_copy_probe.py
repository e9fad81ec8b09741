"""Run on the GPU box: wall clock of ONE pageable hipMemcpy (host -> device and back) by size -- where does the runtime change
its strategy?  (profiles/r04_criterion_shapes.txt: the 32 MiB vectors of f128_quad/2^20 move 2.5 x slower per byte than 16 MiB.)
Also the same bytes through wf_fft_evaluate_poly (f128, quadratic extension), whose transfers this probe explains."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import starkpack_winterfell_amd.capi as capi  # noqa: E402

dev = torch.device("cuda", 0)
ctx = capi.Context(0)


def med(fn, reps=7, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


for mib in (4, 8, 12, 16, 20, 24, 28, 32, 40, 48, 64):
    n = mib << 20
    h = np.random.default_rng(1).integers(0, 255, size=n, dtype=np.uint8)
    ht = torch.from_numpy(h)
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    back = torch.empty(n, dtype=torch.uint8)

    def h2d():
        d.copy_(ht)
        torch.cuda.synchronize()

    def d2h():
        back.copy_(d)
        torch.cuda.synchronize()

    a, b = med(h2d), med(d2h)
    print(f"pageable {mib:3d} MiB: H2D {a:7.3f} ms ({n / a / 1e6:6.1f} GB/s)   D2H {b:7.3f} ms ({n / b / 1e6:6.1f} GB/s)", flush=True)

for logn in (18, 19, 20, 21):
    v = np.random.default_rng(2).integers(0, 2**62, size=(1 << logn) * 2 * 2, dtype=np.uint64)
    t = med(lambda: ctx.fft_evaluate_poly(capi.F128, 2, v))          # the Python wrapper: copies into a FRESH array per call
    buf = v.copy()
    n = 1 << logn
    t2 = med(lambda: capi._check(capi.load().wf_fft_evaluate_poly(ctx._h, capi.F128, 2, capi._p(buf), n)))  # the C entry point, in place
    t3 = med(lambda: v.copy())
    print(f"wf_fft_evaluate_poly f128 quad 2^{logn} ({v.nbytes >> 20} MiB each way): wrapper {t:.3f} ms, C entry point on a reused buffer {t2:.3f} ms, "
          f"numpy .copy() of the input alone {t3:.3f} ms", flush=True)
