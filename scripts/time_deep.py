"""Times the DEEP composition on resident commitments (wf_deep_compose) at proof shapes: main trace of 2^logR x C base
columns, K constraint composition columns over the quadratic extension, composition over the quadratic extension.
Wall clock around the C call (stream synchronised inside it); "into FRI" includes the LDE of the composed polynomial
(blowup 8) into the prover's first layer, "to host" the copy of the n coefficients instead.
    python scripts/time_deep.py [logR] [C] [K] [n_traces]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import starkpack_winterfell_amd.capi as capi

logR = int(sys.argv[1]) if len(sys.argv) > 1 else 20
C = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
T = int(sys.argv[4]) if len(sys.argv) > 4 else 1
ext, logb = 2, 3
n = 1 << logR
ctx = capi.Context(0)
rng = np.random.default_rng(1)
r = lambda *shape: rng.integers(0, 2**62, size=shape, dtype=np.uint64)
trace, _ = ctx.trace_commit_resident(capi.make_params(capi.F64, 1, logR, logb, C, T), [r(n) for _ in range(C * T)])
cons = ctx.constraint_commit_resident(capi.make_params(capi.F64, ext, logR, logb, K, 1), [r(n * ext) for _ in range(K)])
z, cct, ccc = r(ext), r(C * T * ext), r(K * ext)
fri = capi.FriProver(ctx, capi.F64, ext, 4, 1 << logb, 127, 7)
for rep in range(5):
    t0 = time.perf_counter()
    poly = ctx.deep_compose(capi.F64, ext, n, [trace], cons, z, cct, ccc)
    t1 = time.perf_counter()
    ctx.deep_compose(capi.F64, ext, n, [trace], cons, z, cct, ccc, want_poly=False, fri=fri, lde_blowup=1 << logb)
    t2 = time.perf_counter()
    fri.reset()
    fri.begin_poly(poly, 1 << logb)
    t3 = time.perf_counter()
    fri.reset()
    print(f"rep {rep}: 2^{logR} x ({T} x {C} base + {K} E columns): compose to host {(t1 - t0) * 1e3:.3f} ms, compose into FRI (with LDE) "
          f"{(t2 - t1) * 1e3:.3f} ms; begin_poly from the host copy alone {(t3 - t2) * 1e3:.3f} ms", flush=True)
ctx.profile_enable(2)
ctx.deep_compose(capi.F64, ext, n, [trace], cons, z, cct, ccc)
marks = ctx.profile_read()
ctx.profile_enable(0)
print("device time of the composition's launches:", [(k, round(v, 4)) for k, v in marks if k.startswith("deep")])
bytes_read = n * 8 * (C * T + K * ext)
print(f"columns read per composition: {bytes_read / 2**20:.0f} MiB")
