"""Wall clock of wf_trace_commit_resident from HOST columns for a wide trace (several segments), with the upload running
under the kernels (default) -- run once more with WF_EXP_ENABLE=1 WF_EXP_NO_PIPELINE=1 for the serial order.
    python scripts/time_wide_resident.py [logR] [cols] [polys]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import starkpack_winterfell_amd.capi as capi

logR = int(sys.argv[1]) if len(sys.argv) > 1 else 22
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 64
WANT_POLYS = len(sys.argv) > 3 and sys.argv[3] == "polys"   # also bring the polynomials back to the host
ctx = capi.Context(0)
rng = np.random.default_rng(1)
trace = [rng.integers(0, 2**62, size=1 << logR, dtype=np.uint64) for _ in range(cols)]
p = capi.make_params(capi.F64, 1, logR, 3, cols, 1)
mode = "serial (WF_EXP_NO_PIPELINE)" if os.environ.get("WF_EXP_NO_PIPELINE") else "upload under the kernels"
roots = set()
polys = [np.zeros_like(c) for c in trace] if WANT_POLYS else None
extra = ", polynomials back" if WANT_POLYS else ""
for rep in range(4):
    if WANT_POLYS:  # into arrays that exist already (fresh ones would be page-faulted in by the copy: 2 GiB = 200 ms)
        h = C.c_void_p()
        t0 = time.perf_counter()
        capi._check(capi.load().wf_trace_commit_resident(ctx._h, C.byref(p), capi._ptr_array(trace), capi._ptr_array(polys), C.byref(h)))
        t1 = time.perf_counter()
        com = capi.Commitment(h, p.field, keep_alive=ctx)
    else:
        t0 = time.perf_counter()
        com, _ = ctx.trace_commit_resident(p, trace)
        t1 = time.perf_counter()
    roots.add(com.root())
    com.close()
    print(f"rep {rep}: 2^{logR} x {cols} f64 from host columns ({cols << logR >> 17} MiB{extra}), {mode}: {(t1 - t0) * 1e3:.2f} ms", flush=True)
print("root", next(iter(roots)).hex()[:16], "stable" if len(roots) == 1 else "UNSTABLE")
