"""Bring-up check of the larger plans ([11,10], [11,11], 3 passes) against the threaded oracle, and PCIe-inclusive timing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import rand_cols
from oracle import oracle as O
import starkpack_winterfell_amd.capi as capi
ctx = capi.Context(0)
for (field, ext, logR, logB, n_cols, n_traces) in [(1,1,21,1,2,1),(1,1,22,1,1,1),(1,1,23,1,1,1),(2,1,21,1,1,1),(2,1,16,3,10,1)]:
    rng = np.random.default_rng(5)
    R = 1 << logR
    traces = [rand_cols(rng, field, n_cols, R * ext) for _ in range(n_traces)]
    t0=time.time(); want = O.build_trace_commitment(field, traces, ext, logR, logB, 7 if field == 1 else 3, threads=16); t1=time.time()
    p = capi.make_params(field, ext, logR, logB, n_cols, n_traces)
    got = ctx.trace_commit(p, [c for t in traces for c in t]); t2=time.time()
    ok = got["root"] == want["root"] and all(np.array_equal(a,b) for a,b in zip(got["lde"], want["lde"]))
    print((field, ext, logR, logB, n_cols, n_traces), "OK" if ok else "MISMATCH", f"cpu {t1-t0:.2f}s gpu(host api) {t2-t1:.2f}s", flush=True)
# PCIe-inclusive cfg 2
rng = np.random.default_rng(1)
cols = rand_cols(rng, 1, 8, 1 << 20)
p = capi.make_params(1, 1, 20, 3, 8, 1)
ctx.trace_commit(p, cols)
t0 = time.time()
for _ in range(3):
    ctx.trace_commit(p, cols)
print(f"cfg2 host-buffer form (H2D 64 MiB + D2H 1.09 GiB, pageable numpy buffers): {(time.time()-t0)/3*1e3:.1f} ms per commitment")
t0 = time.time()
for _ in range(3):
    ctx.trace_commit(p, cols, want_lde=False, want_polys=False)
print(f"cfg2 host-buffer form without LDE/polys copy-out (leaves+nodes only): {(time.time()-t0)/3*1e3:.1f} ms")
