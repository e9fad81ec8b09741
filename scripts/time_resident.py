"""Wall clock of the resident form: wf_trace_commit_resident (host columns in, handle out) + 50 queried rows with their
batch proof + destroy, repeated (what a prover pays per proof around the device-side 1.3 ms).
    python scripts/time_resident.py [logR] [cols]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import starkpack_winterfell_amd.capi as capi

logR = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = capi.Context(0)
rng = np.random.default_rng(1)
trace = [rng.integers(0, 2**62, size=1 << logR, dtype=np.uint64) for _ in range(cols)]
p = capi.make_params(capi.F64, 1, logR, 3, cols, 1)
pos = np.sort(rng.choice(1 << (logR + 3), size=50, replace=False)).astype(np.uint64)
for rep in range(5):
    t0 = time.perf_counter()
    com, _ = ctx.trace_commit_resident(p, trace, want_polys=False)
    t1 = time.perf_counter()
    rows = com.read_rows(pos)
    proof = com.prove_batch(pos)
    t2 = time.perf_counter()
    com.close()
    t3 = time.perf_counter()
    print(f"rep {rep}: commit {(t1 - t0) * 1e3:.2f} ms, 50 rows + batch proof {(t2 - t1) * 1e3:.2f} ms, destroy {(t3 - t2) * 1e3:.2f} ms")
