"""Times the constraint side at cfg 2's proof shape (trace 2^20, constraint evaluation domain 2^21, quadratic extension, two
composition columns): wf_constraint_commit_from_tables (one packed trace: a transition column and a single-step assertion
column) and wf_constraint_commit_from_evaluations, wall clock and per-launch device times.
    python scripts/time_constraint_side.py [logR] [log_ce_blowup]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import starkpack_winterfell_amd.capi as capi

logR = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lce = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ext, logB, n_cols = 2, 3, 2
R, ce = 1 << logR, 1 << (logR + lce)
ctx = capi.Context(0)
rng = np.random.default_rng(1)
r = lambda *shape: rng.integers(0, 2**62, size=shape, dtype=np.uint64)
one = np.array([0xFFFFFFFF], dtype=np.uint64)          # 1 in Montgomery form
p = capi.make_params(capi.F64, ext, logR, logB, n_cols, 1)
table = [(r(ce * ext), (R, one, r(1))), (r(ce * ext), (1, one, None))]
comb = r(ce * ext)
for rep in range(4):
    t0 = time.perf_counter()
    c1, _ = ctx.constraint_commit_from_tables(p, [table])
    t1 = time.perf_counter()
    c2, _ = ctx.constraint_commit_from_evaluations(p, [comb])
    t2 = time.perf_counter()
    c1.close(); c2.close()
    print(f"rep {rep}: from the table (2 columns of {ce * ext * 8 >> 20} MiB up) {(t1 - t0) * 1e3:.2f} ms; from the combined column (1 up) {(t2 - t1) * 1e3:.2f} ms", flush=True)
ctx.profile_enable(2)
c1, _ = ctx.constraint_commit_from_tables(p, [table])
acc = {}
for k, v in ctx.profile_read():
    acc.setdefault(k, []).append(round(v, 4))
print({k: v for k, v in acc.items() if k != "between_calls"})
