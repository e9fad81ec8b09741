"""A stream of proofs from HOST columns (cfg 2: 2^20 x 8 f64, blowup 8): wall clock per commitment for
  one   : one context, commitments back to back (upload, then kernels, in stream order)
  two   : two host threads with a context each (bench.py's `resident_pipelined*`)
  batch : ONE context, wf_trace_commit_resident_batch (upload of proof k + 1 on the copy stream under the kernels of proof k)
from pageable and from pinned host memory.  Run under
    rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --output-format csv -d <dir> -- python3 scripts/two_contexts.py trace
for the timeline scripts/timeline_overlap.py summarises ("trace": fewer commitments, one memory kind at a time)."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import starkpack_winterfell_amd.capi as capi

LOG_R, LOG_B, C = 20, 3, 8
params = capi.make_params(capi.F64, 1, LOG_R, LOG_B, C, 1)
rng = np.random.default_rng(1)
cols = [rng.integers(0, 2**62, size=1 << LOG_R, dtype=np.uint64) for _ in range(C)]
pinned = [torch.from_numpy(c.view(np.int64)).pin_memory().numpy().view(np.uint64) for c in cols]
mode = sys.argv[1] if len(sys.argv) > 1 else "time"
n_each = 4 if mode == "trace" else 12


def one(columns):
    ctx = capi.Context(0)
    com, _ = ctx.trace_commit_resident(params, columns)
    root = com.root()
    com.close()
    t0 = time.perf_counter()
    for _ in range(2 * n_each):
        com, _ = ctx.trace_commit_resident(params, columns)
        com.close()
    ms = (time.perf_counter() - t0) * 1e3 / (2 * n_each)
    ctx.close()
    return ms, root


def two(columns):
    barrier = threading.Barrier(3)
    roots = []

    def worker():
        c2 = capi.Context(0)
        com, _ = c2.trace_commit_resident(params, columns)
        com.close()
        barrier.wait()
        for _ in range(n_each):
            com, _ = c2.trace_commit_resident(params, columns)
            roots.append(com.root())
            com.close()
        c2.close()

    th = [threading.Thread(target=worker) for _ in range(2)]
    for t in th:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    return (time.perf_counter() - t0) * 1e3 / (2 * n_each), roots


def batch(columns):
    ctx = capi.Context(0)
    if not hasattr(ctx, "trace_commit_resident_batch"):
        return None, []
    coms = ctx.trace_commit_resident_batch(params, [columns] * 2)
    for c in coms:
        c.close()
    t0 = time.perf_counter()
    coms = ctx.trace_commit_resident_batch(params, [columns] * (2 * n_each))
    ms = (time.perf_counter() - t0) * 1e3 / (2 * n_each)
    roots = [c.root() for c in coms]
    for c in coms:
        c.close()
    ctx.close()
    return ms, roots


only = sys.argv[3] if len(sys.argv) > 3 else None   # trace mode: one | two | batch -- a timeline of that variant alone
for name, columns in (("pageable", cols), ("pinned", pinned)):
    if mode == "trace" and len(sys.argv) > 2 and sys.argv[2] != name:
        continue
    a, root = one(columns) if only in (None, "one") else (float("nan"), None)
    b, roots2 = two(columns) if only in (None, "two") else (float("nan"), [])
    c, roots3 = batch(columns) if only in (None, "batch") else (float("nan"), [])
    root = root or (roots2 + roots3 + [None])[0]
    ok = all(r == root for r in roots2) and all(r == root for r in roots3)
    print(f"{name:9s}: one context {a:.3f} ms / commitment | two contexts, two threads {b:.3f} | "
          f"one context, wf_trace_commit_resident_async (upload under compute) {('%.3f' % c) if c is not None else 'n/a'} | roots agree: {ok}", flush=True)
