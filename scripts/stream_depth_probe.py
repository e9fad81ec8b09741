"""Run on the GPU box: at what depth does the FIRST batch of asynchronous commitments from pinned columns stall (scripts/stream_probe.py)?
One fresh process per depth: pageable warm-up (2 commitments), pin, then ONE pinned batch of n with per-call and per-wait wall clock.
    python scripts/stream_depth_probe.py <n> [warm_pinned 0|1]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import starkpack_winterfell_amd.capi as capi  # noqa: E402

n = int(sys.argv[1])
warm_pinned = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = capi.Context(0)
params = capi.make_params(capi.F64, 1, 20, 3, 8, 1)
rng = np.random.default_rng(1)
cols = [rng.integers(0, 2**62, size=1 << 20, dtype=np.uint64) for _ in range(8)]
for c in ctx.trace_commit_resident_batch(params, [cols] * 2):
    c.close()
for c in ctx.trace_commit_resident_batch(params, [cols] * n):  # buffers of n commitments exist (parked) before the pinned batch
    c.close()
pinned = [torch.from_numpy(c.view(np.int64)).pin_memory().numpy().view(np.uint64) for c in cols]
if warm_pinned:
    for c in ctx.trace_commit_resident_batch(params, [pinned] * 2):
        c.close()
t0 = time.perf_counter()
coms = [ctx.trace_commit_resident_async(params, pinned) for _ in range(n)]
t_calls = (time.perf_counter() - t0) * 1e3
waits = []
for c in coms:
    t = time.perf_counter()
    c.wait()
    waits.append((time.perf_counter() - t) * 1e3)
total = (time.perf_counter() - t0) * 1e3
for c in coms:
    c.close()
print(f"depth {n:3d} warm_pinned {warm_pinned}: first pinned batch {total:9.2f} ms = {total / n:8.3f} / commitment; calls {t_calls:6.2f} ms; "
      f"slowest wait {max(waits):9.2f} ms at #{int(np.argmax(waits))}", flush=True)
