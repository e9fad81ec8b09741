"""Run on the GPU box: where does the first batch of wf_trace_commit_resident_async calls with PINNED host columns lose its time?
(bench.py's `resident_stream_pinned_batches`, round 5: 109 ms per commitment in the first batch of 12, 1.41-1.43 in the next four; round 4's
single timed batch read 3.83 ms on the driver's box.)  Per-call wall clock of the async call and of the wait, for: pageable columns, pinned
columns, pinned columns again -- with and without the two extra contexts bench.py's `pipelined()` creates in between.
    python scripts/stream_probe.py [extra_contexts 0|1]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import starkpack_winterfell_amd.capi as capi  # noqa: E402

extra = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = capi.Context(0)
params = capi.make_params(capi.F64, 1, 20, 3, 8, 1)
rng = np.random.default_rng(1)
cols = [rng.integers(0, 2**62, size=1 << 20, dtype=np.uint64) for _ in range(8)]


def batch(columns, n=12, label=""):
    t0 = time.perf_counter()
    calls, coms = [], []
    for _ in range(n):
        t = time.perf_counter()
        coms.append(ctx.trace_commit_resident_async(params, columns))
        calls.append((time.perf_counter() - t) * 1e3)
    waits = []
    for c in coms:
        t = time.perf_counter()
        c.wait()
        waits.append((time.perf_counter() - t) * 1e3)
    total = (time.perf_counter() - t0) * 1e3
    t = time.perf_counter()
    for c in coms:
        c.close()
    closing = (time.perf_counter() - t) * 1e3
    print(f"{label:34s} total {total:9.2f} ms ({total / n:7.3f} / commitment)  slowest call {max(calls):8.2f} (#{int(np.argmax(calls))})  "
          f"slowest wait {max(waits):8.2f} (#{int(np.argmax(waits))})  sum of calls {sum(calls):8.2f}  close {closing:7.2f}", flush=True)


batch(cols, 2, "pageable warm-up (2)")
for i in range(3):
    batch(cols, 12, f"pageable batch {i}")
t = time.perf_counter()
pinned = [torch.from_numpy(c.view(np.int64)).pin_memory().numpy().view(np.uint64) for c in cols]
print(f"pin_memory of the 8 columns: {(time.perf_counter() - t) * 1e3:.2f} ms", flush=True)
if extra:
    for k in range(2):
        c2 = capi.Context(0)
        com, _ = c2.trace_commit_resident(params, pinned)
        com.close()
        c2.close()
    print("two extra contexts created, used once with the pinned columns, closed", flush=True)
batch(pinned, 2, "pinned warm-up (2)")
for i in range(4):
    batch(pinned, 12, f"pinned batch {i}")
for i in range(2):
    batch(cols, 12, f"pageable again {i}")
