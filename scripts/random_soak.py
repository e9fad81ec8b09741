"""One-off differential soak: many random shapes (wider ranges than tests/test_gpu_parity.py::test_random_shapes, incl.
many narrow traces and rows beyond one BLAKE3 chunk) through the device-buffer form into poisoned buffers, compared in
full with the oracle.   python scripts/random_soak.py [n] [seed] [log2 rows from] [log2 rows below] [log2 of the element cap]
(defaults 200 1 3 15 23; e.g. `40 7 15 21 27` sends 40 large shapes through the two- and three-pass plans)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import oracle as orc
import starkpack_winterfell_amd.capi as capi
from conftest import rand_cols

n, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 200), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lo_r, hi_r, cap = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((3, 3), (4, 15), (5, 23)))
rng = np.random.default_rng(seed)
ctx = capi.Context(0)
dev = torch.device("cuda", 0)
bad = 0
for it in range(n):
    field = int(rng.integers(1, 3))
    ext = int(rng.integers(1, 4 if field == 1 else 3))
    logR = int(rng.integers(lo_r, hi_r))
    logB = int(rng.integers(1, 5))
    kind = int(rng.integers(0, 4))
    if kind == 0:   n_cols, n_traces = int(rng.integers(1, 9)), 1
    elif kind == 1: n_cols, n_traces = int(rng.integers(1, 5)), int(rng.integers(2, 40))
    elif kind == 2: n_cols, n_traces = int(rng.integers(9, 256)), 1
    else:           n_cols, n_traces = int(rng.integers(5, 30)), int(rng.integers(2, 12))
    while (1 << (logR + logB)) * n_cols * ext * n_traces > (1 << cap) and logR > 3:
        logR -= 1
    offset = int(rng.integers(2, 2**62))
    R, N = 1 << logR, 1 << (logR + logB)
    traces = [rand_cols(rng, field, n_cols, R * ext) for _ in range(n_traces)]
    want = orc.build_trace_commitment(field, traces, ext, logR, logB, offset, threads=16 if logR >= 15 else 1)
    p = capi.make_params(field, ext, logR, logB, n_cols, n_traces, offset)
    flat = np.concatenate([np.ascontiguousarray(c).reshape(-1) for t in traces for c in t]).view(np.int64)
    d_trace = torch.from_numpy(flat.copy()).to(dev)
    d_polys = torch.full_like(d_trace, -1)
    want_lde = np.stack([np.ascontiguousarray(l) for l in want["lde"]])
    d_lde = torch.full((want_lde.size,), -1, dtype=torch.int64, device=dev)
    d_leaves = torch.full((N, 32), 255, dtype=torch.uint8, device=dev)
    d_nodes = torch.full((N, 32), 255, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.trace_commit_dev(p, d_trace.data_ptr(), d_polys.data_ptr(), d_lde.data_ptr(), d_leaves.data_ptr(), d_nodes.data_ptr())
    torch.cuda.synchronize()
    ok = (np.array_equal(d_lde.cpu().numpy().view(np.uint64).reshape(want_lde.shape), want_lde)
          and np.array_equal(d_leaves.cpu().numpy(), np.ascontiguousarray(want["leaves"]).view(np.uint8).reshape(N, 32))
          and np.array_equal(d_nodes.cpu().numpy(), np.ascontiguousarray(want["nodes"]).view(np.uint8).reshape(N, 32))
          and np.array_equal(d_polys.cpu().numpy().view(np.uint64),
                             np.concatenate([np.ascontiguousarray(c).reshape(-1) for t in want["polys"] for c in t])))
    if not ok:
        bad += 1
        print("MISMATCH", (field, ext, logR, logB, n_cols, n_traces, offset), flush=True)
    if it % 25 == 24 or logR >= 15:
        print(f"{it + 1} shapes, {bad} mismatches", flush=True)
print(f"done: {n} shapes, {bad} mismatches")
sys.exit(1 if bad else 0)
