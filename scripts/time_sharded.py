"""Per-rank cost of the coset-sharded packed commitment on ONE GPU: what a rank of W computes, timed with
per-launch HIP events through the single-GPU entry point of the same kernels (wf_trace_commit_shard_dev on that rank's
coset range).  The interpolation mark covers ALL segments (this entry point interpolates everything); under
wf_trace_commit_sharded_dev a rank transforms n_seg / W of them, a linear share.  Exchanges are not in these numbers.
    python scripts/time_sharded.py [n_traces] [n_cols] [log2 R]
Also: the constraint side of BASELINE configs[2] (4 columns of the quadratic extension at 2^22)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starkpack_winterfell_amd.capi as capi

n_traces = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_cols = int(sys.argv[2]) if len(sys.argv) > 2 else 8
logR = int(sys.argv[3]) if len(sys.argv) > 3 else 20
logB = 3
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
R, N = 1 << logR, 1 << (logR + logB)
p = capi.make_params(capi.F64, 1, logR, logB, n_cols, n_traces)
g = torch.Generator(device=dev)
g.manual_seed(1)
trace = torch.randint(0, 2**62, (n_traces * n_cols * R,), dtype=torch.int64, device=dev, generator=g)
polys = torch.empty_like(trace)
rw = 8 * ((n_cols + 7) // 8)
s = torch.cuda.Stream(device=dev)


def timed(fn, K=5):
    ctx.profile_enable(2)
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ctx.profile_read()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(K):
            fn()
        e1.record(s)
        torch.cuda.synchronize()
    acc = {}
    for k, v in ctx.profile_read():
        acc.setdefault(k, []).append(v)
    return e0.elapsed_time(e1) / K, {k: round(sum(v) / len(v), 4) for k, v in acc.items()}


print(f"# {n_traces} packed traces of 2^{logR} x {n_cols} f64, blowup 8, one MI355X; ms per call")
for W in (1, 2, 4, 8):
    per = 8 // W
    lde = torch.empty(n_traces * R * per * rw, dtype=torch.int64, device=dev)
    leaves = torch.empty((R * per, 32), dtype=torch.uint8, device=dev)
    ms, parts = timed(lambda: ctx.trace_commit_shard_dev(p, 0, per, trace.data_ptr(), polys.data_ptr(), lde.data_ptr(),
                                                         leaves.data_ptr(), s.cuda_stream))
    interp = sum(v for k, v in parts.items() if k.startswith(("interpolate", "layout")))
    ev = sum(v for k, v in parts.items() if k.startswith(("evaluate", "hash")))
    sub = torch.empty((N // W, 32), dtype=torch.uint8, device=dev)
    nodes = torch.empty_like(sub)
    mt, _ = timed(lambda: ctx.merkle_build_dev(sub.data_ptr(), N // W, nodes.data_ptr(), s.cuda_stream))
    n_seg = (n_traces * n_cols + 7) // 8
    share = interp / W if n_seg % W == 0 else interp
    print(f"W={W}: cosets/rank {per}: interpolate all segments {interp:.3f} (rank's share {share:.3f}), evaluate+hash {ev:.3f}, "
          f"sub-tree of N/W leaves {mt:.3f}  => rank total {share + ev + mt:.3f}")
    del lde, leaves, sub, nodes
torch.cuda.empty_cache()

# cfg 3 constraint side
pc = capi.make_params(capi.F64, 2, 22, 3, 4, 1)
cp = torch.randint(0, 2**62, (4 * (1 << 22) * 2,), dtype=torch.int64, device=dev, generator=g)
lde = torch.empty((1 << 25) * 8, dtype=torch.int64, device=dev)
leaves = torch.empty(((1 << 25), 32), dtype=torch.uint8, device=dev)
nodes = torch.empty_like(leaves)
ms, parts = timed(lambda: ctx.constraint_commit_dev(pc, cp.data_ptr(), lde.data_ptr(), leaves.data_ptr(), nodes.data_ptr(), s.cuda_stream), K=3)
print(f"# constraint commitment, 4 columns of the quadratic extension at 2^22, blowup 8: {ms:.3f} ms {parts}")
ctx.close()
