"""Determinism soak (bring-up aid): the bench workload committed N times back to back; every root must be the same.
    python scripts/soak.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starkpack_winterfell_amd.capi as capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
logR, logB, cols = 20, 3, 8
R, N = 1 << logR, 1 << (logR + logB)
gen = torch.Generator(device=dev); gen.manual_seed(7)
trace = torch.randint(0, 2**62, (cols * R,), dtype=torch.int64, device=dev, generator=gen)
polys = torch.empty_like(trace)
lde = torch.empty(N * 8, dtype=torch.int64, device=dev)
leaves = torch.empty((N, 32), dtype=torch.uint8, device=dev)
nodes = torch.empty((N, 32), dtype=torch.uint8, device=dev)
p = capi.make_params(capi.F64, 1, logR, logB, cols, 1)
s = torch.cuda.Stream(device=dev)
roots = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
with torch.cuda.stream(s):
    for k in range(n):
        ctx.trace_commit_dev(p, trace.data_ptr(), polys.data_ptr(), lde.data_ptr(), leaves.data_ptr(), nodes.data_ptr(), s.cuda_stream)
        roots[k].copy_(nodes[1], non_blocking=True)
    torch.cuda.synchronize()
u = torch.unique(roots, dim=0)
print(f"{n} commitments, {u.shape[0]} distinct root(s): {bytes(u[0].cpu().numpy()).hex()[:16]}")
sys.exit(0 if u.shape[0] == 1 else 1)
