"""Experiment: is the device-buffer form capturable into a HIP graph (via torch.cuda.CUDAGraph) and what does replay save?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starkpack_winterfell_amd.capi as capi

dev = torch.device("cuda", 0)
ctx = capi.Context(0)
logR, logB, C = int(sys.argv[1]), 3, int(sys.argv[2])
R, N = 1 << logR, 1 << (logR + logB)
gen = torch.Generator(device=dev); gen.manual_seed(1)
trace = torch.randint(0, 2**62, (C * R,), dtype=torch.int64, device=dev, generator=gen)
polys = torch.empty_like(trace); lde = torch.empty(N * 8 * ((C + 7) // 8), dtype=torch.int64, device=dev)
leaves = torch.empty((N, 32), dtype=torch.uint8, device=dev); nodes = torch.empty((N, 32), dtype=torch.uint8, device=dev)
p = capi.make_params(1, 1, logR, logB, C, 1)
s = torch.cuda.Stream(device=dev)
def call(st): ctx.trace_commit_dev(p, trace.data_ptr(), polys.data_ptr(), lde.data_ptr(), leaves.data_ptr(), nodes.data_ptr(), st)
with torch.cuda.stream(s):
    for _ in range(3): call(s.cuda_stream)
    torch.cuda.synchronize()
    root_direct = bytes(nodes[1].cpu().numpy())
    K = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K): call(s.cuda_stream)
    e1.record(); torch.cuda.synchronize()
    print(f"direct: {e0.elapsed_time(e1)/K:.4f} ms")
leaves_direct, lde_direct = leaves.clone(), lde.clone()
g = torch.cuda.CUDAGraph()
nodes.zero_()
with torch.cuda.graph(g, stream=s):
    call(torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
for it in range(3):   # every replay must redo all the work: outputs are wiped in between
    nodes.zero_(); leaves.zero_(); lde.zero_()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print(f"replay {it}: root {bytes(nodes[1].cpu().numpy()) == root_direct} leaves {torch.equal(leaves, leaves_direct)} lde {torch.equal(lde, lde_direct)}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"graph replay: {e0.elapsed_time(e1)/K:.4f} ms")
