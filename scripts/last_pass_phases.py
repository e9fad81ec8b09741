"""Diagnostic (needs build/exp_stamps/libwf_lde.so = the library built with -DWF_EXPERIMENTS -DWF_EXP_STAMPS): where a work-group of the
persistent last pass (k_seg_last_hash) spends its cycles, phase by phase, on cfg 2.  Thread 0 of every work-group sums
s_memtime differences between phase boundaries over its tiles.
    WF_LDE_LIB=build/exp_stamps/libwf_lde.so python scripts/last_pass_phases.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import starkpack_winterfell_amd.capi as capi

logR, logB, n_cols = 20, 3, 8
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
R, N = 1 << logR, 1 << (logR + logB)
gen = torch.Generator(device=dev); gen.manual_seed(1)
trace = torch.randint(0, 2**62, (n_cols * R,), dtype=torch.int64, device=dev, generator=gen)
polys = torch.empty_like(trace)
lde = torch.empty(N * 8, dtype=torch.int64, device=dev)
leaves = torch.empty((N, 32), dtype=torch.uint8, device=dev)
nodes = torch.empty((N, 32), dtype=torch.uint8, device=dev)
p = capi.make_params(1, 1, logR, logB, n_cols, 1)
s = torch.cuda.Stream(device=dev)
L = capi.load()
L.wf_exp_stamps_read.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros((4096, 8), dtype=np.uint64)
with torch.cuda.stream(s):
    for _ in range(3):
        ctx.trace_commit_dev(p, trace.data_ptr(), polys.data_ptr(), lde.data_ptr(), leaves.data_ptr(), nodes.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    assert L.wf_exp_stamps_read(buf.ctypes.data, 1) == 0
    K = 10
    for _ in range(K):
        ctx.trace_commit_dev(p, trace.data_ptr(), polys.data_ptr(), lde.data_ptr(), leaves.data_ptr(), nodes.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    assert L.wf_exp_stamps_read(buf.ctypes.data, 1) == 0
wg = buf[buf[:, 6] > 0]
names = ["tile into LDS (waits for the prefetch)", "transform", "row stores issued", "ticket + next tile requested", "hashing",
         "end-of-tile barrier"]
tiles = wg[:, 6].sum()
tot = wg[:, :6].sum()
print(f"{len(wg)} work-groups, {tiles / K:.0f} tiles per launch, {tot / tiles:.0f} s_memtime ticks per tile (100 MHz ticks if constant-rate)")
for i, n in enumerate(names):
    print(f"  {n:42s} {wg[:, i].sum() / tiles:9.1f} per tile  {100.0 * wg[:, i].sum() / tot:5.1f} %")
per_wg = wg[:, :6].sum(axis=1) / K
print(f"per work-group busy total: min {per_wg.min():.0f}  median {np.median(per_wg):.0f}  max {per_wg.max():.0f}")
print(f"tiles per work-group: min {wg[:, 6].min() / K:.1f}  max {wg[:, 6].max() / K:.1f}")
