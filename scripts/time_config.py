"""Times the device-buffer form of the trace commitment for an arbitrary configuration (bring-up / tuning aid).
    python scripts/time_config.py <field 1|2> <ext> <log2 R> <log2 blowup> <n_cols> <n_traces>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starkpack_winterfell_amd.capi as capi

field, ext, logR, logB, n_cols, n_traces = (int(x) for x in sys.argv[1:7])
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
w = 1 if field == 1 else 2
R, N = 1 << logR, 1 << (logR + logB)
gen = torch.Generator(device=dev); gen.manual_seed(1)
trace = torch.randint(0, 2**62, (n_traces * n_cols * R * ext * w,), dtype=torch.int64, device=dev, generator=gen)
polys = torch.empty_like(trace)
rw = 8 * ((n_cols * ext + 7) // 8)
lde = torch.empty(n_traces * N * rw * w, dtype=torch.int64, device=dev)
leaves = torch.empty((N, 32), dtype=torch.uint8, device=dev)
nodes = torch.empty((N, 32), dtype=torch.uint8, device=dev)
p = capi.make_params(field, ext, logR, logB, n_cols, n_traces)
s = torch.cuda.Stream(device=dev)
ctx.profile_enable(2)
with torch.cuda.stream(s):
    for _ in range(2):
        ctx.trace_commit_dev(p, trace.data_ptr(), polys.data_ptr(), lde.data_ptr(), leaves.data_ptr(), nodes.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize(); ctx.profile_read()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 5
    e0.record()
    for _ in range(K):
        ctx.trace_commit_dev(p, trace.data_ptr(), polys.data_ptr(), lde.data_ptr(), leaves.data_ptr(), nodes.data_ptr(), s.cuda_stream)
    e1.record(); torch.cuda.synchronize()
acc = {}
for k, v in ctx.profile_read():
    acc.setdefault(k, []).append(v)
print(f"field={field} ext={ext} R=2^{logR} blowup={1<<logB} cols={n_cols} traces={n_traces}: {e0.elapsed_time(e1)/K:.3f} ms",
      {k: round(sum(v)/len(v), 4) for k, v in acc.items()})
