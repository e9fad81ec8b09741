"""Times the device-buffer trace commitment over a grid of shapes in one process (tuning aid: anomalies show up as jumps
in the time per LDE element).   python scripts/sweep.py [f64|f128] > gpurun_out/sweep.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starkpack_winterfell_amd.capi as capi

field = 2 if len(sys.argv) > 1 and sys.argv[1] == "f128" else 1
w = 1 if field == 1 else 2
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
logB = 3
widths = [1, 2, 4, 8, 16, 64] if field == 1 else [1, 2, 4, 10, 32]
print("log2R " + " ".join(f"{c:>16d}" for c in widths) + "   (ms | ps per LDE element)")
for logR in range(10, 24):
    line = f"{logR:5d} "
    for n_cols in widths:
        R, N = 1 << logR, 1 << (logR + logB)
        rw = 8 * ((n_cols + 7) // 8)
        if N * rw * w * 8 > 24 << 30:
            line += f"{'-':>16s} "
            continue
        trace = torch.randint(0, 2**62, (n_cols * R * w,), dtype=torch.int64, device=dev)
        polys = torch.empty_like(trace)
        lde = torch.empty(N * rw * w, dtype=torch.int64, device=dev)
        leaves = torch.empty((N, 32), dtype=torch.uint8, device=dev)
        nodes = torch.empty((N, 32), dtype=torch.uint8, device=dev)
        p = capi.make_params(field, 1, logR, logB, n_cols, 1)
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            for _ in range(2):
                ctx.trace_commit_dev(p, trace.data_ptr(), polys.data_ptr(), lde.data_ptr(), leaves.data_ptr(), nodes.data_ptr(), s.cuda_stream)
            torch.cuda.synchronize()
            K = 5 if logR >= 20 else 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(K):
                ctx.trace_commit_dev(p, trace.data_ptr(), polys.data_ptr(), lde.data_ptr(), leaves.data_ptr(), nodes.data_ptr(), s.cuda_stream)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        line += f"{ms:8.3f} {ms * 1e9 / (N * n_cols):7.1f} "
        del trace, polys, lde, leaves, nodes
    print(line, flush=True)
