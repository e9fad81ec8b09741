"""Experiment: independent commitments (cfg 2, device-resident) issued alternately on two contexts / two streams from one
host thread, against the same number on one -- does the second stream fill the idle tails (Merkle top levels,
interpolation tails, launch gaps) of the first?
    python scripts/time_two_streams.py [n_streams]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starkpack_winterfell_amd.capi as capi

logR, logB, n_cols = 20, 3, 8
dev = torch.device("cuda", 0)
R, N = 1 << logR, 1 << (logR + logB)
p = capi.make_params(1, 1, logR, logB, n_cols, 1)
gen = torch.Generator(device=dev); gen.manual_seed(1)


class Lane:
    def __init__(self):
        self.ctx = capi.Context(0)
        self.stream = torch.cuda.Stream(device=dev)
        self.trace = torch.randint(0, 2**62, (n_cols * R,), dtype=torch.int64, device=dev, generator=gen)
        self.polys = torch.empty_like(self.trace)
        self.lde = torch.empty(N * 8, dtype=torch.int64, device=dev)
        self.leaves = torch.empty((N, 32), dtype=torch.uint8, device=dev)
        self.nodes = torch.empty((N, 32), dtype=torch.uint8, device=dev)

    def commit(self):
        self.ctx.trace_commit_dev(p, self.trace.data_ptr(), self.polys.data_ptr(), self.lde.data_ptr(), self.leaves.data_ptr(),
                                  self.nodes.data_ptr(), self.stream.cuda_stream)


def run(lanes, K):
    for l in lanes:
        l.commit()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(K):
        lanes[k % len(lanes)].commit()
    for l in lanes:
        torch.cuda.current_stream().wait_stream(l.stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K


lanes = [Lane() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2)]
for rep in range(3):
    print(f"rep {rep}: one stream {run(lanes[:1], 40):.4f} ms per commitment; {len(lanes)} streams {run(lanes, 40):.4f} ms per commitment", flush=True)
