"""Scratch timing of the device-resident path on one GPU (used during bring-up; bench.py is the real harness)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starkpack_winterfell_amd.capi as capi

def rand_f64(n, gen):
    v = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device="cuda", generator=gen)
    bad = (v >> 32) == -1
    return torch.where(bad, v & 0x7FFFFFFFFFFFFFFF, v)

def main():
    logR = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    logB = 3
    ctx = capi.Context(0)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    R = 1 << logR; N = R << logB
    p = capi.make_params(capi.F64, 1, logR, logB, ncols, 1)
    rw = 8 * ((ncols + 7) // 8)
    trace = rand_f64(ncols * R, gen)
    polys = torch.empty_like(trace)
    lde = torch.empty(N * rw, dtype=torch.int64, device="cuda")
    leaves = torch.empty(N * 32, dtype=torch.uint8, device="cuda")
    nodes = torch.empty(N * 32, dtype=torch.uint8, device="cuda")
    ts = torch.cuda.Stream()
    torch.cuda.synchronize()
    torch.cuda.set_stream(ts)
    st = ts.cuda_stream
    assert st != 0
    for it in range(3):
        ctx.trace_commit_dev(p, trace.data_ptr(), polys.data_ptr(), lde.data_ptr(), leaves.data_ptr(), nodes.data_ptr(), st)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    K = 10
    e0.record()
    for it in range(K):
        ctx.trace_commit_dev(p, trace.data_ptr(), polys.data_ptr(), lde.data_ptr(), leaves.data_ptr(), nodes.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    balg = R*ncols*8*2 + N*rw*8 + N*64
    print(f"logR={logR} cols={ncols}: {ms:.3f} ms/commit, B_alg={balg/2**20:.0f} MiB -> {balg/ms/1e9:.3f} TB/s, root={bytes(nodes[32:64].cpu().numpy()).hex()}")

main()
