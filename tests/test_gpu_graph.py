"""GPU: the device-buffer form captured into a HIP graph (torch.cuda.CUDAGraph) and replayed.  Nothing in a commitment
may depend on host-side work between launches: every replay has to redo all of it -- outputs are wiped before each one
and compared with a direct call.  (The ticket counters of the persistent last pass reset themselves for this reason;
a 32-byte memset node in front of the kernel did not survive replays.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F64, F128 = 1, 2


@pytest.mark.parametrize("field,logR,logB,n_cols,n_traces", [
    (F64, 20, 3, 8, 1),     # the bench workload: persistent one-segment last pass (tickets)
    (F64, 14, 3, 8, 1),     # one work-group per tile, leaves from the pass
    (F64, 13, 1, 17, 2),    # packed traces: persistent multi-segment pass, padded rows
    (F64, 12, 1, 200, 1),   # rows of two BLAKE3 chunks: chunk chaining values + merge
    (F64, 12, 3, 1, 1),     # coset-packed lanes + separate row hashing
    (F128, 12, 3, 1, 1),    # coset-packed f128: the one shape class that still clears its LDE with a memset
    (F128, 10, 3, 10, 8),   # single pass, long gathered rows
])
def test_commitment_replays_from_a_graph(capi, ctx, field, logR, logB, n_cols, n_traces):
    import torch
    dev = torch.device("cuda", 0)
    w = 1 if field == F64 else 2
    R, N = 1 << logR, 1 << (logR + logB)
    rw = 8 * ((n_cols + 7) // 8)
    gen = torch.Generator(device=dev)
    gen.manual_seed(logR * 100 + n_cols)
    trace = torch.randint(0, 2**62, (n_traces * n_cols * R * w,), dtype=torch.int64, device=dev, generator=gen)
    polys = torch.empty_like(trace)
    lde = torch.empty(n_traces * N * rw * w, dtype=torch.int64, device=dev)
    leaves = torch.empty((N, 32), dtype=torch.uint8, device=dev)
    nodes = torch.empty((N, 32), dtype=torch.uint8, device=dev)
    p = capi.make_params(field, 1, logR, logB, n_cols, n_traces)
    s = torch.cuda.Stream(device=dev)

    def call(stream):
        ctx.trace_commit_dev(p, trace.data_ptr(), polys.data_ptr(), lde.data_ptr(), leaves.data_ptr(), nodes.data_ptr(), stream)

    with torch.cuda.stream(s):
        for _ in range(2):   # scratch buffers and tables exist before the capture
            call(s.cuda_stream)
        torch.cuda.synchronize()
    want = [t.clone() for t in (polys, lde, leaves, nodes)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        call(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for _ in range(3):
        for t in (polys, lde, leaves, nodes):
            t.fill_(-1 if t.dtype == torch.int64 else 255)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        for got, exp, name in zip((polys, lde, leaves, nodes), want, ("polys", "lde", "leaves", "nodes")):
            assert torch.equal(got, exp), name
    # and a direct call after the replays still works (counters left clean)
    for t in (polys, lde, leaves, nodes):
        t.fill_(-1 if t.dtype == torch.int64 else 255)
    call(0)
    torch.cuda.synchronize()
    assert torch.equal(nodes, want[3]) and torch.equal(lde, want[1])
