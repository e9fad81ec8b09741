import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import rand_cols
from oracle import oracle as O
import starkpack_winterfell_amd.capi as capi
ctx = capi.Context(0)
for (field, ext, logR, logB, n_cols, n_traces) in [(1,1,13,1,17,2),(1,1,11,1,17,2),(1,1,11,1,8,2),(1,1,11,1,9,2),(1,1,11,3,17,3),(2,1,12,3,10,2)]:
    rng = np.random.default_rng(5)
    R = 1 << logR
    traces = [rand_cols(rng, field, n_cols, R * ext) for _ in range(n_traces)]
    want = O.build_trace_commitment(field, traces, ext, logR, logB, 7 if field == 1 else 3)
    p = capi.make_params(field, ext, logR, logB, n_cols, n_traces)
    got = ctx.trace_commit(p, [c for t in traces for c in t])
    for t in range(n_traces):
        g, w = got["lde"][t], want["lde"][t]
        if field == 2: g = g[..., 0]; w = w[..., 0]
        bad = np.argwhere(g != w)
        print((field, ext, logR, logB, n_cols, n_traces), "trace", t, "mismatches", len(bad), "of", g.size)
        if len(bad):
            print("  first", bad[:5].tolist(), "cols", sorted(set(bad[:, 1].tolist()))[:30], "rows min/max", bad[:,0].min(), bad[:,0].max())
            r, c = bad[0]
            # does got value appear elsewhere in want?
            loc = np.argwhere(w == g[r, c])
            print("  got value found in want at", loc[:3].tolist())
