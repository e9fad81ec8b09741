"""GPU: single-pass evaluations whose combined rows are longer than one BLAKE3 chunk -- many packed traces of few steps,
the reference's own example at its defaults (examples/src/lib.rs:97-135: 512 do_work traces of 2^10 steps, 80-chunk rows).
Default route: plain evaluation pass + the chunk hashing kernels.  Opt-in route (WF_EXP_SINGLE_FUSED, read at wf_ctx_create):
leaves hashed inside the evaluation pass with the chunk chaining values handed from work-group to work-group through memory
(k_seg_single_hash -- bit-exact but measured slower): every leaf, node and LDE row against the oracle both
ways, both ways of dealing the (coset, chunk) pairs to the XCDs, both fields, ragged last chunks."""
import os

import numpy as np
import pytest

from conftest import rand_cols

pytestmark = pytest.mark.gpu
F64, F128 = 1, 2

SHAPES = [
    (F128, 10, 3, 10, 512),   # the do_work default: 1280 segments, rows of 80 chunks, padded rows of 16 elements
    (F128, 9, 3, 4, 300),     # 2^9-row tiles (four work-groups per CU), 300 segments: 18 full chunks + one of 12 blocks
    (F64, 10, 3, 8, 200),     # f64: 200 segments = 12.5 chunks
    (F64, 9, 4, 5, 300),      # 16 cosets (two per XCD), ragged columns: 1500 base columns = 187.5 segments
    (F64, 9, 2, 8, 600),      # 4 cosets: (coset, chunk) pairs dealt to the XCDs one by one
]


@pytest.mark.parametrize("field,logR,logB,n_cols,n_traces", SHAPES)
def test_single_pass_long_rows(orc, capi, field, logR, logB, n_cols, n_traces):
    os.environ["WF_EXP_SINGLE_FUSED"] = "1"
    try:
        ctx = capi.Context(0)
    finally:
        del os.environ["WF_EXP_SINGLE_FUSED"]
    rng = np.random.default_rng(logR * 1000 + n_cols * 10 + n_traces)
    R = 1 << logR
    traces = [rand_cols(rng, field, n_cols, R) for _ in range(n_traces)]
    want = orc.build_trace_commitment(field, traces, 1, logR, logB, 7 if field == F64 else 3, threads=min(32, len(os.sched_getaffinity(0))))
    params = capi.make_params(field, 1, logR, logB, n_cols, n_traces)
    cols = [c for t in traces for c in t]
    for rep in range(2):  # the second call reuses the chaining flags with the next epoch
        got = ctx.trace_commit(params, cols)
        assert got["root"] == want["root"], rep
        assert np.array_equal(got["leaves"], want["leaves"])
        assert np.array_equal(got["nodes"], want["nodes"])
    for t in (0, 1, n_traces // 2, n_traces - 1):
        assert np.array_equal(got["lde"][t], want["lde"][t]), ("lde of trace", t)
    # resident + device error word path, and the same shape through the separate hashing kernels
    com, _ = ctx.trace_commit_resident(params, cols)
    assert com.root() == want["root"]
    com.close()
    ctx.close()
    plain = capi.Context(0)  # the default route
    got2 = plain.trace_commit(params, cols, want_lde=False, want_polys=False)
    assert got2["root"] == want["root"] and np.array_equal(got2["leaves"], want["leaves"])
    plain.close()
