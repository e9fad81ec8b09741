"""GPU: resident commitments of WIDE traces built from host columns (several segments: the shapes whose upload runs under
the kernels, trace_commit_pipelined) against the oracle: root, polynomials, queried rows and proofs."""
import numpy as np
import pytest

from conftest import rand_cols

pytestmark = pytest.mark.gpu
F64, F128 = 1, 2


@pytest.mark.parametrize("field,logR,logB,n_cols,n_traces", [
    (F64, 11, 3, 20, 1),     # 3 segments, the last one ragged
    (F64, 12, 2, 9, 3),      # packed traces across segment borders
    (F64, 11, 3, 64, 1),     # 8 full segments
    (F64, 13, 1, 130, 1),    # 17 segments: rows longer than one BLAKE3 chunk
    (F128, 11, 3, 10, 1),    # f128: 4 lanes per segment
    (F128, 12, 2, 5, 2),
    (F64, 8, 3, 24, 1),      # single-pass size (pipelined only under WF_EXP_MAX_DIGIT)
])
def test_wide_resident_commitment(ctx, orc, capi, field, logR, logB, n_cols, n_traces):
    rng = np.random.default_rng(logR * 1000 + n_cols)
    off = 7 if field == F64 else 3
    traces = [rand_cols(rng, field, n_cols, 1 << logR) for _ in range(n_traces)]
    want = orc.build_trace_commitment(field, traces, 1, logR, logB, off, threads=8)
    com, polys = ctx.trace_commit_resident(capi.make_params(field, 1, logR, logB, n_cols, n_traces),
                                           [c for t in traces for c in t], want_polys=True)
    assert com.root() == want["root"]
    for t in range(n_traces):
        for c in range(n_cols):
            assert np.array_equal(polys[t * n_cols + c], want["polys"][t][c]), f"polynomial {t}.{c}"
    N = 1 << (logR + logB)
    pos = np.unique(rng.integers(0, N, size=16))
    rows, proof = com.query(pos)
    want_rows = np.concatenate([want["lde"][t][pos][:, :n_cols] for t in range(n_traces)], axis=1)
    assert np.array_equal(rows.reshape(want_rows.shape), want_rows)
    assert proof == orc.merkle_prove_batch(want["nodes"], want["leaves"], [int(p) for p in pos])
    # and once more on the same context (staging buffers and events are reused)
    com2, _ = ctx.trace_commit_resident(capi.make_params(field, 1, logR, logB, n_cols, n_traces), [c for t in traces for c in t])
    assert com2.root() == want["root"]
    com.close()
    com2.close()


@pytest.mark.parametrize("field,logR,n_cols", [(F64, 17, 24), (F64, 18, 20), (F128, 16, 10)])
def test_wide_resident_commitment_at_the_default_threshold(ctx, orc, capi, field, logR, n_cols):
    """Columns of 1 MiB and more take the pipelined route without any switch: root, polynomials and sampled rows against the
    threaded oracle."""
    rng = np.random.default_rng(logR + n_cols)
    off = 7 if field == F64 else 3
    trace = rand_cols(rng, field, n_cols, 1 << logR)
    want = orc.build_trace_commitment(field, [trace], 1, logR, 3, off, threads=16)
    com, polys = ctx.trace_commit_resident(capi.make_params(field, 1, logR, 3, n_cols, 1), trace, want_polys=True)
    assert com.root() == want["root"]
    for c in range(n_cols):
        assert np.array_equal(polys[c], want["polys"][0][c])
    N = 1 << (logR + 3)
    pos = np.unique(rng.integers(0, N, size=32))
    rows, proof = com.query(pos)
    assert np.array_equal(rows.reshape(len(pos), -1), want["lde"][0].reshape(N, -1)[pos][:, :rows.reshape(len(pos), -1).shape[1]])
    assert proof == orc.merkle_prove_batch(want["nodes"], want["leaves"], [int(p) for p in pos])
    com.close()
