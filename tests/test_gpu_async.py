"""GPU: wf_trace_commit_resident_async / wf_commitment_wait -- a stream of proofs from host columns, the upload of proof
k + 1 under the kernels of proof k (STARKPack proves many traces one after the other: examples/src/lib.rs:97-135).
Every commitment must be the one the synchronous entry point and the oracle produce."""
import ctypes as C

import numpy as np
import pytest

from conftest import rand_cols

pytestmark = pytest.mark.gpu
F64, F128 = 1, 2


@pytest.mark.parametrize("field,logR,logB,n_cols,n_traces,n_proofs", [
    (F64, 12, 3, 8, 1, 5), (F64, 10, 2, 3, 2, 4), (F128, 10, 3, 10, 1, 3), (F64, 14, 3, 20, 1, 3), (F64, 8, 1, 1, 1, 6)])
def test_stream_of_proofs_matches_oracle(orc, capi, field, logR, logB, n_cols, n_traces, n_proofs):
    ctx = capi.Context(0)
    rng = np.random.default_rng(logR * 100 + n_cols)
    R = 1 << logR
    params = capi.make_params(field, 1, logR, logB, n_cols, n_traces)
    proofs = [[rand_cols(rng, field, n_cols, R) for _ in range(n_traces)] for _ in range(n_proofs)]
    wants = [orc.build_trace_commitment(field, tr, 1, logR, logB, 7 if field == F64 else 3) for tr in proofs]
    coms = [ctx.trace_commit_resident_async(params, [c for t in tr for c in t]) for tr in proofs]
    # queries are ordered behind the kernels on the context's stream: legal before the handle has been waited for
    N = R << logB
    pos = [0, 1, N // 2 + 1, N - 1]
    for com, want in zip(coms, wants):
        rows, proof = com.query(pos)
        want_rows = np.concatenate([want["lde"][t][pos][:, :n_cols] for t in range(n_traces)], axis=1)
        assert np.array_equal(rows.reshape(want_rows.shape), want_rows)
        assert proof == orc.merkle_prove_batch(want["nodes"], want["leaves"], pos)
    for com, want in zip(coms, wants):
        com.wait()
        assert com.root() == want["root"]
        com.wait()  # idempotent
    # interleaved with the synchronous entry point on the same context
    sync, _ = ctx.trace_commit_resident(params, [c for t in proofs[0] for c in t])
    again = ctx.trace_commit_resident_async(params, [c for t in proofs[-1] for c in t])
    assert sync.root() == wants[0]["root"] and again.root() == wants[-1]["root"]  # root() waits by itself
    for c in coms + [sync, again]:
        c.close()
    ctx.close()


def test_pinned_columns_and_destroy_while_pending(orc, capi):
    import torch
    ctx = capi.Context(0)
    rng = np.random.default_rng(5)
    logR, logB, n_cols = 16, 3, 8
    params = capi.make_params(F64, 1, logR, logB, n_cols, 1)
    cols = rand_cols(rng, F64, n_cols, 1 << logR)
    pinned = [torch.from_numpy(c.view(np.int64)).pin_memory().numpy().view(np.uint64) for c in cols]
    want = orc.build_trace_commitment(F64, [cols], 1, logR, logB, 7, threads=8)
    coms = [ctx.trace_commit_resident_async(params, pinned) for _ in range(6)]
    coms[0].close()          # destroyed while its kernels may still run: waits, parks the buffers
    coms[3].close()
    for c in coms[1:3] + coms[4:]:
        assert c.root() == want["root"]
        c.close()
    ctx.close()


def test_errors_leave_the_context_usable(orc, capi):
    ctx = capi.Context(0)
    L = capi.load()
    rng = np.random.default_rng(6)
    params = capi.make_params(F64, 1, 10, 3, 8, 1)
    cols = rand_cols(rng, F64, 8, 1 << 10)
    h = C.c_void_p()
    arr = capi._ptr_array(cols)
    arr[5] = None
    assert L.wf_trace_commit_resident_async(ctx._h, C.byref(params), arr, C.byref(h)) == -19
    assert L.wf_trace_commit_resident_async(ctx._h, C.byref(params), None, C.byref(h)) == -19
    bad = capi.make_params(F64, 1, 2, 3, 8, 1)
    assert L.wf_trace_commit_resident_async(ctx._h, C.byref(bad), capi._ptr_array(cols), C.byref(h)) == -12
    assert L.wf_commitment_wait(None) == -19
    want = orc.build_trace_commitment(F64, [cols], 1, 10, 3, 7)
    # more handles in flight than pinned root slots: WF_ERR_BUSY, and everything queued so far still completes
    coms = []
    for _ in range(256):
        coms.append(ctx.trace_commit_resident_async(params, cols))
    with pytest.raises(capi.WfError) as e:
        ctx.trace_commit_resident_async(params, cols)
    assert e.value.code == -20
    assert coms[0].root() == want["root"]           # frees a slot
    extra = ctx.trace_commit_resident_async(params, cols)
    assert extra.root() == want["root"] and coms[-1].root() == want["root"]
    for c in coms + [extra]:
        c.close()
    ctx.close()
