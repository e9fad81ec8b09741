"""GPU: the rules of the boundary that protect a host from itself (include/wf_lde.h, wf_ctx_create):
one call at a time per context (WF_ERR_BUSY for a second thread, nothing corrupted), calls on different streams are
ordered by the library (they share the context's scratch), allocation failures come back as WF_ERR_HIP and leave the
context usable, parked buffers are given back under memory pressure."""
import threading

import numpy as np
import pytest

from conftest import rand_cols
from loopback import run_ranks

pytestmark = pytest.mark.gpu
F64 = 1


def test_second_thread_gets_busy_not_corruption(ctx, orc, capi):
    """Thread A is inside a long host-form commitment (2^18 rows, ~1 GiB copied out); thread B hammering the same
    context is refused with WF_ERR_BUSY whenever it meets A inside; A's result is right and B succeeds once A has left."""
    rng = np.random.default_rng(9)
    cols = rand_cols(rng, F64, 8, 1 << 10)
    want = orc.build_trace_commitment(F64, [cols], 1, 10, 3, 7)
    params = capi.make_params(F64, 1, 10, 3, 8, 1)
    started = threading.Event()
    results = {}

    def long_call():
        started.set()
        while True:   # (B may be inside at the very moment A arrives: A is then the one refused, and simply comes again)
            try:
                results["a"] = ctx.trace_commit(capi.make_params(F64, 1, 18, 3, 8, 1), big)
                return
            except capi.WfError as e:
                assert e.code == -20, e
                results["a_busy"] = results.get("a_busy", 0) + 1

    big = rand_cols(rng, F64, 8, 1 << 18)
    ta = threading.Thread(target=long_call)
    ta.start()
    started.wait(timeout=60)
    busy = 0
    while ta.is_alive():
        try:
            got = ctx.trace_commit(params, cols, want_lde=False, want_polys=False)
            assert got["root"] == want["root"]                      # entered between A's calls: still right
        except capi.WfError as e:
            assert e.code == -20, e
            busy += 1
    ta.join()
    assert busy + results.get("a_busy", 0) >= 1, "the two threads never met inside the context"
    want_big = orc.build_trace_commitment(F64, [big], 1, 18, 3, 7, threads=16)
    assert results["a"]["root"] == want_big["root"] and np.array_equal(results["a"]["nodes"], want_big["nodes"])
    assert ctx.trace_commit(params, cols)["root"] == want["root"]  # B succeeds once A has left


def test_calls_on_different_streams_are_ordered(ctx, orc, capi):
    """A trace commitment on stream A and a constraint commitment on stream B right behind it share the context's
    scratch: the library makes B wait for A (no host synchronisation in between)."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(4)
    logR, logB = 17, 3
    R, N = 1 << logR, 1 << (logR + logB)
    cols = rand_cols(rng, F64, 8, R)
    polys = rand_cols(rng, F64, 2, R * 2)                       # two columns of the quadratic extension
    want_t = orc.build_trace_commitment(F64, [cols], 1, logR, logB, 7, threads=16)
    want_c = orc.build_constraint_commitment(F64, polys, 2, logR, logB, 7, threads=16)
    pt, pc = capi.make_params(F64, 1, logR, logB, 8, 1), capi.make_params(F64, 2, logR, logB, 2, 1)
    d_trace = torch.from_numpy(np.concatenate(cols).view(np.int64)).to(dev)
    d_cpolys = torch.from_numpy(np.concatenate(polys).view(np.int64)).to(dev)
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    for _ in range(3):
        o = [torch.empty_like(d_trace), torch.empty(N * 8, dtype=torch.int64, device=dev),
             torch.empty((N, 32), dtype=torch.uint8, device=dev), torch.empty((N, 32), dtype=torch.uint8, device=dev)]
        c = [torch.empty(N * 8, dtype=torch.int64, device=dev), torch.empty((N, 32), dtype=torch.uint8, device=dev),
             torch.empty((N, 32), dtype=torch.uint8, device=dev)]
        torch.cuda.synchronize()
        ctx.trace_commit_dev(pt, d_trace.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), sa.cuda_stream)
        ctx.constraint_commit_dev(pc, d_cpolys.data_ptr(), c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), sb.cuda_stream)
        ctx.trace_commit_dev(pt, d_trace.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), sa.cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(o[3].cpu().numpy(), want_t["nodes"])
        assert np.array_equal(c[2].cpu().numpy(), want_c["nodes"])
        assert np.array_equal(c[0].cpu().numpy().view(np.uint64).reshape(N, 8), want_c["lde"])


def test_allocation_failure_is_reported_and_survivable(ctx, orc, capi):
    """Parameters whose scratch exceeds the device: WF_ERR_HIP, no abort, and the context keeps working -- also when
    buffers of destroyed resident commitments were parked (they are released for the retry)."""
    rng = np.random.default_rng(2)
    cols = rand_cols(rng, F64, 8, 1 << 12)
    params = capi.make_params(F64, 1, 12, 3, 8, 1)
    want = orc.build_trace_commitment(F64, [cols], 1, 12, 3, 7)
    com, _ = ctx.trace_commit_resident(params, cols)
    com.close()                                                 # its buffers are parked in the context now
    huge = capi.make_params(F64, 1, 25, 7, 255, 1)              # 128 GiB of segment scratch + 8 TiB of intermediate
    with pytest.raises(capi.WfError) as e:
        ctx.trace_commit_dev(huge, 256, 256, 256, 256, 256)     # never dereferenced: the scratch allocation fails first
    assert e.value.code == -30 and "hipMalloc" in str(e.value)
    got = ctx.trace_commit(params, cols)
    assert got["root"] == want["root"] and np.array_equal(got["nodes"], want["nodes"])
    com, _ = ctx.trace_commit_resident(params, cols)            # and the resident form allocates afresh
    assert com.root() == want["root"]
    com.close()


def test_every_thread_its_own_context_is_fine(orc, capi):
    rng = np.random.default_rng(1)
    cols = rand_cols(rng, F64, 3, 1 << 10)
    want = orc.build_trace_commitment(F64, [cols], 1, 10, 2, 7)
    params = capi.make_params(F64, 1, 10, 2, 3, 1)

    def fn(_r):
        c = capi.Context(0)
        roots = [c.trace_commit(params, cols, want_lde=False)["root"] for _ in range(5)]
        c.close()
        return roots

    assert run_ranks(4, fn) == [[want["root"]] * 5] * 4


def test_handles_destroyed_after_their_context_do_not_touch_it(orc, capi):
    """The rule is commitments and provers first, the context last -- but a host with a garbage collector may finalise in
    any order at shutdown.  Destroying late must not write into the dead context (the library keeps a registry of live
    contexts and frees the buffers directly); destroying a context twice is ignored."""
    import ctypes as C
    L = capi.load()
    rng = np.random.default_rng(5)
    cols = rand_cols(rng, F64, 4, 1 << 9)
    params = capi.make_params(F64, 1, 9, 2, 4, 1)
    c = capi.Context(0)
    com, _ = c.trace_commit_resident(params, cols)
    pr = capi.FriProver(c, F64, 1, 4, 4, 7, 7)
    pr.begin(rand_cols(rng, F64, 1, 1 << 9)[0])
    pr.commit_layer()
    h_ctx, h_com, h_pr = c._h, com._h, pr._h
    c._children = []                       # take the Python-side ordering out of the picture
    c._h = C.c_void_p()
    com._h = C.c_void_p()
    pr._h = C.c_void_p()
    L.wf_ctx_destroy(h_ctx)
    L.wf_ctx_destroy(h_ctx)                # twice: ignored
    L.wf_commitment_destroy(h_com)
    L.wf_fri_prover_destroy(h_pr)
    c2 = capi.Context(0)                   # the heap is intact: a full commitment on a fresh context
    want = orc.build_trace_commitment(F64, [cols], 1, 9, 2, 7)
    assert c2.trace_commit(params, cols)["root"] == want["root"]
    c2.close()


def test_failed_pipelined_upload_drains_its_streams_and_leaves_the_context_usable(orc, capi):
    """The pipelined upload of a multi-segment trace (trace_commit_pipelined) fails part-way -- WF_EXP_FAIL_AFTER_SEGMENT, a
    switch the library reads once at wf_ctx_create -- with copies of the caller's columns and kernels already queued.  The
    call must come back with the error AFTER both streams have drained (the columns may be freed at once), release what
    it allocated, and leave the context serving other shapes."""
    import os
    from conftest import rand_cols
    os.environ["WF_EXP_ENABLE"] = "1"
    os.environ["WF_EXP_FAIL_AFTER_SEGMENT"] = "2"
    os.environ["WF_EXP_PIPELINE_MIN_BYTES"] = "0"
    try:
        ctx = capi.Context(0)
    finally:
        del os.environ["WF_EXP_FAIL_AFTER_SEGMENT"], os.environ["WF_EXP_PIPELINE_MIN_BYTES"], os.environ["WF_EXP_ENABLE"]
    rng = np.random.default_rng(9)
    wide = capi.make_params(1, 1, 12, 3, 40, 1)      # five segments, two passes: the pipelined route
    cols = rand_cols(rng, 1, 40, 1 << 12)
    for _ in range(3):
        with pytest.raises(capi.WfError) as e:
            ctx.trace_commit_resident(wide, cols)
        assert e.value.code == -30 and "injected failure" in str(e.value)
        cols = rand_cols(rng, 1, 40, 1 << 12)          # the old columns are garbage-collected while nothing may still read them
    # other routes of the same context are untouched: one segment (not pipelined), and the host-buffer form of the wide shape
    narrow = capi.make_params(1, 1, 12, 3, 8, 1)
    com, _ = ctx.trace_commit_resident(narrow, cols[:8])
    assert com.root() == orc.build_trace_commitment(1, [cols[:8]], 1, 12, 3, 7)["root"]
    com.close()
    got = ctx.trace_commit(wide, cols)
    assert got["root"] == orc.build_trace_commitment(1, [cols], 1, 12, 3, 7)["root"]
    ctx.close()
