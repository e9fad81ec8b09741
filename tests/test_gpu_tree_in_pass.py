"""GPU: the first three levels of the Merkle tree built inside the fused last evaluation pass (k_seg_last_hash<.., TREE>,
seg_kernels.hpp): one segment of one trace, blowup 8, 32-byte digests -- the work-group that finishes the last of a row
block's eight cosets hashes that block's leaves three levels up, and the tree kernels start from there (run_merkle's
`skip`, path.hip).  By default the route is taken where the persistent last pass runs anyway (one-segment f64 traces of
2^17 rows and up; cfg 2 of the benchmark); contexts created with WF_EXP_PERSISTENT_ALWAYS send small shapes through it.
Everything is compared with the oracle into POISONED node buffers, twice in a row (the row-block counters reset
themselves), and with a context that has the route switched off (WF_EXP_NO_TREE_IN_PASS)."""
import os

import numpy as np
import pytest

from conftest import rand_cols

pytestmark = pytest.mark.gpu
F64, F128 = 1, 2


def make_ctx(capi, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    os.environ["WF_EXP_ENABLE"] = "1"
    try:
        return capi.Context(0)
    finally:
        del os.environ["WF_EXP_ENABLE"]
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.fixture(scope="module")
def forced(capi):
    c = make_ctx(capi, WF_EXP_PERSISTENT_ALWAYS=1, WF_EXP_MAX_DIGIT=7)
    yield c
    c.close()


def commit_poisoned(ctx, capi, field, logR, logB, cols, digest_bytes=32, calls=2):
    import torch
    dev = torch.device("cuda", 0)
    n_cols = len(cols)
    N = 1 << (logR + logB)
    params = capi.make_params(field, 1, logR, logB, n_cols, 1, digest_bytes=digest_bytes)
    w = 1 if field == F64 else 2
    rw = 8 * ((n_cols + 7) // 8)
    flat = np.concatenate([np.ascontiguousarray(c).reshape(-1) for c in cols]).view(np.int64)
    d_trace = torch.from_numpy(flat.copy()).to(dev)
    d_polys = torch.empty_like(d_trace)
    d_lde = torch.full((N * rw * w,), -1, dtype=torch.int64, device=dev)
    out = []
    for _ in range(calls):  # fresh poisoned outputs per call: a level that is not written shows
        d_leaves = torch.full((N, 32), 0xEE, dtype=torch.uint8, device=dev)
        d_nodes = torch.full((N, 32), 0xEE, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.trace_commit_dev(params, d_trace.data_ptr(), d_polys.data_ptr(), d_lde.data_ptr(), d_leaves.data_ptr(), d_nodes.data_ptr())
        ctx.synchronize()
        out.append((d_leaves.cpu().numpy(), d_nodes.cpu().numpy()))
    return out


@pytest.mark.parametrize("field,logR,n_cols", [
    (F64, 14, 8), (F64, 15, 5), (F64, 16, 8), (F64, 17, 7),   # f64: one segment = up to 8 columns
    (F128, 14, 4), (F128, 15, 3), (F128, 16, 4),             # f128: up to 4
])
def test_tree_levels_from_the_last_pass_match_the_oracle(forced, orc, capi, field, logR, n_cols):
    rng = np.random.default_rng(hash((field, logR, n_cols)) % 2**32)
    cols = rand_cols(rng, field, n_cols, 1 << logR)
    want = orc.build_trace_commitment(field, [cols], 1, logR, 3, 7 if field == F64 else 3)
    for leaves, nodes in commit_poisoned(forced, capi, field, logR, 3, cols):
        assert np.array_equal(leaves, want["leaves"])
        assert np.array_equal(nodes, want["nodes"])


@pytest.mark.parametrize("field,logR,n_cols", [(F64, 20, 8), (F64, 18, 8), (F64, 21, 6), (F128, 20, 4)])
def test_default_route_equals_the_route_without_it(ctx, capi, field, logR, n_cols):
    """Shapes the DEFAULT context sends through the tree-building pass (cfg 2 is the first) against a context with the
    route off: same leaves, same nodes at every level, two calls in a row."""
    rng = np.random.default_rng(logR * 100 + n_cols)
    cols = rand_cols(rng, field, n_cols, 1 << logR)
    plain = make_ctx(capi, WF_EXP_NO_TREE_IN_PASS=1)
    try:
        a = commit_poisoned(ctx, capi, field, logR, 3, cols)
        b = commit_poisoned(plain, capi, field, logR, 3, cols, calls=1)
    finally:
        plain.close()
    for leaves, nodes in a:
        assert np.array_equal(leaves, b[0][0])
        assert np.array_equal(nodes, b[0][1])


@pytest.mark.parametrize("logB,digest_bytes", [(2, 32), (4, 32), (1, 32), (3, 24)])
def test_shapes_outside_the_route_are_unchanged(forced, orc, capi, logB, digest_bytes):
    """Other blowups and the 24-byte hasher keep the tree kernels for every level."""
    rng = np.random.default_rng(logB * 10 + digest_bytes)
    cols = rand_cols(rng, F64, 8, 1 << 14)
    with orc.digest_size(digest_bytes):
        want = orc.build_trace_commitment(F64, [cols], 1, 14, logB, 7)
    for leaves, nodes in commit_poisoned(forced, capi, F64, 14, logB, cols, digest_bytes=digest_bytes):
        if digest_bytes == 32:
            assert np.array_equal(leaves, want["leaves"]) and np.array_equal(nodes, want["nodes"])
        else:
            assert np.array_equal(leaves[:, :24], want["leaves"]) and np.array_equal(nodes[1:, :24], want["nodes"][1:])
