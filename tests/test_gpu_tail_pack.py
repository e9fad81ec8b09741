"""GPU: the tail-packed evaluation (k_seg_last_hash_tp + the coset-packed strided launch of the tail segment, path.hip):
one trace of several segments whose last segment is at most half full -- f128: 5, 6, 9, 10, 13, 14 .. columns, f64:
9 .. 12, 17 .. 20 .. -- has that segment evaluated two cosets per tile row and finished by one tail tile per coset pair.
By default only shapes of >= 2^20 LDE rows take the route (the ticket kernel's own rule); contexts created with
WF_EXP_PERSISTENT_ALWAYS (+ a digit cap, so that shorter traces run multi-pass plans) send small shapes through it.
Everything is compared with the oracle into POISONED buffers (the tail tile also writes the rows' zero padding), and with
a context that has the route switched off (WF_EXP_NO_TAIL_PACK)."""
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


@pytest.fixture(scope="module")
def forced9(capi):
    c = make_ctx(capi, WF_EXP_PERSISTENT_ALWAYS=1, WF_EXP_MAX_DIGIT=9)
    yield c
    c.close()


def commit_poisoned(ctx, capi, field, ext, logR, logB, cols, digest_bytes=32):
    import torch
    dev = torch.device("cuda", 0)
    n_cols = len(cols)
    N = 1 << (logR + logB)
    params = capi.make_params(field, ext, logR, logB, n_cols, 1, digest_bytes=digest_bytes)
    w = 1 if field == F64 else 2
    rw = 8 * ((n_cols * ext + 7) // 8)
    flat = np.concatenate([np.ascontiguousarray(c).reshape(-1) for c in cols]).view(np.int64)
    d_trace = torch.from_numpy(flat.copy()).to(dev)
    d_polys = torch.empty_like(d_trace)
    d_lde = torch.full((N * rw * w,), -1, dtype=torch.int64, device=dev)
    d_leaves = torch.full((N, 32), 0xEE, dtype=torch.uint8, device=dev)
    d_nodes = torch.full((N, 32), 0xEE, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for _ in range(2):  # a second call replays with warm tables and self-reset ticket counters
        ctx.trace_commit_dev(params, d_trace.data_ptr(), d_polys.data_ptr(), d_lde.data_ptr(), d_leaves.data_ptr(), d_nodes.data_ptr())
        ctx.synchronize()
    return (d_lde.cpu().numpy().view(np.uint64), d_leaves.cpu().numpy(), d_nodes.cpu().numpy())


@pytest.mark.parametrize("field,ext,logR,logB,n_cols", [
    (F128, 1, 14, 3, 10),   # the do_work shape: 2 full segments + 2 tail columns
    (F128, 1, 14, 3, 9), (F128, 1, 14, 1, 5), (F128, 1, 15, 2, 6), (F128, 1, 14, 3, 13), (F128, 1, 14, 3, 62),
    (F128, 2, 14, 3, 5),    # quadratic extension: 10 base columns
    (F64, 1, 14, 3, 9), (F64, 1, 14, 3, 10), (F64, 1, 15, 1, 11), (F64, 1, 14, 3, 12), (F64, 1, 14, 2, 17), (F64, 1, 14, 3, 20),
    (F64, 1, 14, 3, 124),   # 15 full segments + 4 tail columns: the longest row of one BLAKE3 chunk
    (F64, 2, 14, 3, 5), (F64, 3, 14, 3, 3),
])
def test_tail_packed_commitment_matches_the_oracle(forced, orc, capi, field, ext, logR, logB, n_cols):
    rng = np.random.default_rng(hash((field, ext, logR, logB, n_cols)) % 2**32)
    cols = rand_cols(rng, field, n_cols, (1 << logR) * ext)
    want = orc.build_trace_commitment(field, [cols], ext, logR, logB, 7 if field == F64 else 3)
    lde, leaves, nodes = commit_poisoned(forced, capi, field, ext, logR, logB, cols)
    assert np.array_equal(lde.reshape(-1), np.ascontiguousarray(want["lde"][0]).reshape(-1))
    assert np.array_equal(leaves, want["leaves"])
    assert np.array_equal(nodes, want["nodes"])


@pytest.mark.parametrize("field,logR,n_cols", [(F128, 18, 10), (F64, 18, 10), (F128, 17, 6), (F64, 19, 12)])
def test_tail_packing_default_route_equals_the_route_without_it(ctx, capi, field, logR, n_cols):
    """Shapes the DEFAULT context sends through the tail-packed kernels (>= 2^20 LDE rows; cfg 5 is the first) against a
    context with the route off: same LDE (padding included), leaves and tree."""
    rng = np.random.default_rng(logR * 100 + n_cols)
    cols = rand_cols(rng, field, n_cols, 1 << logR)
    plain = make_ctx(capi, WF_EXP_NO_TAIL_PACK=1)
    try:
        a = commit_poisoned(ctx, capi, field, 1, logR, 3, cols)
        b = commit_poisoned(plain, capi, field, 1, logR, 3, cols)
    finally:
        plain.close()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_tail_packed_2_9_row_tiles_and_blake3_192(forced9, orc, capi):
    """2^9-row tiles (the specialised instantiation) and the 24-byte hasher through the tail tile."""
    for field, n_cols in ((F128, 10), (F64, 10)):
        rng = np.random.default_rng(7 + field)
        cols = rand_cols(rng, field, n_cols, 1 << 16)
        with orc.digest_size(24):
            want = orc.build_trace_commitment(field, [cols], 1, 16, 3, 7 if field == F64 else 3)
        lde, leaves, nodes = commit_poisoned(forced9, capi, field, 1, 16, 3, cols, digest_bytes=24)
        assert np.array_equal(lde.reshape(-1), np.ascontiguousarray(want["lde"][0]).reshape(-1))
        assert np.array_equal(leaves[:, :24], want["leaves"]) and not leaves[:, 24:].any()
        assert np.array_equal(nodes[1:, :24], want["nodes"][1:])


@pytest.mark.parametrize("field,logR,n_cols,world", [(F128, 14, 10, 2), (F64, 14, 10, 4), (F128, 14, 6, 4)])
def test_tail_packed_coset_shards_match_the_oracle(forced, orc, capi, field, logR, n_cols, world):
    """The tail-packed route through wf_trace_commit_shard_dev with coset0 != 0 (a rank of a coset-sharded commitment evaluates
    blowup / W cosets: the tail work region sits behind n_cosets * (n_seg - 1) segments of the SHARD's work buffer and the leaf
    slots that park the first coset's chaining values are the shard's own rows).  Every shard's LDE rows (padding written into
    poisoned buffers) and leaves against the oracle's unsharded commitment."""
    import torch
    from starkpack_winterfell_amd import shard
    logB = 3
    rng = np.random.default_rng(field * 1000 + n_cols * 10 + world)
    R, blowup = 1 << logR, 1 << logB
    cols = rand_cols(rng, field, n_cols, R)
    want = orc.build_trace_commitment(field, [cols], 1, logR, logB, 7 if field == F64 else 3)
    params = capi.make_params(field, 1, logR, logB, n_cols, 1)
    w = 1 if field == F64 else 2
    rw = 8 * ((n_cols + 7) // 8)
    dev = torch.device("cuda", 0)
    d_trace = torch.from_numpy(np.concatenate([c.reshape(-1) for c in cols]).view(np.int64)).to(dev)
    full = want["lde"][0].reshape((R, blowup, rw) + ((2,) if w == 2 else ()))
    leaves_full = want["leaves"].reshape(R, blowup, 32)
    for rank in range(world):
        c0, nc = shard.cosets_of_rank(blowup, rank, world)
        d_polys = torch.empty_like(d_trace)
        d_lde = torch.full((R * nc * rw * w,), -1, dtype=torch.int64, device=dev)
        d_leaves = torch.full((R * nc, 32), 0xEE, dtype=torch.uint8, device=dev)
        for _ in range(2):
            forced.trace_commit_shard_dev(params, c0, nc, d_trace.data_ptr(), d_polys.data_ptr(), d_lde.data_ptr(), d_leaves.data_ptr())
            forced.synchronize()
        lde = d_lde.cpu().numpy().view(np.uint64).reshape((R, nc, rw) + ((2,) if w == 2 else ()))
        assert np.array_equal(lde, full[:, c0:c0 + nc]), f"rank {rank}"
        assert np.array_equal(d_leaves.cpu().numpy().reshape(R, nc, 32), leaves_full[:, c0:c0 + nc]), f"rank {rank}"
