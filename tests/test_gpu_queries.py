"""GPU: resident commitments and the query service (rows + Merkle proofs read from HBM).

Reference behaviour: TraceCommitment::query / build_segment_queries (prover/src/trace/commitment.rs:87-111,135-190),
ConstraintCommitment::query (prover/src/constraints/commitment.rs:54-69), MerkleTree::prove / prove_batch
(crypto/src/merkle/mod.rs:192-284); proof checks as in crypto/src/merkle/tests.rs:94-131,154-208."""
import numpy as np
import pytest

from conftest import rand_cols, rand_f64

pytestmark = pytest.mark.gpu
F64, F128 = 1, 2


@pytest.mark.parametrize("field,ext,logR,logB,n_cols,n_traces", [
    (F64, 1, 10, 3, 8, 1), (F64, 1, 12, 2, 5, 3), (F128, 1, 9, 3, 10, 2), (F64, 2, 11, 3, 3, 1)])
def test_resident_trace_commitment_queries(ctx, orc, capi, field, ext, logR, logB, n_cols, n_traces):
    rng = np.random.default_rng(11 * logR + n_cols)
    traces = [rand_cols(rng, field, n_cols, (1 << logR) * ext) for _ in range(n_traces)]
    offset = 7 if field == F64 else 3
    want = orc.build_trace_commitment(field, traces, ext, logR, logB, offset)
    params = capi.make_params(field, ext, logR, logB, n_cols, n_traces)
    com, polys = ctx.trace_commit_resident(params, [c for t in traces for c in t], want_polys=True)
    N, epr = 1 << (logR + logB), n_cols * ext
    assert com.root() == want["root"]
    assert (com.n_rows, com.row_elems, com.depth) == (N, epr * n_traces, logR + logB)
    for t in range(n_traces):
        for c in range(n_cols):
            assert np.array_equal(polys[t * n_cols + c], want["polys"][t][c])

    # rows: row p of trace 0 || trace 1 || ..  (comb_states of build_segment_queries)
    positions = [0, 1, N - 1, 5, 4, N // 2 + 3] + [int(x) for x in rng.choice(N, size=40, replace=False)]
    positions = list(dict.fromkeys(positions))
    rows = com.read_rows(positions)
    for i, p in enumerate(positions):
        for t in range(n_traces):
            assert np.array_equal(rows[i, t * epr:(t + 1) * epr], want["lde"][t][p, :epr])
        # the queried rows hash to the committed leaf (what the verifier re-checks, air/src/proof/queries.rs:128)
        assert orc.hash_elements(field, rows[i]) == bytes(want["leaves"][p])

    # single paths
    for p in positions[:8]:
        proof = com.prove(p)
        assert proof == orc.merkle_prove(want["nodes"], want["leaves"], p)
        assert orc.merkle_verify(want["root"], p, proof)

    # batch ("octopus") proofs, including adjacent positions and a single position
    for sel in (positions, positions[:1], [6, 7], [N - 2, N - 1, 0], sorted(positions)[:17]):
        got = com.prove_batch(sel)
        assert got == orc.merkle_prove_batch(want["nodes"], want["leaves"], sel)
        q_rows, q_proof = com.query(sel)                 # both in one round trip (TraceCommitment::query)
        assert np.array_equal(q_rows, com.read_rows(sel)) and q_proof == got
    com.close()


def test_resident_constraint_commitment(ctx, orc, capi):
    rng = np.random.default_rng(3)
    polys = rand_cols(rng, F64, 4, (1 << 10) * 2)
    want = orc.build_constraint_commitment(F64, polys, 2, 10, 3, 7)
    com = ctx.constraint_commit_resident(capi.make_params(F64, 2, 10, 3, 4, 1), polys)
    assert com.root() == want["root"]
    pos = [3, 8191, 77, 4096]
    rows = com.read_rows(pos)
    for i, p in enumerate(pos):
        assert np.array_equal(rows[i], want["lde"][p, :8])
    assert com.prove_batch(pos) == orc.merkle_prove_batch(want["nodes"], want["leaves"], pos)
    q_rows, q_proof = com.query(pos)
    assert np.array_equal(q_rows, rows) and q_proof == com.prove_batch(pos)
    com.close()


def test_query_errors(ctx, capi):
    rng = np.random.default_rng(4)
    cols = rand_cols(rng, F64, 2, 64)
    com, _ = ctx.trace_commit_resident(capi.make_params(F64, 1, 6, 1, 2, 1), cols)
    for bad, code in (([], -19), (list(range(128)) * 2, -19), ([1, 1], -18), ([128], -18)):
        with pytest.raises(capi.WfError) as e:
            com.prove_batch(bad)
        with pytest.raises(capi.WfError) as e:
            com.query(bad)
        assert e.value.code == code
    with pytest.raises(capi.WfError) as e:
        com.read_rows([500])
    assert e.value.code == -18
    with pytest.raises(capi.WfError) as e:
        com.prove(128)
    assert e.value.code == -18
    assert len(com.prove_batch(list(range(128)))[0]) == 128   # every leaf: no internal nodes needed
    com.close()


def test_parked_buffers_are_reused_and_released(ctx, orc, capi):
    """A destroyed resident commitment leaves its device buffers with the context (wf_ctx_release_cached hands them back);
    the next commitment of the same shape must not see anything of the previous one."""
    import torch
    rng = np.random.default_rng(5)
    p = capi.make_params(F64, 1, 12, 3, 8, 1)
    roots = []
    for rep in range(3):
        cols = rand_cols(rng, F64, 8, 1 << 12)
        want = orc.build_trace_commitment(F64, [cols], 1, 12, 3, 7)
        com, _ = ctx.trace_commit_resident(p, cols)
        assert com.root() == want["root"]
        pos = [1, 77, (1 << 15) - 1]
        rows, proof = com.query(pos)
        for i, q in enumerate(pos):
            assert np.array_equal(rows[i], want["lde"][0][q, :8])
        assert proof == orc.merkle_prove_batch(want["nodes"], want["leaves"], pos)
        roots.append(com.root())
        com.close()
    assert len(set(roots)) == 3
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info(0)[0]
    ctx.release_cached()
    free_after = torch.cuda.mem_get_info(0)[0]
    assert free_after >= free_before + (1 << 15) * 64       # at least the parked LDE (2 MiB) came back
    ctx.release_cached()                                      # idempotent


def test_many_small_columns_are_staged(ctx, orc, capi):
    """640 host columns of 64 KiB (40 MiB: more than one 32 MiB staging piece) in, their polynomials back out: the staged
    upload / download route of the host-column entry points (upload_columns / download_columns)."""
    rng = np.random.default_rng(9)
    logR, n_cols, n_traces = 13, 8, 80
    traces = [rand_cols(rng, F64, n_cols, 1 << logR) for _ in range(n_traces)]
    want = orc.build_trace_commitment(F64, traces, 1, logR, 1, 7, threads=16)
    com, polys = ctx.trace_commit_resident(capi.make_params(F64, 1, logR, 1, n_cols, n_traces),
                                           [c for t in traces for c in t], want_polys=True)
    assert com.root() == want["root"]
    for t in (0, 41, 79):
        for c in (0, 7):
            assert np.array_equal(polys[t * n_cols + c], want["polys"][t][c])
    com.close()


def test_read_lde_ranges(ctx, orc, capi):
    """wf_commitment_read_lde: contiguous row ranges of one trace's matrix exactly as RowMatrix stores it (padding lanes
    included), for trace and constraint commitments; out-of-range requests come back as status codes."""
    rng = np.random.default_rng(8)
    logR, logB = 9, 3
    N = 1 << (logR + logB)
    traces = [rand_cols(rng, 1, 5, 1 << logR) for _ in range(2)]
    want = orc.build_trace_commitment(1, traces, 1, logR, logB, 7)
    com, _ = ctx.trace_commit_resident(capi.make_params(1, 1, logR, logB, 5, 2), [c for t in traces for c in t])
    for t, r0, n in ((0, 0, N), (1, 17, 300), (1, N - 1, 1), (0, 5, 0)):
        assert np.array_equal(com.read_lde(t, r0, n), want["lde"][t][r0:r0 + n])
    for t, r0, n, code in ((2, 0, 1, -16), (0, N, 1, -18), (0, N - 1, 2, -18)):
        with pytest.raises(capi.WfError) as e:
            com.read_lde(t, r0, n)
        assert e.value.code == code
    com.close()
    polys = rand_cols(rng, 1, 1, (1 << logR) * 2)                       # one column of the quadratic extension: dense rows
    wantc = orc.build_constraint_commitment(1, polys, 2, logR, logB, 7)
    comc = ctx.constraint_commit_resident(capi.make_params(1, 2, logR, logB, 1, 1), polys)
    got = comc.read_lde(0, 40, 64)
    assert np.array_equal(got[:, :2], wantc["lde"][40:104, :2]) and got.shape[1] in (2, 8)
    comc.close()


def test_query_many_equals_single_queries(ctx, orc, capi):
    """wf_commitment_query_many: the trace tree, the constraint tree and FRI layers of different shapes answered in one round
    trip -- the same rows and BatchMerkleProofs as one wf_commitment_query each; a bad list fails the whole call."""
    rng = np.random.default_rng(77)
    logR, logB = 10, 3
    N = 1 << (logR + logB)
    traces = [rand_cols(rng, F64, 5, 1 << logR) for _ in range(2)]
    tcom, _ = ctx.trace_commit_resident(capi.make_params(F64, 1, logR, logB, 5, 2), [c for t in traces for c in t])
    ccom = ctx.constraint_commit_resident(capi.make_params(F64, 2, logR, logB, 3, 1), rand_cols(rng, F64, 3, 2 << logR))
    fri = capi.FriProver(ctx, F64, 2, 4, 1 << logB, 7, 7)
    fri.begin(rand_cols(rng, F64, 1, 2 * N)[0])
    n_layers = capi.fri_num_layers(4, 1 << logB, 7, N)
    for _ in range(n_layers):
        fri.commit_layer()
        fri.fold(rand_f64(rng, 2))
    pos = np.unique(rng.integers(0, N, size=40)).astype(np.uint64)
    requests = [(tcom, pos, True), (ccom, pos, True)]
    p, size = pos, N
    for i in range(n_layers):
        p = capi.fri_fold_positions(p, size, 4)
        requests.append((fri.layer(i), p, True))
        size //= 4
    requests.append((tcom, pos[:3], False))  # proof only
    got = capi.query_many(requests)
    assert len(got) == len(requests)
    for (com, positions, want_rows), (rows, proof) in zip(requests, got):
        if want_rows:
            want_r, want_p = com.query(positions)
            assert np.array_equal(rows, want_r)
        else:
            assert rows is None
            want_p = com.prove_batch(positions)
        assert proof == want_p
    with pytest.raises(capi.WfError) as e:  # duplicate positions in the third list: DuplicateLeafIndex for the whole call
        capi.query_many(requests[:2] + [(requests[2][0], np.array([1, 1], dtype=np.uint64), True)])
    assert e.value.code == -18
    with pytest.raises(capi.WfError):
        capi.query_many([])
    other = capi.Context(0)
    ocom, _ = other.trace_commit_resident(capi.make_params(F64, 1, 5, 1, 2, 1), rand_cols(rng, F64, 2, 32))
    with pytest.raises(capi.WfError):  # two contexts in one call
        capi.query_many([(tcom, pos[:2], True), (ocom, np.array([1], dtype=np.uint64), True)])
    ocom.close()
    other.close()
    fri.close()
    tcom.close()
    ccom.close()


def test_read_lde_strided(ctx, orc, capi):
    """The constraint evaluation domain's rows of a resident trace LDE (every (lde blowup / ce blowup)-th row) in one read."""
    rng = np.random.default_rng(5)
    logR, logB = 9, 3
    N = 1 << (logR + logB)
    for field, n_cols in ((F64, 5), (F128, 3)):
        traces = [rand_cols(rng, field, n_cols, 1 << logR) for _ in range(2)]
        want = orc.build_trace_commitment(field, traces, 1, logR, logB, 7 if field == F64 else 3)
        com, _ = ctx.trace_commit_resident(capi.make_params(field, 1, logR, logB, n_cols, 2), [c for t in traces for c in t])
        for t in range(2):
            for begin, stride in ((0, 4), (0, 2), (3, 8), (5, 1), (N - 1, 7)):
                n = (N - 1 - begin) // stride + 1
                got = com.read_lde(t, begin, n, stride)
                assert np.array_equal(got, want["lde"][t][begin::stride][:n])
            assert com.read_lde(t, 8, 3, 16).shape[0] == 3
        with pytest.raises(capi.WfError) as e:
            com.read_lde(0, 1, N // 4 + 1, 4)      # the last row would be N + 1
        assert e.value.code == -18
        with pytest.raises(capi.WfError):
            com.read_lde(2, 0, 1, 4)
        com.close()
