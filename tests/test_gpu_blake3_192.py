"""GPU: the Blake3_192 hasher (crypto/src/hash/blake/mod.rs:68-114) through the C ABI: wf_params::digest_bytes = 24 for the
commitments and their queries, wf_ctx_set_digest_bytes(24) for wf_hash_rows / wf_merkle_build / the FRI entry points.
Against tests/golden/blake3_192.json (LLVM's BLAKE3 + Python integers, made without the oracle and without the HIP code)
and against the oracle in its 24-byte mode on larger shapes (every kernel that writes leaves or merges nodes)."""
import numpy as np
import pytest

import golden_util as G
from conftest import rand_cols

pytestmark = pytest.mark.gpu
F64, F128 = 1, 2


def hex24(a):
    return [bytes(x).hex() for x in np.asarray(a, dtype=np.uint8).reshape(-1, 24)]


def test_golden_blake3_192(capi):
    g = G.load("blake3_192.json")
    ctx = capi.Context(0)
    ctx.set_digest_bytes(24)
    for k in g["kat"]:  # a byte string of 16-byte words that are valid f128 elements is hashed raw
        data = np.frombuffer(bytes(i % 251 for i in range(k["len"])), dtype=np.uint64)
        assert bytes(ctx.hash_rows(capi.F128, data, 1, k["len"] // 16)[0]).hex() == k["digest"], k["len"]
    for t in g["trees"]:
        leaves = np.frombuffer(b"".join(bytes.fromhex(x) for x in t["leaves"]), dtype=np.uint8).reshape(-1, 24)
        assert hex24(ctx.merkle_build(leaves)) == t["nodes"]
    m = g["merge"]
    two = np.frombuffer(bytes.fromhex(m["left"]) + bytes.fromhex(m["right"]), dtype=np.uint8).reshape(2, 24)
    assert bytes(ctx.merkle_build(two)[1]).hex() == m["digest"]
    for case in g["commits"]:
        field, fid, ext = case["field"], G.field_id(case["field"]), case["ext"]
        traces = [[G.to_mem(field, col) for col in tr] for tr in case["traces"]]
        n_cols = len(traces[0])
        params = capi.make_params(fid, ext, case["log2_trace_len"], case["log2_blowup"], n_cols, len(traces),
                                  int(case["offset"]), digest_bytes=24)
        got = ctx.trace_commit(params, [c for t in traces for c in t])
        assert hex24(got["leaves"]) == case["leaves"], case["name"]
        assert hex24(got["nodes"]) == case["nodes"] and got["root"].hex() == case["root"]
    ctx.set_digest_bytes(32)   # and back: the 256-bit hasher is untouched
    g256 = G.load("blake3_kat.json")
    t = g256["trees"][0]
    leaves = np.frombuffer(b"".join(bytes.fromhex(x) for x in t["leaves"]), dtype=np.uint8).reshape(-1, 32)
    assert G.hexrows(ctx.merkle_build(leaves)) == t["nodes"]
    ctx.close()


@pytest.mark.parametrize("field,logR,logB,n_cols,n_traces", [
    (F64, 14, 3, 8, 1),     # two passes, one segment: leaves from the persistent last pass, every tree kernel (2^17 leaves)
    (F64, 18, 3, 8, 1),     # 2^21 leaves: the two-level launches + the LDS subtree levels incl. the four-lane form
    (F64, 12, 3, 20, 1),    # three segments: chaining values across segments
    (F64, 10, 2, 3, 5),     # single pass, several traces: k_hash_rows
    (F64, 11, 1, 200, 1),   # rows of two BLAKE3 chunks: chunk kernels + merge
    (F128, 12, 3, 10, 2),   # f128, packed traces
    (F64, 9, 3, 1, 1),      # coset-packed lanes: leaves hashed by k_hash_rows / in the packed pass
    (F128, 10, 3, 1, 1),
])
def test_commitments_and_queries_blake3_192(orc, capi, field, logR, logB, n_cols, n_traces):
    ctx = capi.Context(0)
    rng = np.random.default_rng(192 + logR + n_cols)
    R, N = 1 << logR, 1 << (logR + logB)
    traces = [rand_cols(rng, field, n_cols, R) for _ in range(n_traces)]
    with orc.digest_size(24):
        want = orc.build_trace_commitment(field, traces, 1, logR, logB, 7 if field == F64 else 3, threads=8)
        params = capi.make_params(field, 1, logR, logB, n_cols, n_traces, digest_bytes=24)
        cols = [c for t in traces for c in t]
        got = ctx.trace_commit(params, cols, want_lde=False, want_polys=False)
        assert got["leaves"].shape == (N, 24) and np.array_equal(got["leaves"], want["leaves"])
        assert np.array_equal(got["nodes"], want["nodes"]) and got["root"] == want["root"]
        com, _ = ctx.trace_commit_resident(params, cols)
        assert com.root() == want["root"]
        pos = sorted({0, 1, N - 1, N // 2, 5 % N, (N // 3) | 1})
        rows, proof = com.query(pos)
        assert proof == orc.merkle_prove_batch(want["nodes"], want["leaves"], pos)
        assert com.prove(pos[-1]) == orc.merkle_prove(want["nodes"], want["leaves"], pos[-1])
        com.close()
        if n_traces == 1:
            cw = orc.build_constraint_commitment(field, want["polys"][0], 1, logR, logB, 7 if field == F64 else 3, threads=8)
            cg = ctx.constraint_commit(params, want["polys"][0], want_lde=False)
            assert cg["root"] == cw["root"] and np.array_equal(cg["nodes"], cw["nodes"])
    ctx.close()


def test_fri_prover_blake3_192(orc, capi):
    """The resident FRI prover with the context's hasher set to Blake3_192: every layer root, the remainder commitment and a
    layer's batch proof (24-byte entries) against the oracle's FriProver restatement in its 24-byte mode."""
    ctx = capi.Context(0)
    ctx.set_digest_bytes(24)
    rng = np.random.default_rng(7)
    n, folding, blowup, max_rem, offset = 1 << 12, 4, 8, 7, 7
    ev = rand_cols(rng, F64, 1, n)[0]
    L = orc.lib()
    pr = capi.FriProver(ctx, F64, 1, folding, blowup, max_rem, offset)
    pr.begin(ev)
    n_layers = capi.fri_num_layers(folding, blowup, max_rem, n)
    with orc.digest_size(24):
        size, cur, layers = n, ev, []
        for i in range(n_layers):
            want = orc.fri_layer_commit(F64, cur, size, 1, folding)
            root = pr.commit_layer()
            assert len(root) == 24 and root == want["root"], f"layer {i}"
            alpha = np.array([L.orc_f64_new(1234567 + i)], dtype=np.uint64)
            cur = orc.apply_drp(F64, want["transposed"], size // folding, 1, folding, offset, alpha)
            pr.fold(alpha)
            layers.append(want)
            size //= folding
        rem, digest = pr.set_remainder(size)
        want_rem = cur.copy()
        orc.interpolate_poly_with_offset(F64, want_rem, size, 1, orc.get_twiddles(F64, size, inverse=True), L.orc_f64_new(offset))
        assert digest == orc.hash_elements(F64, want_rem.reshape(-1)[:size // blowup])
        lay0 = pr.layer(0)
        pos = [0, 3, n // folding - 1]
        rows, proof = lay0.query(pos)
        assert proof == orc.merkle_prove_batch(layers[0]["nodes"], layers[0]["leaves"], pos)
        assert lay0.digest_bytes == 24 and lay0.root() == layers[0]["root"]
    pr.close()
    ctx.close()
