"""GPU: the HIP path against the committed golden vectors (tests/golden, made by oracle/gen_golden.py without the
C oracle and without the HIP code)."""
import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu
F64, F128 = 1, 2


@pytest.mark.parametrize("case", G.load("lde_commit_small.json"), ids=lambda c: c["name"])
def test_commit_golden(ctx, capi, case):
    field, fid, ext = case["field"], G.field_id(case["field"]), case["ext"]
    traces = [[G.to_mem(field, col) for col in tr] for tr in case["traces"]]
    n_cols = len(traces[0])
    params = capi.make_params(fid, ext, case["log2_trace_len"], case["log2_blowup"], n_cols, len(traces),
                              int(case["offset"]))
    got = ctx.trace_commit(params, [c for t in traces for c in t])
    rw = 8 * ((n_cols * ext + 7) // 8)
    for t in range(len(traces)):
        for c in range(n_cols):
            assert np.array_equal(got["polys"][t * n_cols + c], G.to_mem(field, case["polys"][t][c]))
        assert np.array_equal(got["lde"][t], G.lde_to_mem(field, case["lde"][t], rw))
    assert G.hexrows(got["leaves"]) == case["leaves"]
    assert G.hexrows(got["nodes"]) == case["nodes"]
    assert got["root"].hex() == case["root"]
    if len(traces) == 1:
        polys = [G.to_mem(field, col) for col in case["polys"][0]]
        p1 = capi.make_params(fid, ext, case["log2_trace_len"], case["log2_blowup"], n_cols, 1, int(case["offset"]))
        assert ctx.constraint_commit(p1, polys)["root"].hex() == case["root"]


def test_blake3_golden_through_hash_rows(ctx, capi):
    """Official-implementation digests: a byte string whose 16-byte words are valid f128 elements is hashed raw by
    hash_elements (blake/mod.rs:47-51), so wf_hash_rows(F128) must reproduce the BLAKE3 KATs."""
    g = G.load("blake3_kat.json")
    n = 0
    for k in g["kat"]:
        if k["len"] == 0 or k["len"] % 16:
            continue
        data = np.frombuffer(bytes(i % 251 for i in range(k["len"])), dtype=np.uint64)
        got = ctx.hash_rows(capi.F128, data, 1, k["len"] // 16)
        assert bytes(got[0]).hex() == k["digest"], k["len"]
        n += 1
    assert n >= 10
    for t in g["trees"]:
        leaves = np.frombuffer(b"".join(bytes.fromhex(x) for x in t["leaves"]), dtype=np.uint8).reshape(-1, 32)
        assert G.hexrows(ctx.merkle_build(leaves)) == t["nodes"]
    m = g["merge"]
    two = np.frombuffer(bytes.fromhex(m["left"]) + bytes.fromhex(m["right"]), dtype=np.uint8).reshape(2, 32)
    assert bytes(ctx.merkle_build(two)[1]).hex() == m["digest"]


def test_reference_literal_inputs_golden(ctx, capi):
    """The reference's own literal test inputs through the HIP path: LEAVES4 / LEAVES8 (crypto/src/merkle/tests.rs:13-65)
    -> wf_merkle_build and resident proofs; the FRI test polynomial (fri/src/prover/tests.rs:58-69) -> wf_fft_evaluate_poly
    and the first layer commitment.  Expected values: tests/golden/reference_inputs.json (see tests/golden/README.md)."""
    g = G.load("reference_inputs.json")
    for t in g["trees"]:
        leaves = np.frombuffer(b"".join(bytes.fromhex(x) for x in t["leaves"]), dtype=np.uint8).reshape(-1, 32)
        nodes = ctx.merkle_build(leaves)
        assert G.hexrows(nodes) == t["nodes"] and bytes(nodes[1]).hex() == t["root"]
    f = g["fri"]
    tl, blowup, folding = f["trace_length"], f["lde_blowup"], f["folding"]
    n = tl * blowup
    p = np.zeros((n, 2), dtype=np.uint64)
    p[:tl, 0] = np.arange(tl, dtype=np.uint64)
    ev = ctx.fft_evaluate_poly(capi.F128, 1, p.reshape(-1))
    want = G.to_mem("f128", f["evaluations"])
    assert np.array_equal(ev.reshape(-1, 2), want)
    layer = ctx.fri_layer_commit(capi.F128, 1, ev, folding)
    assert np.array_equal(layer["transposed"].reshape(-1, 2), G.to_mem("f128", [v for r in f["transposed"] for v in r]))
    assert G.hexrows(layer["leaves"]) == f["leaves"] and layer["root"].hex() == f["root"]
    pr = capi.FriProver(ctx, capi.F128, 1, folding, blowup, 7, 3)
    pr.begin(ev)
    assert pr.commit_layer().hex() == f["root"]
    pr.close()


def test_reference_held_known_answers(ctx, orc):
    """The vectors the reference's own tests hold (tests/golden/reference_kat.json), through the C ABI: the device's extension
    product as P(z) of the polynomial a * x at z = b (wf_evaluate_columns_at: Horner in E multiplies coefficient and point with the
    kernels' one ext_mul, csrc/fri_kernels.hpp, which FRI folding, DEEP and the constraint combination share), and the leaf hash
    of the f128 row 1, 2, 3, 4 (wf_hash_rows: canonical bytes as f128/tests.rs:165-181 lists them, then BLAKE3)."""
    g = G.load("reference_kat.json")
    ints = lambda v: [int(x) for x in v]  # noqa: E731
    for ext, key in ((2, "f64_quad_mul"), (3, "f64_cube_mul")):
        for c in g[key]:
            col = np.zeros(8 * ext, dtype=np.uint64)          # eight coefficients of `ext` coordinates: 0, a, 0, ...
            col[ext:2 * ext] = orc.f64_new(ints(c["a"]))
            got = ctx.evaluate_columns_at(F64, ext, [col], orc.f64_new(ints(c["b"])), ext)
            assert [int(x) for x in orc.f64_as_int(got[0])] == ints(c["expected"]), c["where"]
    c = g["f128_elements_as_bytes"]
    row = orc.f128_from_ints(ints(c["source"]))
    assert ctx.hash_rows(F128, row, 1, 4)[0].tobytes().hex() == c["blake3_256_of_expected_bytes"]
    # transpose_slice's doc test through the FRI layer commitment (the layer's rows are the transposed evaluations)
    c = g["transpose_slice"]
    layer = ctx.fri_layer_commit(F64, 1, np.array(c["values"], dtype=np.uint64), c["N"])
    assert np.asarray(layer["transposed"]).reshape(-1, c["N"]).tolist() == c["expected"], c["where"]

