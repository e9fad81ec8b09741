"""GPU: the HIP path against the committed golden vectors (tests/golden, made by oracle/gen_golden.py without the
C oracle and without the HIP code)."""
import numpy as np
import pytest

import golden_util as G
from conftest import rand_f128

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



def test_reference_held_known_answers_division_split_and_base_field(ctx, orc, capi):
    """The rest of the vectors the reference's own tests hold, through the C ABI (round 4: these were checked on the oracle only).
      * polynom::syn_div (math/src/polynom/tests.rs:178-207) -> wf_deep_compose: with the trace coefficients zero the DEEP
        composition polynomial is exactly syn_div(H, 1, z) of the one composition column H (prover/src/composer/mod.rs:168-186:
        H'(x) = (H(x) - H(z)) / (x - z), the constant dropped with the remainder) -- the reference's dividend, divisor root and
        quotient, over f128 as in its test;
      * the composition polynomial's column split (prover/src/constraints/composition_poly.rs:109-123, values 0 .. 4 n - 1 cut
        into four columns) -> wf_constraint_commit_from_evaluations, at the smallest trace length the reference's TraceInfo
        admits (8; the literal itself uses 4, which no entry point accepts): coefficients 0 .. 31 evaluated over the constraint
        evaluation domain by wf_fft_evaluate_poly_with_offset, interpolated back and cut into columns by the device;
      * the base-field assertions with literal operands (math/src/field/f64/tests.rs:17-86, f128/tests.rs:19-88): products as
        P(z) of a * x at z = b (wf_evaluate_columns_at, extension degree 1), sums and differences as the two outputs
        a + b, a - b of a 2-point transform (wf_fft_evaluate_poly: w_2 = -1)."""
    g = G.load("reference_kat.json")
    ints = lambda v: [int(x) for x in v]  # noqa: E731
    n = 8
    # ---- syn_div through the DEEP composition
    for c in g["f128_syn_div"]:
        poly = ints(c["poly"]) + [0] * (n - len(c["poly"]))
        H = orc.f128_from_ints(poly)
        rng = np.random.default_rng(1)
        trace = rand_f128(rng, n)
        c_trace, _ = ctx.trace_commit_resident(capi.make_params(F128, 1, 3, 1, 1, 1), [trace])
        c_cons = ctx.constraint_commit_resident(capi.make_params(F128, 1, 3, 1, 1, 1), [H])
        got = ctx.deep_compose(F128, 1, n, [c_trace], c_cons, orc.f128_from_ints([int(c["b"])]), orc.f128_from_ints([0]),
                               orc.f128_from_ints([1]))
        q = orc.f128_to_ints(got)
        want = ints(c["expected"])
        assert q[:len(want)] == want and not any(q[len(want):]), c["where"]
        c_trace.close()
        c_cons.close()
    # ---- CompositionPoly::new's split, the literal's pattern at trace length 8
    c = g["f128_composition_segment"]
    trace_len, num_cols = 8, c["num_cols"]
    coeffs = orc.f128_from_ints(list(range(trace_len * num_cols)))
    p = capi.make_params(F128, 1, 3, 1, num_cols, 1)
    evals = ctx.fft_evaluate_poly_with_offset(F128, 1, coeffs, 3, 1)   # over the constraint evaluation domain (offset 3)
    com, cols = ctx.constraint_commit_from_evaluations(p, [evals], want_polys=True)
    assert [orc.f128_to_ints(col) for col in cols] == [list(range(i * trace_len, (i + 1) * trace_len)) for i in range(num_cols)], c["where"]
    assert c["expected"] == [list(range(i * 4, (i + 1) * 4)) for i in range(4)]   # (the same pattern as the literal's)
    com.close()
    # ---- base-field literals
    M64, M128 = 2**64 - 2**32 + 1, 2**128 - 45 * 2**40 + 1
    for field, M, new, val in ((F64, M64, lambda v: orc.f64_new([x % M64 for x in v]),
                                lambda a: [int(x) for x in orc.f64_as_int(np.asarray(a).reshape(-1))]),
                               (F128, M128, lambda v: orc.f128_from_ints([x % M128 for x in v]), lambda a: orc.f128_to_ints(a))):
        def mul(a, b):
            col = new([0, a, 0, 0, 0, 0, 0, 0])
            return val(ctx.evaluate_columns_at(field, 1, [col], new([b]), 1))[0]

        def add_sub(a, b):
            return val(ctx.fft_evaluate_poly(field, 1, new([a, b])))

        t = M - 1
        assert mul(5, 3) == 15 and mul(t, t) == 1 and mul(t, 2) == M - 2 and mul(t, 4) == M - 4      # tests.rs: mul, overflow cases
        assert mul((M + 1) // 2, 2) == 1
        assert mul(0, 12345) == 0 and mul(1, 12345) == 12345                                       # identities
        assert add_sub(2, 3) == [5, M - 1]
        assert add_sub(t, 1) == [0, M - 2] and add_sub(t, 2)[0] == 1                                 # add: overflow
        assert add_sub(5, 3)[1] == 2 and add_sub(3, 5)[1] == M - 2                                   # sub: underflow
