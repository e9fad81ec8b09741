"""GPU: DEEP composition polynomial built from resident commitments (wf_deep_compose) against the oracle's literal
restatement of DeepCompositionPoly::add_trace_polys / add_composition_poly (prover/src/composer/mod.rs:62-193,
acc_trace_poly + syn_div_in_place + merge_trace_compositions), bit for bit."""
import numpy as np
import pytest

from conftest import rand_cols, rand_f64, rand_f128

pytestmark = pytest.mark.gpu
F64, F128 = 1, 2


def rand_e(rng, field, ext, n=1):
    return rand_f64(rng, n * ext) if field == F64 else rand_f128(rng, n * ext)


def setup(ctx, orc, capi, rng, field, ext, logn, n_main, n_aux, n_traces, n_cons, logb=1):
    """Resident commitments of n_traces packed traces (main segment + optional auxiliary segment over E) and of the
    constraint composition columns; returns them with the oracle-side tables and the coefficient permutation."""
    n = 1 << logn
    off = 7 if field == F64 else 3
    main = [rand_cols(rng, field, n_main, n) for _ in range(n_traces)]
    want_main = orc.build_trace_commitment(field, main, 1, logn, logb, off)
    c_main, _ = ctx.trace_commit_resident(capi.make_params(field, 1, logn, logb, n_main, n_traces), [c for t in main for c in t])
    handles = [c_main]
    aux_polys = None
    if n_aux:
        aux = [rand_cols(rng, field, n_aux, n * ext) for _ in range(n_traces)]
        aux_polys = orc.build_trace_commitment(field, aux, ext, logn, logb, off)["polys"]
        c_aux, _ = ctx.trace_commit_resident(capi.make_params(field, ext, logn, logb, n_aux, n_traces), [c for t in aux for c in t])
        handles.append(c_aux)
    cons = rand_cols(rng, field, n_cons, n * ext) if n_cons else []
    c_cons = ctx.constraint_commit_resident(capi.make_params(field, ext, logn, logb, n_cons, 1), cons) if n_cons else None
    # the reference's TracePolyTable t: main columns of trace t, then its auxiliary columns (composer/mod.rs:84-128)
    tables = []
    for t in range(n_traces):
        tab = [(want_main["polys"][t][c], 1) for c in range(n_main)]
        if n_aux:
            tab += [(aux_polys[t][c], ext) for c in range(n_aux)]
        tables.append(tab)
    # coefficients: the C ABI takes them handle by handle (main segments of all traces, then auxiliary ones)
    w = 1 if field == F64 else 2
    cc_tables = [[rand_e(rng, field, ext) for _ in tab] for tab in tables]
    abi = [cc_tables[t][c] for t in range(n_traces) for c in range(n_main)]
    abi += [cc_tables[t][n_main + c] for t in range(n_traces) for c in range(n_aux)]
    cc_cons = [rand_e(rng, field, ext) for _ in range(n_cons)]
    flat = lambda xs: np.concatenate([np.asarray(x).reshape(-1, w) for x in xs]).reshape((-1, w) if w > 1 else -1)
    return dict(n=n, handles=handles, c_cons=c_cons, tables=tables, cons=cons,
                cc_oracle=[c for tab in cc_tables for c in tab], cc_abi=flat(abi),
                cc_cons=cc_cons, cc_cons_abi=flat(cc_cons) if n_cons else None)


def close(s):
    for h in s["handles"]:
        h.close()
    if s["c_cons"] is not None:
        s["c_cons"].close()


@pytest.mark.parametrize("field,ext", [(F64, 1), (F64, 2), (F64, 3), (F128, 1), (F128, 2)])
@pytest.mark.parametrize("logn,n_main,n_aux,n_traces,n_cons", [
    (3, 2, 0, 1, 1),      # smallest trace the reference admits
    (5, 3, 2, 2, 2),
    (10, 4, 1, 1, 3),     # exactly one scan block
    (11, 2, 2, 3, 2),     # two blocks: a carry crosses
    (13, 3, 0, 1, 4),
    (4, 10, 3, 40, 8),    # many short columns: the linear combination runs in column chunks
    (16, 2, 1, 1, 1),     # 64 blocks
])
def test_deep_compose(ctx, orc, capi, field, ext, logn, n_main, n_aux, n_traces, n_cons):
    if ext == 1:
        n_aux = 0  # no extension: the auxiliary segment would be another base-field segment; covered by n_main
    rng = np.random.default_rng(logn * 100 + ext * 10 + field)
    s = setup(ctx, orc, capi, rng, field, ext, logn, n_main, n_aux, n_traces, n_cons)
    z = rand_e(rng, field, ext)
    want = orc.deep_compose(field, ext, s["n"], s["tables"], s["cons"], z, s["cc_oracle"], s["cc_cons"])
    got = ctx.deep_compose(field, ext, s["n"], s["handles"], s["c_cons"], z, s["cc_abi"], s["cc_cons_abi"])
    assert np.array_equal(got, want)
    w = 1 if field == F64 else 2
    top = got.reshape(s["n"], ext, w)[-1]
    assert not top.any()  # degree n - 2 (composer/mod.rs:151)
    close(s)


def test_deep_compose_without_constraints_and_large(ctx, orc, capi):
    """2^18 coefficients, 8 main columns, quadratic extension: 256 blocks (the carry kernel's threads own one block each)."""
    rng = np.random.default_rng(7)
    s = setup(ctx, orc, capi, rng, F64, 2, 18, 8, 0, 1, 0)
    z = rand_e(rng, F64, 2)
    want = orc.deep_compose(F64, 2, s["n"], s["tables"], [], z, s["cc_oracle"], [])
    got = ctx.deep_compose(F64, 2, s["n"], s["handles"], None, z, s["cc_abi"])
    assert np.array_equal(got, want)
    close(s)


def test_deep_compose_more_blocks_than_threads(ctx, orc, capi):
    """2^20 coefficients: 1024 scan blocks, four per thread of the carry kernel."""
    rng = np.random.default_rng(8)
    s = setup(ctx, orc, capi, rng, F64, 2, 20, 2, 0, 1, 1)
    z = rand_e(rng, F64, 2)
    want = orc.deep_compose(F64, 2, s["n"], s["tables"], s["cons"], z, s["cc_oracle"], s["cc_cons"])
    got = ctx.deep_compose(F64, 2, s["n"], s["handles"], s["c_cons"], z, s["cc_abi"], s["cc_cons_abi"])
    assert np.array_equal(got, want)
    close(s)


@pytest.mark.parametrize("field,ext", [(F64, 2), (F128, 1)])
def test_deep_compose_into_fri(ctx, orc, capi, field, ext):
    """The polynomial handed to the FRI prover in HBM: the first layer's root equals the one a prover started from the
    oracle's coefficients (wf_fri_prover_begin_poly) commits to; DeepCompositionPoly::evaluate, composer/mod.rs:198-205."""
    rng = np.random.default_rng(11)
    logn, blowup = 10, 8
    s = setup(ctx, orc, capi, rng, field, ext, logn, 3, 2 if ext > 1 else 0, 2, 2, logb=3)
    z = rand_e(rng, field, ext)
    want = orc.deep_compose(field, ext, s["n"], s["tables"], s["cons"], z, s["cc_oracle"], s["cc_cons"])
    off = 7 if field == F64 else 3
    a = capi.FriProver(ctx, field, ext, 4, blowup, 7, off)
    b = capi.FriProver(ctx, field, ext, 4, blowup, 7, off)
    got = ctx.deep_compose(field, ext, s["n"], s["handles"], s["c_cons"], z, s["cc_abi"], s["cc_cons_abi"], fri=a, lde_blowup=blowup)
    assert np.array_equal(got, want)
    b.begin_poly(want, blowup)
    assert a.commit_layer() == b.commit_layer()
    # and without the copy to the host
    a.reset()
    assert ctx.deep_compose(field, ext, s["n"], s["handles"], s["c_cons"], z, s["cc_abi"], s["cc_cons_abi"], want_poly=False,
                            fri=a, lde_blowup=blowup) is None
    b.reset()
    b.begin_poly(want, blowup)
    assert a.commit_layer() == b.commit_layer()
    a.close()
    b.close()
    close(s)


def test_deep_compose_errors(ctx, orc, capi):
    rng = np.random.default_rng(3)
    s = setup(ctx, orc, capi, rng, F64, 2, 6, 2, 1, 1, 1)
    z = rand_e(rng, F64, 2)
    args = (F64, 2, s["n"], s["handles"], s["c_cons"])
    with pytest.raises(capi.WfError):  # syn_div_in_place: "constant cannot be zero"
        ctx.deep_compose(*args, np.zeros(2, dtype=np.uint64), s["cc_abi"], s["cc_cons_abi"])
    with pytest.raises(capi.WfError):  # not a field element
        ctx.deep_compose(*args, np.array([2**64 - 1, 1], dtype=np.uint64), s["cc_abi"], s["cc_cons_abi"])
    bad = s["cc_abi"].copy()
    bad[0] = np.uint64(2**64 - 1)
    with pytest.raises(capi.WfError):
        ctx.deep_compose(*args, z, bad, s["cc_cons_abi"])
    with pytest.raises(capi.WfError):  # neither an output nor a FRI prover
        ctx.deep_compose(*args, z, s["cc_abi"], s["cc_cons_abi"], want_poly=False)
    with pytest.raises(capi.WfError):  # composition over the cubic extension, auxiliary columns over the quadratic one
        ctx.deep_compose(F64, 3, s["n"], s["handles"], None, rand_e(rng, F64, 3), rand_e(rng, F64, 3, 3))
    other, _ = ctx.trace_commit_resident(capi.make_params(F64, 1, 7, 1, 2, 1), rand_cols(rng, F64, 2, 128))
    with pytest.raises(capi.WfError):  # polynomials of another length
        ctx.deep_compose(F64, 2, s["n"], [s["handles"][0], other], None, z, rand_e(rng, F64, 2, 4))
    with pytest.raises(capi.WfError):  # constraint coefficients missing
        ctx.deep_compose(*args, z, s["cc_abi"], None)
    # a FRI layer holds no polynomials
    fri = capi.FriProver(ctx, F64, 2, 4, 8, 7, 7)
    ctx.deep_compose(*args, z, s["cc_abi"], s["cc_cons_abi"], want_poly=False, fri=fri, lde_blowup=8)
    fri.commit_layer()
    fri.fold(rand_e(rng, F64, 2))
    with pytest.raises(capi.WfError):
        ctx.deep_compose(F64, 2, s["n"], [fri.layer(0)], None, z, rand_e(rng, F64, 2, 4))
    with pytest.raises(capi.WfError):  # "a prior proof generation request has not been completed yet"
        ctx.deep_compose(*args, z, s["cc_abi"], s["cc_cons_abi"], want_poly=False, fri=fri, lde_blowup=8)
    fri.close()
    other.close()
    close(s)
