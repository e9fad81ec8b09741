"""GPU: the constraint side from combined constraint evaluations to the resident commitment in one call
(wf_constraint_commit_from_evaluations) against the oracle's chain of the reference's steps: interpolate_poly_with_offset
over the constraint evaluation domain (constraints/evaluation_table.rs:178-185), STARKPack's combination with powers of
final_coeff (prover/src/lib.rs:442-453), CompositionPoly::new / segment (constraints/composition_poly.rs:21-41, 86-98),
build_constraint_commitment (lib.rs:680-715)."""
import numpy as np
import pytest

from conftest import rand_cols, rand_f64, rand_f128

pytestmark = pytest.mark.gpu
F64, F128 = 1, 2


@pytest.mark.parametrize("field,ext", [(F64, 1), (F64, 2), (F64, 3), (F128, 1), (F128, 2)])
@pytest.mark.parametrize("logR,log_ce_blowup,n_cols,n_tables,logB", [
    (3, 1, 1, 1, 1),      # smallest: one column, ce domain = 2 x trace
    (6, 2, 3, 2, 2),      # fewer columns than the ce blowup: the top chunk is dropped
    (10, 3, 8, 3, 3),     # ce domain = LDE domain
    (12, 2, 4, 1, 3),
    (13, 1, 2, 4, 2),
])
def test_constraint_commit_from_evaluations(ctx, orc, capi, field, ext, logR, log_ce_blowup, n_cols, n_tables, logB):
    rng = np.random.default_rng(logR * 7 + ext + field * 100)
    off = 7 if field == F64 else 3
    ce = 1 << (logR + log_ce_blowup)
    tables = [rand_cols(rng, field, 1, ce * ext)[0] for _ in range(n_tables)]
    fc = rand_f64(rng, ext) if field == F64 else rand_f128(rng, ext)
    want_cols = orc.composition_poly_from_evaluations(field, ext, tables, logR, n_cols, off, fc)
    want = orc.build_constraint_commitment(field, want_cols, ext, logR, logB, off)
    p = capi.make_params(field, ext, logR, logB, n_cols, 1)
    com, polys = ctx.constraint_commit_from_evaluations(p, tables, fc if n_tables > 1 else None, want_polys=True)
    assert com.root() == want["root"]
    for c in range(n_cols):
        assert np.array_equal(polys[c], want_cols[c]), f"column {c}"
    # the handle answers like a constraint commitment built from the columns
    N = 1 << (logR + logB)
    pos = np.unique(rng.integers(0, N, size=min(12, N)))
    rows, proof = com.query(pos)
    assert proof == orc.merkle_prove_batch(want["nodes"], want["leaves"], [int(q) for q in pos])
    assert np.array_equal(rows.reshape(len(pos), -1), want["lde"].reshape(N, -1)[pos][:, :rows.reshape(len(pos), -1).shape[1]])
    z = rand_f64(rng, ext) if field == F64 else rand_f128(rng, ext)
    ood = com.evaluate_polys_at(z, ext, n_cols)
    for c in range(n_cols):
        assert np.array_equal(ood[c], orc.eval_column_at(field, want_cols[c], ext, z, ext))
    com.close()


def test_constraint_commit_from_evaluations_errors(ctx, capi):
    rng = np.random.default_rng(5)
    p = capi.make_params(F64, 2, 6, 2, 4, 1)
    tabs = [rand_cols(rng, F64, 1, 256 * 2)[0] for _ in range(2)]
    with pytest.raises(capi.WfError):   # two tables need final_coeff
        ctx.constraint_commit_from_evaluations(p, tabs)
    with pytest.raises(capi.WfError):   # "trace length must be smaller than size of composition polynomial"
        ctx.constraint_commit_from_evaluations(p, [rand_cols(rng, F64, 1, 64 * 2)[0]])
    with pytest.raises(capi.WfError):   # 4 columns of 64 do not fit 128 coefficients
        ctx.constraint_commit_from_evaluations(p, [rand_cols(rng, F64, 1, 128 * 2)[0]])
    with pytest.raises(capi.WfError):   # not a field element
        ctx.constraint_commit_from_evaluations(p, tabs, np.array([2**64 - 1, 0], dtype=np.uint64))
    com, _ = ctx.constraint_commit_from_evaluations(p, tabs, rand_f64(rng, 2))   # the context is fine afterwards
    com.close()


def _divisors(orc, field, logR, n_ex, rng):
    """A transition divisor (x^n - 1) / prod (x - g^(n-i)) (ConstraintDivisor::from_transition, air/src/air/divisor.rs:52-66),
    an assertion at one step (x - g^step) (from_assertion :68-96, a = 1: as many inverses as the domain has points) and a
    periodic assertion (x^(n/stride) - g^(offset n / stride))."""
    n = 1 << logR
    L = orc.lib()
    if field == F64:
        g = L.orc_f64_get_root_of_unity(logR)
        mem = lambda v: np.array([v], dtype=np.uint64)                 # noqa: E731
        gpow = lambda e: mem(L.orc_f64_exp(g, e))                      # noqa: E731
        one = mem(L.orc_f64_new(1))
    else:
        gi = orc.f128_root_of_unity(logR)
        P = 2**128 - 45 * 2**40 + 1
        gpow = lambda e: orc.f128_from_ints([pow(gi, e, P)])           # noqa: E731
        one = orc.f128_from_ints([1])
    exemptions = np.concatenate([gpow(n - 1 - i) for i in range(n_ex)]) if n_ex else None
    return [(n, one, exemptions), (1, gpow(int(rng.integers(0, n))), None), (n // 4, gpow((3 * n) // 4 % n), None)]


@pytest.mark.parametrize("field,ext", [(F64, 1), (F64, 2), (F64, 3), (F128, 1), (F128, 2)])
@pytest.mark.parametrize("logR,log_ce_blowup,n_cols,n_tables,n_ex", [(4, 1, 2, 1, 1), (7, 2, 3, 2, 2), (11, 3, 4, 3, 1), (12, 1, 2, 1, 3)])
def test_constraint_commit_from_tables(ctx, orc, capi, field, ext, logR, log_ce_blowup, n_cols, n_tables, n_ex):
    """All of ConstraintEvaluationTable::into_comb_poly on the device: acc_column + get_inv_evaluation
    (constraints/evaluation_table.rs:335-426) per column, then the chain above."""
    rng = np.random.default_rng(logR * 11 + ext + field * 50)
    off = 7 if field == F64 else 3
    ce = 1 << (logR + log_ce_blowup)
    tables, combined = [], []
    for _ in range(n_tables):
        divs = _divisors(orc, field, logR, n_ex, rng)
        cols = rand_cols(rng, field, len(divs), ce * ext)
        tables.append(list(zip(cols, divs)))
        combined.append(orc.combine_evaluation_table(field, ext, cols, divs, off))
    fc = rand_f64(rng, ext) if field == F64 else rand_f128(rng, ext)
    want_cols = orc.composition_poly_from_evaluations(field, ext, combined, logR, n_cols, off, fc)
    want = orc.build_constraint_commitment(field, want_cols, ext, logR, 3, off)
    com, polys = ctx.constraint_commit_from_tables(capi.make_params(field, ext, logR, 3, n_cols, 1), tables,
                                                   fc if n_tables > 1 else None, want_polys=True)
    for c in range(n_cols):
        assert np.array_equal(polys[c], want_cols[c]), f"column {c}"
    assert com.root() == want["root"]
    com.close()


def test_constraint_commit_from_tables_errors(ctx, orc, capi):
    rng = np.random.default_rng(9)
    p = capi.make_params(F64, 2, 6, 2, 2, 1)
    col = rand_cols(rng, F64, 1, 256 * 2)[0]
    one = np.array([orc.lib().orc_f64_new(1)], dtype=np.uint64)
    for bad in [(3, one, None),                                            # numerator degree not a power of two
                (512, one, None),                                          # larger than the domain
                (64, np.array([2**64 - 1], dtype=np.uint64), None),        # not a field element
                (64, one, np.full(9, 5, dtype=np.uint64)),                 # more exemption points than supported
                (64, one, np.array([2**64 - 2], dtype=np.uint64))]:        # invalid exemption point
        with pytest.raises(capi.WfError):
            ctx.constraint_commit_from_tables(p, [[(col, bad)]])
    com, _ = ctx.constraint_commit_from_tables(p, [[(col, (64, one, None))]])
    com.close()
