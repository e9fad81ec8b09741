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
