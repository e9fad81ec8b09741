"""GPU: out-of-domain evaluation of polynomial columns against the oracle's Horner restatement
(ColMatrix::evaluate_columns_at, prover/src/matrix/col_matrix.rs:249-254; get_ood_frame, trace/poly_table.rs:60-73)."""
import numpy as np
import pytest

from conftest import rand_cols, rand_f64, rand_f128

pytestmark = pytest.mark.gpu
F64, F128 = 1, 2


@pytest.mark.parametrize("field,ext_c,ext_z", [(F64, 1, 1), (F64, 1, 2), (F64, 1, 3), (F64, 2, 2), (F64, 3, 3),
                                               (F128, 1, 1), (F128, 1, 2), (F128, 2, 2)])
@pytest.mark.parametrize("logn", [3, 7, 8, 12, 13, 17])   # below / at / above one 4096-coefficient block, many blocks
def test_evaluate_columns_at(ctx, orc, field, ext_c, ext_z, logn):
    rng = np.random.default_rng(logn * 10 + ext_z)
    n = 1 << logn
    cols = rand_cols(rng, field, 5, n * ext_c)
    z = rand_f64(rng, ext_z) if field == F64 else rand_f128(rng, ext_z)
    got = ctx.evaluate_columns_at(field, ext_c, cols, z, ext_z)
    for i, c in enumerate(cols):
        assert np.array_equal(got[i], orc.eval_column_at(field, c, ext_c, z, ext_z)), f"column {i}"


def test_ood_frame_of_resident_commitment(ctx, orc, capi):
    """get_ood_frame: trace polynomials (base field) at z and z*g in the quadratic extension, straight from HBM."""
    rng = np.random.default_rng(1)
    logR, n_cols, n_traces = 12, 6, 2
    traces = [rand_cols(rng, F64, n_cols, 1 << logR) for _ in range(n_traces)]
    want = orc.build_trace_commitment(F64, traces, 1, logR, 3, 7)
    com, _ = ctx.trace_commit_resident(capi.make_params(F64, 1, logR, 3, n_cols, n_traces), [c for t in traces for c in t])
    z = rand_f64(rng, 2)
    L = orc.lib()
    g = L.orc_f64_get_root_of_unity(logR)
    zg = np.array([L.orc_f64_mul(int(z[0]), g), L.orc_f64_mul(int(z[1]), g)], dtype=np.uint64)  # z * E::from(g)
    for point in (z, zg):
        got = com.evaluate_polys_at(point, 2, n_cols * n_traces)
        for t in range(n_traces):
            for c in range(n_cols):
                assert np.array_equal(got[t * n_cols + c], orc.eval_column_at(F64, want["polys"][t][c], 1, point, 2))
    # the frame in one call
    frame = com.evaluate_polys_at_points(np.concatenate([z, zg]), 2, 2, n_cols * n_traces)
    for q, point in enumerate((z, zg)):
        assert np.array_equal(frame[q], com.evaluate_polys_at(point, 2, n_cols * n_traces))
    four = com.evaluate_polys_at_points(np.concatenate([zg, z, z, zg]), 4, 2, n_cols * n_traces)
    assert np.array_equal(four[0], frame[1]) and np.array_equal(four[1], frame[0]) and np.array_equal(four[3], frame[1])
    with pytest.raises(capi.WfError):
        com.evaluate_polys_at_points(np.concatenate([z] * 5), 5, 2, n_cols * n_traces)
    with pytest.raises(capi.WfError):
        com.evaluate_polys_at(np.array([2**64 - 1, 0], dtype=np.uint64), 2, n_cols * n_traces)  # invalid element
    com.close()
