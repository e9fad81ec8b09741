"""GPU: the small coset evaluations the constraint evaluator prepares before its row loop, both
plain calls of fft::evaluate_poly_with_offset and therefore of wf_fft_evaluate_poly_with_offset:
PeriodicValueTable::new (prover/src/constraints/periodic_table.rs:44-66) -- each periodic column's polynomial (length a
divisor of the trace length, as small as 2) over its own coset offset^(trace_length / poly_size), blown up by the
constraint-evaluation factor, then interleaved row by row; LargePolyConstraint::new (constraints/boundary.rs:426-433) --
a boundary polynomial over the whole constraint evaluation domain."""
import numpy as np
import pytest

from conftest import rand_cols

pytestmark = pytest.mark.gpu
F64, F128 = 1, 2


def _mem(orc, field, canonical):
    return orc.lib().orc_f64_new(canonical) if field == F64 else canonical


def _pow(orc, field, base, e):
    L = orc.lib()
    if field == F64:
        return int(L.orc_f64_as_int(L.orc_f64_exp(L.orc_f64_new(base), e)))
    p = 2**128 - 45 * 2**40 + 1
    return pow(base, e, p)


@pytest.mark.parametrize("field,offset", [(F64, 7), (F128, 3)])
@pytest.mark.parametrize("trace_length,ce_blowup,sizes", [(64, 2, [2, 4, 16]), (256, 4, [8, 2, 256, 32]), (32, 8, [32]), (16, 1, [2, 4, 8, 16])])
def test_periodic_value_table(ctx, orc, field, offset, trace_length, ce_blowup, sizes):
    rng = np.random.default_rng(trace_length + ce_blowup + field)
    polys = [rand_cols(rng, field, 1, n)[0] for n in sizes]
    got_cols, want_cols = [], []
    for poly, n in zip(polys, sizes):
        off = _pow(orc, field, offset, trace_length // n)            # periodic_table.rs:48-49
        want_cols.append(orc.evaluate_poly_with_offset(field, poly, n, 1, orc.get_twiddles(field, n), _mem(orc, field, off),
                                                       ce_blowup))
        got_cols.append(ctx.fft_evaluate_poly_with_offset(field, 1, poly, off, ce_blowup))
        assert np.array_equal(got_cols[-1], want_cols[-1])
    # the table itself (periodic_table.rs:57-66): values[i * width + j] = column_j[i % len_j]
    length = max(sizes) * ce_blowup
    table = lambda cols: np.stack([np.stack([c[i % len(c)] for c in cols]) for i in range(length)])  # noqa: E731
    assert np.array_equal(table(got_cols), table(want_cols))


@pytest.mark.parametrize("field,offset,ext", [(F64, 7, 1), (F64, 7, 2), (F128, 3, 1)])
def test_large_boundary_polynomial(ctx, orc, field, offset, ext):
    rng = np.random.default_rng(99 + field + ext)
    n, ce_domain = 64, 512                                           # SMALL_POLY_DEGREE = 63 < 64 <= trace length
    poly = rand_cols(rng, field, 1, n * ext)[0]
    want = orc.evaluate_poly_with_offset(field, poly, n, ext, orc.get_twiddles(field, n), _mem(orc, field, offset),
                                         ce_domain // n)
    assert np.array_equal(ctx.fft_evaluate_poly_with_offset(field, ext, poly, offset, ce_domain // n), want)
