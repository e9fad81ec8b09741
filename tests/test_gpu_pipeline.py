"""GPU: the path and its "next" rows chained the way Prover::generate_proof chains them (prover/src/lib.rs:240-610),
all handles alive at once on one context: packed trace commitment -> OOD frame -> constraint commitment -> DEEP
evaluations -> FRI commit phase -> queries of all three kinds at the same positions.  What the reference computes with
user code (AIR constraint evaluation, DEEP composition) is replaced by seeded random polynomials of the right shape;
the Fiat-Shamir channel by a seeded generator.  Every device result is compared with the oracle."""
import numpy as np
import pytest

from conftest import rand_cols

pytestmark = pytest.mark.gpu
F64 = 1


def test_prove_shaped_pipeline(ctx, orc, capi):
    L = orc.lib()
    rng = np.random.default_rng(20241003)
    logR, logB, n_cols, n_traces, ext = 10, 3, 6, 2, 2
    R, N, offset = 1 << logR, 1 << (logR + logB), 7
    folding, max_rem = 4, 7

    # 1. build_trace_commitment (lib.rs:267-268), STARKPack: two traces under one tree
    traces = [rand_cols(rng, F64, n_cols, R) for _ in range(n_traces)]
    want_t = orc.build_trace_commitment(F64, traces, 1, logR, logB, offset)
    tcom, polys = ctx.trace_commit_resident(capi.make_params(F64, 1, logR, logB, n_cols, n_traces),
                                            [c for t in traces for c in t], want_polys=True)
    assert tcom.root() == want_t["root"]

    # 2. out-of-domain frame (lib.rs:485-489): every trace polynomial at z and z*g, z in the quadratic extension
    z = rand_cols(rng, F64, 1, ext)[0]
    g = np.array([L.orc_f64_get_root_of_unity(logR), 0], dtype=np.uint64)
    zg = orc.ext_mul(F64, ext, z, g)
    for point in (z, zg):
        got = tcom.evaluate_polys_at(point, ext, n_cols * n_traces)
        for t in range(n_traces):
            for c in range(n_cols):
                assert np.array_equal(got[t * n_cols + c], orc.eval_column_at(F64, want_t["polys"][t][c], 1, point, ext))

    # 3. build_constraint_commitment (lib.rs:333-334): composition polynomial columns over E
    comp = rand_cols(rng, F64, 2, R * ext)
    want_c = orc.build_constraint_commitment(F64, comp, ext, logR, logB, offset)
    ccom = ctx.constraint_commit_resident(capi.make_params(F64, ext, logR, logB, 2, 1), comp)
    assert ccom.root() == want_c["root"]

    # 4. DEEP composition polynomial evaluated over the LDE domain (composer/mod.rs:198-205), then the FRI commit phase
    deep = rand_cols(rng, F64, 1, R * ext)[0]
    deep_evals = ctx.fft_evaluate_poly_with_offset(F64, ext, deep, offset, 1 << logB)
    want_e = orc.evaluate_poly_with_offset(F64, deep, R, ext, orc.get_twiddles(F64, R), L.orc_f64_new(offset), 1 << logB)
    assert np.array_equal(deep_evals, want_e)
    fri = capi.FriProver(ctx, F64, ext, folding, 1 << logB, max_rem, offset)
    fri.begin_poly(deep, 1 << logB)                      # the evaluations never leave the device
    cur, size, want_layers = deep_evals, N, []
    for i in range(capi.fri_num_layers(folding, 1 << logB, max_rem, N)):
        want = orc.fri_layer_commit(F64, cur, size, ext, folding)
        assert fri.commit_layer() == want["root"]
        alpha = rand_cols(rng, F64, 1, ext)[0]           # channel.draw_fri_alpha()
        cur = orc.apply_drp(F64, want["transposed"], size // folding, ext, folding, offset, alpha)
        fri.fold(alpha)
        want_layers.append(want)
        size //= folding
    rem, rem_digest = fri.set_remainder(size)
    want_rem = cur.copy()
    orc.interpolate_poly_with_offset(F64, want_rem, size, ext, orc.get_twiddles(F64, size, inverse=True), L.orc_f64_new(offset))
    keep = (size >> logB) * ext
    assert np.array_equal(rem.reshape(-1), want_rem[:keep]) and rem_digest == orc.hash_elements(F64, want_rem[:keep])

    # 5. queries (lib.rs:560-600): the same positions against the trace tree, the constraint tree and every FRI layer
    positions = [int(p) for p in rng.choice(N, size=27, replace=False)]
    rows = tcom.read_rows(positions)
    for i, p in enumerate(positions):
        comb = np.concatenate([want_t["lde"][t][p, :n_cols] for t in range(n_traces)])
        assert np.array_equal(rows[i], comb) and orc.hash_elements(F64, rows[i]) == bytes(want_t["leaves"][p])
    assert tcom.prove_batch(positions) == orc.merkle_prove_batch(want_t["nodes"], want_t["leaves"], positions)
    crow = ccom.read_rows(positions)
    for i, p in enumerate(positions):
        assert np.array_equal(crow[i], want_c["lde"][p, :2 * ext])
    assert ccom.prove_batch(positions) == orc.merkle_prove_batch(want_c["nodes"], want_c["leaves"], positions)
    folded, domain = np.array(positions, dtype=np.uint64), N
    for i, want in enumerate(want_layers):
        folded = capi.fri_fold_positions(folded, domain, folding)
        layer = fri.layer(i)
        tr = want["transposed"].reshape(domain // folding, folding * ext)
        assert np.array_equal(layer.read_rows(folded), tr[folded.astype(np.int64)])
        assert layer.prove_batch(folded) == orc.merkle_prove_batch(want["nodes"], want["leaves"], [int(p) for p in folded])
        domain //= folding
    fri.close()
    ccom.close()
    tcom.close()


def test_prove_shaped_pipeline_resident_chain(ctx, orc, capi):
    """The same chain with this round's entry points, nothing but the (stand-in) constraint evaluator on the host: packed
    trace commitment -> strided read of the constraint evaluation domain's rows -> constraint commitment from the
    evaluation TABLES of both packed traces (divisors, interpolation, final_coeff combination) -> out-of-domain frames in
    one call each -> DEEP composition straight into the FRI prover -> commit phase -> every tree queried in one round trip."""
    L = orc.lib()
    rng = np.random.default_rng(20261004)
    logR, logB, log_ce, n_cols, n_traces, ext, n_comp = 9, 3, 1, 5, 2, 2, 2
    R, N, ce, offset = 1 << logR, 1 << (logR + logB), 1 << (logR + log_ce), 7
    folding, max_rem = 4, 7
    re = lambda k=1: rand_cols(rng, F64, 1, ext * k)[0]                         # noqa: E731

    traces = [rand_cols(rng, F64, n_cols, R) for _ in range(n_traces)]
    want_t = orc.build_trace_commitment(F64, traces, 1, logR, logB, offset)
    tcom, _ = ctx.trace_commit_resident(capi.make_params(F64, 1, logR, logB, n_cols, n_traces), [c for t in traces for c in t])
    assert tcom.root() == want_t["root"]

    # the evaluator's input: rows of the constraint evaluation domain = every (lde / ce)-th row of the extended trace
    stride = N // ce
    for t in range(n_traces):
        assert np.array_equal(tcom.read_lde(t, 0, ce, stride), want_t["lde"][t][::stride])

    # the evaluator's output (stand-in): per packed trace a transition column and a boundary column over the ce domain
    g_trace = L.orc_f64_get_root_of_unity(logR)
    one = np.array([L.orc_f64_new(1)], dtype=np.uint64)
    divisors = [(R, one, np.array([L.orc_f64_exp(g_trace, R - 1)], dtype=np.uint64)),       # (x^n - 1) / (x - g^(n-1))
                (1, one, None)]                                                               # x - 1: an assertion at step 0
    tables = [[(rand_cols(rng, F64, 1, ce * ext)[0], d) for d in divisors] for _ in range(n_traces)]
    final_coeff = re()
    combined = [orc.combine_evaluation_table(F64, ext, [c for c, _ in tab], [d for _, d in tab], offset) for tab in tables]
    comp = orc.composition_poly_from_evaluations(F64, ext, combined, logR, n_comp, offset, final_coeff)
    want_c = orc.build_constraint_commitment(F64, comp, ext, logR, logB, offset)
    ccom, _ = ctx.constraint_commit_from_tables(capi.make_params(F64, ext, logR, logB, n_comp, 1), tables, final_coeff)
    assert ccom.root() == want_c["root"]

    # out-of-domain frames
    z = re()
    zg = orc.ext_mul(F64, ext, z, np.array([g_trace, 0], dtype=np.uint64))
    frame = tcom.evaluate_polys_at_points(np.concatenate([z, zg]), 2, ext, n_cols * n_traces)
    for q, point in enumerate((z, zg)):
        for t in range(n_traces):
            for c in range(n_cols):
                assert np.array_equal(frame[q][t * n_cols + c], orc.eval_column_at(F64, want_t["polys"][t][c], 1, point, ext))
    ood_c = ccom.evaluate_polys_at(z, ext, n_comp)
    for c in range(n_comp):
        assert np.array_equal(ood_c[c], orc.eval_column_at(F64, comp[c], ext, z, ext))

    # DEEP composition into the FRI prover, commit phase
    cc_t = [re() for _ in range(n_cols * n_traces)]
    cc_c = [re() for _ in range(n_comp)]
    want_deep = orc.deep_compose(F64, ext, R, [[(p, 1) for p in want_t["polys"][t]] for t in range(n_traces)], comp, z, cc_t, cc_c)
    fri = capi.FriProver(ctx, F64, ext, folding, 1 << logB, max_rem, offset)
    got_deep = ctx.deep_compose(F64, ext, R, [tcom], ccom, z, np.concatenate(cc_t), np.concatenate(cc_c), fri=fri, lde_blowup=1 << logB)
    assert np.array_equal(got_deep, want_deep)
    cur = orc.evaluate_poly_with_offset(F64, want_deep, R, ext, orc.get_twiddles(F64, R), L.orc_f64_new(offset), 1 << logB)
    size, want_layers = N, []
    for i in range(capi.fri_num_layers(folding, 1 << logB, max_rem, N)):
        want = orc.fri_layer_commit(F64, cur, size, ext, folding)
        assert fri.commit_layer() == want["root"]
        alpha = re()
        cur = orc.apply_drp(F64, want["transposed"], size // folding, ext, folding, offset, alpha)
        fri.fold(alpha)
        want_layers.append(want)
        size //= folding
    fri.set_remainder(size)

    # query phase: one round trip
    positions = np.sort(rng.choice(N, size=23, replace=False)).astype(np.uint64)
    requests = [(tcom, positions, True), (ccom, positions, True)]
    folded, domain = positions, N
    for i in range(len(want_layers)):
        folded = capi.fri_fold_positions(folded, domain, folding)
        requests.append((fri.layer(i), folded, True))
        domain //= folding
    answers = capi.query_many(requests)
    plist = [int(p) for p in positions]
    rows_t, proof_t = answers[0]
    for i, p in enumerate(plist):
        assert np.array_equal(rows_t[i], np.concatenate([want_t["lde"][t][p, :n_cols] for t in range(n_traces)]))
    assert proof_t == orc.merkle_prove_batch(want_t["nodes"], want_t["leaves"], plist)
    rows_c, proof_c = answers[1]
    assert np.array_equal(rows_c.reshape(len(plist), -1), want_c["lde"].reshape(N, -1)[positions.astype(np.int64)][:, :n_comp * ext])
    assert proof_c == orc.merkle_prove_batch(want_c["nodes"], want_c["leaves"], plist)
    domain = N
    for (com, pos, _), (rows, proof), want in zip(requests[2:], answers[2:], want_layers):
        tr = want["transposed"].reshape(domain // folding, folding * ext)
        assert np.array_equal(rows, tr[pos.astype(np.int64)])
        assert proof == orc.merkle_prove_batch(want["nodes"], want["leaves"], [int(p) for p in pos])
        domain //= folding
    fri.close()
    ccom.close()
    tcom.close()
