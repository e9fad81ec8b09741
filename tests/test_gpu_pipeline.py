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
