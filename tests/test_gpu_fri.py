"""GPU: FRI layer commitments against the oracle.

Reference: FriProver::build_layers / build_layer / set_remainder (fri/src/prover/mod.rs:172-226), apply_drp
(fri/src/folding/mod.rs:85-117).  The channel (root -> alpha) stays on the host; here alpha is derived from the
root with merge_with_int so that the layer chain is deterministic."""
import numpy as np
import pytest

from conftest import rand_cols, rand_f64, rand_f128

pytestmark = pytest.mark.gpu
F64, F128 = 1, 2


def _rand(rng, field, n):
    return rand_f64(rng, n) if field == F64 else rand_f128(rng, n)


@pytest.mark.parametrize("field,ext", [(F64, 1), (F64, 2), (F64, 3), (F128, 1), (F128, 2)])
@pytest.mark.parametrize("folding", [2, 4, 8, 16])
@pytest.mark.parametrize("logn", [6, 11])
def test_layer_commit_and_drp(ctx, orc, field, ext, folding, logn):
    rng = np.random.default_rng(100 * folding + logn + ext)
    n = 1 << logn
    ev = _rand(rng, field, n * ext)
    offset = 7 if field == F64 else 3
    want = orc.fri_layer_commit(field, ev, n, ext, folding)
    got = ctx.fri_layer_commit(field, ext, ev, folding)
    assert np.array_equal(got["transposed"], want["transposed"])
    assert np.array_equal(got["leaves"], want["leaves"])
    assert np.array_equal(got["nodes"], want["nodes"])
    assert got["root"] == want["root"]
    alpha = _rand(rng, field, ext)
    want_next = orc.apply_drp(field, want["transposed"], n // folding, ext, folding, offset, alpha)
    got_next = ctx.fri_apply_drp(field, ext, got["transposed"], folding, offset, alpha)
    assert np.array_equal(got_next, want_next)


def test_build_layers_chain_f64_quad(ctx, orc):
    """A whole commit phase as build_layers runs it (prover/mod.rs:172-189): evaluations of a degree < 2^10 polynomial
    (coefficients 0..trace_length-1 as in fri/src/prover/tests.rs:58-69) over a blowup-8 domain, folding factor 4,
    remainder of max degree 31; every layer root, every folded layer and the remainder commitment match."""
    field, ext, folding, blowup, max_rem = F64, 2, 4, 8, 31
    L = orc.lib()
    trace_len = 1 << 10
    n = trace_len * blowup
    coeffs = np.zeros(n * ext, dtype=np.uint64)
    coeffs[0:trace_len * ext:ext] = [L.orc_f64_new(i) for i in range(trace_len)]
    tw = orc.get_twiddles(field, n)
    ev = coeffs.copy()
    orc.evaluate_poly(field, ev, n, ext, tw)            # build_evaluations
    ev_gpu = ev.copy()
    size, layers = n, 0
    while size > (max_rem + 1) * blowup:                # FriOptions::num_fri_layers (fri/src/options.rs:85-93)
        want = orc.fri_layer_commit(field, ev, size, ext, folding)
        got = ctx.fri_layer_commit(field, ext, ev_gpu, folding)
        assert got["root"] == want["root"], f"layer {layers}"
        seed = orc.merge_with_int(want["root"], layers)  # stand-in for channel.draw_fri_alpha()
        alpha = np.frombuffer(seed[:16], dtype=np.uint64) % np.uint64(2**63)
        alpha = np.array([L.orc_f64_new(int(a)) for a in alpha], dtype=np.uint64)
        ev = orc.apply_drp(field, want["transposed"], size // folding, ext, folding, 7, alpha)
        ev_gpu = ctx.fri_apply_drp(field, ext, got["transposed"], folding, 7, alpha)
        assert np.array_equal(ev, ev_gpu), f"layer {layers}"
        size //= folding
        layers += 1
    assert layers == 3 and size == 128
    # set_remainder (prover/mod.rs:218-226): interpolate with offset, keep size/blowup coefficients, hash them
    rem = ev.copy()
    orc.interpolate_poly_with_offset(field, rem, size, ext, orc.get_twiddles(field, size, inverse=True), L.orc_f64_new(7))
    rem_gpu = ctx.fft_interpolate_poly_with_offset(field, ext, ev_gpu, 7)
    assert np.array_equal(rem, rem_gpu)
    keep = (size // blowup) * ext
    assert bytes(ctx.hash_rows(field, rem_gpu[:keep], 1, keep)[0]) == orc.hash_elements(field, rem[:keep])
    assert not rem[keep:].any()                          # the folded polynomial really has low degree


def test_fri_argument_errors(ctx, capi):
    ev = np.zeros(64, dtype=np.uint64)
    for folding, code in ((3, -19), (32, -19)):
        with pytest.raises(capi.WfError) as e:
            ctx.fri_layer_commit(F64, 1, ev, folding)
        assert e.value.code == code
    with pytest.raises(capi.WfError) as e:
        ctx.fri_layer_commit(F64, 1, ev[:4], 4)       # too few evaluations for one tree
    assert e.value.code == -12
    with pytest.raises(capi.WfError) as e:
        ctx.fri_apply_drp(F64, 1, ev, 4, 0, np.zeros(1, dtype=np.uint64))
    assert e.value.code == -17


@pytest.mark.parametrize("field,ext,folding,blowup,max_rem,log_trace", [
    (F64, 2, 4, 8, 31, 10), (F64, 1, 2, 4, 7, 9), (F128, 1, 8, 8, 15, 9), (F64, 3, 16, 8, 7, 9)])
def test_resident_fri_prover(ctx, orc, capi, field, ext, folding, blowup, max_rem, log_trace):
    """FriProver with everything resident in HBM (wf_fri_prover): build_layers (prover/mod.rs:172-227) and the query
    phase build_proof / query_layer (:232-300) -- every root, every queried [E; N] row, every batch proof and the
    remainder equal what the oracle computes layer by layer on the host."""
    L = orc.lib()
    rng = np.random.default_rng(field * 1000 + folding * 10 + ext)
    trace_len = 1 << log_trace
    n = trace_len * blowup
    w = 1 if field == F64 else 2
    offset = 7 if field == F64 else 3
    # evaluations of a polynomial of degree < trace_len over the (unshifted) domain, as fri/src/prover/tests.rs builds them
    coeffs = np.zeros((n, ext) + ((2,) if w == 2 else ()), dtype=np.uint64)
    low = rand_cols(rng, field, 1, trace_len * ext)[0].reshape((trace_len, ext) + ((2,) if w == 2 else ()))
    coeffs[:trace_len] = low
    ev = coeffs.reshape(-1).copy()
    orc.evaluate_poly(field, ev, n, ext, orc.get_twiddles(field, n))

    n_layers = capi.fri_num_layers(folding, blowup, max_rem, n)
    size, want_layers = n, []
    while size > (max_rem + 1) * blowup:
        size //= folding
        want_layers.append(None)
    assert n_layers == len(want_layers) and n_layers >= 1

    pr = capi.FriProver(ctx, field, ext, folding, blowup, max_rem, offset)
    pr.begin(ev)
    with pytest.raises(capi.WfError):
        pr.begin(ev)                                     # "a prior proof generation request has not been completed yet"
    size, cur = n, ev
    for i in range(n_layers):
        want = orc.fri_layer_commit(field, cur, size, ext, folding)
        root = pr.commit_layer()
        assert root == want["root"], f"layer {i}"
        seed = orc.merge_with_int(want["root"], i)       # stand-in for channel.draw_fri_alpha()
        raw = np.frombuffer(seed[:8 * ext], dtype=np.uint64) % np.uint64(2**62)
        if field == F64:
            alpha = np.array([L.orc_f64_new(int(a)) for a in raw], dtype=np.uint64)
        else:
            alpha = np.stack([raw, np.zeros_like(raw)], axis=1).reshape(-1)
        cur = orc.apply_drp(field, want["transposed"], size // folding, ext, folding, offset, alpha)
        pr.fold(alpha)
        want_layers[i] = want
        size //= folding
    assert pr.num_layers() == n_layers
    rem, digest = pr.set_remainder(size)
    want_rem = cur.copy()
    off_elem = L.orc_f64_new(offset) if field == F64 else offset
    orc.interpolate_poly_with_offset(field, want_rem, size, ext, orc.get_twiddles(field, size, inverse=True), off_elem)
    keep = (size // blowup) * ext * w
    assert np.array_equal(rem.reshape(-1), want_rem.reshape(-1)[:keep])
    assert digest == orc.hash_elements(field, want_rem.reshape(-1)[:keep])

    # query phase: positions drawn over the LDE domain, folded layer by layer (fold_positions, folding/mod.rs:158-175)
    positions = np.unique(rng.integers(0, n, size=20)).astype(np.uint64)
    rng.shuffle(positions)
    domain = n
    for i in range(n_layers):
        positions = capi.fri_fold_positions(positions, domain, folding)
        target = domain // folding
        seen = []
        for p in positions:                              # the reference's definition, restated
            assert p < target and p not in seen
            seen.append(int(p))
        layer = pr.layer(i)
        assert layer.n_rows == target and layer.row_elems == folding * ext and layer.root() == want_layers[i]["root"]
        rows = layer.read_rows(positions)
        tr = want_layers[i]["transposed"].reshape((target, folding * ext) + ((2,) if w == 2 else ()))
        assert np.array_equal(rows, tr[positions.astype(np.int64)])
        leaves, nodes, depth = layer.prove_batch(positions)
        wl, wn, wd = orc.merkle_prove_batch(want_layers[i]["nodes"], want_layers[i]["leaves"], [int(p) for p in positions])
        assert leaves == wl and nodes == wn and depth == wd
        domain = target
    pr.reset()
    assert pr.num_layers() == 0
    pr.begin(ev)                                         # usable again after reset
    pr.reset()
    # begin_poly: the polynomial's coset LDE (offset, blowup) is produced on the device and committed as layer 0
    pr.begin_poly(low.reshape(-1), blowup)
    lde = np.ascontiguousarray(orc.evaluate_poly_with_offset(field, low.reshape(-1), trace_len, ext,
                                                             orc.get_twiddles(field, trace_len), off_elem, blowup))
    assert pr.commit_layer() == orc.fri_layer_commit(field, lde, n, ext, folding)["root"]
    pr.close()
