"""GPU: the constraint / DEEP / FRI side of the path at PRODUCTION size (BASELINE.json configs[2] names a quadratic-
extension constraint evaluation on a 2^22-row trace; rounds so far had compared these entry points with the oracle only
up to 2^12..2^13).

  (i)   Prover::build_constraint_commitment (prover/src/lib.rs:680-715) with E = quadratic extension:
        2^20 x 4 columns of E in full against the threaded oracle; 2^22 x 4 columns of E (cfg 3's constraint side, 2 GiB of
        LDE) through the device-buffer form: by definition on samples, the whole tree rebuilt by the oracle from the GPU
        leaves, and (round 4) the whole commitment -- root, nodes, leaves, LDE -- against the threaded oracle;
  (ii)  fft::interpolate_poly_with_offset (math/src/fft/mod.rs:362; ConstraintEvaluationTable::into_comb_poly,
        prover/src/constraints/evaluation_table.rs:166-186) and fft::evaluate_poly_with_offset (mod.rs:171;
        DeepCompositionPoly::evaluate, prover/src/composer/mod.rs:198-205) at 2^21..2^23 -- the 3-pass column-layout
        plans with the offset series -- element for element against the oracle;
  (iii) FriProver::build_layers (fri/src/prover/mod.rs:172-230) from the DEEP polynomial itself: 2^20 coefficients of E,
        blowup 8, folding 4: every layer root, the remainder and its commitment against the oracle.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import rand_f64

pytestmark = pytest.mark.gpu
F64 = 1
THREADS = 16


def _fast_layer_commit(orc, ev, n, ext, folding):
    """orc.fri_layer_commit with the row hashing done by the C oracle's threaded commit_to_comb_rows (hash_elements of
    every row + the tree) instead of one Python call per row."""
    tr = orc.transpose_slice(F64, ev, n, ext, folding)
    rows = n // folding
    leaves = np.empty((rows, 32), dtype=np.uint8)
    nodes = np.empty((rows, 32), dtype=np.uint8)
    ptrs = (C.c_void_p * 1)(tr.ctypes.data)
    rc = orc.lib().orc_commit_to_comb_rows(F64, ptrs, 1, rows, folding * ext, folding * ext, leaves.ctypes.data_as(C.c_void_p),
                                           nodes.ctypes.data_as(C.c_void_p), THREADS)
    assert rc == 0
    return dict(transposed=tr, leaves=leaves, nodes=nodes, root=bytes(nodes[1]))


def test_constraint_commitment_quadratic_2_20_full(ctx, orc, capi):
    logR, logB, n_cols, ext = 20, 3, 4, 2
    rng = np.random.default_rng(2020)
    polys = [rand_f64(rng, (1 << logR) * ext) for _ in range(n_cols)]
    want = orc.build_constraint_commitment(F64, polys, ext, logR, logB, 7, threads=THREADS)
    got = ctx.constraint_commit(capi.make_params(F64, ext, logR, logB, n_cols, 1), polys)
    assert got["root"] == want["root"]
    assert np.array_equal(got["lde"], want["lde"])
    assert np.array_equal(got["leaves"], want["leaves"])
    assert np.array_equal(got["nodes"], want["nodes"])


def test_constraint_commitment_quadratic_2_22_device(ctx, orc, capi):
    import torch
    logR, logB, n_cols, ext = 22, 3, 4, 2
    R, N, B = 1 << logR, 1 << (logR + logB), n_cols * ext
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(22)
    polys = torch.randint(-2**63, 2**63 - 1, (n_cols * R * ext,), dtype=torch.int64, device=dev, generator=gen)
    polys = torch.where((polys >> 32) == -1, polys & 0x7FFFFFFFFFFFFFFF, polys)
    lde = torch.full((N * 8,), -1, dtype=torch.int64, device=dev)
    leaves = torch.empty((N, 32), dtype=torch.uint8, device=dev)
    nodes = torch.empty((N, 32), dtype=torch.uint8, device=dev)
    params = capi.make_params(F64, ext, logR, logB, n_cols, 1)
    ctx.constraint_commit_dev(params, polys.data_ptr(), lde.data_ptr(), leaves.data_ptr(), nodes.data_ptr())
    ctx.synchronize()
    L = orc.lib()
    u64 = lambda t: t.cpu().numpy().view(np.uint64)  # noqa: E731
    lde2 = lde.view(N, 8)
    g = L.orc_f64_get_root_of_unity(logR + logB)
    off = L.orc_f64_new(7)
    js = [0, 1, 9, N - 1, 23456789, 8 * 4321 + 5]
    xs = np.array([L.orc_f64_mul(off, L.orc_f64_exp(g, j)) for j in js], dtype=np.uint64)
    rows = u64(lde2[js])
    for c in range(n_cols):
        col = u64(polys[c * R * ext:(c + 1) * R * ext]).reshape(R, ext)
        for e in range(ext):  # a polynomial over E at a base-field point: coordinate by coordinate (quadratic.rs:26-28)
            assert np.array_equal(orc.eval_many(F64, np.ascontiguousarray(col[:, e]), xs), rows[:, c * ext + e]), (c, e)
    lh = leaves.cpu().numpy()
    for j, row in zip(js, rows):
        assert bytes(lh[j]) == orc.hash_elements(F64, row[:B])
    assert np.array_equal(orc.build_merkle_nodes(lh, threads=THREADS), nodes.cpu().numpy())

    # the WHOLE constraint commitment of cfg 3 against the threaded oracle (Prover::build_constraint_commitment,
    # prover/src/lib.rs:680-715 restated): root, every node, every leaf and the 2 GiB of LDE -- as the trace side is in
    # test_gpu_fullsize.py.  ~4 GiB of host memory for the oracle's outputs.
    import os
    avail = 0.0
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) / (1 << 20)
    except OSError:
        pass
    if avail < 12:
        del lde, lde2, leaves, nodes, polys
        torch.cuda.empty_cache()
        pytest.skip(f"sampled checks and the tree passed; the full comparison needs ~6 GiB of host memory, MemAvailable is {avail:.0f} GiB")
    ph = u64(polys).reshape(n_cols, R * ext)
    threads = min(64, len(os.sched_getaffinity(0)))
    want = orc.build_constraint_commitment(F64, [ph[c] for c in range(n_cols)], ext, logR, logB, 7, threads=threads)
    nh = nodes.cpu().numpy()
    assert bytes(nh[1]) == want["root"], "cfg 3 constraint side: root differs from the oracle's"
    assert np.array_equal(nh, want["nodes"])
    assert np.array_equal(lh, want["leaves"])
    assert np.array_equal(u64(lde2), np.asarray(want["lde"]).reshape(N, 8))
    del want, ph, nh, lh, lde, lde2, leaves, nodes, polys
    torch.cuda.empty_cache()


@pytest.mark.parametrize("logn,ext", [(21, 1), (21, 2), (22, 1), (22, 2), (23, 1)])
def test_interpolate_poly_with_offset_large(ctx, orc, logn, ext):
    rng = np.random.default_rng(logn * 10 + ext)
    n = 1 << logn
    ev = rand_f64(rng, n * ext)
    want = ev.copy()
    orc.interpolate_poly_with_offset(F64, want, n, ext, orc.get_twiddles(F64, n, inverse=True), orc.lib().orc_f64_new(7))
    got = ctx.fft_interpolate_poly_with_offset(F64, ext, ev, 7)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("logn,ext,blowup", [(20, 2, 8), (21, 1, 4), (19, 3, 8), (22, 1, 2)])
def test_evaluate_poly_with_offset_large(ctx, orc, logn, ext, blowup):
    rng = np.random.default_rng(logn + 7 * ext)
    n = 1 << logn
    poly = rand_f64(rng, n * ext)
    want = orc.evaluate_poly_with_offset(F64, poly, n, ext, orc.get_twiddles(F64, n), orc.lib().orc_f64_new(7), blowup)
    got = ctx.fft_evaluate_poly_with_offset(F64, ext, poly, 7, blowup)
    assert np.array_equal(got.reshape(-1), np.ascontiguousarray(want).reshape(-1))


def test_fri_commit_phase_from_deep_poly_2_20(ctx, orc, capi):
    ext, folding, blowup, max_rem, log_trace = 2, 4, 8, 31, 20
    L = orc.lib()
    rng = np.random.default_rng(77)
    trace_len = 1 << log_trace
    n = trace_len * blowup
    poly = rand_f64(rng, trace_len * ext)
    off = L.orc_f64_new(7)
    cur = np.ascontiguousarray(orc.evaluate_poly_with_offset(F64, poly, trace_len, ext, orc.get_twiddles(F64, trace_len), off, blowup)).reshape(-1)
    n_layers = capi.fri_num_layers(folding, blowup, max_rem, n)
    assert n_layers == 8
    pr = capi.FriProver(ctx, F64, ext, folding, blowup, max_rem, 7)
    pr.begin_poly(poly, blowup)                 # DeepCompositionPoly::evaluate on the device, straight into layer 0
    size = n
    for i in range(n_layers):
        want = _fast_layer_commit(orc, cur, size, ext, folding)
        assert pr.commit_layer() == want["root"], f"layer {i}"
        seed = orc.merge_with_int(want["root"], i)   # stand-in for channel.draw_fri_alpha()
        raw = np.frombuffer(seed[:8 * ext], dtype=np.uint64) % np.uint64(2**62)
        alpha = np.array([L.orc_f64_new(int(a)) for a in raw], dtype=np.uint64)
        cur = orc.apply_drp(F64, want["transposed"], size // folding, ext, folding, 7, alpha, threads=THREADS)
        pr.fold(alpha)
        if i in (0, 3):                              # a few queried rows + their batch proof from the resident layer
            pos = np.array([0, 5, size // folding - 1, (size // folding) // 3], dtype=np.uint64)
            layer = pr.layer(i)
            assert np.array_equal(layer.read_rows(pos), want["transposed"].reshape(size // folding, folding * ext)[pos.astype(np.int64)])
            assert layer.prove_batch(pos) == orc.merkle_prove_batch(want["nodes"], want["leaves"], [int(p) for p in pos])
        size //= folding
    rem, digest = pr.set_remainder(size)
    want_rem = cur.copy()
    orc.interpolate_poly_with_offset(F64, want_rem, size, ext, orc.get_twiddles(F64, size, inverse=True), off)
    keep = (size // blowup) * ext
    assert np.array_equal(rem.reshape(-1), want_rem[:keep]) and not want_rem[keep:].any()
    assert digest == orc.hash_elements(F64, want_rem[:keep])
    pr.close()


def test_proof_chain_cfg2_constraint_side_and_deep_composition(ctx, orc, capi):
    """cfg 2's proof shape, the resident chain that follows the trace commitment: the constraint side from combined evaluations
    over a constraint evaluation domain of 2^21 points (interpolation with offset, STARKPack combination of two packed
    traces, two columns of the quadratic extension, commitment: prover/src/lib.rs:435-472) and the DEEP composition over the
    2^20 x 8 main trace and those columns (composer/mod.rs:62-193), both in full against the oracle."""
    logR, logB, ext, n_cols = 20, 3, 2, 8
    R = 1 << logR
    rng = np.random.default_rng(77)
    trace = [rand_f64(rng, R) for _ in range(n_cols)]
    want_t = orc.build_trace_commitment(F64, [trace], 1, logR, logB, 7, threads=THREADS)
    tcom, _ = ctx.trace_commit_resident(capi.make_params(F64, 1, logR, logB, n_cols, 1), trace)
    assert tcom.root() == want_t["root"]
    tables = [rand_f64(rng, 2 * R * ext) for _ in range(2)]
    fc = rand_f64(rng, ext)
    want_cols = orc.composition_poly_from_evaluations(F64, ext, tables, logR, 2, 7, fc)
    want_c = orc.build_constraint_commitment(F64, want_cols, ext, logR, logB, 7, threads=THREADS)
    ccom, polys = ctx.constraint_commit_from_evaluations(capi.make_params(F64, ext, logR, logB, 2, 1), tables, fc, want_polys=True)
    assert ccom.root() == want_c["root"]
    assert all(np.array_equal(a, b) for a, b in zip(polys, want_cols))
    z = rand_f64(rng, ext)
    cc_t = [rand_f64(rng, ext) for _ in range(n_cols)]
    cc_c = [rand_f64(rng, ext) for _ in range(2)]
    want_d = orc.deep_compose(F64, ext, R, [[(p, 1) for p in want_t["polys"][0]]], want_cols, z, cc_t, cc_c)
    got_d = ctx.deep_compose(F64, ext, R, [tcom], ccom, z, np.concatenate(cc_t), np.concatenate(cc_c))
    assert np.array_equal(got_d, want_d)
    tcom.close()
    ccom.close()
