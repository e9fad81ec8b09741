"""One-off state soak: a random sequence of calls of different kinds and shapes on ONE context (host-buffer, resident +
queries, constraint commitments, FRI commit phases, OOD evaluations, stand-alone transforms), every result checked
against the oracle -- scratch growth, the parked commitment buffers, the FRI arena and the cached tables all get
exercised in arbitrary order.   python scripts/mixed_soak.py [n] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle as orc
import starkpack_winterfell_amd.capi as capi
from conftest import rand_cols, rand_f64, rand_f128

n, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 150), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rng = np.random.default_rng(seed)
ctx = capi.Context(0)
L = orc.lib()
F64, F128 = 1, 2
live = []          # resident commitments kept alive across iterations: (handle, want)
counts = {}


def shape():
    field = int(rng.integers(1, 3))
    ext = int(rng.integers(1, 4 if field == F64 else 3))
    logR, logB = int(rng.integers(3, 13)), int(rng.integers(1, 4))
    n_cols, n_traces = int(rng.integers(1, 12)), int(rng.integers(1, 4))
    return field, ext, logR, logB, n_cols, n_traces, (7 if field == F64 else 3)


for it in range(n):
    kind = ["host", "resident", "constraint", "fri", "ood", "fft", "deep", "cevals", "drop"][int(rng.integers(0, 9))]
    counts[kind] = counts.get(kind, 0) + 1
    field, ext, logR, logB, n_cols, n_traces, offset = shape()
    R, N = 1 << logR, 1 << (logR + logB)
    if kind == "host":
        traces = [rand_cols(rng, field, n_cols, R * ext) for _ in range(n_traces)]
        want = orc.build_trace_commitment(field, traces, ext, logR, logB, offset)
        got = ctx.trace_commit(capi.make_params(field, ext, logR, logB, n_cols, n_traces), [c for t in traces for c in t])
        assert got["root"] == want["root"] and np.array_equal(got["nodes"], want["nodes"])
        for t in range(n_traces):
            assert np.array_equal(got["lde"][t], want["lde"][t])
    elif kind == "resident":
        traces = [rand_cols(rng, field, n_cols, R * ext) for _ in range(n_traces)]
        want = orc.build_trace_commitment(field, traces, ext, logR, logB, offset)
        com, _ = ctx.trace_commit_resident(capi.make_params(field, ext, logR, logB, n_cols, n_traces), [c for t in traces for c in t])
        assert com.root() == want["root"]
        live.append((com, want))
    elif kind == "constraint":
        n_cols = min(n_cols, 4)
        polys = rand_cols(rng, field, n_cols, R * ext)
        want = orc.build_constraint_commitment(field, polys, ext, logR, logB, offset)
        com = ctx.constraint_commit_resident(capi.make_params(field, ext, logR, logB, n_cols, 1), polys)
        assert com.root() == want["root"]
        live.append((com, want))
    elif kind == "fri":
        folding = [2, 4, 8, 16][int(rng.integers(0, 4))]
        logN = max(logR + logB, 7)
        size, max_rem = 1 << logN, 7
        ev = rand_cols(rng, field, 1, size * ext)[0]
        pr = capi.FriProver(ctx, field, ext, folding, 1 << logB, max_rem, offset)
        pr.begin(ev)
        cur = ev
        for i in range(capi.fri_num_layers(folding, 1 << logB, max_rem, 1 << logN)):
            want = orc.fri_layer_commit(field, cur, size, ext, folding)
            assert pr.commit_layer() == want["root"], "fri layer"
            alpha = rand_f64(rng, ext) if field == F64 else rand_f128(rng, ext)
            cur = orc.apply_drp(field, want["transposed"], size // folding, ext, folding, offset, alpha)
            pr.fold(alpha)
            size //= folding
        rem, _ = pr.set_remainder(size)
        want_rem = cur.copy()
        orc.interpolate_poly_with_offset(field, want_rem, size, ext, orc.get_twiddles(field, size, inverse=True),
                                         L.orc_f64_new(offset) if field == F64 else offset)
        w = 1 if field == F64 else 2
        assert np.array_equal(rem.reshape(-1), want_rem.reshape(-1)[:(size >> logB) * ext * w])
        pr.close()
    elif kind == "ood":
        cols = rand_cols(rng, field, n_cols, R)
        ez = int(rng.integers(1, 4 if field == F64 else 3))
        z = rand_f64(rng, ez) if field == F64 else rand_f128(rng, ez)
        got = ctx.evaluate_columns_at(field, 1, cols, z, ez)
        for c in range(n_cols):
            assert np.array_equal(got[c], orc.eval_column_at(field, cols[c], 1, z, ez))
    elif kind == "fft":
        p = rand_cols(rng, field, 1, R * ext)[0]
        want = orc.evaluate_poly_with_offset(field, p, R, ext, orc.get_twiddles(field, R),
                                             L.orc_f64_new(offset) if field == F64 else offset, 1 << logB)
        assert np.array_equal(ctx.fft_evaluate_poly_with_offset(field, ext, p, offset, 1 << logB), want)
    elif kind == "cevals":
        # constraint side from combined evaluations: interpolation over the ce domain, final_coeff combination, commitment
        ce_log = logR + int(rng.integers(1, 4))
        n_cols = min(n_cols, 1 << (ce_log - logR))
        tabs = [rand_cols(rng, field, 1, (1 << ce_log) * ext)[0] for _ in range(n_traces)]
        fc = rand_f64(rng, ext) if field == F64 else rand_f128(rng, ext)
        cols = orc.composition_poly_from_evaluations(field, ext, tabs, logR, n_cols, offset, fc)
        want = orc.build_constraint_commitment(field, cols, ext, logR, logB, offset)
        com, _ = ctx.constraint_commit_from_evaluations(capi.make_params(field, ext, logR, logB, n_cols, 1), tabs, fc if n_traces > 1 else None)
        assert com.root() == want["root"], "constraint commitment from evaluations"
        live.append((com, want))
    elif kind == "deep":
        # DEEP composition over a fresh main segment (+ auxiliary segment) and constraint columns, then into a FRI prover
        if ext == 1:
            n_aux = 0
        else:
            n_aux = int(rng.integers(0, 3))
        n_cons = int(rng.integers(0, 4))
        main = [rand_cols(rng, field, n_cols, R) for _ in range(n_traces)]
        mp = orc.build_trace_commitment(field, main, 1, logR, logB, offset)["polys"]
        cm, _ = ctx.trace_commit_resident(capi.make_params(field, 1, logR, logB, n_cols, n_traces), [c for t in main for c in t])
        handles, tables = [cm], [[(mp[t][c], 1) for c in range(n_cols)] for t in range(n_traces)]
        if n_aux:
            aux = [rand_cols(rng, field, n_aux, R * ext) for _ in range(n_traces)]
            ap = orc.build_trace_commitment(field, aux, ext, logR, logB, offset)["polys"]
            ca, _ = ctx.trace_commit_resident(capi.make_params(field, ext, logR, logB, n_aux, n_traces), [c for t in aux for c in t])
            handles.append(ca)
            for t in range(n_traces):
                tables[t] += [(ap[t][c], ext) for c in range(n_aux)]
        cons = rand_cols(rng, field, n_cons, R * ext) if n_cons else []
        cc = ctx.constraint_commit_resident(capi.make_params(field, ext, logR, logB, n_cons, 1), cons) if n_cons else None
        re = lambda k=1: rand_f64(rng, ext * k) if field == F64 else rand_f128(rng, ext * k)
        z = re()
        cct = [[re() for _ in tab] for tab in tables]
        abi = [cct[t][c] for t in range(n_traces) for c in range(n_cols)] + [cct[t][n_cols + c] for t in range(n_traces) for c in range(n_aux)]
        ccc = [re() for _ in range(n_cons)]
        w = 1 if field == F64 else 2
        flat = lambda xs: np.concatenate([np.asarray(x).reshape(-1, w) for x in xs]).reshape((-1, w) if w > 1 else -1)
        want = orc.deep_compose(field, ext, R, tables, cons, z, [c for tab in cct for c in tab], ccc)
        pr = capi.FriProver(ctx, field, ext, 2, 1 << logB, 3, offset)
        got = ctx.deep_compose(field, ext, R, handles, cc, z, flat(abi), flat(ccc) if n_cons else None, fri=pr, lde_blowup=1 << logB)
        assert np.array_equal(got, want), "deep composition"
        pr2 = capi.FriProver(ctx, field, ext, 2, 1 << logB, 3, offset)
        pr2.begin_poly(want, 1 << logB)
        assert pr.commit_layer() == pr2.commit_layer(), "deep composition into FRI"
        pr.close(); pr2.close()
        for h in handles:
            h.close()
        if cc is not None:
            cc.close()
    # queries against a random live commitment, then maybe drop some
    if live:
        com, want = live[int(rng.integers(0, len(live)))]
        Nl = com.n_rows
        pos = np.unique(rng.integers(0, Nl, size=min(20, Nl)))
        rows, proof = com.query(pos)
        assert proof == orc.merkle_prove_batch(want["nodes"], want["leaves"], [int(q) for q in pos])
    if kind == "drop" or len(live) > 6:
        while len(live) > 2:
            live.pop(int(rng.integers(0, len(live))))[0].close()
        if rng.integers(0, 3) == 0:
            ctx.release_cached()
    if it % 25 == 24:
        print(f"{it + 1} steps ok  {counts}", flush=True)
print("done", counts)
