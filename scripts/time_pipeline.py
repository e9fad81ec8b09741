"""Wall clock of the GPU side of one proof at the bench scale, chained as Prover::generate_proof chains it
(prover/src/lib.rs:240-610) with every handle resident: trace commitment -> OOD frame -> constraint commitment ->
DEEP composition (wf_deep_compose, straight into the FRI prover) -> FRI commit phase -> queries of all trees.  Constraint
evaluation (user code in the reference) is replaced by random combined evaluations over a constraint evaluation domain of
2 R points; the constraint commitment starts from them (wf_constraint_commit_from_evaluations: interpolation included).
    python scripts/time_pipeline.py [logR] [cols] [n_traces]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import starkpack_winterfell_amd.capi as capi

logR = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n_traces = int(sys.argv[3]) if len(sys.argv) > 3 else 1
logB, ext, folding, max_rem, n_queries = 3, 2, 4, 127, 50
R, N = 1 << logR, 1 << (logR + logB)
ctx = capi.Context(0)
rng = np.random.default_rng(1)
trace = [rng.integers(0, 2**62, size=R, dtype=np.uint64) for _ in range(cols * n_traces)]
comb_evals = rng.integers(0, 2**62, size=(2 * R, ext), dtype=np.uint64)   # combined constraint evaluations over a ce domain of 2 R points
cc_t = rng.integers(0, 2**62, size=(cols * n_traces, ext), dtype=np.uint64)
cc_c = rng.integers(0, 2**62, size=(2, ext), dtype=np.uint64)
z = rng.integers(0, 2**62, size=ext, dtype=np.uint64)
pos = np.sort(rng.choice(N, size=n_queries, replace=False)).astype(np.uint64)
fri = capi.FriProver(ctx, capi.F64, ext, folding, 1 << logB, max_rem, 7)
n_layers = capi.fri_num_layers(folding, 1 << logB, max_rem, N)
alphas = [rng.integers(0, 2**62, size=ext, dtype=np.uint64) for _ in range(n_layers)]
for rep in range(4):
    t = [time.perf_counter()]
    tcom, _ = ctx.trace_commit_resident(capi.make_params(capi.F64, 1, logR, logB, cols, n_traces), trace, want_polys=False)
    t.append(time.perf_counter())
    tcom.evaluate_polys_at_points(np.concatenate([z, z]), 2, ext, cols * n_traces)   # the frame: z and z * g in one call
    t.append(time.perf_counter())
    ccom, _ = ctx.constraint_commit_from_evaluations(capi.make_params(capi.F64, ext, logR, logB, 2, 1), [comb_evals])   # iNTT over the ce domain included
    t.append(time.perf_counter())
    ccom.evaluate_polys_at(z, ext, 2)                    # composition columns at z
    t.append(time.perf_counter())
    ctx.deep_compose(capi.F64, ext, R, [tcom], ccom, z, cc_t, cc_c, want_poly=False, fri=fri, lde_blowup=1 << logB)
    t.append(time.perf_counter())
    for i in range(n_layers):
        fri.commit_layer()
        fri.fold(alphas[i])
    fri.set_remainder(1 << 12)
    t.append(time.perf_counter())
    requests = [(tcom, pos, True), (ccom, pos, True)]
    p, size = pos, N
    for i in range(n_layers):
        p = capi.fri_fold_positions(p, size, folding)
        requests.append((fri.layer(i), p, True))
        size //= folding
    capi.query_many(requests, parse=False)               # every tree in one round trip (C call only)
    t.append(time.perf_counter())
    fri.reset(); tcom.close(); ccom.close()
    t.append(time.perf_counter())
    d = [(b - a) * 1e3 for a, b in zip(t, t[1:])]
    print(f"rep {rep}: trace commit {d[0]:.2f}  OOD frame {d[1]:.2f}  constraint commit from evaluations {d[2]:.2f}  its OOD {d[3]:.2f}  DEEP composition + LDE {d[4]:.2f}  "
          f"FRI layers {d[5]:.2f}  queries {d[6]:.2f}  release {d[7]:.2f}  total {sum(d):.2f} ms")
