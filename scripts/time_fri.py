"""Times the resident FRI commit phase (wf_fri_prover_*) at the bench scale: DEEP polynomial of 2^logR coefficients over
the quadratic extension -> LDE (blowup 8) -> layers (folding 4 or 8) down to the remainder.  Wall clock, host in the loop
(commit_layer returns each root to the host, as the Fiat-Shamir channel needs it).
    python scripts/time_fri.py [logR] [folding] [field 1|2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import starkpack_winterfell_amd.capi as capi

logR = int(sys.argv[1]) if len(sys.argv) > 1 else 20
folding = int(sys.argv[2]) if len(sys.argv) > 2 else 4
field = int(sys.argv[3]) if len(sys.argv) > 3 else capi.F64
ext, blowup, max_rem = 2, 8, 127
ctx = capi.Context(0)
rng = np.random.default_rng(1)
shape = ((1 << logR), ext) if field == capi.F64 else ((1 << logR), ext, 2)   # f128: (lo, hi) words, any value below 2^126
poly = rng.integers(0, 2**62, size=shape, dtype=np.uint64)
fri = capi.FriProver(ctx, field, ext, folding, blowup, max_rem, 7 if field == capi.F64 else 3)
n_layers = capi.fri_num_layers(folding, blowup, max_rem, (1 << logR) * blowup)
alphas = [rng.integers(0, 2**62, size=shape[1:], dtype=np.uint64) for _ in range(n_layers)]
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fri.begin_poly(poly, blowup)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    per = []
    for i in range(n_layers):
        s = time.perf_counter()
        fri.commit_layer()
        fri.fold(alphas[i])
        per.append((time.perf_counter() - s) * 1e3)
    rem, digest = fri.set_remainder(1 << 12)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    fri.reset()
    print(f"rep {rep}: begin_poly (H2D {poly.nbytes >> 20} MiB + LDE) {(t1 - t0) * 1e3:.2f} ms, {n_layers} layers + remainder {(t2 - t1) * 1e3:.2f} ms",
          [round(x, 3) for x in per])

# per-launch device times of one more run (events around every launch)
ctx.profile_enable(2)
fri.begin_poly(poly, blowup)
for i in range(n_layers):
    fri.commit_layer()
    fri.fold(alphas[i])
fri.set_remainder(1 << 12)
marks = ctx.profile_read()
ctx.profile_enable(0)
acc = {}
for k, v in marks:
    acc.setdefault(k, []).append(round(v, 4))
print({k: v[:3] for k, v in acc.items()})
