"""Device times of the stand-alone transform path (run_transform, path.hip): per-launch HIP events of
  * fft::interpolate_poly_with_offset on one column of E, 2^21 .. 2^23 (the reference calls it once per proof, on the
    constraint evaluation domain: prover/src/constraints/evaluation_table.rs:180-181),
  * fft::interpolate_poly / evaluate_poly at the reference's criterion sizes (math/benches/fft.rs: 2^18 .. 2^20),
  * the same interpolation inside wf_constraint_commit_from_evaluations at cfg 2's and cfg 3's proof shapes
    (ce domain 2 x trace length, quadratic extension, two composition columns),
next to the segment-kernel interpolation of an 8-column matrix of the same length (what a commitment's K1 costs).
    python scripts/time_transforms.py [f64|f128]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import starkpack_winterfell_amd.capi as capi

field = capi.F128 if len(sys.argv) > 1 and sys.argv[1] == "f128" else capi.F64
fname = "f128" if field == capi.F128 else "f64"
w = capi.ELEM_WORDS[field]
ctx = capi.Context(0)
rng = np.random.default_rng(3)


def rand(n, ext):
    v = rng.integers(0, 2**62, size=n * ext * w, dtype=np.uint64)
    return v.reshape((n, ext, w)) if w > 1 else v.reshape((n, ext))


def launches(fn, reps=3):
    """per-launch device times (ms), median over reps, of what fn() queues on the context"""
    fn()
    acc = {}
    for _ in range(reps):
        ctx.profile_enable(2)
        fn()
        one = {}
        for k, v in ctx.profile_read():
            if k != "between_calls":
                one[k] = one.get(k, 0.0) + v
        for k, v in one.items():
            acc.setdefault(k, []).append(v)
    ctx.profile_enable(0)
    return {k: round(sorted(v)[len(v) // 2], 4) for k, v in acc.items()}


print(f"# stand-alone transforms, {fname}; per-launch device time in ms (HIP events on the launch stream)")
for logn in (18, 20, 21, 22, 23):
    for ext in ((1, 2, 3) if field == capi.F64 else (1, 2)):
        if logn >= 22 and ext == 3:
            continue
        x = rand(1 << logn, ext)
        t = launches(lambda: ctx.fft_interpolate_poly_with_offset(field, ext, x.copy(), 7 if field == capi.F64 else 3))
        fft = {k: v for k, v in t.items() if k.startswith("fft.")}
        tot = sum(fft.values())
        elems = (1 << logn) * ext
        print(f"interpolate_poly_with_offset 2^{logn} ext {ext}: {fft}  total {tot:.4f} ms  = {tot * 1e6 / (elems * logn):.3f} ns / (base element x bit)", flush=True)
    if logn <= 20:
        x = rand(1 << logn, 1)
        t = launches(lambda: ctx.fft_evaluate_poly(field, 1, x.copy()))
        print(f"evaluate_poly 2^{logn} ext 1: {({k: v for k, v in t.items() if k.startswith('fft.')})}", flush=True)

# the interpolation inside the constraint side (combined evaluations in, two composition columns out)
if field == capi.F64:
    for logR in (20, 22):
        ce = 1 << (logR + 1)
        p = capi.make_params(field, 2, logR, 3, 2, 1)
        comb = rng.integers(0, 2**62, size=ce * 2, dtype=np.uint64)

        def run():
            c, _ = ctx.constraint_commit_from_evaluations(p, [comb])
            c.close()

        t = launches(run)
        fft = sum(v for k, v in t.items() if k.startswith("fft."))
        rest = sum(v for k, v in t.items() if not k.startswith("fft."))
        print(f"constraint_commit_from_evaluations trace 2^{logR}, ce 2^{logR + 1}, ext 2, 2 columns: interpolation {fft:.4f} ms of {fft + rest:.4f} ms of kernels  {t}", flush=True)
        ctx.release_cached()

# yardstick: the commitment path's own interpolation (segment kernels) of 8 base columns of the same lengths
for logn in (20, 21, 22):
    n_cols = 8 if field == capi.F64 else 4
    p = capi.make_params(field, 1, logn, 1, n_cols, 1)
    cols = [rand(1 << logn, 1).reshape(-1, w) if w > 1 else rand(1 << logn, 1).reshape(-1) for _ in range(n_cols)]

    def run():
        c, _ = ctx.trace_commit_resident(p, cols)
        c.close()

    t = launches(run)
    it = sum(v for k, v in t.items() if k.startswith(("interpolate", "layout")))
    print(f"segment kernels, interpolation of 2^{logn} x {n_cols} (one segment): {it:.4f} ms = {it * 1e6 / ((1 << logn) * n_cols * logn):.3f} ns / (base element x bit)  "
          f"{({k: v for k, v in t.items() if k.startswith(('interpolate', 'layout'))})}", flush=True)
    ctx.release_cached()
ctx.close()
