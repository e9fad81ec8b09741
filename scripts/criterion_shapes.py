"""The reference's own criterion benchmark shapes on the GPU, so that whoever runs `cargo bench` on the reference can line
the numbers up (run on the GPU box; writes gpurun_out/criterion_shapes.txt, copied to profiles/ per round):

  prover/benches/row_matrix.rs:15-17   matrix_evaluate_matrix: RowMatrix::evaluate_polys::<8>, 2^19 x {32,64,96} f64
                                       polynomials, blowup {2,4,8}            -> wf_constraint_commit_dev without leaves
  math/benches/fft.rs:15               fft_evaluate_poly simple / with_offset, fft_interpolate_poly, sizes 2^18..2^20,
                                       f64 / f128 / quadratic extension       -> wf_fft_* (host buffers: PCIe included)
  crypto/benches/merkle.rs:18          merkle tree construction, 65536..262144 Blake3 leaves -> wf_merkle_build_dev
Times are medians of 7 after 2 warm-ups; device-buffer forms are timed with HIP events, host forms with the wall clock."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import starkpack_winterfell_amd.capi as capi  # noqa: E402

dev = torch.device("cuda", 0)
ctx = capi.Context(0)
out = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "criterion_shapes.txt"), "w")


def say(s):
    print(s, flush=True)
    out.write(s + "\n")
    out.flush()


stream = torch.cuda.Stream(device=dev)  # the launches and the events that bracket them share this stream


def dev_time(fn, reps=7, warm=2):
    with torch.cuda.stream(stream):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            fn()
            e1.record(stream)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def host_time(fn, reps=7, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


def rand_dev(n):
    g = torch.Generator(device=dev)
    g.manual_seed(n & 0xFFFF)
    return torch.randint(0, 2**62, (n,), dtype=torch.int64, device=dev, generator=g)


say("# reference criterion shapes on one MI355X (scripts/criterion_shapes.py)")
say("# prover/benches/row_matrix.rs matrix_evaluate_matrix (RowMatrix::evaluate_polys::<8>), 2^19 rows, f64, device-resident")
for n_poly in (32, 64, 96):
    polys = rand_dev(n_poly << 19)
    for blowup in (2, 4, 8):
        p = capi.make_params(capi.F64, 1, 19, blowup.bit_length() - 1, n_poly, 1)
        lde = torch.empty((blowup << 19) * n_poly, dtype=torch.int64, device=dev)
        ms = dev_time(lambda: ctx.constraint_commit_dev(p, polys.data_ptr(), lde.data_ptr(), 0, 0, stream.cuda_stream))
        say(f"matrix_evaluate_matrix/524288 num_poly: {n_poly}, blowup_factor: {blowup}: {ms:.3f} ms")
        del lde
    del polys
torch.cuda.empty_cache()

say("# math/benches/fft.rs (host buffers in and out: PCIe included), blowup 8")
rng = np.random.default_rng(1)
for name, field, ext in (("f64", capi.F64, 1), ("f64_quad", capi.F64, 2), ("f128", capi.F128, 1), ("f128_quad", capi.F128, 2)):
    w = 1 if field == capi.F64 else 2
    for size in (1 << 18, 1 << 19, 1 << 20):
        small = rng.integers(0, 2**62, size=(size // 8) * ext * w, dtype=np.uint64)
        full = np.zeros(size * ext * w, dtype=np.uint64)
        full[:small.size] = small
        # in place on a reused buffer, as the reference's `&mut [E]` entry points work (round 5: the Python wrapper's default of copying
        # into a FRESH array per call put ~3 ms of page faults into the 32 MiB shapes -- glibc mmaps allocations from 32 MiB up --
        # which round 4's table showed as a "5 x for 2 x the size" jump; the library's own transfers are linear in size)
        work = full.copy()
        ms = host_time(lambda: ctx.fft_evaluate_poly(field, ext, work, inplace=True))
        say(f"{name}/fft_evaluate_poly/simple/{size}: {ms:.3f} ms")
        res = np.empty(size * ext * w, dtype=np.uint64).reshape((-1, w) if w > 1 else (-1,))
        ms = host_time(lambda: ctx.fft_evaluate_poly_with_offset(field, ext, small, 7 if field == capi.F64 else 3, 8, out=res))
        say(f"{name}/fft_evaluate_poly/with_offset/{size}: {ms:.3f} ms")
        ev = rng.integers(0, 2**62, size=size * ext * w, dtype=np.uint64)
        ms = host_time(lambda: ctx.fft_interpolate_poly(field, ext, ev, inplace=True))
        say(f"{name}/fft_interpolate_poly/simple/{size}: {ms:.3f} ms")
        ms = host_time(lambda: ctx.fft_interpolate_poly_with_offset(field, ext, ev, 7 if field == capi.F64 else 3, inplace=True))
        say(f"{name}/fft_interpolate_poly/with_offset/{size}: {ms:.3f} ms")

say("# crypto/benches/merkle.rs merkle tree construction (Blake3_256), device-resident")
for n in (65536, 131072, 262144, 1 << 23):
    leaves = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev)
    nodes = torch.empty_like(leaves)
    ms = dev_time(lambda: ctx.merkle_build_dev(leaves.data_ptr(), n, nodes.data_ptr(), stream.cuda_stream))
    say(f"merkle tree construction/{n}: {ms * 1e3:.1f} us")
ctx.close()
