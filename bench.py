#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: trace-LDE + Merkle-commit of a 2^20 x 8 f64 trace at blowup 8.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path (Prover::build_trace_commitment, /root/reference/prover/src/lib.rs:615-670)
over one synthetic trace that is already resident in HBM: interpolate 8 columns -> evaluate over the 8 cosets into
the row-major LDE matrix -> hash 2^23 rows -> build the Merkle tree.  At N > 1 every rank commits its own
independent proof (weak scaling, BASELINE.json configs[3]) and the step ends with the path's one collective, an
all-gather of the 32-byte roots over RCCL.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     : dominant logical kernel, its algorithmic bytes per launch
                 / its HIP-event duration measured inside the timed region, against the 8 TB/s HBM peak
  cpu_baseline : the CPU oracle (oracle/, "port" of the reference's concurrent path) timed on this box's host
                 cores on one full commitment of the same workload (rank 0, N == 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_R, LOG_B, N_COLS = 20, 3, 8
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# integer-VALU rates of the path's two instruction sequences in isolation on one MI355X (scripts/microbench.hip; the
# better of the boxes measured this round, profiles/r01_microbench.txt): Goldilocks butterflies, BLAKE3 compressions
ALU_BFLY_PER_S = 1.59e12
ALU_COMPRESS_PER_S = 57.0e9


def work_model(log_r=LOG_R, log_b=LOG_B, c=N_COLS, e=8):
    """algorithmic bytes and field operations of one commitment."""
    R, N = 1 << log_r, 1 << (log_r + log_b)
    beta = 1 << log_b
    bytes_k = {
        "interpolate": R * c * e + R * c * e,          # K1: read trace, write polys
        "evaluate": R * c * e + N * c * e,             # K2: read polys, write LDE
        "hash_rows": N * c * e + N * 32,               # K3: read LDE rows, write leaves
        "merkle": N * 32 + N * 32,                     # K4: read leaves, write nodes
    }
    b_alg = R * c * e * 2 + N * c * e + N * 32 * 2     # §8(d): each datum of the path exactly once
    butterflies = c * (R // 2) * log_r * (1 + beta)
    muls = butterflies + c * R + beta * c * R
    field_ops = 3 * butterflies + c * R + beta * c * R
    compressions = N * ((c * e + 63) // 64) + (N - 1)
    return dict(bytes_per_kernel=bytes_k, b_alg=b_alg, field_ops=field_ops, modmuls=muls, compressions=compressions,
                butterflies=butterflies)


def rand_f64_dev(torch, n, seed, device):
    """n uniform Montgomery residues in [0, p) on the device (int64 storage of the u64 bit patterns)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    v = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=device, generator=gen)
    bad = (v >> 32) == -1  # top 32 bits all ones: may be >= p = 2^64 - 2^32 + 1
    return torch.where(bad, v & 0x7FFFFFFFFFFFFFFF, v)


def cpu_baseline(model, trace_host=None, gpu_root=None):
    """One full commitment of the bench workload (the very trace rank 0 committed on the GPU) with the threaded CPU
    oracle -- test infrastructure, used here only as the reported CPU baseline and as a last parity gate."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    threads = min(os.cpu_count() or 1, 64)
    try:
        threads = min(threads, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    rng = np.random.default_rng(0x57415446)
    p = np.uint64(2**64 - 2**32 + 1)

    def cols(log_r):
        out = []
        for _ in range(N_COLS):
            v = rng.integers(0, 2**64 - 1, size=1 << log_r, dtype=np.uint64, endpoint=True)
            v[v >= p] -= p
            out.append(v)
        return out

    O.build_trace_commitment(O.F64, [cols(14)], 1, 14, LOG_B, 7, threads=threads)  # warm-up (threads, page cache)
    data = cols(LOG_R) if trace_host is None else [np.ascontiguousarray(c) for c in trace_host]
    t0 = time.perf_counter()
    res = O.build_trace_commitment(O.F64, [data], 1, LOG_R, LOG_B, 7, threads=threads)
    dt = time.perf_counter() - t0
    return dict(value=model["field_ops"] / dt, unit="field-ops/s", cores=threads, kind="port",
                sample=f"1 full commitment (2^{LOG_R} x {N_COLS} f64, blowup {1 << LOG_B}) after a 2^14-row warm-up; "
                       f"{dt * 1e3:.0f} ms wall incl. output allocation",
                ms=dt * 1e3, root=res["root"].hex(),
                root_matches_gpu=(None if gpu_root is None else res["root"].hex() == gpu_root))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-launch", action="store_true", help="also report HIP-event times of every kernel launch")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import starkpack_winterfell_amd.capi as capi
    from starkpack_winterfell_amd import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # one rank per GPU; the modulo only matters for rehearsals of the multi-rank control flow on a box with fewer GPUs
    # than ranks (WF_BENCH_BACKEND=gloo) -- RCCL itself refuses two ranks on one device
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    backend = os.environ.get("WF_BENCH_BACKEND", "nccl")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    model = work_model()
    ctx = capi.Context(dev_index)
    params = capi.make_params(capi.F64, 1, LOG_R, LOG_B, N_COLS, 1)
    R, N = 1 << LOG_R, 1 << (LOG_R + LOG_B)
    proof_id = shard.proofs_of_rank(world, rank, world)[0]
    trace = rand_f64_dev(torch, N_COLS * R, shard.seed_of_proof(0x57415446, proof_id), device)
    polys = torch.empty_like(trace)
    lde = torch.empty(N * 8, dtype=torch.int64, device=device)
    leaves = torch.empty((N, 32), dtype=torch.uint8, device=device)
    nodes = torch.empty((N, 32), dtype=torch.uint8, device=device)

    stream = torch.cuda.Stream(device=device)
    torch.cuda.synchronize()
    ctx.profile_enable(1)  # HIP events at the logical-kernel boundaries of every timed step (5 per step, ~1 %)
    per_launch = {}

    # the roots of the K timed commitments of this rank; ranks run free of each other (independent proofs, no data-path
    # collective) and exchange all their roots once, inside the timed region: the path's single exchange
    roots = torch.zeros((max(args.steps, args.warmup, 1), 32), dtype=torch.uint8, device=device)

    def step(k):
        ctx.trace_commit_dev(params, trace.data_ptr(), polys.data_ptr(), lde.data_ptr(), leaves.data_ptr(),
                             nodes.data_ptr(), stream.cuda_stream)
        roots[k].copy_(nodes[1], non_blocking=True)

    with torch.cuda.stream(stream):
        for k in range(args.warmup):
            step(k)
        if world > 1:
            shard.all_gather_roots(roots)  # warm the communicator up as well
        torch.cuda.synchronize()
        ctx.profile_read()  # drop the warm-up events
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k)
        all_roots = shard.all_gather_roots(roots[:args.steps]) if world > 1 else roots[:args.steps]
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        # HIP events recorded on the launch stream in front of every kernel of the K timed steps
        for name, ms in ctx.profile_read():
            per_launch.setdefault(name, []).append(ms)

    t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    root_hex = bytes(nodes[1].cpu().numpy()).hex()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        commits = world * args.steps
        value = commits * model["field_ops"] / elapsed
        avg = {k: sum(v) / len(v) for k, v in per_launch.items()}
        logical = {
            "interpolate": sum(v for k, v in avg.items() if k.startswith(("interpolate", "layout"))),
            "evaluate": sum(v for k, v in avg.items() if k.startswith("evaluate")),
            "hash_rows": avg.get("hash_rows", 0.0),
            "merkle": avg.get("merkle", 0.0),
        }
        if args.per_launch:  # a separate, finer pass outside the timed region: one event per kernel launch
            ctx.profile_enable(2)
            with torch.cuda.stream(stream):
                for k in range(args.steps):
                    step(k)
                torch.cuda.synchronize()
            fine = {}
            for name, ms in ctx.profile_read():
                fine.setdefault(name, []).append(ms)
            avg = {k: sum(v) / len(v) for k, v in fine.items()}
        # one segment, one trace: the leaves are hashed by the last evaluation pass itself (no k_hash_rows launch, the
        # LDE is not read back); the evaluate kernel then also owns the write of the leaves
        bytes_k = dict(model["bytes_per_kernel"])
        fused_hash = "hash_rows" not in avg
        if fused_hash:
            bytes_k["evaluate"] += N * 32
            bytes_k["hash_rows"] = 0
        dom = max(logical, key=logical.get)
        dom_ms = logical[dom]
        achieved = bytes_k[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom, {}).get("hbm_bytes_per_step")
            except Exception:
                traffic = None
        out = {
            "metric": "trace-LDE + Merkle-commit field-ops/s (wall-clock ms in ms_per_step), 2^20x8 f64 trace blowup=8",
            "value": value,
            "unit": "field-ops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 (Goldilocks, Montgomery form) + u32 (BLAKE3)",
            "data": "synthetic (seeded uniform field elements, resident in HBM)",
            "config": {"workload": "BASELINE.json configs[1]: 2^20 rows x 8 cols f64, blowup 8, BLAKE3-256 Merkle; "
                                   "one independent commitment per GPU per step"
                                   + (", the roots of all steps all-gathered over RCCL once per run" if world > 1 else ""),
                       "log2_trace_len": LOG_R, "n_cols": N_COLS, "blowup": 1 << LOG_B, "n_traces": 1},
            "commits_per_s": commits / elapsed,
            # the path is integer-VALU bound: time of its butterflies and BLAKE3 compressions at the rates the same
            # instruction sequences reach in isolation (scripts/microbench.hip, profiles/r01_microbench.txt) vs the step
            "alu": {"butterflies": model["butterflies"], "blake3_compressions": model["compressions"],
                    "microbench_butterflies_per_s": ALU_BFLY_PER_S, "microbench_compressions_per_s": ALU_COMPRESS_PER_S,
                    "ideal_ms": (model["butterflies"] / ALU_BFLY_PER_S + model["compressions"] / ALU_COMPRESS_PER_S) * 1e3,
                    "frac": (model["butterflies"] / ALU_BFLY_PER_S + model["compressions"] / ALU_COMPRESS_PER_S) * 1e3 / ms_per_step},
            "path": {"b_alg_bytes": model["b_alg"], "hbm_frac": model["b_alg"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "field_ops": model["field_ops"], "blake3_compressions": model["compressions"]},
            "roofline": {"bound": "hbm", "kernel": dom + (" (leaf hashing fused into its last pass)" if fused_hash and dom == "evaluate" else ""),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes": bytes_k[dom], "avg_ms": dom_ms},
            "launch_ms": {k: round(v, 4) for k, v in avg.items()},
            "root": root_hex,
            "roots_gathered": int(all_roots.shape[0]),
        }
        if world == 1 and not args.no_cpu_baseline:
            th = trace.cpu().numpy().view("uint64").reshape(N_COLS, R)
            out["cpu_baseline"] = cpu_baseline(model, th, root_hex)
            if out["cpu_baseline"]["root_matches_gpu"] is False:
                raise SystemExit("PARITY FAILURE: CPU oracle root != GPU root on the bench workload")
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
