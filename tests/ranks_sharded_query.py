"""One rank of a REAL multi-process run of the sharded commitment and its collective query service, checked against the
CPU oracle.  Started by tests/test_gpu_multi_device.py as

    python -m torch.distributed.run --nproc-per-node W tests/ranks_sharded_query.py

  * default transport: RCCL inside libwf_lde.so (wf_comm_create; the unique id travels through the launcher's TCP store),
    one rank per GPU -- what an 8-GPU node runs;
  * WF_BENCH_BACKEND=gloo: the same ranks share the devices there are (rehearsal on a one-GPU box) and the bytes of the
    collectives travel over a gloo group plugged in as wf_transport.

Every rank: wf_trace_commit_sharded_resident on the same seeded host columns (2 packed traces of 2^12 x 8 f64, blowup 8,
and a 4-column f128 pair), the root against the oracle's single tree, wf_sharded_commitment_query at positions on every
rank's cosets and leaf ranges against the oracle's rows and merkle_prove_batch, the out-of-domain hand-over (polys view).
Exit code 0 and one line "RANK r OK" per rank; anything else is a failure.  (A file under tests/ that is not collected by
pytest: no test_ prefix.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import starkpack_winterfell_amd.capi as capi
    from starkpack_winterfell_amd import shard
    from oracle import oracle as O
    from conftest import rand_cols

    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("WF_BENCH_BACKEND", "rccl")
    dev = local_rank % max(1, torch.cuda.device_count())
    ctx = capi.Context(dev)
    if backend == "gloo":
        import torch.distributed as dist
        dist.init_process_group("gloo")
        comm = shard.Comm.with_process_group(ctx)
    else:
        assert torch.cuda.device_count() >= world, "RCCL needs one device per rank"
        store = shard.store_from_env(rank, world)
        comm = shard.Comm.with_store(ctx, store, rank, world)
        assert comm.transport == "rccl" and capi.load().wf_comm_rccl_version() > 0
        assert capi.load().wf_comm_rccl_path()
    assert (comm.rank, comm.world) == (rank, world)
    O.build()

    for field, log_r, log_b, n_cols, n_traces, offset in ((capi.F64, 12, 3, 8, 2, 7), (capi.F128, 10, 3, 4, 2, 3)):
        rng = np.random.default_rng(20260 + field)  # the same columns on every rank
        R, blowup = 1 << log_r, 1 << log_b
        N = R * blowup
        traces = [rand_cols(rng, field, n_cols, R) for _ in range(n_traces)]
        want = O.build_trace_commitment(field, traces, 1, log_r, log_b, offset)
        params = capi.make_params(field, 1, log_r, log_b, n_cols, n_traces)
        com = comm.trace_commit_sharded_resident(params, [c for t in traces for c in t])
        assert com.root() == want["root"], f"rank {rank}: sharded root != oracle root"
        # positions on every coset (= every row owner) and in every leaf range (= every tree owner), plus neighbours
        pos = sorted({(k * blowup + c) % N for c in range(blowup) for k in (1, R // 2 + c)} |
                     {r * (N // world) + d for r in range(world) for d in (0, 1, N // world - 1)})
        rows, proof = com.query(pos)
        want_rows = np.concatenate([want["lde"][t][pos][:, :n_cols] for t in range(n_traces)], axis=1)
        assert np.array_equal(rows.reshape(want_rows.shape), want_rows), f"rank {rank}: queried rows differ from the oracle's LDE"
        assert proof == O.merkle_prove_batch(want["nodes"], want["leaves"], [int(x) for x in pos]), \
            f"rank {rank}: batch proof differs from the oracle's"
        # the out-of-domain frame needs no exchange: the polynomials are complete on every rank
        z = rand_cols(np.random.default_rng(5), field, 1, 1)[0]
        ood = com.polys().evaluate_polys_at(z, 1, n_cols * n_traces)
        want_ood = np.stack([O.eval_column_at(field, c, 1, z, 1) for t in want["polys"] for c in t])
        assert np.array_equal(ood.reshape(want_ood.shape), want_ood), f"rank {rank}: out-of-domain values"
        com.close()
        comm.barrier()
    worst = comm.max_f64(float(rank))
    assert worst == float(world - 1)
    comm.close()
    ctx.close()
    if backend == "gloo":
        import torch.distributed as dist
        dist.destroy_process_group()
    print(f"RANK {rank} OK", flush=True)


if __name__ == "__main__":
    main()
