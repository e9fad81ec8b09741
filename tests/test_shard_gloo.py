"""Multi-process CPU test (gloo, world_size 2) of the N > 1 path: proof partition + the one all-gather of roots."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from starkpack_winterfell_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_proofs = 2 * world
    mine = shard.proofs_of_rank(n_proofs, rank, world)
    # a stand-in "root" per proof: deterministic bytes derived from the proof id (no GPU compute in this test)
    roots = torch.stack([torch.full((32,), pid + 1, dtype=torch.uint8) + torch.arange(32, dtype=torch.uint8)
                         for pid in mine])
    gathered = shard.all_gather_roots(roots)
    out_q.put((rank, mine, gathered.numpy().tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_roots_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    assert results[0][1] == [0, 1] and results[1][1] == [2, 3]          # disjoint, ordered partition
    want = [[(pid + 1 + i) % 256 for i in range(32)] for pid in range(4)]
    for _, _, gathered in results:                                        # every rank sees all roots, rank-major
        assert gathered == want


def test_partition_and_seeds():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from starkpack_winterfell_amd import shard
    for n, w in ((8, 8), (8, 3), (5, 2), (1, 4), (0, 2)):
        parts = [shard.proofs_of_rank(n, r, w) for r in range(w)]
        assert sorted(x for p in parts for x in p) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    seeds = {shard.seed_of_proof(7, i) for i in range(64)}
    assert len(seeds) == 64
    one = torch.zeros((1, 32), dtype=torch.uint8)
    assert torch.equal(shard.all_gather_roots(one), one)                  # world 1: identity, no process group


def _leaf_worker(rank, world, port, out_q):
    import sys
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from starkpack_winterfell_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # every rank derives the same "full" natural-order leaves and keeps only the rows of its cosets
    R, blowup = 64, 8
    full = (np.arange(R * blowup * 32, dtype=np.int64) * 2654435761 % 251).astype(np.uint8).reshape(R * blowup, 32)
    c0, nc = shard.cosets_of_rank(blowup, rank, world)
    local = full.reshape(R, blowup, 32)[:, c0:c0 + nc].reshape(R * nc, 32)
    got = shard.all_gather_leaf_shards(torch.from_numpy(local.copy()), R, nc)
    out_q.put((rank, bool((got.numpy() == full).all())))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_leaf_shards_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_leaf_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == [(0, True), (1, True)]


def test_coset_partition():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import pytest
    from starkpack_winterfell_amd import shard
    assert [shard.cosets_of_rank(8, r, 4) for r in range(4)] == [(0, 2), (2, 2), (4, 2), (6, 2)]
    assert shard.cosets_of_rank(8, 0, 1) == (0, 8)
    with pytest.raises(ValueError):
        shard.cosets_of_rank(8, 0, 3)
    g = torch.arange(2 * 4 * 3 * 32, dtype=torch.int64).remainder(256).to(torch.uint8).view(-1, 32)
    out = shard.interleave_leaf_shards(g, 2, 4, 3)     # world 2, R = 4, 3 cosets per rank
    v = g.view(2, 4, 3, 32)
    assert torch.equal(out.view(4, 6, 32)[1, 4], v[1, 1, 1]) and torch.equal(out.view(4, 6, 32)[3, 2], v[0, 3, 2])
