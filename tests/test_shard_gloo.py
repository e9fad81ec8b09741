"""CPU tests of the N > 1 path: the partition rules of the C ABI (wf_shard_*), the wf_transport callbacks over a gloo
group of two processes, and the whole distributed-tree bookkeeping of wf_trace_commit_sharded_dev (coset shards ->
all-to-all of digests -> per-rank sub-tree -> all-gather of sub-roots -> top levels) driven with the partition code of
the library and checked against the oracle's single tree.  No GPU compute here: the kernels themselves are covered by
tests/test_gpu_comm.py."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(worker, world, *extra):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q) + extra) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def interleave(recv: np.ndarray, n_k: int, world: int, per: int) -> np.ndarray:
    """numpy restatement of k_interleave_leaves (csrc/comm.hip): src[s][k][lc] -> dst[k][s * per + lc]."""
    return recv.reshape(world, n_k, per, 32).transpose(1, 0, 2, 3).reshape(n_k * world * per, 32)


def sharded_tree_of_rank(shard, O, want, rank, world, R, blowup, a2a, ag):
    """What wf_trace_commit_sharded_dev does after hashing, on the host with the oracle as the compression function;
    returns (leaves, nodes, top) of this rank."""
    N = R * blowup
    c0, per = shard.cosets_of_rank(blowup, rank, world)
    mine = want["leaves"].reshape(R, blowup, 32)[:, c0:c0 + per].reshape(-1)       # (k, local coset) order
    recv = a2a(np.ascontiguousarray(mine))                                         # R / W * per digests from every rank
    leaves = interleave(recv, R // world, world, per)
    nodes = O.build_merkle_nodes(leaves) if N // world >= 2 else leaves.copy()
    sub_root = nodes[1] if N // world >= 2 else leaves[0]
    subs = ag(np.ascontiguousarray(sub_root)).reshape(world, 32)
    top = np.zeros((2 * world, 32), dtype=np.uint8)
    top[world:] = subs
    if world > 1:
        top[:world] = O.build_merkle_nodes(subs)
    else:
        top[:2] = nodes[:2]
    return leaves, nodes, top


def check_rank_against_single_tree(shard, want, rank, world, R, blowup, leaves, nodes, top):
    N = R * blowup
    n_local = N // world
    assert np.array_equal(leaves, want["leaves"][rank * n_local:(rank + 1) * n_local])
    # node i of a level with n >= W nodes is local node i - n - r * n / W + n / W (include/wf_lde.h)
    n = world
    while n < N:
        lo = n + rank * n // world
        assert np.array_equal(nodes[n // world:2 * n // world], want["nodes"][lo:lo + n // world]), n
        n *= 2
    assert np.array_equal(top[1:world], want["nodes"][1:world]) and not top[0].any()
    for s in range(world):
        assert np.array_equal(top[world + s], want["nodes"][world + s] if world < N else want["leaves"][s])
    # routing of queried positions
    for pos in (0, 1, N - 1, N // 2 + 3, 5 * blowup + 2):
        rr, rl, tr, ll = shard.route(N.bit_length() - 1, blowup, world, pos)
        c0, per = shard.cosets_of_rank(blowup, rr, world)
        assert c0 <= pos % blowup < c0 + per and rl == (pos // blowup) * per + pos % blowup - c0
        assert tr == pos // n_local and ll == pos % n_local
        if tr == rank:
            assert np.array_equal(leaves[ll], want["leaves"][pos])


def _transport_worker(rank, world, port, q):
    _init(rank, world, port)
    from starkpack_winterfell_amd import shard
    ag, a2a = shard.process_group_collectives()
    cbs = shard.transport_callbacks(world, ag, a2a, host_memory=True)
    n = 96
    send = (np.arange(n, dtype=np.uint8) + 17 * rank).astype(np.uint8)
    recv = np.zeros(world * n, dtype=np.uint8)
    assert cbs[0](None, send.ctypes.data, recv.ctypes.data, n, None) == 0
    ok = all(np.array_equal(recv[s * n:(s + 1) * n], (np.arange(n) + 17 * s).astype(np.uint8)) for s in range(world))
    send2 = np.concatenate([np.full(n, 10 * rank + s, dtype=np.uint8) for s in range(world)])   # block s is for rank s
    recv2 = np.zeros(world * n, dtype=np.uint8)
    assert cbs[1](None, send2.ctypes.data, recv2.ctypes.data, n, None) == 0
    ok = ok and all((recv2[s * n:(s + 1) * n] == 10 * s + rank).all() for s in range(world))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_transport_callbacks_world2():
    """The wf_transport a gloo group provides: all_gather is rank-major, all_to_all delivers block s to rank s."""
    assert _spawn(_transport_worker, 2) == [(0, True), (1, True)]


def _tree_worker(rank, world, port, q, logR, logB, n_cols, n_traces):
    _init(rank, world, port)
    from oracle import oracle as O
    from starkpack_winterfell_amd import shard
    from conftest import rand_cols
    rng = np.random.default_rng(5)
    R, blowup = 1 << logR, 1 << logB
    traces = [rand_cols(rng, 1, n_cols, R) for _ in range(n_traces)]      # the same packed traces on every rank
    want = O.build_trace_commitment(O.F64, traces, 1, logR, logB, 7)
    ag, a2a = shard.process_group_collectives()
    leaves, nodes, top = sharded_tree_of_rank(shard, O, want, rank, world, R, blowup, a2a, ag)
    check_rank_against_single_tree(shard, want, rank, world, R, blowup, leaves, nodes, top)
    q.put((rank, bytes(top[1]) == want["root"]))
    dist.barrier()
    dist.destroy_process_group()


def test_distributed_tree_world2():
    """Two gloo ranks assemble one packed commitment's tree from coset shards; every rank ends with the oracle's root,
    its leaf range, its sub-tree and the replicated top levels."""
    assert _spawn(_tree_worker, 2, 6, 3, 3, 2) == [(0, True), (1, True)]


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_distributed_tree_threads(orc, world):
    """The same bookkeeping for every world size the blowup admits, ranks as threads (tests/loopback.py)."""
    from loopback import Loopback, run_ranks
    from starkpack_winterfell_amd import shard
    from conftest import rand_cols
    rng = np.random.default_rng(world)
    logR, logB = 5, 3
    R, blowup = 1 << logR, 1 << logB
    traces = [rand_cols(rng, 1, 2, R) for _ in range(3)]
    want = orc.build_trace_commitment(orc.F64, traces, 1, logR, logB, 7)
    lb = Loopback(world, timeout=60)

    def rank_fn(r):
        ag, a2a = lb.collectives(r)
        leaves, nodes, top = sharded_tree_of_rank(shard, orc, want, r, world, R, blowup, a2a, ag)
        check_rank_against_single_tree(shard, want, r, world, R, blowup, leaves, nodes, top)
        return bytes(top[1])

    assert run_ranks(world, rank_fn) == [want["root"]] * world


def test_partition_rules():
    from starkpack_winterfell_amd import capi, shard
    for n, w in ((8, 8), (8, 3), (5, 2), (1, 4), (0, 2)):
        parts = [shard.proofs_of_rank(n, r, w) for r in range(w)]
        assert sorted(x for p in parts for x in p) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert shard.proofs_of_rank(4, 0, 2) == [0, 1] and shard.proofs_of_rank(4, 1, 2) == [2, 3]
    assert len({shard.seed_of_proof(7, i) for i in range(64)}) == 64
    assert [shard.cosets_of_rank(8, r, 4) for r in range(4)] == [(0, 2), (2, 2), (4, 2), (6, 2)]
    assert shard.cosets_of_rank(8, 0, 1) == (0, 8)
    for bad in ((8, 0, 3), (8, 0, 16), (6, 0, 2)):
        with pytest.raises(ValueError):
            shard.cosets_of_rank(*bad)
    L = capi.load()
    first, count = C.c_uint32(), C.c_uint32()
    assert L.wf_shard_proofs(4, 2, 2, C.byref(first), C.byref(count)) == -19          # rank outside the world
    assert L.wf_shard_route(4, 8, 2, 16, None, None, None, None) == -18               # position outside the domain
    assert shard.route(23, 8, 8, 12345) == (1, 1543, 0, 12345)
    assert shard.route(10, 8, 2, 1023) == (1, 127 * 4 + 3, 1, 511)


def test_comm_entry_points_reject_bad_arguments_without_a_device(capi):
    L = capi.load()
    h = C.c_void_p()
    assert L.wf_comm_create(None, None, 0, 1, C.byref(h)) == -19
    assert L.wf_comm_create_with_transport(None, None, 0, 1, C.byref(h)) == -19
    assert L.wf_comm_barrier(None) == -19 and L.wf_comm_rank(None) == -1 and L.wf_comm_world(None) == 0
    assert L.wf_comm_all_gather_roots(None, None, 1, None, None) == -19
    assert L.wf_trace_commit_sharded_dev(None, None, None, None, None, None, None, None, None) == -19
