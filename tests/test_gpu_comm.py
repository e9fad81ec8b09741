"""GPU: the multi-GPU layer through the C ABI (wf_comm_*, wf_trace_commit_sharded_dev).

One GPU is what the test box has, so:
  * RCCL itself runs on a world of one (communicator from a unique id, all-gathers, barrier, the sharded commitment);
  * worlds of 2, 4 and 8 run as threads of this process, one context per rank on the same device, with the in-process
    transport of tests/loopback.py plugged in as wf_transport -- the partitioning, the staging buffers and every kernel
    are the ones RCCL drives, only the bytes of the two collectives travel through host memory.
Every rank's outputs are compared bit for bit with the oracle's single commitment."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import rand_cols
from loopback import Loopback, run_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
F64, F128 = 1, 2


def check_rank(shard, want, rank, world, logR, logB, n_cols, n_traces, field, polys, lde, leaves, nodes, top):
    R, blowup = 1 << logR, 1 << logB
    N, n_local = R * blowup, (R * blowup) // world
    c0, per = shard.cosets_of_rank(blowup, rank, world)
    w = 1 if field == F64 else 2
    rw = 8 * ((n_cols + 7) // 8)
    tail = (2,) if w == 2 else ()
    got_polys = polys.reshape((n_traces * n_cols, R) + tail)
    for t in range(n_traces):
        for c in range(n_cols):
            assert np.array_equal(got_polys[t * n_cols + c], want["polys"][t][c]), ("poly", t, c)
        full = want["lde"][t].reshape((R, blowup, rw) + tail)
        assert np.array_equal(lde.reshape((n_traces, R, per, rw) + tail)[t], full[:, c0:c0 + per]), ("lde", t)
    assert np.array_equal(leaves, want["leaves"][rank * n_local:(rank + 1) * n_local])
    n = world
    while n < N:  # node i of a level with n >= W nodes is local node i - n - r * n / W + n / W
        lo = n + rank * n // world
        assert np.array_equal(nodes[n // world:2 * n // world], want["nodes"][lo:lo + n // world]), ("level", n)
        n *= 2
    assert np.array_equal(top[1:2 * world], want["nodes"][1:2 * world]) and not top[0].any()


@pytest.mark.parametrize("field,logR,logB,n_cols,n_traces,world", [
    (F64, 10, 3, 8, 8, 8),      # 8 segments over 8 ranks: interpolation sharded by segment, polys all-gathered
    (F64, 11, 3, 8, 4, 4), (F64, 10, 3, 8, 2, 2), (F64, 12, 3, 16, 1, 2),
    (F64, 11, 3, 8, 1, 8),      # one segment: every rank interpolates (nothing to split), cosets one per rank
    (F64, 10, 3, 5, 3, 4),      # ragged: 15 base columns = 2 segments over 4 ranks -> replicated interpolation
    (F64, 10, 2, 3, 1, 4), (F64, 9, 3, 1, 1, 2),
    (F128, 10, 3, 4, 4, 4),     # 4 f128 segments over 4 ranks
    (F128, 10, 2, 10, 2, 2), (F128, 9, 3, 1, 1, 8),
    (F64, 10, 3, 200, 1, 4),    # rows longer than one BLAKE3 chunk
    (F64, 10, 3, 8, 2, 1)])
def test_sharded_commitment_ranks_as_threads(orc, capi, field, logR, logB, n_cols, n_traces, world):
    import torch
    from starkpack_winterfell_amd import shard
    rng = np.random.default_rng(1000 * world + logR + n_cols)
    R, blowup = 1 << logR, 1 << logB
    N = R * blowup
    traces = [rand_cols(rng, field, n_cols, R) for _ in range(n_traces)]
    want = orc.build_trace_commitment(field, traces, 1, logR, logB, 7 if field == F64 else 3)
    params = capi.make_params(field, 1, logR, logB, n_cols, n_traces)
    w = 1 if field == F64 else 2
    rw = 8 * ((n_cols + 7) // 8)
    host = np.concatenate([c.reshape(-1) for t in traces for c in t]).view(np.int64)
    dev = torch.device("cuda", 0)
    lb = Loopback(world)

    def rank_fn(r):
        ctx = capi.Context(0)
        ag, a2a = lb.collectives(r)
        comm = shard.Comm.with_transport(ctx, r, world, ag, a2a)
        per = blowup // world
        d_trace = torch.from_numpy(host).to(dev)
        d_polys = torch.full_like(d_trace, -1)
        d_lde = torch.full((n_traces * R * per * rw * w,), -1, dtype=torch.int64, device=dev)   # padding must be written
        d_leaves = torch.full((N // world, 32), 0xEE, dtype=torch.uint8, device=dev)
        d_nodes = torch.full((N // world, 32), 0xEE, dtype=torch.uint8, device=dev)
        d_top = torch.full((2 * world, 32), 0xEE, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for _ in range(2):  # a second call reuses staging buffers and tables
            comm.trace_commit_sharded_dev(params, d_trace.data_ptr(), d_polys.data_ptr(), d_lde.data_ptr(),
                                          d_leaves.data_ptr(), d_nodes.data_ptr(), d_top.data_ptr())
            ctx.synchronize()
        out = (d_polys.cpu().numpy().view(np.uint64), d_lde.cpu().numpy().view(np.uint64), d_leaves.cpu().numpy(),
               d_nodes.cpu().numpy(), d_top.cpu().numpy())
        comm.close()
        ctx.close()
        return out

    for r, (polys, lde, leaves, nodes, top) in enumerate(run_ranks(world, rank_fn)):
        check_rank(shard, want, r, world, logR, logB, n_cols, n_traces, field, polys, lde, leaves, nodes, top)
        assert bytes(top[1]) == want["root"]


def test_roots_and_leaf_shards_ranks_as_threads(capi):
    """wf_comm_all_gather_roots / _leaf_shards / barrier / max over 4 ranks (threads, loopback transport)."""
    import torch
    from starkpack_winterfell_amd import shard
    world, R, per = 4, 64, 2
    dev = torch.device("cuda", 0)
    full = (np.arange(R * world * per * 32, dtype=np.int64) * 2654435761 % 251).astype(np.uint8).reshape(R, world, per, 32)
    lb = Loopback(world)

    def rank_fn(r):
        ctx = capi.Context(0)
        comm = shard.Comm.with_transport(ctx, r, world, *lb.collectives(r))
        roots = torch.from_numpy(np.full((3, 32), r + 1, dtype=np.uint8)).to(dev)
        allr = torch.zeros((world * 3, 32), dtype=torch.uint8, device=dev)
        comm.all_gather_roots(roots.data_ptr(), 3, allr.data_ptr())
        mine = torch.from_numpy(np.ascontiguousarray(full[:, r])).to(dev)          # (k, local coset) order
        nat = torch.zeros((R * world * per, 32), dtype=torch.uint8, device=dev)
        comm.all_gather_leaf_shards(mine.data_ptr(), R, per, nat.data_ptr())
        comm.barrier()
        m = comm.max_f64(10.0 + r)
        ctx.synchronize()
        out = allr.cpu().numpy(), nat.cpu().numpy(), m
        comm.close()
        ctx.close()
        return out

    for allr, nat, m in run_ranks(world, rank_fn):
        assert np.array_equal(allr.reshape(world, 3, 32), np.repeat(np.arange(1, world + 1, dtype=np.uint8), 96).reshape(world, 3, 32))
        assert np.array_equal(nat, full.reshape(-1, 32))
        assert m == 13.0


RCCL_SCRIPT = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["WF_ROOT"])
sys.path.insert(0, os.path.join(os.environ["WF_ROOT"], "tests"))
import torch
import starkpack_winterfell_amd.capi as capi
from starkpack_winterfell_amd import shard
from oracle import oracle as O
from conftest import rand_cols
assert capi.load().wf_comm_rccl_version() > 0
ctx = capi.Context(0)
store = shard.store_from_env(0, 1)
comm = shard.Comm.with_store(ctx, store, 0, 1)            # wf_comm_unique_id + ncclCommInitRank inside libwf_lde.so
assert (comm.rank, comm.world, comm.transport) == (0, 1, "rccl")
dev = torch.device("cuda", 0)
roots = torch.arange(3 * 32, dtype=torch.uint8, device=dev).reshape(3, 32)
out = torch.zeros_like(roots)
comm.all_gather_roots(roots.data_ptr(), 3, out.data_ptr())
comm.barrier()
assert comm.max_f64(2.5) == 2.5
assert comm.gather_f64(7.25) == [7.25]
assert comm.info() == {"transport": "rccl", "count": 1, "user_rank": 0, "device": 0}, comm.info()   # ncclCommCount / UserRank / CuDevice
ctx.synchronize()
assert torch.equal(out, roots)
rng = np.random.default_rng(3)
traces = [rand_cols(rng, 1, 8, 1 << 10) for _ in range(2)]
want = O.build_trace_commitment(O.F64, traces, 1, 10, 3, 7)
p = capi.make_params(capi.F64, 1, 10, 3, 8, 2)
host = np.concatenate([c.reshape(-1) for t in traces for c in t]).view(np.int64)
d_trace = torch.from_numpy(host).to(dev)
d_polys = torch.empty_like(d_trace)
d_lde = torch.empty(2 * (1 << 13) * 8, dtype=torch.int64, device=dev)
d_leaves = torch.empty(((1 << 13), 32), dtype=torch.uint8, device=dev)
d_nodes = torch.empty_like(d_leaves)
d_top = torch.empty((2, 32), dtype=torch.uint8, device=dev)
comm.trace_commit_sharded_dev(p, d_trace.data_ptr(), d_polys.data_ptr(), d_lde.data_ptr(), d_leaves.data_ptr(),
                              d_nodes.data_ptr(), d_top.data_ptr())
ctx.synchronize()
assert np.array_equal(d_nodes.cpu().numpy(), want["nodes"]) and bytes(d_top[1].cpu().numpy()) == want["root"]
nat = torch.empty_like(d_leaves)
comm.all_gather_leaf_shards(d_leaves.data_ptr(), 1 << 10, 8, nat.data_ptr())
ctx.synchronize()
assert torch.equal(nat, d_leaves)
comm.close()
ctx.close()
print("RCCL OK")
"""


def test_rccl_world_of_one_through_the_c_abi(capi):
    """RCCL inside libwf_lde.so on the one device of this box (in a child process: a communicator owns threads)."""
    capi.load()
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "WF_ROOT": ROOT, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
    out = subprocess.run([sys.executable, "-c", RCCL_SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "RCCL OK" in out.stdout


def test_comm_argument_errors(ctx, capi):
    from starkpack_winterfell_amd import shard
    L = capi.load()
    h = C.c_void_p()
    uid = (C.c_uint8 * 128)()
    assert L.wf_comm_create(ctx._h, uid, 2, 2, C.byref(h)) == -19          # rank outside the world
    assert L.wf_comm_create_with_transport(ctx._h, C.byref(capi.Transport()), 0, 2, C.byref(h)) == -19   # empty table
    lb = Loopback(1)
    comm = shard.Comm.with_transport(ctx, 0, 1, *lb.collectives(0))
    p = capi.make_params(F64, 1, 4, 2, 1, 1)
    with pytest.raises(capi.WfError) as e:
        comm.trace_commit_sharded_dev(p, 8, 0, 0, 8, 8, 8)                   # null LDE buffer
    assert e.value.code == -19
    comm.close()


def test_failing_transport_is_reported(capi):
    """A transport error surfaces as WF_ERR_COMM from every rank; the contexts stay usable."""
    import torch
    from starkpack_winterfell_amd import shard
    world = 2
    dev = torch.device("cuda", 0)

    def boom(_mine):
        raise RuntimeError("link down")

    def rank_fn(r):
        ctx = capi.Context(0)
        comm = shard.Comm.with_transport(ctx, r, world, boom, boom)
        a = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
        b = torch.zeros((2, 32), dtype=torch.uint8, device=dev)
        with pytest.raises(capi.WfError) as e:
            comm.all_gather_roots(a.data_ptr(), 1, b.data_ptr())
        code = e.value.code
        ctx.synchronize()
        comm.close()
        ctx.close()
        return code

    assert run_ranks(world, rank_fn) == [-32, -32]


@pytest.mark.parametrize("field,logR,logB,n_cols,n_traces,world", [
    (F64, 10, 3, 8, 2, 4), (F64, 9, 3, 3, 3, 8), (F64, 10, 2, 8, 1, 2), (F128, 9, 3, 4, 2, 4), (F64, 8, 3, 8, 1, 1)])
def test_sharded_resident_commitment_and_collective_queries(orc, capi, field, logR, logB, n_cols, n_traces, world):
    """wf_trace_commit_sharded_resident + wf_sharded_commitment_query on W ranks (threads): every rank gets the rows and
    the batch proof the unsharded commitment answers with (TraceCommitment::query), the reference's error cases come back
    as status codes, and the out-of-domain frame is evaluated locally from the complete polynomials."""
    from starkpack_winterfell_amd import shard
    rng = np.random.default_rng(31 * world + logR)
    R, N = 1 << logR, 1 << (logR + logB)
    traces = [rand_cols(rng, field, n_cols, R) for _ in range(n_traces)]
    want = orc.build_trace_commitment(field, traces, 1, logR, logB, 7 if field == F64 else 3)
    params = capi.make_params(field, 1, logR, logB, n_cols, n_traces)
    cols = [c for t in traces for c in t]
    positions = np.union1d(rng.integers(0, N, size=24), [0, N - 1]).astype(np.uint64)   # distinct, both ends included
    rng.shuffle(positions)
    z = rand_cols(rng, field, 1, 1)[0]
    cc = rand_cols(rng, field, 1, n_cols * n_traces)[0]
    lb = Loopback(world)

    def rank_fn(r):
        ctx = capi.Context(0)
        comm = shard.Comm.with_transport(ctx, r, world, *lb.collectives(r))
        com = comm.trace_commit_sharded_resident(params, cols)
        root = com.root()
        rows, proof = com.query(positions)
        rows1, proof1 = com.query(positions[:1])
        codes = []
        for bad in (np.array([], dtype=np.uint64), np.array([N], dtype=np.uint64), np.array([3, 3], dtype=np.uint64)):
            try:
                com.query(bad)
                codes.append(0)
            except capi.WfError as e:
                codes.append(e.code)
        ood = com.polys().evaluate_polys_at(z, 1, n_cols * n_traces)
        # the DEEP composition needs no exchange either: every rank holds the complete polynomials
        deep = ctx.deep_compose(field, 1, R, [com.polys()], None, z, cc)
        com.close()
        comm.close()
        ctx.close()
        return root, rows, proof, rows1, proof1, codes, ood, deep

    pos = positions.astype(np.int64)
    want_rows = np.concatenate([want["lde"][t][pos][:, :n_cols] for t in range(n_traces)], axis=1)
    want_proof = orc.merkle_prove_batch(want["nodes"], want["leaves"], [int(p) for p in positions])
    want_proof1 = orc.merkle_prove_batch(want["nodes"], want["leaves"], [int(positions[0])])
    want_ood = np.stack([orc.eval_column_at(field, c, 1, z, 1) for t in want["polys"] for c in t])
    w = 1 if field == F64 else 2
    want_deep = orc.deep_compose(field, 1, R, [[(c, 1) for c in t] for t in want["polys"]], [], z,
                                 list(cc.reshape(n_cols * n_traces, -1) if w > 1 else cc.reshape(-1, 1)), [])
    for root, rows, proof, rows1, proof1, codes, ood, deep in run_ranks(world, rank_fn):
        assert np.array_equal(deep, want_deep)
        assert root == want["root"]
        assert np.array_equal(rows.reshape(want_rows.shape), want_rows)
        assert proof == want_proof
        assert np.array_equal(rows1.reshape(-1), want_rows[0].reshape(-1)) and proof1 == want_proof1
        assert codes == [-19, -18, -18]
        assert np.array_equal(ood.reshape(want_ood.shape), want_ood)


def test_watchdog_covers_the_blocking_calls_when_a_peer_never_arrives(orc, capi, monkeypatch):
    """WF_COMM_TIMEOUT_S at the C level: rank 1 leaves (it never enters the next collective), so what rank 0's transport
    queues on the stream cannot complete -- here a host function that blocks the stream.  wf_comm_barrier and
    wf_sharded_commitment_query must come back with WF_ERR_COMM after the time-out instead of hanging inside a copy into
    pageable memory, every later collective must be refused, and the copies that were still queued when the calls gave up
    must land in memory that is still there (the communicator's pinned buffer) once the stream is released."""
    import threading
    import time
    from starkpack_winterfell_amd import shard
    monkeypatch.setenv("WF_COMM_TIMEOUT_S", "2")
    world, logR, logB, n_cols = 2, 9, 3, 8
    rng = np.random.default_rng(5)
    cols = rand_cols(rng, F64, n_cols, 1 << logR)
    params = capi.make_params(F64, 1, logR, logB, n_cols, 1)
    want = orc.build_trace_commitment(F64, [cols], 1, logR, logB, 7)
    hip = shard._hip_runtime()
    HOSTFN = C.CFUNCTYPE(None, C.c_void_p)
    hip.hipLaunchHostFunc.argtypes = [C.c_void_p, HOSTFN, C.c_void_p]
    release, peer_gone, done = threading.Event(), threading.Event(), threading.Event()
    blocker = HOSTFN(lambda _u: (release.wait(60), None)[1])
    lb = Loopback(world)
    out = {}

    def rank_fn(r):
        ctx = capi.Context(0)
        ag, a2a = shard.transport_callbacks(world, *lb.collectives(r))

        def guarded(inner):
            def cb(user, d_send, d_recv, nbytes, stream):
                if r == 0 and peer_gone.is_set():  # the peer never arrives: the exchange stays queued for ever
                    return hip.hipLaunchHostFunc(C.c_void_p(stream), blocker, None)
                return inner(user, d_send, d_recv, nbytes, stream)
            return capi.TRANSPORT_FN(cb)

        cbs = (guarded(ag), guarded(a2a))
        tr = capi.Transport(None, cbs[0], cbs[1])
        h = C.c_void_p()
        capi._check(capi.load().wf_comm_create_with_transport(ctx._h, C.byref(tr), r, world, C.byref(h)))
        comm = shard.Comm(h, ctx, keep=(cbs, tr, ag, a2a))
        com = comm.trace_commit_sharded_resident(params, cols)
        assert com.root() == want["root"]
        rows, _ = com.query(np.array([1, 77], dtype=np.uint64))   # a working collective query first
        comm.barrier()
        if r == 1:
            peer_gone.set()
            done.wait(120)   # keeps its handles alive; never calls another collective
        else:
            peer_gone.wait(60)
            t0 = time.perf_counter()
            with pytest.raises(capi.WfError) as e:
                com.query(np.array([5, 9], dtype=np.uint64))
            waited = time.perf_counter() - t0
            out["query"] = (e.value.code, waited)
            t1 = time.perf_counter()
            for call in (comm.barrier, lambda: comm.max_f64(1.0), lambda: com.query(np.array([5], dtype=np.uint64))):
                with pytest.raises(capi.WfError) as e2:   # a dead communicator refuses further collectives at once
                    call()
                out.setdefault("after", []).append(e2.value.code)
            out["after_s"] = time.perf_counter() - t1   # "at once": while the stream is STILL blocked behind the dead exchange
            release.set()       # the stream drains: the copies queued behind the dead exchange run now
            ctx.synchronize()
            done.set()
        com.close()
        comm.close()
        ctx.close()
        return True

    try:
        assert run_ranks(world, rank_fn) == [True, True]
    finally:
        release.set()
        done.set()
    code, waited = out["query"]
    assert code == -32 and 1.5 < waited < 30, out
    assert out["after"] == [-32, -32, -32]
    assert out["after_s"] < 5.0, out   # (round 5: the query used to wait for the blocked stream -- a minute here -- before refusing)


def test_comm_info_and_gather_f64(capi):
    """wf_comm_info / wf_comm_gather_f64 on the loopback transport (what bench.py puts into `collective`)."""
    from starkpack_winterfell_amd import shard
    world = 4
    lb = Loopback(world)

    def rank_fn(r):
        ctx = capi.Context(0)
        comm = shard.Comm.with_transport(ctx, r, world, *lb.collectives(r))
        got = comm.gather_f64(1.5 * r), comm.info()
        comm.close()
        ctx.close()
        return got

    for r, (vals, info) in enumerate(run_ranks(world, rank_fn)):
        assert vals == [0.0, 1.5, 3.0, 4.5]
        assert info == {"transport": "caller", "count": world, "user_rank": r, "device": 0}
