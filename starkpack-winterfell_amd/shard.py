"""Multi-GPU layout of the path -- a thin caller of the wf_comm part of the C ABI (include/wf_lde.h, csrc/comm.hip).

The reference has no distributed code ("Distributed communication backend: none").  One process per
GPU; the partition rules (wf_shard_*), the exchanges (RCCL inside libwf_lde.so) and the kernels around them all
live behind the C ABI, so a Rust host binds exactly what this module binds:

  * independent proofs, one per GPU (BASELINE.json configs[3]): no data-path collective, ONE all-gather of the
    32-byte roots -- `Comm.all_gather_roots`;
  * one STARKPack commitment sharded by coset -- `Comm.trace_commit_sharded_dev`.

Rendezvous: rank 0 draws the RCCL unique id (`wf_comm_unique_id`) and publishes it through a key-value store (the
TCPStore that `torch.distributed.run` sets up is used here; any channel of the host will do).  For rehearsals on
machines without one GPU per rank, `Comm.with_process_group` plugs a `torch.distributed` group (gloo) into the
library as a caller-supplied transport (`wf_transport`): partitioning, staging and kernels are the ones RCCL runs
with, only the bytes travel differently.
"""
from __future__ import annotations

import ctypes as C
import os

from . import capi


# ---- partition rules (wf_shard_*: no device needed) ---------------------------------------------------------------
def proofs_of_rank(n_proofs: int, rank: int, world: int):
    """Contiguous block partition of proof ids over ranks (first ranks take the remainder)."""
    first, count = C.c_uint32(), C.c_uint32()
    capi._check(capi.load().wf_shard_proofs(n_proofs, rank, world, C.byref(first), C.byref(count)))
    return list(range(first.value, first.value + count.value))


def seed_of_proof(base_seed: int, proof_id: int) -> int:
    """Synthetic-input seed of one proof (seeds offset by proof / rank)."""
    return (base_seed + 0x9E3779B97F4A7C15 * (proof_id + 1)) & 0x7FFFFFFFFFFFFFFF


def cosets_of_rank(blowup: int, rank: int, world: int):
    """(first coset, count) of a rank when ONE packed commitment is sharded by coset (world must divide blowup)."""
    first, count = C.c_uint32(), C.c_uint32()
    rc = capi.load().wf_shard_cosets(blowup, rank, world, C.byref(first), C.byref(count))
    if rc:
        raise ValueError(capi.load().wf_last_error().decode())
    return first.value, count.value


def route(log2_lde_rows: int, blowup: int, world: int, position: int):
    """(row_rank, row_local, tree_rank, leaf_local) of an LDE row of a sharded commitment (wf_shard_route)."""
    rr, tr = C.c_uint32(), C.c_uint32()
    rl, ll = C.c_uint64(), C.c_uint64()
    capi._check(capi.load().wf_shard_route(log2_lde_rows, blowup, world, position, C.byref(rr), C.byref(rl),
                                           C.byref(tr), C.byref(ll)))
    return rr.value, rl.value, tr.value, ll.value


# ---- rendezvous ---------------------------------------------------------------------------------------------------
def unique_id() -> bytes:
    """ncclGetUniqueId through the library (wf_comm_unique_id): the 128 bytes rank 0 hands to every rank."""
    uid = (C.c_uint8 * 128)()
    capi._check(capi.load().wf_comm_unique_id(uid))
    return bytes(uid)


class Loopback:
    """An in-process transport for wf_comm -- W ranks as W threads of one process, the bytes of the two collectives handed
    over through host memory (`collectives(rank)` -> the pair `Comm.with_transport` takes).  Lets one GPU (or none, for the
    callbacks alone) run the multi-rank code paths of libwf_lde.so exactly as RCCL would drive them: rehearsals of the
    thread-per-GPU route of bench.py and the tests."""

    def __init__(self, world: int, timeout: float = 120.0):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world, timeout=timeout)
        self.slots = [None] * world

    def collectives(self, rank: int):
        import numpy as np
        world = self.world

        def all_gather(mine):
            self.slots[rank] = np.array(mine, copy=True)
            self.barrier.wait()
            out = np.concatenate(self.slots)
            self.barrier.wait()  # nobody overwrites a slot before everybody has read it
            return out

        def all_to_all(mine):
            self.slots[rank] = np.array(mine, copy=True)
            self.barrier.wait()
            n = mine.size // world
            out = np.concatenate([self.slots[s][rank * n:(rank + 1) * n] for s in range(world)])
            self.barrier.wait()
            return out

        return all_gather, all_to_all


def store_from_env(rank: int, world: int, timeout_s: float = 300.0):
    """The TCP key-value store at MASTER_ADDR:MASTER_PORT (rank 0 hosts it) -- the env torch.distributed.run sets."""
    import datetime
    from torch.distributed import TCPStore
    # under torch.distributed.run the launcher already hosts the store on MASTER_PORT: every rank is a client then
    agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "False") == "True"
    return TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")), world,
                    rank == 0 and not agent_store, timeout=datetime.timedelta(seconds=timeout_s), wait_for_workers=False)


class Comm:
    """wf_comm wrapper: one per (context, rank)."""

    def __init__(self, handle, ctx, keep=()):
        self._h = handle
        self.ctx = ctx
        self._keep = keep  # callback objects of a custom transport must outlive the communicator
        self.rank = capi.load().wf_comm_rank(self._h)
        self.world = capi.load().wf_comm_world(self._h)
        self.transport = "transport" if keep else "rccl"
        ctx._adopt(self)  # closed before the context (capi.Context.close), whatever order finalisers run in

    _store_seq = 0  # communicators created through a store by this process: part of the key (no stale ids)

    @classmethod
    def with_store(cls, ctx, store, rank: int, world: int, key: str | None = None):
        """RCCL communicator; the unique id travels through `store` (set/get of bytes).  Rank 0 ALWAYS publishes
        something under the key -- the id, or an error marker that every peer turns into an exception -- so a rank 0
        that cannot load RCCL does not leave the others waiting for the store's timeout.  The default key carries a
        per-process sequence number (every rank creates its communicators in the same order), so a second communicator
        on the same store never reads the first one's id."""
        L = capi.load()
        if key is None:
            key = f"wf_comm_id/{cls._store_seq}"
        cls._store_seq += 1
        if rank == 0:
            uid = (C.c_uint8 * 128)()
            rc = L.wf_comm_unique_id(uid)
            if rc != 0:
                msg = L.wf_last_error().decode()
                store.set(key, b"ERR:" + msg.encode())
                raise capi.WfError(rc, msg)
            store.set(key, b"UID:" + bytes(uid))
        raw = bytes(store.get(key))
        if raw[:4] != b"UID:":
            raise capi.WfError(-32, "rank 0 could not draw an RCCL unique id: " + raw[4:].decode(errors="replace"))
        uid = (C.c_uint8 * 128).from_buffer_copy(raw[4:132])
        h = C.c_void_p()
        capi._check(L.wf_comm_create(ctx._h, uid, rank, world, C.byref(h)))
        return cls(h, ctx)

    @classmethod
    def with_unique_id(cls, ctx, uid: bytes, rank: int, world: int):
        """RCCL communicator from an id the caller already holds (ranks that are threads of one process hand it over in
        memory: `unique_id()` on one thread, this on every thread -- ncclCommInitRank returns once all `world` have called)."""
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(uid[:128])
        capi._check(capi.load().wf_comm_create(ctx._h, buf, rank, world, C.byref(h)))
        return cls(h, ctx)

    @classmethod
    def with_transport(cls, ctx, rank: int, world: int, all_gather, all_to_all):
        """Caller-supplied transport (wf_transport).  `all_gather(mine)` takes this rank's bytes (numpy uint8 [n]) and
        returns everybody's, rank-major [world * n]; `all_to_all(mine)` takes [world * n] (block s is for rank s) and
        returns [world * n] (block s came from rank s).  The device bytes are staged through the host."""
        cbs = transport_callbacks(world, all_gather, all_to_all)
        tr = capi.Transport(None, cbs[0], cbs[1])
        h = C.c_void_p()
        capi._check(capi.load().wf_comm_create_with_transport(ctx._h, C.byref(tr), rank, world, C.byref(h)))
        return cls(h, ctx, keep=(cbs, tr))

    @classmethod
    def with_process_group(cls, ctx, group=None):
        """The same over a torch.distributed group whose backend handles host tensors (gloo).  Rehearsal / bring-up
        of the multi-rank control flow on machines without one GPU per rank."""
        import torch.distributed as dist
        ag, a2a = process_group_collectives(group)
        return cls.with_transport(ctx, dist.get_rank(group), dist.get_world_size(group), ag, a2a)

    def close(self):
        if self._h:
            if capi._alive(self.ctx):
                capi.load().wf_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- collectives (device pointers: e.g. torch.Tensor.data_ptr()) ---------------------------------------------------
    def barrier(self):
        capi._check(capi.load().wf_comm_barrier(self._h))

    def stream_wait(self, stream: int = 0):
        """Blocking wait on `stream` under the communicator's watchdog (WfError -32 instead of a hang)."""
        capi._check(capi.load().wf_comm_stream_wait(self._h, stream or None))

    def max_f64(self, value: float) -> float:
        v = C.c_double(value)
        capi._check(capi.load().wf_comm_max_f64(self._h, C.byref(v)))
        return v.value

    def gather_f64(self, value: float):
        """One double of every rank on every rank, rank-major (wf_comm_gather_f64)."""
        out = (C.c_double * self.world)()
        capi._check(capi.load().wf_comm_gather_f64(self._h, float(value), out))
        return list(out)

    def info(self):
        """What the transport reports about this communicator (wf_comm_info): for RCCL ncclCommCount / ncclCommUserRank /
        ncclCommCuDevice of the ncclComm_t in use."""
        t, n, r, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        capi._check(capi.load().wf_comm_info(self._h, C.byref(t), C.byref(n), C.byref(r), C.byref(d)))
        return {"transport": "rccl" if t.value == 0 else "caller", "count": n.value, "user_rank": r.value, "device": d.value}

    def all_gather_roots(self, d_roots: int, n_roots: int, d_all: int, stream: int = 0):
        """The one collective of the independent-proofs sharding: [n_roots][32] per rank -> [world][n_roots][32]."""
        capi._check(capi.load().wf_comm_all_gather_roots(self._h, d_roots, n_roots, d_all, stream or None))

    def all_gather_leaf_shards(self, d_shard: int, trace_len: int, per_rank: int, d_leaves: int, stream: int = 0):
        capi._check(capi.load().wf_comm_all_gather_leaf_shards(self._h, d_shard, trace_len, per_rank, d_leaves,
                                                               stream or None))

    def trace_commit_sharded_resident(self, params, trace_cols) -> "ShardedCommitment":
        """Collective: host columns (the same on every rank) -> a resident sharded commitment."""
        import numpy as np
        cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in trace_cols]
        h = C.c_void_p()
        capi._check(capi.load().wf_trace_commit_sharded_resident(self._h, C.byref(params), capi._ptr_array(cols), C.byref(h)))
        return ShardedCommitment(h, self, params)

    def trace_commit_sharded_dev(self, params, d_trace: int, d_polys: int, d_lde_shard: int, d_leaves: int,
                                 d_nodes: int, d_top: int, stream: int = 0):
        capi._check(capi.load().wf_trace_commit_sharded_dev(self._h, C.byref(params), d_trace, d_polys or None,
                                                            d_lde_shard, d_leaves, d_nodes, d_top, stream or None))


class ShardedCommitment:
    """wf_sharded_commitment wrapper: one packed commitment resident on the ranks of a communicator."""

    def __init__(self, handle, comm, params):
        self._h, self.comm, self.params = handle, comm, params
        self.n_rows = 1 << (params.log2_trace_len + params.log2_blowup)
        self.depth = params.log2_trace_len + params.log2_blowup
        self.row_elems = params.n_cols * params.ext_degree * params.n_traces
        comm.ctx._adopt(self)  # after its communicator in the context's list: closed before it

    def close(self):
        if self._h:
            if capi._alive(self.comm):
                capi.load().wf_sharded_commitment_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def root(self) -> bytes:
        out = (C.c_uint8 * 32)()
        capi._check(capi.load().wf_sharded_commitment_root(self._h, out))
        return bytes(out)[:self.params.digest_bytes]

    def query(self, positions):
        """Collective: (rows, (leaves, nodes, depth)) -- the answer of Commitment.query on the unsharded commitment."""
        import numpy as np
        pos = np.ascontiguousarray(positions, dtype=np.uint64)
        n = len(pos)
        w = capi.ELEM_WORDS[self.params.field]
        rows = np.empty((n, self.row_elems, w) if w > 1 else (n, self.row_elems), dtype=np.uint64)
        cap = max(1, n) * (self.depth + 1)
        leaves = np.empty((max(1, n), self.params.digest_bytes), dtype=np.uint8)
        nodes = np.empty((cap, self.params.digest_bytes), dtype=np.uint8)
        counts = np.zeros(max(1, n), dtype=np.uint32)
        n_vec, n_nodes, depth = C.c_size_t(), C.c_size_t(), C.c_uint32()
        capi._check(capi.load().wf_sharded_commitment_query(self._h, capi._p(pos), n, capi._p(rows), capi._p(leaves),
                                                            capi._p(nodes), cap, capi._p(counts), C.byref(n_vec),
                                                            C.byref(n_nodes), C.byref(depth)))
        out, k = [], 0
        for i in range(n_vec.value):
            out.append([bytes(nodes[k + j]) for j in range(int(counts[i]))])
            k += int(counts[i])
        return rows, ([bytes(x) for x in leaves[:n]], out, depth.value)

    def polys(self):
        """The polynomials as a capi.Commitment (evaluate_polys_at); owned by this object."""
        h = C.c_void_p()
        capi._check(capi.load().wf_sharded_commitment_polys(self._h, C.byref(h)))
        return capi.Commitment(h, self.params.field, owned=False, keep_alive=self, digest_bytes=self.params.digest_bytes)


def process_group_collectives(group=None):
    """(all_gather, all_to_all) on numpy byte arrays over a torch.distributed group (host tensors: gloo)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)

    def all_gather(mine):
        out = torch.empty(world * mine.size, dtype=torch.uint8)
        dist.all_gather_into_tensor(out, torch.from_numpy(np.ascontiguousarray(mine)), group=group)
        return out.numpy()

    def all_to_all(mine):
        # gloo has no all_to_all on every build: all-gather everything, keep the blocks addressed to this rank
        n = mine.size // world
        return all_gather(mine).reshape(world, world, n)[:, rank, :].reshape(-1)

    return all_gather, all_to_all


def transport_callbacks(world: int, all_gather, all_to_all, host_memory: bool = False):
    """The two C callbacks of a wf_transport around byte-array collectives.  host_memory: the pointers are host
    pointers (CPU tests of the callbacks themselves); otherwise device pointers, staged with hipMemcpy."""
    import numpy as np

    def fetch(ptr, n, stream):
        buf = np.empty(n, dtype=np.uint8)
        if host_memory:
            C.memmove(buf.ctypes.data, ptr, n)
        else:
            hip = _hip_runtime()
            _hip_check(hip.hipStreamSynchronize(C.c_void_p(stream)))
            _hip_check(hip.hipMemcpy(C.c_void_p(buf.ctypes.data), C.c_void_p(ptr), C.c_size_t(n), 2))  # DeviceToHost
        return buf

    def store(ptr, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        if host_memory:
            C.memmove(ptr, arr.ctypes.data, arr.size)
        else:
            _hip_check(_hip_runtime().hipMemcpy(C.c_void_p(ptr), C.c_void_p(arr.ctypes.data), C.c_size_t(arr.size), 1))  # HostToDevice

    def cb_all_gather(_user, d_send, d_recv, nbytes, stream):
        try:  # never let an exception cross the C boundary
            out = all_gather(fetch(d_send, nbytes, stream))
            assert out.size == world * nbytes
            store(d_recv, out)
            return 0
        except Exception as e:
            print("wf_transport.all_gather:", repr(e), flush=True)
            return 1

    def cb_all_to_all(_user, d_send, d_recv, nbytes, stream):
        try:
            out = all_to_all(fetch(d_send, world * nbytes, stream))
            assert out.size == world * nbytes
            store(d_recv, out)
            return 0
        except Exception as e:
            print("wf_transport.all_to_all:", repr(e), flush=True)
            return 1

    return capi.TRANSPORT_FN(cb_all_gather), capi.TRANSPORT_FN(cb_all_to_all)


_hip = None


def _hip_runtime():
    """The HIP runtime this process already holds (libwf_lde.so is linked against it), for the staging copies of the
    rehearsal transport."""
    global _hip
    if _hip is None:
        capi.load()
        for name in ("libamdhip64.so.7", "libamdhip64.so"):  # already mapped by libwf_lde.so: dlopen returns that copy
            try:
                _hip = C.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise RuntimeError("the HIP runtime (libamdhip64.so) could not be opened")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    return _hip


def _hip_check(rc: int):
    if rc != 0:
        raise RuntimeError(f"HIP runtime call failed with {rc}")
