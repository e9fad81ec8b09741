"""ctypes binding of libwf_lde.so -- the C ABI of include/wf_lde.h, nothing more.

Arrays follow the reference's in-memory element types: f64 = numpy uint64 Montgomery residues; f128 = numpy
uint64 with a trailing dimension of 2 (lo, hi).  There is no fallback: if the library is missing or no HIP device
is present, loading / context creation raises WfError.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import numpy as np

from .build import lib_path

F64, F128 = 1, 2
ELEM_WORDS = {F64: 1, F128: 2}

SYMBOLS = [
    "wf_ctx_create", "wf_ctx_destroy", "wf_last_error", "wf_device_count", "wf_ctx_synchronize", "wf_ctx_stream",
    "wf_ctx_release_cached", "wf_ctx_set_digest_bytes", "wf_plan_digits", "wf_commitment_query", "wf_ctx_profile_enable", "wf_ctx_profile_read", "wf_params_check", "wf_elem_bytes", "wf_row_width", "wf_column_bytes", "wf_lde_bytes", "wf_digests_bytes",
    "wf_trace_commit", "wf_constraint_commit", "wf_trace_commit_dev", "wf_constraint_commit_dev",
    "wf_trace_commit_shard_dev", "wf_merkle_build_dev", "wf_trace_commit_resident", "wf_trace_commit_resident_async", "wf_commitment_wait", "wf_constraint_commit_resident", "wf_commitment_destroy", "wf_commitment_root",
    "wf_commitment_info", "wf_commitment_read_rows", "wf_commitment_read_lde", "wf_commitment_read_lde_strided", "wf_deep_compose", "wf_commitment_evaluate_polys_at_points", "wf_constraint_commit_from_evaluations", "wf_constraint_commit_from_tables", "wf_commitment_query_many", "wf_commitment_prove", "wf_commitment_prove_batch",
    "wf_evaluate_columns_at", "wf_commitment_evaluate_polys_at", "wf_fri_layer_commit", "wf_fri_apply_drp", "wf_fri_layer_commit_dev", "wf_fri_apply_drp_dev",
    "wf_fri_prover_create", "wf_fri_prover_destroy", "wf_fri_num_layers", "wf_fri_prover_begin", "wf_fri_prover_begin_dev", "wf_fri_prover_begin_poly",
    "wf_fri_prover_commit_layer", "wf_fri_prover_fold", "wf_fri_prover_set_remainder", "wf_fri_prover_num_layers",
    "wf_fri_prover_layer", "wf_fri_prover_reset", "wf_fri_fold_positions",
    "wf_fft_evaluate_poly", "wf_fft_evaluate_poly_with_offset", "wf_fft_interpolate_poly",
    "wf_fft_interpolate_poly_with_offset", "wf_evaluate_polys_over", "wf_hash_rows", "wf_merkle_build",
    "wf_comm_unique_id", "wf_comm_create", "wf_comm_create_with_transport", "wf_comm_destroy", "wf_comm_rank",
    "wf_comm_world", "wf_comm_rccl_version", "wf_comm_rccl_path", "wf_comm_stream_wait", "wf_comm_all_gather", "wf_comm_all_gather_roots", "wf_comm_barrier",
    "wf_comm_max_f64", "wf_comm_gather_f64", "wf_comm_info", "wf_shard_proofs", "wf_shard_cosets", "wf_shard_route", "wf_comm_all_gather_leaf_shards",
    "wf_trace_commit_sharded_dev", "wf_trace_commit_sharded_resident", "wf_sharded_commitment_destroy",
    "wf_sharded_commitment_root", "wf_sharded_commitment_query", "wf_sharded_commitment_polys",
]


class WfError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"wf_lde error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [
        ("field", C.c_uint32), ("ext_degree", C.c_uint32), ("log2_trace_len", C.c_uint32),
        ("log2_blowup", C.c_uint32), ("n_cols", C.c_uint32), ("n_traces", C.c_uint32),
        ("digest_bytes", C.c_uint32), ("reserved", C.c_uint32), ("domain_offset", C.c_uint8 * 16),
    ]


TRANSPORT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class Transport(C.Structure):
    """wf_transport: caller-supplied collectives (device pointers)."""
    _fields_ = [("user", C.c_void_p), ("all_gather", TRANSPORT_FN), ("all_to_all", TRANSPORT_FN)]


class Divisor(C.Structure):
    """wf_divisor: (x^a - b) / prod_k (x - e_k), one numerator term (ConstraintDivisor, air/src/air/divisor.rs:26-29)."""
    _fields_ = [("numerator_degree", C.c_uint64), ("numerator_constant", C.c_uint8 * 16), ("exemptions", C.c_void_p),
                ("n_exemptions", C.c_uint32)]


class EvaluationTable(C.Structure):
    """wf_evaluation_table: the columns of one ConstraintEvaluationTable with their divisors."""
    _fields_ = [("columns", C.c_void_p), ("divisors", C.POINTER(Divisor)), ("n_columns", C.c_uint32)]


class Query(C.Structure):
    """wf_query (include/wf_lde.h): one commitment's share of wf_commitment_query_many."""
    _fields_ = [("commitment", C.c_void_p), ("positions", C.c_void_p), ("n", C.c_size_t), ("rows_out", C.c_void_p),
                ("leaves_out", C.c_void_p), ("nodes_out", C.c_void_p), ("nodes_capacity", C.c_size_t),
                ("node_counts", C.c_void_p), ("n_vectors", C.c_size_t), ("n_nodes", C.c_size_t), ("depth", C.c_uint32)]


def make_params(field, ext_degree, log2_trace_len, log2_blowup, n_cols, n_traces=1, offset=None, digest_bytes=32) -> Params:
    """digest_bytes: 32 = Blake3_256, 24 = Blake3_192 (host arrays of digests are then 24 bytes per entry)."""
    if offset is None:
        offset = 7 if field == F64 else 3  # ProofOptions::domain_offset = B::GENERATOR, air/src/options.rs:199-201
    p = Params(field, ext_degree, log2_trace_len, log2_blowup, n_cols, n_traces, digest_bytes, 0)
    p.domain_offset[:] = list(int(offset).to_bytes(16, "little"))
    return p


_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7; a process can only hold one HIP runtime, and it must be
    the one torch was built with if torch is used later in the same process (device buffers, RCCL).  Loading torch's
    copy first makes libwf_lde.so bind to it as well (same SONAME), whatever the import order."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """dlopen libwf_lde.so (raises if it has not been built)."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            # a fresh checkout: compile in-tree if the ROCm toolchain is there (hipcc cross-compiles without a GPU)
            import shutil
            if os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc"):
                from .build import build
                build()
        if not os.path.exists(path):
            raise WfError(-30, f"{path} not built (run __graft_entry__.build() / make -C csrc)")
        _preload_torch_hip_runtime()
        L = C.CDLL(path)
        vp, sz, u32, i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int
        PP = C.POINTER(Params)
        L.wf_last_error.restype = C.c_char_p
        L.wf_ctx_create.argtypes = [i32, C.POINTER(vp)]
        L.wf_ctx_destroy.argtypes = [vp]
        L.wf_ctx_destroy.restype = None
        L.wf_ctx_synchronize.argtypes = [vp]
        L.wf_ctx_stream.argtypes = [vp]
        L.wf_ctx_stream.restype = vp
        L.wf_ctx_profile_enable.argtypes = [vp, i32]
        L.wf_ctx_profile_read.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_float)]
        L.wf_params_check.argtypes = [PP, i32]
        L.wf_elem_bytes.argtypes = [u32]
        L.wf_elem_bytes.restype = sz
        for n in ("wf_row_width", "wf_column_bytes", "wf_lde_bytes", "wf_digests_bytes"):
            getattr(L, n).argtypes = [PP]
            getattr(L, n).restype = sz
        L.wf_trace_commit.argtypes = [vp, PP, vp, vp, vp, vp, vp, vp]
        L.wf_constraint_commit.argtypes = [vp, PP, vp, vp, vp, vp, vp]
        L.wf_trace_commit_dev.argtypes = [vp, PP, vp, vp, vp, vp, vp, vp]
        L.wf_constraint_commit_dev.argtypes = [vp, PP, vp, vp, vp, vp, vp]
        L.wf_trace_commit_shard_dev.argtypes = [vp, PP, u32, u32, vp, vp, vp, vp, vp]
        L.wf_merkle_build_dev.argtypes = [vp, vp, sz, vp, vp]
        L.wf_trace_commit_resident.argtypes = [vp, PP, vp, vp, C.POINTER(vp)]
        L.wf_constraint_commit_resident.argtypes = [vp, PP, vp, C.POINTER(vp)]
        L.wf_commitment_destroy.argtypes = [vp]
        L.wf_commitment_destroy.restype = None
        L.wf_commitment_root.argtypes = [vp, vp]
        L.wf_commitment_info.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(u32)]
        L.wf_commitment_read_rows.argtypes = [vp, vp, sz, vp]
        L.wf_commitment_read_lde.argtypes = [vp, u32, C.c_uint64, C.c_uint64, vp, C.POINTER(C.c_uint64)]
        L.wf_commitment_read_lde_strided.argtypes = [vp, u32, C.c_uint64, C.c_uint64, C.c_uint64, vp, C.POINTER(C.c_uint64)]
        L.wf_deep_compose.argtypes = [vp, vp, C.c_size_t, vp, vp, u32, vp, vp, vp, vp, C.c_size_t]
        L.wf_commitment_evaluate_polys_at_points.argtypes = [vp, vp, u32, u32, vp]
        L.wf_commitment_query_many.argtypes = [C.POINTER(Query), C.c_size_t]
        L.wf_constraint_commit_from_evaluations.argtypes = [vp, C.POINTER(Params), vp, C.c_size_t, C.c_size_t, vp, vp, vp]
        L.wf_constraint_commit_from_tables.argtypes = [vp, C.POINTER(Params), C.POINTER(EvaluationTable), C.c_size_t, C.c_size_t, vp, vp, vp]
        L.wf_commitment_prove.argtypes = [vp, C.c_uint64, vp]
        L.wf_commitment_prove_batch.argtypes = [vp, vp, sz, vp, vp, sz, vp, C.POINTER(sz), C.POINTER(sz),
                                                C.POINTER(u32)]
        L.wf_evaluate_columns_at.argtypes = [vp, u32, u32, vp, sz, sz, vp, u32, vp]
        L.wf_commitment_evaluate_polys_at.argtypes = [vp, vp, u32, vp]
        L.wf_fri_layer_commit.argtypes = [vp, u32, u32, vp, sz, u32, vp, vp, vp, vp]
        L.wf_fri_apply_drp.argtypes = [vp, u32, u32, vp, sz, u32, vp, vp, vp]
        L.wf_fri_layer_commit_dev.argtypes = [vp, u32, u32, vp, sz, u32, vp, vp, vp, vp]
        L.wf_fri_apply_drp_dev.argtypes = [vp, u32, u32, vp, sz, u32, vp, vp, vp, vp]
        L.wf_fri_prover_create.argtypes = [vp, u32, u32, u32, u32, u32, vp, C.POINTER(vp)]
        L.wf_fri_prover_destroy.argtypes = [vp]
        L.wf_fri_prover_destroy.restype = None
        L.wf_fri_num_layers.argtypes = [u32, u32, u32, sz]
        L.wf_fri_num_layers.restype = sz
        L.wf_fri_prover_begin.argtypes = [vp, vp, sz]
        L.wf_fri_prover_begin_dev.argtypes = [vp, vp, sz, vp]
        L.wf_fri_prover_begin_poly.argtypes = [vp, vp, sz, sz]
        L.wf_fri_prover_commit_layer.argtypes = [vp, vp]
        L.wf_fri_prover_fold.argtypes = [vp, vp]
        L.wf_fri_prover_set_remainder.argtypes = [vp, vp, sz, C.POINTER(sz), vp]
        L.wf_fri_prover_num_layers.argtypes = [vp]
        L.wf_fri_prover_num_layers.restype = sz
        L.wf_fri_prover_layer.argtypes = [vp, sz, C.POINTER(vp)]
        L.wf_fri_prover_reset.argtypes = [vp]
        L.wf_fri_fold_positions.argtypes = [vp, sz, sz, u32, vp, C.POINTER(sz)]
        L.wf_fft_evaluate_poly.argtypes = [vp, u32, u32, vp, sz]
        L.wf_fft_interpolate_poly.argtypes = [vp, u32, u32, vp, sz]
        L.wf_fft_interpolate_poly_with_offset.argtypes = [vp, u32, u32, vp, sz, vp]
        L.wf_fft_evaluate_poly_with_offset.argtypes = [vp, u32, u32, vp, sz, vp, sz, vp]
        L.wf_evaluate_polys_over.argtypes = [vp, PP, vp, vp]
        L.wf_hash_rows.argtypes = [vp, u32, vp, sz, sz, vp]
        L.wf_merkle_build.argtypes = [vp, vp, sz, vp]
        L.wf_device_count.argtypes = []
        L.wf_ctx_release_cached.argtypes = [vp]
        L.wf_ctx_set_digest_bytes.argtypes = [vp, u32]
        L.wf_plan_digits.argtypes = [u32, u32, u32, C.POINTER(u32)]
        L.wf_commitment_query.argtypes = [vp, vp, sz, vp, vp, vp, sz, vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(u32)]
        pu32, pu64 = C.POINTER(u32), C.POINTER(C.c_uint64)
        L.wf_comm_unique_id.argtypes = [vp]
        L.wf_comm_create.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
        L.wf_comm_create_with_transport.argtypes = [vp, C.POINTER(Transport), i32, i32, C.POINTER(vp)]
        L.wf_comm_destroy.argtypes = [vp]
        L.wf_comm_destroy.restype = None
        L.wf_comm_rank.argtypes = [vp]
        L.wf_comm_world.argtypes = [vp]
        L.wf_comm_rccl_version.argtypes = []
        L.wf_trace_commit_resident_async.argtypes = [vp, PP, vp, C.POINTER(vp)]
        L.wf_commitment_wait.argtypes = [vp]
        L.wf_comm_rccl_path.argtypes = []
        L.wf_comm_rccl_path.restype = C.c_char_p
        L.wf_comm_stream_wait.argtypes = [vp, vp]
        L.wf_comm_all_gather.argtypes = [vp, vp, vp, sz, vp]
        L.wf_comm_all_gather_roots.argtypes = [vp, vp, sz, vp, vp]
        L.wf_comm_barrier.argtypes = [vp]
        L.wf_comm_max_f64.argtypes = [vp, C.POINTER(C.c_double)]
        L.wf_comm_gather_f64.argtypes = [vp, C.c_double, C.POINTER(C.c_double)]
        L.wf_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        L.wf_shard_proofs.argtypes = [u32, u32, u32, pu32, pu32]
        L.wf_shard_cosets.argtypes = [u32, u32, u32, pu32, pu32]
        L.wf_shard_route.argtypes = [u32, u32, u32, C.c_uint64, pu32, pu64, pu32, pu64]
        L.wf_comm_all_gather_leaf_shards.argtypes = [vp, vp, sz, u32, vp, vp]
        L.wf_trace_commit_sharded_dev.argtypes = [vp, PP, vp, vp, vp, vp, vp, vp, vp]
        L.wf_trace_commit_sharded_resident.argtypes = [vp, PP, vp, C.POINTER(vp)]
        L.wf_sharded_commitment_destroy.argtypes = [vp]
        L.wf_sharded_commitment_destroy.restype = None
        L.wf_sharded_commitment_root.argtypes = [vp, vp]
        L.wf_sharded_commitment_query.argtypes = [vp, vp, sz, vp, vp, vp, sz, vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(u32)]
        L.wf_sharded_commitment_polys.argtypes = [vp, C.POINTER(vp)]
        _lib = L
    return _lib


def _check(rc: int):
    if rc != 0:
        raise WfError(rc, load().wf_last_error().decode())


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _ptr_array(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def _off16(offset: int):
    return (C.c_uint8 * 16).from_buffer_copy(int(offset).to_bytes(16, "little"))


def _alive(owner) -> bool:
    """Is the object a handle lives in (a Context, or something that itself lives in one) still open?"""
    seen = 0
    while owner is not None and seen < 4:
        if isinstance(owner, Context):
            return bool(owner._h)
        if not getattr(owner, "_h", True):
            return False
        owner = getattr(owner, "_ctx", None) or getattr(owner, "ctx", None) or getattr(owner, "_keep_alive", None) or getattr(owner, "comm", None)
        seen += 1
    return True


def device_count() -> int:
    return load().wf_device_count()


class Context:
    """wf_ctx wrapper: one per GPU."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        _check(load().wf_ctx_create(device, C.byref(self._h)))
        self.device = device
        self.digest_bytes = 32  # hasher of the entry points without wf_params (set_digest_bytes)
        # Handles created on this context (commitments, FRI provers, communicators): the C ABI wants them destroyed
        # before it.  References from the children keep the context alive in normal operation, but the interpreter's
        # final garbage collection runs the finalisers of a dead cycle in no particular order -- so close() takes the
        # children down first, newest first, and their own close() afterwards is a no-op.
        self._children = []

    def _adopt(self, child):
        self._children.append(weakref.ref(child))
        if len(self._children) > 64:
            self._children = [r for r in self._children if r() is not None]

    def set_digest_bytes(self, n: int):
        """wf_ctx_set_digest_bytes: 32 = Blake3_256 (default), 24 = Blake3_192 for hash_rows / merkle_build / the FRI entry points."""
        _check(load().wf_ctx_set_digest_bytes(self._h, n))
        self.digest_bytes = n

    def release_cached(self):
        """Return the parked buffers of destroyed resident commitments to the driver (wf_ctx_release_cached)."""
        _check(load().wf_ctx_release_cached(self._h))

    def close(self):
        if self._h:
            for ref in reversed(self._children):
                child = ref()
                if child is not None:
                    child.close()
            self._children = []
            load().wf_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(load().wf_ctx_synchronize(self._h))

    def profile_enable(self, level: int = 1):
        """0 off, 1 one event per logical kernel, 2 one event per launch."""
        _check(load().wf_ctx_profile_enable(self._h, int(level)))

    def profile_read(self):
        """[(launch name, milliseconds)] of the *_commit_dev calls since the last read (waits for them)."""
        cap = 16384
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        n = load().wf_ctx_profile_read(self._h, cap, names, ms)
        if n < 0:
            _check(n)
        return [(names[i].decode(), float(ms[i])) for i in range(n)]

    @property
    def stream(self) -> int:
        return load().wf_ctx_stream(self._h) or 0

    # -- the path, host buffers ------------------------------------------------------------------------------
    def trace_commit(self, params: Params, trace_cols, want_lde=True, want_polys=True):
        """trace_cols: flat list [trace][col] of column arrays.  Returns dict(polys, lde, leaves, nodes, root)."""
        L = load()
        _check(L.wf_params_check(C.byref(params), 0))
        w = ELEM_WORDS[params.field]
        cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in trace_cols]
        n_rows = 1 << (params.log2_trace_len + params.log2_blowup)
        rw = L.wf_row_width(C.byref(params))
        polys = [np.empty_like(c) for c in cols] if want_polys else None
        lde = ([np.empty((n_rows, rw, w) if w > 1 else (n_rows, rw), dtype=np.uint64) for _ in range(params.n_traces)]
               if want_lde else None)
        db = params.digest_bytes
        leaves = np.empty((n_rows, db), dtype=np.uint8)
        nodes = np.empty((n_rows, db), dtype=np.uint8)
        root = np.empty(32, dtype=np.uint8)
        _check(L.wf_trace_commit(self._h, C.byref(params), _ptr_array(cols),
                                 _ptr_array(polys) if polys else None, _ptr_array(lde) if lde else None,
                                 _p(leaves), _p(nodes), _p(root)))
        return dict(polys=polys, lde=lde, leaves=leaves, nodes=nodes, root=bytes(root)[:db])

    def constraint_commit(self, params: Params, poly_cols, want_lde=True):
        L = load()
        _check(L.wf_params_check(C.byref(params), 1))
        w = ELEM_WORDS[params.field]
        cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in poly_cols]
        n_rows = 1 << (params.log2_trace_len + params.log2_blowup)
        rw = L.wf_row_width(C.byref(params))
        lde = np.empty((n_rows, rw, w) if w > 1 else (n_rows, rw), dtype=np.uint64) if want_lde else None
        db = params.digest_bytes
        leaves = np.empty((n_rows, db), dtype=np.uint8)
        nodes = np.empty((n_rows, db), dtype=np.uint8)
        root = np.empty(32, dtype=np.uint8)
        _check(L.wf_constraint_commit(self._h, C.byref(params), _ptr_array(cols), _p(lde), _p(leaves), _p(nodes),
                                      _p(root)))
        return dict(lde=lde, leaves=leaves, nodes=nodes, root=bytes(root)[:db])

    # -- the path, device buffers (raw device addresses, e.g. torch.Tensor.data_ptr()) ---------------------------
    def trace_commit_dev(self, params: Params, d_trace: int, d_polys: int, d_lde: int, d_leaves: int, d_nodes: int,
                         stream: int = 0):
        _check(load().wf_trace_commit_dev(self._h, C.byref(params), d_trace, d_polys, d_lde, d_leaves, d_nodes,
                                          stream or None))

    def constraint_commit_dev(self, params: Params, d_polys: int, d_lde: int, d_leaves: int, d_nodes: int,
                              stream: int = 0):
        _check(load().wf_constraint_commit_dev(self._h, C.byref(params), d_polys, d_lde, d_leaves, d_nodes,
                                               stream or None))

    def trace_commit_shard_dev(self, params: Params, coset_begin: int, coset_count: int, d_trace: int, d_polys: int,
                               d_lde_shard: int, d_leaves_shard: int, stream: int = 0):
        _check(load().wf_trace_commit_shard_dev(self._h, C.byref(params), coset_begin, coset_count, d_trace,
                                                d_polys or None, d_lde_shard, d_leaves_shard, stream or None))

    def merkle_build_dev(self, d_leaves: int, n_leaves: int, d_nodes: int, stream: int = 0):
        _check(load().wf_merkle_build_dev(self._h, d_leaves, n_leaves, d_nodes, stream or None))

    # -- the path, resident form -------------------------------------------------------------------------------------
    def trace_commit_resident(self, params: Params, trace_cols, want_polys=False):
        L = load()
        _check(L.wf_params_check(C.byref(params), 0))
        cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in trace_cols]
        polys = [np.empty_like(c) for c in cols] if want_polys else None
        h = C.c_void_p()
        _check(L.wf_trace_commit_resident(self._h, C.byref(params), _ptr_array(cols),
                                          _ptr_array(polys) if polys else None, C.byref(h)))
        return Commitment(h, params.field, keep_alive=self, digest_bytes=params.digest_bytes), polys

    def trace_commit_resident_async(self, params: Params, trace_cols):
        """wf_trace_commit_resident_async: returns a Commitment whose kernels may still run; .wait() / .root() complete it.
        The column arrays are kept alive by the returned object (pinned columns are read by the DMA engine after the
        call has returned)."""
        L = load()
        cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in trace_cols]
        h = C.c_void_p()
        _check(L.wf_trace_commit_resident_async(self._h, C.byref(params), _ptr_array(cols), C.byref(h)))
        com = Commitment(h, params.field, keep_alive=self, digest_bytes=params.digest_bytes)
        com._inputs = cols
        return com

    def trace_commit_resident_batch(self, params: Params, proofs):
        """A stream of proofs: proofs = [columns of proof 0, columns of proof 1, ...] -> their Commitments, all complete.
        Upload of proof k + 1 under the kernels of proof k (wf_trace_commit_resident_async + wf_commitment_wait)."""
        coms = [self.trace_commit_resident_async(params, cols) for cols in proofs]
        for c in coms:
            c.wait()
        return coms

    def constraint_commit_resident(self, params: Params, poly_cols):
        L = load()
        _check(L.wf_params_check(C.byref(params), 1))
        cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in poly_cols]
        h = C.c_void_p()
        _check(L.wf_constraint_commit_resident(self._h, C.byref(params), _ptr_array(cols), C.byref(h)))
        return Commitment(h, params.field, keep_alive=self, digest_bytes=params.digest_bytes)

    def constraint_commit_from_evaluations(self, params: Params, combined_evaluations, final_coeff=None, want_polys=False):
        """wf_constraint_commit_from_evaluations: [n_tables] combined constraint evaluations over the constraint evaluation
        domain -> resident constraint commitment (and, on request, the composition polynomial's columns)."""
        L = load()
        _check(L.wf_params_check(C.byref(params), 1))
        tabs = [np.ascontiguousarray(t, dtype=np.uint64) for t in combined_evaluations]
        w = ELEM_WORDS[params.field]
        ce = tabs[0].size // (w * params.ext_degree)
        fc = np.ascontiguousarray(final_coeff, dtype=np.uint64) if final_coeff is not None else None
        R = 1 << params.log2_trace_len
        polys = None
        if want_polys:
            shape = (R * params.ext_degree, w) if w > 1 else (R * params.ext_degree,)
            polys = [np.empty(shape, dtype=np.uint64) for _ in range(params.n_cols)]
        h = C.c_void_p()
        _check(L.wf_constraint_commit_from_evaluations(self._h, C.byref(params), _ptr_array(tabs), len(tabs), ce,
                                                       _p(fc) if fc is not None else None,
                                                       _ptr_array(polys) if polys else None, C.byref(h)))
        return Commitment(h, params.field, keep_alive=self, digest_bytes=params.digest_bytes), polys

    def constraint_commit_from_tables(self, params: Params, tables, final_coeff=None, want_polys=False):
        """wf_constraint_commit_from_tables: tables = [[(column, (a, b, exemptions)), ..] per packed trace] -- the whole of
        ConstraintEvaluationTable::into_comb_poly, the final_coeff combination and the commitment, resident."""
        L = load()
        _check(L.wf_params_check(C.byref(params), 1))
        w = ELEM_WORDS[params.field]
        keep, c_tables = [], (EvaluationTable * len(tables))()
        ce = None
        for t, table in zip(c_tables, tables):
            cols = [np.ascontiguousarray(c, dtype=np.uint64) for c, _ in table]
            ce = cols[0].size // (w * params.ext_degree)
            divs = (Divisor * len(table))()
            for d, (_, (a, b, ex)) in zip(divs, table):
                d.numerator_degree = a
                bb = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1)
                raw = bb.tobytes().ljust(16, b"\0")
                d.numerator_constant[:] = list(raw[:16])
                if ex is not None and len(ex):
                    e = np.ascontiguousarray(ex, dtype=np.uint64)
                    keep.append(e)
                    d.exemptions = e.ctypes.data
                    d.n_exemptions = e.size // w
            ptrs = _ptr_array(cols)
            keep += [cols, divs, ptrs]
            t.columns = C.cast(ptrs, C.c_void_p)
            t.divisors = divs
            t.n_columns = len(table)
        fc = np.ascontiguousarray(final_coeff, dtype=np.uint64) if final_coeff is not None else None
        R = 1 << params.log2_trace_len
        polys = None
        if want_polys:
            shape = (R * params.ext_degree, w) if w > 1 else (R * params.ext_degree,)
            polys = [np.empty(shape, dtype=np.uint64) for _ in range(params.n_cols)]
        h = C.c_void_p()
        _check(L.wf_constraint_commit_from_tables(self._h, C.byref(params), c_tables, len(tables), ce,
                                                  _p(fc) if fc is not None else None, _ptr_array(polys) if polys else None,
                                                  C.byref(h)))
        return Commitment(h, params.field, keep_alive=self, digest_bytes=params.digest_bytes), polys

    def deep_compose(self, field, ext, n, trace_commitments, constraint_commitment, z, trace_coeffs, constraint_coeffs=None,
                     want_poly=True, fri: "FriProver" = None, lde_blowup: int = 0):
        """DeepCompositionPoly::add_trace_polys + add_composition_poly (prover/src/composer/mod.rs:62-193) on resident
        commitments; returns the n coefficients of E (or None), and starts `fri` from the polynomial's LDE when given."""
        w = ELEM_WORDS[field]
        handles = (C.c_void_p * len(trace_commitments))(*[c._h for c in trace_commitments])
        zz = np.ascontiguousarray(z, dtype=np.uint64)
        cct = np.ascontiguousarray(trace_coeffs, dtype=np.uint64)
        ccc = np.ascontiguousarray(constraint_coeffs, dtype=np.uint64) if constraint_coeffs is not None else None
        out = np.empty((n * ext, w) if w > 1 else (n * ext,), dtype=np.uint64) if want_poly else None
        _check(load().wf_deep_compose(self._h, handles, len(trace_commitments),
                                      constraint_commitment._h if constraint_commitment is not None else None, _p(zz), ext,
                                      _p(cct), _p(ccc) if ccc is not None else None, _p(out) if out is not None else None,
                                      fri._h if fri is not None else None, lde_blowup))
        return out

    # -- building blocks -----------------------------------------------------------------------------------------
    def fft_evaluate_poly(self, field, ext, poly: np.ndarray, inplace: bool = False) -> np.ndarray:
        """inplace=True: transform the caller's (contiguous uint64) array where it lies, as the reference's `&mut [E]` entry point
        does -- no fresh array per call (a fresh 32 MiB numpy allocation alone costs ~3 ms of page faults: glibc mmaps it)."""
        a = np.ascontiguousarray(poly, dtype=np.uint64)
        if not inplace or a is not poly:
            a = a.copy()
        n = a.size // (ELEM_WORDS[field] * ext)
        _check(load().wf_fft_evaluate_poly(self._h, field, ext, _p(a), n))
        return a

    def fft_interpolate_poly(self, field, ext, evals: np.ndarray, inplace: bool = False) -> np.ndarray:
        """inplace=True: transform the caller's (contiguous uint64) array where it lies, as the reference's `&mut [E]` entry point
        does -- no fresh array per call (a fresh 32 MiB numpy allocation alone costs ~3 ms of page faults: glibc mmaps it)."""
        a = np.ascontiguousarray(evals, dtype=np.uint64)
        if not inplace or a is not evals:
            a = a.copy()
        n = a.size // (ELEM_WORDS[field] * ext)
        _check(load().wf_fft_interpolate_poly(self._h, field, ext, _p(a), n))
        return a

    def fft_interpolate_poly_with_offset(self, field, ext, evals: np.ndarray, offset: int, inplace: bool = False) -> np.ndarray:
        a = np.ascontiguousarray(evals, dtype=np.uint64)
        if not inplace or a is not evals:
            a = a.copy()
        n = a.size // (ELEM_WORDS[field] * ext)
        _check(load().wf_fft_interpolate_poly_with_offset(self._h, field, ext, _p(a), n, _off16(offset)))
        return a

    def fft_evaluate_poly_with_offset(self, field, ext, poly: np.ndarray, offset: int, blowup: int, out: np.ndarray = None) -> np.ndarray:
        a = np.ascontiguousarray(poly, dtype=np.uint64)
        w = ELEM_WORDS[field]
        n = a.size // (w * ext)
        if out is None:  # (callers that time the call pass a preallocated result)
            out = np.empty((n * blowup * ext, w) if w > 1 else (n * blowup * ext,), dtype=np.uint64)
        _check(load().wf_fft_evaluate_poly_with_offset(self._h, field, ext, _p(a), n, _off16(offset), blowup, _p(out)))
        return out

    def evaluate_polys_over(self, params: Params, poly_cols) -> np.ndarray:
        L = load()
        _check(L.wf_params_check(C.byref(params), 1))
        w = ELEM_WORDS[params.field]
        cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in poly_cols]
        n_rows = 1 << (params.log2_trace_len + params.log2_blowup)
        rw = L.wf_row_width(C.byref(params))
        lde = np.empty((n_rows, rw, w) if w > 1 else (n_rows, rw), dtype=np.uint64)
        _check(L.wf_evaluate_polys_over(self._h, C.byref(params), _ptr_array(cols), _p(lde)))
        return lde

    def evaluate_columns_at(self, field, ext, poly_cols, z: np.ndarray, z_ext: int) -> np.ndarray:
        cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in poly_cols]
        w = ELEM_WORDS[field]
        n = cols[0].size // (w * ext)
        zz = np.ascontiguousarray(z, dtype=np.uint64)
        out = np.empty((len(cols), z_ext, w) if w > 1 else (len(cols), z_ext), dtype=np.uint64)
        _check(load().wf_evaluate_columns_at(self._h, field, ext, _ptr_array(cols), len(cols), n, _p(zz), z_ext, _p(out)))
        return out

    def fri_layer_commit(self, field, ext, evals: np.ndarray, folding: int):
        a = np.ascontiguousarray(evals, dtype=np.uint64)
        n = a.size // (ELEM_WORDS[field] * ext)
        rows = max(1, n // max(1, folding))
        tr = np.empty_like(a)
        leaves = np.empty((rows, self.digest_bytes), dtype=np.uint8)
        nodes = np.empty((rows, self.digest_bytes), dtype=np.uint8)
        root = np.empty(32, dtype=np.uint8)
        _check(load().wf_fri_layer_commit(self._h, field, ext, _p(a), n, folding, _p(tr), _p(leaves), _p(nodes),
                                          _p(root)))
        return dict(transposed=tr, leaves=leaves, nodes=nodes, root=bytes(root)[:self.digest_bytes])

    def fri_apply_drp(self, field, ext, transposed: np.ndarray, folding: int, offset: int, alpha: np.ndarray):
        a = np.ascontiguousarray(transposed, dtype=np.uint64)
        w = ELEM_WORDS[field]
        rows = a.size // (w * ext * folding)
        al = np.ascontiguousarray(alpha, dtype=np.uint64)
        out = np.empty((rows * ext, w) if w > 1 else (rows * ext,), dtype=np.uint64)
        _check(load().wf_fri_apply_drp(self._h, field, ext, _p(a), rows, folding, _off16(offset), _p(al), _p(out)))
        return out

    def hash_rows(self, field, rows: np.ndarray, n_rows: int, row_elems: int) -> np.ndarray:
        a = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.empty((n_rows, self.digest_bytes), dtype=np.uint8)
        _check(load().wf_hash_rows(self._h, field, _p(a), n_rows, row_elems, _p(out)))
        return out

    def merkle_build(self, leaves: np.ndarray) -> np.ndarray:
        lv = np.ascontiguousarray(leaves, dtype=np.uint8).reshape(-1, self.digest_bytes)
        nodes = np.empty_like(lv)
        _check(load().wf_merkle_build(self._h, _p(lv), lv.shape[0], _p(nodes)))
        return nodes


class Commitment:
    """wf_commitment wrapper: LDE + tree resident in HBM; rows and Merkle proofs are read from there."""

    def __init__(self, handle, field, owned=True, keep_alive=None, digest_bytes=32):
        self._h = handle
        self.field = field
        self.digest_bytes = digest_bytes  # entries of the digest arrays this handle's queries fill (32, or 24 for Blake3_192)
        self._owned = owned  # layers of a FriProver belong to the prover
        self._keep_alive = keep_alive  # the Context (or FriProver) this handle lives in: destroyed after it, never before
        self._inputs = None  # host columns of an asynchronous commitment, until it has completed
        if owned and isinstance(keep_alive, Context):
            keep_alive._adopt(self)
        n_rows, row_elems, depth = C.c_uint64(), C.c_uint64(), C.c_uint32()
        _check(load().wf_commitment_info(self._h, C.byref(n_rows), C.byref(row_elems), C.byref(depth)))
        self.n_rows, self.row_elems, self.depth = n_rows.value, row_elems.value, depth.value

    def close(self):
        if self._h:
            # (a context that is already gone took its scratch and pool with it: touching the handle now would be a use
            # after free -- this happens only when the interpreter's final collection finalises the context first)
            if self._owned and _alive(self._keep_alive):
                load().wf_commitment_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def root(self) -> bytes:
        out = (C.c_uint8 * 32)()
        _check(load().wf_commitment_root(self._h, out))
        self._inputs = None
        return bytes(out)[:self.digest_bytes]

    def wait(self):
        """Completes a commitment made by trace_commit_resident_async (no-op otherwise)."""
        _check(load().wf_commitment_wait(self._h))
        self._inputs = None

    def read_rows(self, positions) -> np.ndarray:
        pos = np.ascontiguousarray(positions, dtype=np.uint64)
        w = ELEM_WORDS[self.field]
        out = np.empty((len(pos), self.row_elems, w) if w > 1 else (len(pos), self.row_elems), dtype=np.uint64)
        _check(load().wf_commitment_read_rows(self._h, _p(pos), len(pos), _p(out)))
        return out

    def read_lde(self, trace: int, row_begin: int, n_rows: int, row_stride: int = 1) -> np.ndarray:
        """Rows row_begin + k * row_stride (k < n_rows) of one trace's matrix as stored (padded rows)."""
        rw = C.c_uint64()
        _check(load().wf_commitment_read_lde(self._h, trace, 0, 0, None, C.byref(rw)))
        w = ELEM_WORDS[self.field]
        out = np.empty((n_rows, rw.value, w) if w > 1 else (n_rows, rw.value), dtype=np.uint64)
        if row_stride == 1:
            _check(load().wf_commitment_read_lde(self._h, trace, row_begin, n_rows, _p(out), C.byref(rw)))
        else:
            _check(load().wf_commitment_read_lde_strided(self._h, trace, row_begin, n_rows, row_stride, _p(out), C.byref(rw)))
        return out

    def evaluate_polys_at(self, z: np.ndarray, z_ext: int, n_cols_total: int) -> np.ndarray:
        zz = np.ascontiguousarray(z, dtype=np.uint64)
        w = ELEM_WORDS[self.field]
        out = np.empty((n_cols_total, z_ext, w) if w > 1 else (n_cols_total, z_ext), dtype=np.uint64)
        _check(load().wf_commitment_evaluate_polys_at(self._h, _p(zz), z_ext, _p(out)))
        return out

    def evaluate_polys_at_points(self, points: np.ndarray, n_points: int, z_ext: int, n_cols_total: int) -> np.ndarray:
        """An out-of-domain frame (z, z g) in one round trip: [n_points][n_cols_total] elements of z's field."""
        zz = np.ascontiguousarray(points, dtype=np.uint64)
        w = ELEM_WORDS[self.field]
        out = np.empty((n_points, n_cols_total, z_ext, w) if w > 1 else (n_points, n_cols_total, z_ext), dtype=np.uint64)
        _check(load().wf_commitment_evaluate_polys_at_points(self._h, _p(zz), n_points, z_ext, _p(out)))
        return out

    def prove(self, index: int):
        out = np.empty((self.depth + 1, self.digest_bytes), dtype=np.uint8)
        _check(load().wf_commitment_prove(self._h, index, _p(out)))
        return [bytes(x) for x in out]

    def prove_batch(self, positions):
        """Returns (leaves, nodes, depth) shaped like the reference's BatchMerkleProof."""
        pos = np.ascontiguousarray(positions, dtype=np.uint64)
        n = len(pos)
        cap = max(1, n) * (self.depth + 1)
        leaves = np.empty((max(1, n), self.digest_bytes), dtype=np.uint8)
        nodes = np.empty((cap, self.digest_bytes), dtype=np.uint8)
        counts = np.zeros(max(1, n), dtype=np.uint32)
        n_vec, n_nodes, depth = C.c_size_t(), C.c_size_t(), C.c_uint32()
        _check(load().wf_commitment_prove_batch(self._h, _p(pos), n, _p(leaves), _p(nodes), cap, _p(counts),
                                                C.byref(n_vec), C.byref(n_nodes), C.byref(depth)))
        out, k = [], 0
        for i in range(n_vec.value):
            out.append([bytes(nodes[k + j]) for j in range(int(counts[i]))])
            k += int(counts[i])
        return [bytes(x) for x in leaves[:n]], out, depth.value

    def query(self, positions):
        """TraceCommitment::query: (rows, (leaves, nodes, depth)) in one round trip (wf_commitment_query)."""
        pos = np.ascontiguousarray(positions, dtype=np.uint64)
        n = len(pos)
        w = ELEM_WORDS[self.field]
        rows = np.empty((n, self.row_elems, w) if w > 1 else (n, self.row_elems), dtype=np.uint64)
        cap = max(1, n) * (self.depth + 1)
        leaves = np.empty((max(1, n), self.digest_bytes), dtype=np.uint8)
        nodes = np.empty((cap, self.digest_bytes), dtype=np.uint8)
        counts = np.zeros(max(1, n), dtype=np.uint32)
        n_vec, n_nodes, depth = C.c_size_t(), C.c_size_t(), C.c_uint32()
        _check(load().wf_commitment_query(self._h, _p(pos), n, _p(rows), _p(leaves), _p(nodes), cap, _p(counts),
                                          C.byref(n_vec), C.byref(n_nodes), C.byref(depth)))
        out, k = [], 0
        for i in range(n_vec.value):
            out.append([bytes(nodes[k + j]) for j in range(int(counts[i]))])
            k += int(counts[i])
        return rows, ([bytes(x) for x in leaves[:n]], out, depth.value)


def query_many(requests, parse=True):
    """wf_commitment_query_many: [(commitment, positions, want_rows)] of one context answered in one host round trip;
    returns [(rows or None, (leaves, nodes, depth))] like Commitment.query / prove_batch (parse=False: the raw output
    arrays, for timing the call without the Python-side unpacking)."""
    qs = (Query * len(requests))()
    keep = []
    for q, (com, positions, want_rows) in zip(qs, requests):
        pos = np.ascontiguousarray(positions, dtype=np.uint64)
        n = len(pos)
        w = ELEM_WORDS[com.field]
        rows = np.empty((n, com.row_elems, w) if w > 1 else (n, com.row_elems), dtype=np.uint64) if want_rows else None
        cap = max(1, n) * (com.depth + 1)
        leaves = np.empty((max(1, n), com.digest_bytes), dtype=np.uint8)
        nodes = np.empty((cap, com.digest_bytes), dtype=np.uint8)
        counts = np.zeros(max(1, n), dtype=np.uint32)
        q.commitment, q.positions, q.n = com._h, _p(pos), n
        q.rows_out = _p(rows) if rows is not None else None
        q.leaves_out, q.nodes_out, q.nodes_capacity, q.node_counts = _p(leaves), _p(nodes), cap, _p(counts)
        keep.append((pos, rows, leaves, nodes, counts))
    _check(load().wf_commitment_query_many(qs, len(requests)))
    if not parse:
        return keep
    out = []
    for q, (pos, rows, leaves, nodes, counts) in zip(qs, keep):
        vecs, k = [], 0
        for i in range(q.n_vectors):
            vecs.append([bytes(nodes[k + j]) for j in range(int(counts[i]))])
            k += int(counts[i])
        out.append((rows, ([bytes(x) for x in leaves[:len(pos)]], vecs, int(q.depth))))
    return out


def plan_digits(field: int, log2_n: int, n_segments: int = 1):
    """wf_plan_digits: the digit passes of a 2^log2_n-row transform (needs no GPU)."""
    out = (C.c_uint32 * 4)()
    n = load().wf_plan_digits(field, log2_n, n_segments, out)
    _check(n if n < 0 else 0)
    return [int(out[i]) for i in range(n)]


def fri_num_layers(folding: int, blowup: int, remainder_max_degree: int, domain_size: int) -> int:
    return int(load().wf_fri_num_layers(folding, blowup, remainder_max_degree, domain_size))


def fri_fold_positions(positions, source_domain_size: int, folding: int) -> np.ndarray:
    pos = np.ascontiguousarray(positions, dtype=np.uint64)
    out = np.empty(max(1, len(pos)), dtype=np.uint64)
    n_out = C.c_size_t()
    _check(load().wf_fri_fold_positions(_p(pos), len(pos), source_domain_size, folding, _p(out), C.byref(n_out)))
    return out[:n_out.value].copy()


class FriProver:
    """wf_fri_prover wrapper: the commit phase of FriProver (fri/src/prover/mod.rs) with everything resident in HBM."""

    def __init__(self, ctx: "Context", field, ext, folding, blowup, remainder_max_degree, offset: int):
        self.field, self.ext, self.folding, self.blowup = field, ext, folding, blowup
        self._ctx = ctx  # the prover uses its context until it is destroyed: keep it alive (interpreter exit order)
        self._h = C.c_void_p()
        _check(load().wf_fri_prover_create(ctx._h, field, ext, folding, blowup, remainder_max_degree, _off16(offset),
                                           C.byref(self._h)))
        ctx._adopt(self)

    def close(self):
        if self._h:
            if _alive(self._ctx):
                load().wf_fri_prover_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def begin(self, evals: np.ndarray):
        a = np.ascontiguousarray(evals, dtype=np.uint64)
        n = a.size // (self.ext * ELEM_WORDS[self.field])
        _check(load().wf_fri_prover_begin(self._h, _p(a), n))

    def begin_poly(self, poly: np.ndarray, lde_blowup: int):
        """DEEP composition polynomial (coefficients) -> its LDE evaluations on the device -> first layer."""
        a = np.ascontiguousarray(poly, dtype=np.uint64)
        n = a.size // (self.ext * ELEM_WORDS[self.field])
        _check(load().wf_fri_prover_begin_poly(self._h, _p(a), n, lde_blowup))

    def begin_dev(self, d_ptr: int, n: int, stream: int = 0):
        _check(load().wf_fri_prover_begin_dev(self._h, d_ptr, n, stream))

    def commit_layer(self) -> bytes:
        out = (C.c_uint8 * 32)()
        _check(load().wf_fri_prover_commit_layer(self._h, out))
        return bytes(out)[:self._ctx.digest_bytes]

    def fold(self, alpha: np.ndarray):
        al = np.ascontiguousarray(alpha, dtype=np.uint64)
        _check(load().wf_fri_prover_fold(self._h, _p(al)))

    def set_remainder(self, capacity: int):
        w = ELEM_WORDS[self.field]
        out = np.empty((capacity, self.ext, w) if w > 1 else (capacity, self.ext), dtype=np.uint64)
        n = C.c_size_t()
        digest = (C.c_uint8 * 32)()
        _check(load().wf_fri_prover_set_remainder(self._h, _p(out), capacity, C.byref(n), digest))
        return out[:n.value].copy(), bytes(digest)[:self._ctx.digest_bytes]

    def num_layers(self) -> int:
        return int(load().wf_fri_prover_num_layers(self._h))

    def layer(self, i: int) -> Commitment:
        h = C.c_void_p()
        _check(load().wf_fri_prover_layer(self._h, i, C.byref(h)))
        return Commitment(h, self.field, owned=False, keep_alive=self, digest_bytes=self._ctx.digest_bytes)

    def reset(self):
        _check(load().wf_fri_prover_reset(self._h))

