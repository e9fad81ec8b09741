"""ORACLE (test infrastructure, NOT product code): ctypes/numpy front end of oracle/liboracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
f64 arrays are numpy uint64 (Montgomery residues, as the reference keeps them in memory);
f128 arrays are numpy uint64 with a trailing dimension of 2 (lo, hi words of the canonical u128).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

F64, F128 = 1, 2
ELEM_WORDS = {F64: 1, F128: 2}


def build(force: bool = False) -> str:
    """Compile oracle/liboracle.so with gcc (oracle/Makefile)."""
    import shutil
    if shutil.which("make") and shutil.which("gcc"):  # make is a no-op when the library is up to date
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []) + ["liboracle.so"],
                              stdout=subprocess.DEVNULL)
    elif not os.path.exists(_LIB_PATH):
        raise RuntimeError("oracle/liboracle.so is missing and there is no gcc/make to build it")
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u64, sz, vp, i32, u32 = C.c_uint64, C.c_size_t, C.c_void_p, C.c_int, C.c_uint32
        for name in ("new", "as_int", "inv"):
            getattr(L, f"orc_f64_{name}").restype = u64
            getattr(L, f"orc_f64_{name}").argtypes = [u64]
        for name in ("add", "sub", "mul", "exp"):
            getattr(L, f"orc_f64_{name}").restype = u64
            getattr(L, f"orc_f64_{name}").argtypes = [u64, u64]
        L.orc_f64_get_root_of_unity.restype = u64
        L.orc_f64_get_root_of_unity.argtypes = [u32]
        for name in ("add", "sub", "mul"):
            getattr(L, f"orc_f128_{name}").argtypes = [vp, vp, vp]
        L.orc_f128_inv.argtypes = [vp, vp]
        L.orc_f128_get_root_of_unity.argtypes = [u32, vp]
        L.orc_f64_get_twiddles.argtypes = [vp, sz, i32]
        L.orc_f128_get_twiddles_p.argtypes = [vp, sz, i32]
        for f in ("f64", "f128"):
            getattr(L, f"orc_{f}_permute").argtypes = [vp, sz, sz]
            getattr(L, f"orc_{f}_fft_in_place").argtypes = [vp, sz, sz, vp]
            getattr(L, f"orc_{f}_evaluate_poly").argtypes = [vp, sz, sz, vp]
            getattr(L, f"orc_{f}_interpolate_poly").argtypes = [vp, sz, sz, vp]
            getattr(L, f"orc_{f}_eval_many").argtypes = [vp, sz, vp, sz, vp]
            getattr(L, f"orc_{f}_interpolate_columns").argtypes = [vp, sz, sz, sz, vp, i32]
        L.orc_f64_evaluate_poly_with_offset.argtypes = [vp, sz, sz, vp, u64, sz, vp]
        L.orc_f64_interpolate_poly_with_offset.argtypes = [vp, sz, sz, vp, u64]
        L.orc_f64_evaluate_polys_over.argtypes = [vp, sz, sz, sz, sz, u64, vp, i32]
        L.orc_f128_evaluate_poly_with_offset_p.argtypes = [vp, sz, sz, vp, vp, sz, vp]
        L.orc_f128_interpolate_poly_with_offset_p.argtypes = [vp, sz, sz, vp, vp]
        L.orc_f128_evaluate_polys_over_p.argtypes = [vp, sz, sz, sz, sz, vp, vp, i32]
        L.orc_set_digest_bytes.argtypes = [i32]
        L.orc_hash_elements.argtypes = [i32, vp, sz, vp]
        L.orc_merge.argtypes = [vp, vp, vp]
        L.orc_merge_with_int.argtypes = [vp, u64, vp]
        L.orc_build_merkle_nodes.argtypes = [vp, sz, vp, i32]
        L.orc_commit_to_comb_rows.argtypes = [i32, vp, sz, sz, sz, sz, vp, vp, i32]
        L.orc_build_trace_commitment.argtypes = [i32, sz, C.c_uint, C.c_uint, sz, sz, vp, vp, vp, vp, vp, vp, i32]
        L.orc_build_constraint_commitment.argtypes = [i32, sz, C.c_uint, C.c_uint, sz, vp, vp, vp, vp, vp, i32]
        L.orc_blake3_hash.argtypes = [vp, sz, vp]
        L.orc_eval_column_at.argtypes = [i32, vp, sz, sz, vp, sz, vp]
        L.orc_ext_mul.argtypes = [i32, sz, vp, vp, vp]
        L.orc_syn_div.argtypes = [i32, sz, vp, sz, vp]
        L.orc_syn_div.restype = None
        L.orc_acc_column.argtypes = [i32, vp, sz, sz, sz, vp, vp, sz, vp, vp]
        L.orc_scale_acc.argtypes = [i32, vp, vp, sz, sz, vp, sz]
        L.orc_deep_compose.argtypes = [i32, sz, sz, sz, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp, vp]
        L.orc_transpose_slice.argtypes = [i32, vp, sz, sz, sz, vp]
        L.orc_apply_drp.argtypes = [i32, vp, sz, sz, sz, vp, vp, vp, i32]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _ptr_array(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def _off_bytes(offset: int) -> np.ndarray:
    return np.frombuffer(int(offset).to_bytes(16, "little"), dtype=np.uint8).copy()


# ----------------------------------------------------------------------------------------------- field helpers

def f64_new(values) -> np.ndarray:
    """canonical ints (array-like of < 2^64) -> Montgomery uint64 array."""
    L = lib()
    v = np.asarray(values, dtype=np.uint64)
    out = np.empty_like(v)
    flat_in, flat_out = v.reshape(-1), out.reshape(-1)
    for i in range(flat_in.size):
        flat_out[i] = L.orc_f64_new(int(flat_in[i]))
    return out


def f64_as_int(values) -> np.ndarray:
    L = lib()
    v = np.asarray(values, dtype=np.uint64)
    out = np.empty_like(v)
    flat_in, flat_out = v.reshape(-1), out.reshape(-1)
    for i in range(flat_in.size):
        flat_out[i] = L.orc_f64_as_int(int(flat_in[i]))
    return out


def f128_from_ints(values) -> np.ndarray:
    vals = list(values)
    out = np.empty((len(vals), 2), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i, 0] = v & 0xFFFFFFFFFFFFFFFF
        out[i, 1] = v >> 64
    return out


def f128_to_ints(arr: np.ndarray):
    a = np.asarray(arr, dtype=np.uint64).reshape(-1, 2)
    return [int(a[i, 0]) | (int(a[i, 1]) << 64) for i in range(a.shape[0])]


def f128_op(name: str, a: int, b: int | None = None) -> int:
    L = lib()
    abuf = (C.c_uint8 * 16).from_buffer_copy(int(a).to_bytes(16, "little"))
    out = (C.c_uint8 * 16)()
    if b is None:
        getattr(L, f"orc_f128_{name}")(abuf, out)
    else:
        bbuf = (C.c_uint8 * 16).from_buffer_copy(int(b).to_bytes(16, "little"))
        getattr(L, f"orc_f128_{name}")(abuf, bbuf, out)
    return int.from_bytes(bytes(out), "little")


def f128_root_of_unity(n: int) -> int:
    out = (C.c_uint8 * 16)()
    lib().orc_f128_get_root_of_unity(n, out)
    return int.from_bytes(bytes(out), "little")


# ----------------------------------------------------------------------------------------------- math::fft

def get_twiddles(field: int, n: int, inverse: bool = False) -> np.ndarray:
    w = ELEM_WORDS[field]
    out = np.empty((n // 2, w) if w > 1 else (n // 2,), dtype=np.uint64)
    fn = lib().orc_f64_get_twiddles if field == F64 else lib().orc_f128_get_twiddles_p
    rc = fn(_p(out), n, int(inverse))
    if rc:
        raise ValueError(f"get_twiddles failed: {rc}")
    return out


def _fname(field, name):
    return getattr(lib(), f"orc_{'f64' if field == F64 else 'f128'}_{name}")


def fft_in_place(field: int, v: np.ndarray, n: int, ext: int, tw: np.ndarray):
    _fname(field, "fft_in_place")(_p(v), n, ext, _p(tw))


def permute(field: int, v: np.ndarray, n: int, ext: int = 1):
    _fname(field, "permute")(_p(v), n, ext)


def evaluate_poly(field: int, p: np.ndarray, n: int, ext: int, tw: np.ndarray):
    _fname(field, "evaluate_poly")(_p(p), n, ext, _p(tw))


def interpolate_poly(field: int, v: np.ndarray, n: int, ext: int, inv_tw: np.ndarray):
    _fname(field, "interpolate_poly")(_p(v), n, ext, _p(inv_tw))


def evaluate_poly_with_offset(field: int, p: np.ndarray, n: int, ext: int, tw: np.ndarray, offset_mem, blowup: int):
    """offset_mem: f64 -> Montgomery u64; f128 -> python int."""
    w = ELEM_WORDS[field]
    shape = (n * blowup * ext, w) if w > 1 else (n * blowup * ext,)
    out = np.empty(shape, dtype=np.uint64)
    if field == F64:
        lib().orc_f64_evaluate_poly_with_offset(_p(p), n, ext, _p(tw), int(offset_mem), blowup, _p(out))
    else:
        lib().orc_f128_evaluate_poly_with_offset_p(_p(p), n, ext, _p(tw), _p(_off_bytes(offset_mem)), blowup, _p(out))
    return out


def interpolate_poly_with_offset(field: int, v: np.ndarray, n: int, ext: int, inv_tw: np.ndarray, offset_mem):
    if field == F64:
        lib().orc_f64_interpolate_poly_with_offset(_p(v), n, ext, _p(inv_tw), int(offset_mem))
    else:
        lib().orc_f128_interpolate_poly_with_offset_p(_p(v), n, ext, _p(inv_tw), _p(_off_bytes(offset_mem)))


def eval_many(field: int, p: np.ndarray, xs: np.ndarray) -> np.ndarray:
    w = ELEM_WORDS[field]
    n, m = p.size // w, xs.size // w
    out = np.empty_like(xs)
    _fname(field, "eval_many")(_p(p), n, _p(xs), m, _p(out))
    return out


# ----------------------------------------------------------------------------------------------- hashing / merkle

def blake3(data: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data if data else b"\0")
    lib().orc_blake3_hash(buf, len(data), out)
    return bytes(out)


class digest_size:
    """`with digest_size(24):` -- the hashing functions below act as Blake3_192 (crypto/src/hash/blake/mod.rs:68-114)
    inside the block: 24-byte digests, 48-byte merge inputs, leaf / node arrays of 24-byte entries.  Default 32."""

    def __init__(self, n: int):
        self.n = n

    def __enter__(self):
        if lib().orc_set_digest_bytes(self.n):
            raise ValueError("digest size must be 24 or 32")
        global _DB
        self.prev, _DB = _DB, self.n

    def __exit__(self, *a):
        global _DB
        _DB = self.prev
        lib().orc_set_digest_bytes(self.prev)


_DB = 32


def hash_elements(field: int, elems: np.ndarray) -> bytes:
    out = (C.c_uint8 * 32)()
    e = np.ascontiguousarray(elems, dtype=np.uint64)
    lib().orc_hash_elements(field, _p(e), e.size // ELEM_WORDS[field], out)
    return bytes(out)[:_DB]


def merge(a: bytes, b: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    lib().orc_merge((C.c_uint8 * _DB).from_buffer_copy(a), (C.c_uint8 * _DB).from_buffer_copy(b), out)
    return bytes(out)[:_DB]


def merge_with_int(seed: bytes, value: int) -> bytes:
    out = (C.c_uint8 * 32)()
    lib().orc_merge_with_int((C.c_uint8 * _DB).from_buffer_copy(seed), value, out)
    return bytes(out)[:_DB]


def build_merkle_nodes(leaves: np.ndarray, threads: int = 1) -> np.ndarray:
    leaves = np.ascontiguousarray(leaves, dtype=np.uint8).reshape(-1, _DB)
    nodes = np.empty_like(leaves)
    rc = lib().orc_build_merkle_nodes(_p(leaves), leaves.shape[0], _p(nodes), threads)
    if rc:
        raise ValueError(f"build_merkle_nodes failed: {rc}")
    return nodes


# ----------------------------------------------------------------------------------------------- the path

def row_width(n_cols: int, ext: int) -> int:
    return 8 * ((n_cols * ext + 7) // 8)


def evaluate_polys_over(field: int, polys, ext: int, log2_R: int, log2_blowup: int, offset: int, threads: int = 1):
    """polys: list of per-column arrays.  offset: canonical python int.  Returns (N, row_width[,2]) array."""
    R, blowup = 1 << log2_R, 1 << log2_blowup
    w = ELEM_WORDS[field]
    rw = row_width(len(polys), ext)
    shape = (R * blowup, rw, w) if w > 1 else (R * blowup, rw)
    out = np.zeros(shape, dtype=np.uint64)
    polys = [np.ascontiguousarray(p, dtype=np.uint64) for p in polys]
    ptrs = _ptr_array(polys)
    if field == F64:
        rc = lib().orc_f64_evaluate_polys_over(ptrs, len(polys), ext, R, blowup, lib().orc_f64_new(offset), _p(out),
                                               threads)
    else:
        rc = lib().orc_f128_evaluate_polys_over_p(ptrs, len(polys), ext, R, blowup, _p(_off_bytes(offset)), _p(out),
                                                  threads)
    if rc:
        raise ValueError(f"evaluate_polys_over failed: {rc}")
    return out


def build_trace_commitment(field: int, traces, ext: int, log2_R: int, log2_blowup: int, offset: int,
                           threads: int = 1, out=None):
    """traces: list (per trace) of lists (per column) of arrays with R*ext elements.
    Returns dict(polys=[[...]], lde=[...], leaves, nodes, root).  out: the dict a previous call of the same shape
    returned -- its arrays are written again instead of allocating new ones (timing runs)."""
    R, blowup = 1 << log2_R, 1 << log2_blowup
    w = ELEM_WORDS[field]
    n_traces, n_cols = len(traces), len(traces[0])
    rw = row_width(n_cols, ext)
    N = R * blowup
    cols = [np.ascontiguousarray(c, dtype=np.uint64) for t in traces for c in t]
    if out is not None:
        polys = [c for t in out["polys"] for c in t]
        lde, leaves, nodes = out["lde"], out["leaves"], out["nodes"]
    else:
        polys = [np.empty_like(c) for c in cols]
        lde = [np.zeros((N, rw, w) if w > 1 else (N, rw), dtype=np.uint64) for _ in range(n_traces)]
        leaves = np.empty((N, _DB), dtype=np.uint8)
        nodes = np.empty((N, _DB), dtype=np.uint8)
    rc = lib().orc_build_trace_commitment(field, ext, log2_R, log2_blowup, n_cols, n_traces, _p(_off_bytes(offset)),
                                          _ptr_array(cols), _ptr_array(polys), _ptr_array(lde), _p(leaves), _p(nodes),
                                          threads)
    if rc:
        raise ValueError(f"build_trace_commitment failed: {rc}")
    polys = [polys[t * n_cols:(t + 1) * n_cols] for t in range(n_traces)]
    return dict(polys=polys, lde=lde, leaves=leaves, nodes=nodes, root=bytes(nodes[1]))


def last_phase_ms():
    """(interpolate, evaluate, hash rows, tree) wall clock in ms of the last build_*_commitment of this process."""
    out = (C.c_double * 4)()
    lib().orc_last_phase_ms(out)
    return tuple(out)


def blake3_compress_both(cv, block, counter: int, block_len: int, flags: int):
    """One compression through the SIMD and the scalar form of oracle/blake3_ref.c (test hook)."""
    cv = np.ascontiguousarray(cv, dtype=np.uint32)
    block = np.ascontiguousarray(block, dtype=np.uint32)
    a, b = np.zeros(8, np.uint32), np.zeros(8, np.uint32)
    lib().orc_blake3_compress_both(_p(cv), _p(block), C.c_uint64(counter), C.c_uint32(block_len), C.c_uint32(flags), _p(a), _p(b))
    return a, b


def build_constraint_commitment(field: int, poly_cols, ext: int, log2_R: int, log2_blowup: int, offset: int,
                                threads: int = 1):
    R, blowup = 1 << log2_R, 1 << log2_blowup
    w = ELEM_WORDS[field]
    n_cols = len(poly_cols)
    rw = row_width(n_cols, ext)
    N = R * blowup
    cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in poly_cols]
    lde = np.zeros((N, rw, w) if w > 1 else (N, rw), dtype=np.uint64)
    leaves = np.empty((N, _DB), dtype=np.uint8)
    nodes = np.empty((N, _DB), dtype=np.uint8)
    rc = lib().orc_build_constraint_commitment(field, ext, log2_R, log2_blowup, n_cols, _p(_off_bytes(offset)),
                                               _ptr_array(cols), _p(lde), _p(leaves), _p(nodes), threads)
    if rc:
        raise ValueError(f"build_constraint_commitment failed: {rc}")
    return dict(lde=lde, leaves=leaves, nodes=nodes, root=bytes(nodes[1]))


# ----------------------------------------------------------------------------------------------- merkle proofs

def merkle_prove(nodes: np.ndarray, leaves: np.ndarray, index: int):
    """MerkleTree::prove, crypto/src/merkle/mod.rs:192-212."""
    n = leaves.shape[0]
    if index >= n:
        raise ValueError("leaf index out of bounds")
    proof = [bytes(leaves[index]), bytes(leaves[index ^ 1])]
    i = (index + n) >> 1
    while i > 1:
        proof.append(bytes(nodes[i ^ 1]))
        i >>= 1
    return proof


def merkle_verify(root: bytes, index: int, proof) -> bool:
    """MerkleTree::verify, crypto/src/merkle/mod.rs:295-317."""
    r = index & 1
    v = merge(proof[r], proof[1 - r])
    index = (index + 2 ** (len(proof) - 1)) >> 1
    for p in proof[2:]:
        v = merge(v, p) if index & 1 == 0 else merge(p, v)
        index >>= 1
    return v == root


def merkle_prove_batch(nodes: np.ndarray, leaves: np.ndarray, indexes):
    """MerkleTree::prove_batch, crypto/src/merkle/mod.rs:222-284 (incl. map_indexes :376-395 and
    normalize_indexes :397-403).  Returns (leaves, nodes, depth) like BatchMerkleProof."""
    n = leaves.shape[0]
    depth = n.bit_length() - 1
    if len(indexes) == 0:
        raise ValueError("too few leaf indexes")
    if len(indexes) > 255:
        raise ValueError("too many leaf indexes")
    index_map = {}
    for i, idx in enumerate(indexes):
        index_map[idx] = i
        if idx >= n:
            raise ValueError("leaf index out of bounds")
    if len(index_map) != len(indexes):
        raise ValueError("duplicate leaf index")
    norm = sorted({idx - (idx & 1) for idx in indexes})
    out_leaves = [None] * len(index_map)
    out_nodes = []
    nxt = []
    for index in norm:
        missing = []
        for i in (index, index + 1):
            v = bytes(leaves[i])
            if i in index_map:
                out_leaves[index_map[i]] = v
            else:
                missing.append(v)
        out_nodes.append(missing)
        nxt.append((index + n) >> 1)
    for _ in range(1, depth):
        cur, nxt = nxt, []
        i = 0
        while i < len(cur):
            sib = cur[i] ^ 1
            if i + 1 < len(cur) and cur[i + 1] == sib:
                i += 1
            else:
                out_nodes[i].append(bytes(nodes[sib]))
            nxt.append(sib >> 1)
            i += 1
    return out_leaves, out_nodes, depth


# ----------------------------------------------------------------------------------------------- FRI layer pieces

def syn_div(field: int, ext: int, poly: np.ndarray, b: np.ndarray) -> np.ndarray:
    """polynom::syn_div_in_place(poly, 1, b) (math/src/polynom/mod.rs:535-542): poly / (x - b), the remainder dropped; the
    quotient's n - 1 coefficients followed by a zero, as the reference leaves the slice."""
    out = np.ascontiguousarray(poly, dtype=np.uint64).copy()
    bb = np.ascontiguousarray(b, dtype=np.uint64)
    n = out.size // (ELEM_WORDS[field] * ext)
    lib().orc_syn_div(field, ext, _p(out), n, _p(bb))
    return out


def ext_mul(field: int, ext: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_ext_mul(field, ext, _p(a), _p(b), _p(out))
    return out


def transpose_slice(field: int, src: np.ndarray, n: int, ext: int, N: int) -> np.ndarray:
    """utils/core/src/lib.rs:206-227: result[i][j] = source[i + j * (n / N)]."""
    src = np.ascontiguousarray(src, dtype=np.uint64)
    out = np.empty_like(src)
    lib().orc_transpose_slice(field, _p(src), n, ext, N, _p(out))
    return out


def apply_drp(field: int, values: np.ndarray, rows: int, ext: int, N: int, offset: int, alpha: np.ndarray,
              threads: int = 1) -> np.ndarray:
    """fri/src/folding/mod.rs:85-117.  values: rows x N elements of E (row major); alpha: one element of E."""
    values = np.ascontiguousarray(values, dtype=np.uint64)
    alpha = np.ascontiguousarray(alpha, dtype=np.uint64)
    w = ELEM_WORDS[field]
    out = np.empty((rows * ext, w) if w > 1 else (rows * ext,), dtype=np.uint64)
    lib().orc_apply_drp(field, _p(values), rows, ext, N, _p(_off_bytes(offset)), _p(alpha), _p(out), threads)
    return out


def fri_layer_commit(field: int, evaluations: np.ndarray, n: int, ext: int, N: int, threads: int = 1):
    """FriProver::build_layer, commit half (fri/src/prover/mod.rs:191-203): transpose_slice -> hash_values -> tree."""
    tr = transpose_slice(field, evaluations, n, ext, N)
    rows = n // N
    w = ELEM_WORDS[field]
    flat = tr.reshape(rows, -1)
    leaves = np.empty((rows, _DB), dtype=np.uint8)
    for i in range(rows):
        leaves[i] = np.frombuffer(hash_elements(field, flat[i]), dtype=np.uint8)
    nodes = build_merkle_nodes(leaves, threads)
    return dict(transposed=tr, leaves=leaves, nodes=nodes, root=bytes(nodes[1]))


def eval_column_at(field: int, poly: np.ndarray, ext_c: int, z: np.ndarray, ext_z: int) -> np.ndarray:
    """ColMatrix::evaluate_columns_at for one column (col_matrix.rs:249-254): P(z), coefficients embedded into z's field."""
    poly = np.ascontiguousarray(poly, dtype=np.uint64)
    z = np.ascontiguousarray(z, dtype=np.uint64)
    w = ELEM_WORDS[field]
    n = poly.size // (w * ext_c)
    out = np.empty((ext_z, w) if w > 1 else (ext_z,), dtype=np.uint64)
    lib().orc_eval_column_at(field, _p(poly), n, ext_c, _p(z), ext_z, _p(out))
    return out


# ----------------------------------------------------------------------------------------------- DEEP composition

def deep_compose(field: int, ext: int, n: int, tables, constraint_cols, z: np.ndarray, cc_traces, cc_constraints):
    """DeepCompositionPoly::add_trace_polys + add_composition_poly (prover/src/composer/mod.rs:62-193).

    tables: one list per TracePolyTable of (column, ext_c) pairs, main columns (ext_c = 1) first, then auxiliary ones
    (ext_c = ext); constraint_cols: columns of `ext` coordinates; cc_traces: one element of E per trace column in table
    order; cc_constraints: one per constraint column.  The out-of-domain values the reference is handed
    (T_i(z), T_i(z g), H_i(z)) are computed here with eval_column_at, as the prover does before composing
    (prover/src/lib.rs: get_ood_frame / evaluate_at).  Returns n elements of E.
    """
    w = ELEM_WORDS[field]
    z = np.ascontiguousarray(z, dtype=np.uint64)
    logn = n.bit_length() - 1
    if field == F64:
        g = np.zeros(ext, dtype=np.uint64)
        g[0] = lib().orc_f64_get_root_of_unity(logn)
    else:
        g = np.zeros((ext, w), dtype=np.uint64)
        g[0] = f128_from_ints([f128_root_of_unity(logn)])[0]
    zg = ext_mul(field, ext, z, g)
    cols, col_ext, ood_z, ood_zg = [], [], [], []
    for table in tables:
        for col, ext_c in table:
            col = np.ascontiguousarray(col, dtype=np.uint64)
            cols.append(col)
            col_ext.append(ext_c)
            ood_z.append(eval_column_at(field, col, ext_c, z, ext))
            ood_zg.append(eval_column_at(field, col, ext_c, zg, ext))
    ccols = [np.ascontiguousarray(c, dtype=np.uint64) for c in constraint_cols]
    ood_c = [eval_column_at(field, c, ext, z, ext) for c in ccols]
    per_table = np.array([len(t) for t in tables], dtype=np.uint64)
    col_ext_a = np.array(col_ext, dtype=np.uint64)
    cat = lambda xs: np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.uint64).reshape(-1) for x in xs])) \
        if len(xs) else np.zeros(1, dtype=np.uint64)
    ood_z_a, ood_zg_a, ood_c_a = cat(ood_z), cat(ood_zg), cat(ood_c)
    cct = cat(list(cc_traces))
    ccc = cat(list(cc_constraints))
    out = np.empty((n * ext, w) if w > 1 else (n * ext,), dtype=np.uint64)
    lib().orc_deep_compose(field, ext, n, len(tables), _p(per_table), _ptr_array(cols), _p(col_ext_a), _p(ood_z_a),
                           _p(ood_zg_a), _p(cct), len(ccols), _ptr_array(ccols) if ccols else None, _p(ood_c_a), _p(ccc),
                           _p(z), _p(out))
    return out


# ----------------------------------------------------------------------------------------------- constraint side from evaluations

def composition_poly_from_evaluations(field: int, ext: int, tables, log2_R: int, n_cols: int, offset: int, final_coeff=None):
    """The tail of ConstraintEvaluationTable::into_comb_poly for every packed trace (constraints/evaluation_table.rs:178-185:
    interpolate_poly_with_offset over the constraint evaluation domain), STARKPack's combination
    final = comb_0 + sum comb_i * final_coeff^i (prover/src/lib.rs:442-453) and CompositionPoly::new / segment
    (constraints/composition_poly.rs:21-41, 86-98).  tables: combined constraint evaluations, one array of ce elements of E per
    packed trace.  Returns the n_cols column polynomials (R elements of E each)."""
    w = ELEM_WORDS[field]
    R = 1 << log2_R
    final = None
    for i, t in enumerate(tables):
        v = np.ascontiguousarray(t, dtype=np.uint64).copy()
        ce = v.size // (w * ext)
        interpolate_poly_with_offset(field, v, ce, ext, get_twiddles(field, ce, inverse=True),
                                     lib().orc_f64_new(offset) if field == F64 else offset)
        if i == 0:
            final = v
        else:
            fc = np.ascontiguousarray(final_coeff, dtype=np.uint64)
            lib().orc_scale_acc(field, _p(final), _p(v), ext, ce, _p(fc), i)
    return segment(field, ext, final, R, n_cols)


def segment(field: int, ext: int, coefficients: np.ndarray, R: int, n_cols: int):
    """constraints/composition_poly.rs:86-98 `segment`: the composition polynomial's coefficients in n_cols contiguous runs of
    R elements of E -- column c holds coefficients c R .. (c + 1) R - 1 (the reference's own test: 0..15 -> four columns of
    four, composition_poly.rs:109-123; tests/golden/reference_kat.json)."""
    w = ELEM_WORDS[field]
    flat = np.ascontiguousarray(coefficients, dtype=np.uint64).reshape(-1, ext * w)
    return [np.ascontiguousarray(flat[c * R:(c + 1) * R]).reshape((R * ext, w) if w > 1 else (R * ext,)) for c in range(n_cols)]


def combine_evaluation_table(field: int, ext: int, columns, divisors, offset: int) -> np.ndarray:
    """ConstraintEvaluationTable::into_comb_poly up to the interpolation (evaluation_table.rs:166-176): every column divided by
    its divisor and summed.  divisors: one (a, b, exemptions) per column -- numerator x^a - b with b a base-field element in
    memory representation, exemptions a (possibly empty) array of base-field elements (air/src/air/divisor.rs:26-29)."""
    w = ELEM_WORDS[field]
    cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in columns]
    ce = cols[0].size // (w * ext)
    result = np.zeros_like(cols[0])  # E::zeroed_vector(self.num_rows())
    for col, (a, b, ex) in zip(cols, divisors):
        bb = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1)
        exs = np.ascontiguousarray(ex, dtype=np.uint64) if ex is not None and len(ex) else None
        n_ex = 0 if exs is None else exs.size // w
        lib().orc_acc_column(field, _p(col), ext, ce, a, _p(bb), _p(exs) if exs is not None else None, n_ex,
                             _p(_off_bytes(offset)), _p(result))
    return result
