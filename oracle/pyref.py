"""ORACLE (test infrastructure, NOT product code): definitional big-integer reference.

Independent of both the C oracle and the HIP kernels: plain Python integers, naive O(n^2) polynomial
evaluation.  It states WHAT the reference path computes (SURVEY.md §8a "mathematical contract"):

  * trace column values v_i = P(w_R^i)  ->  coefficients of P          (col_matrix.rs:196-206)
  * LDE row j, base column b = P_b(s * g^j), g = root of unity of order R*blowup
                                                                      (row_matrix.rs:82-98, segments.rs:127-152)
  * leaf_j = BLAKE3(canonical LE bytes of row j of every packed trace) (row_matrix.rs:204-238, blake/mod.rs:46-59)
  * nodes[i] = BLAKE3(nodes[2i] || nodes[2i+1]), nodes[1] = root        (merkle/mod.rs:350-374)

Only small sizes: everything here is pure-Python loops.
"""
from __future__ import annotations

import hashlib  # noqa: F401  (not used for BLAKE3: hashlib has no blake3)

F64_P = 2**64 - 2**32 + 1
F64_GENERATOR = 7
F64_TWO_ADICITY = 32
F64_TWO_ADIC_ROOT = 7277203076849721926

F128_P = 2**128 - 45 * 2**40 + 1
F128_GENERATOR = 3
F128_TWO_ADICITY = 40
F128_TWO_ADIC_ROOT = 23953097886125630542083529559205016746

MONT_R = 2**64


class Field:
    def __init__(self, name: str):
        if name == "f64":
            self.p, self.gen, self.adicity, self.root = F64_P, F64_GENERATOR, F64_TWO_ADICITY, F64_TWO_ADIC_ROOT
            self.elem_bytes, self.field_id = 8, 1
        elif name == "f128":
            self.p, self.gen, self.adicity, self.root = F128_P, F128_GENERATOR, F128_TWO_ADICITY, F128_TWO_ADIC_ROOT
            self.elem_bytes, self.field_id = 16, 2
        else:
            raise ValueError(name)
        self.name = name

    def root_of_unity(self, n: int) -> int:
        """math/src/field/traits.rs:254-263"""
        assert 0 < n <= self.adicity
        return pow(self.root, 1 << (self.adicity - n), self.p)

    # in-memory representation <-> canonical integer
    def to_mem(self, x: int) -> int:
        return (x * MONT_R) % self.p if self.name == "f64" else x % self.p

    def from_mem(self, m: int) -> int:
        return (m * pow(MONT_R, -1, self.p)) % self.p if self.name == "f64" else m

    def inv(self, x: int) -> int:
        return pow(x, self.p - 2, self.p)


def poly_eval(coeffs, x, p):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % p
    return acc


def interpolate(values, F: Field):
    """Coefficients of the degree<n polynomial with P(w^i) = values[i] (naive inverse DFT)."""
    n = len(values)
    w = F.root_of_unity(n.bit_length() - 1)
    winv = F.inv(w)
    ninv = F.inv(n % F.p)
    out = []
    for k in range(n):
        x = pow(winv, k, F.p)
        out.append(poly_eval(values, x, F.p) * ninv % F.p)
    return out


def lde_rows(polys, blowup: int, offset: int, F: Field):
    """polys: list of base columns (canonical coefficient lists, all length R).
    Returns rows[j][b] = P_b(offset * g^j), j < R*blowup (canonical ints)."""
    R = len(polys[0])
    N = R * blowup
    g = F.root_of_unity(N.bit_length() - 1)
    rows = []
    x = offset % F.p
    for _ in range(N):
        rows.append([poly_eval(col, x, F.p) for col in polys])
        x = x * g % F.p
    return rows


def row_bytes(row, F: Field) -> bytes:
    """Canonical little-endian serialisation of one row (blake/mod.rs:46-59, f64/mod.rs:605-610)."""
    return b"".join(int(v).to_bytes(F.elem_bytes, "little") for v in row)


def merkle_nodes(leaves, blake3):
    """merkle/mod.rs:350-374.  leaves: list of 32-byte digests.  Returns list of len(leaves) digests."""
    n = len(leaves) // 2
    nodes = [bytes(32)] * (2 * n)
    for i in range(n):
        nodes[n + i] = blake3(leaves[2 * i] + leaves[2 * i + 1])
    for i in range(n - 1, 0, -1):
        nodes[i] = blake3(nodes[2 * i] + nodes[2 * i + 1])
    return nodes


# --- BLAKE3 written from the public specification; pure Python, slow -----------------

_IV = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
_PERM = [2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8]
_M32 = 0xFFFFFFFF


def _rotr(x, n):
    return ((x >> n) | (x << (32 - n))) & _M32


def _g(v, a, b, c, d, mx, my):
    v[a] = (v[a] + v[b] + mx) & _M32
    v[d] = _rotr(v[d] ^ v[a], 16)
    v[c] = (v[c] + v[d]) & _M32
    v[b] = _rotr(v[b] ^ v[c], 12)
    v[a] = (v[a] + v[b] + my) & _M32
    v[d] = _rotr(v[d] ^ v[a], 8)
    v[c] = (v[c] + v[d]) & _M32
    v[b] = _rotr(v[b] ^ v[c], 7)


def _compress(cv, m, counter, block_len, flags):
    v = list(cv) + _IV[:4] + [counter & _M32, (counter >> 32) & _M32, block_len, flags]
    m = list(m)
    for _ in range(7):
        _g(v, 0, 4, 8, 12, m[0], m[1])
        _g(v, 1, 5, 9, 13, m[2], m[3])
        _g(v, 2, 6, 10, 14, m[4], m[5])
        _g(v, 3, 7, 11, 15, m[6], m[7])
        _g(v, 0, 5, 10, 15, m[8], m[9])
        _g(v, 1, 6, 11, 12, m[10], m[11])
        _g(v, 2, 7, 8, 13, m[12], m[13])
        _g(v, 3, 4, 9, 14, m[14], m[15])
        m = [m[i] for i in _PERM]
    return [v[i] ^ v[i + 8] for i in range(8)]


def _words(block: bytes):
    block = block + bytes(64 - len(block))
    return [int.from_bytes(block[4 * i:4 * i + 4], "little") for i in range(16)]


def _chunk_cv(data: bytes, counter: int, extra: int):
    cv = list(_IV)
    blocks = [data[i:i + 64] for i in range(0, len(data), 64)] or [b""]
    for i, blk in enumerate(blocks):
        flags = (1 if i == 0 else 0) | ((2 | extra) if i == len(blocks) - 1 else 0)
        cv = _compress(cv, _words(blk), counter, len(blk), flags)
    return cv


def blake3_py(data: bytes) -> bytes:
    if len(data) <= 1024:
        cv = _chunk_cv(data, 0, 8)
    else:
        chunks = [data[i:i + 1024] for i in range(0, len(data), 1024)]
        stack = []
        for c, ch in enumerate(chunks[:-1]):
            x = _chunk_cv(ch, c, 0)
            total = c + 1
            while total & 1 == 0:
                x = _compress(_IV, stack.pop() + x, 0, 64, 4)
                total >>= 1
            stack.append(x)
        cv = _chunk_cv(chunks[-1], len(chunks) - 1, 0)
        while stack:
            left = stack.pop()
            cv = _compress(_IV, left + cv, 0, 64, 4 | (8 if not stack else 0))
    return b"".join(w.to_bytes(4, "little") for w in cv)
