"""Generates tests/golden/*.json -- run once in the BUILD container (needs /opt/rocm LLVM for the official BLAKE3).

The vectors are produced WITHOUT the C oracle and WITHOUT the HIP code:
  * field / NTT / LDE values : Python big integers, naive O(n^2) evaluation (oracle/pyref.py);
  * digests                  : the official BLAKE3 C implementation v1.8.x that LLVM bundles, exported by
                               /opt/rocm/lib/llvm/lib/libclang-cpp.so as llvm_blake3_hasher_* (SURVEY.md §8c).
Both the C oracle (tests/test_oracle.py) and the HIP path (tests/test_gpu_golden.py) are checked against them.
The reference itself (Rust) cannot run here; its tests hold no literal expected outputs for this path, so these
vectors are anchored on the path's mathematical contract and on the reference's deterministic test inputs
(Fibonacci trace of prover/src/trace/tests.rs:28-38, LEAVES4/LEAVES8 of crypto/src/merkle/tests.rs:13-65).

    python oracle/gen_golden.py
"""
import ctypes
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import pyref as P  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

_llvm = ctypes.CDLL("/opt/rocm/lib/llvm/lib/libclang-cpp.so")
_llvm.llvm_blake3_version.restype = ctypes.c_char_p


def blake3(data: bytes) -> bytes:
    h = ctypes.create_string_buffer(4096)
    _llvm.llvm_blake3_hasher_init(h)
    _llvm.llvm_blake3_hasher_update(h, data, ctypes.c_size_t(len(data)))
    out = ctypes.create_string_buffer(32)
    _llvm.llvm_blake3_hasher_finalize(h, out, ctypes.c_size_t(32))
    return out.raw


def pattern(n):  # the input pattern of the official BLAKE3 test vectors
    return bytes(i % 251 for i in range(n))


def gen_blake3():
    assert blake3(b"").hex() == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    lens = [0, 1, 2, 7, 8, 63, 64, 65, 127, 128, 129, 512, 1023, 1024, 1025, 2040, 2048, 2049, 3072, 3073, 4080,
            4096, 4097, 5000, 5120, 6144, 7168, 8192, 8193, 16384, 31744, 102400]
    kat = [dict(len=n, digest=blake3(pattern(n)).hex()) for n in lens]
    a, b = blake3(b"left"), blake3(b"right")
    merge = dict(left=a.hex(), right=b.hex(), digest=blake3(a + b).hex())
    mwi = dict(seed=a.hex(), value=0x0123456789ABCDEF, digest=blake3(a + (0x0123456789ABCDEF).to_bytes(8, "little")).hex())
    # crypto/src/merkle/tests.rs:13-65 use fixed 32-byte leaves; here: leaves i = blake3(le32(i))
    leaves4 = [blake3(i.to_bytes(4, "little")) for i in range(4)]
    leaves8 = [blake3(i.to_bytes(4, "little")) for i in range(8)]
    trees = []
    for lv in (leaves4, leaves8):
        nodes = P.merkle_nodes(lv, blake3)
        trees.append(dict(leaves=[x.hex() for x in lv], nodes=[x.hex() for x in nodes]))
    return dict(source=f"LLVM-bundled BLAKE3 {_llvm.llvm_blake3_version().decode()}", input="byte i = i % 251",
                kat=kat, merge=merge, merge_with_int=mwi, trees=trees)


def gen_field():
    rnd = random.Random(20240661)
    out = {}
    for name in ("f64", "f128"):
        F = P.Field(name)
        edge = [0, 1, 2, F.p - 1, F.p - 2, (F.p + 1) // 2, 2**32 - 1 if name == "f64" else 2**64 - 1,
                2**32 if name == "f64" else 2**64, F.gen]
        vals = edge + [rnd.randrange(F.p) for _ in range(40)]
        ops = []
        for i in range(len(vals)):
            a, b = vals[i], vals[(i * 7 + 3) % len(vals)]
            ops.append(dict(a=str(a), b=str(b), add=str((a + b) % F.p), sub=str((a - b) % F.p),
                            mul=str(a * b % F.p), inv=str(F.inv(a) if a else 0), mem_a=str(F.to_mem(a))))
        roots = {str(n): str(F.root_of_unity(n)) for n in (1, 2, 3, 6, 10, 20, 23, F.adicity)}
        out[name] = dict(modulus=str(F.p), generator=F.gen, two_adicity=F.adicity, ops=ops, roots_of_unity=roots)
    return out


def commit_case(name, field, ext, blowup, traces, offset=None):
    """traces: list of traces; trace = list of columns; column = list of canonical ints (len R*ext)."""
    F = P.Field(field)
    offset = F.gen if offset is None else offset
    R = len(traces[0][0]) // ext
    all_polys, all_rows = [], []
    for tr in traces:
        base_cols = []  # base column b = col*ext + coord
        for col in tr:
            for e in range(ext):
                base_cols.append(col[e::ext])
        polys = [P.interpolate(c, F) for c in base_cols]
        rows = P.lde_rows(polys, blowup, offset, F)
        # back to per-column interleaved coefficient layout
        polys_cols = []
        for ci in range(len(tr)):
            inter = [0] * (R * ext)
            for e in range(ext):
                inter[e::ext] = polys[ci * ext + e]
            polys_cols.append(inter)
        all_polys.append(polys_cols)
        all_rows.append(rows)
    N = R * blowup
    leaves = [blake3(b"".join(P.row_bytes(all_rows[t][j], F) for t in range(len(traces)))) for j in range(N)]
    nodes = P.merkle_nodes(leaves, blake3)
    return dict(name=name, field=field, ext=ext, log2_trace_len=R.bit_length() - 1, log2_blowup=blowup.bit_length() - 1,
                offset=str(offset),
                traces=[[[str(v) for v in col] for col in tr] for tr in traces],
                polys=[[[str(v) for v in col] for col in tr] for tr in all_polys],
                lde=[[[str(v) for v in row] for row in rows] for rows in all_rows],
                leaves=[x.hex() for x in leaves], nodes=[x.hex() for x in nodes], root=nodes[1].hex())


def gen_commit():
    rnd = random.Random(661)
    fib = [[1, 2, 5, 13, 34, 89, 233, 610], [1, 3, 8, 21, 55, 144, 377, 987]]  # prover/src/trace/tests.rs:28-38
    cases = []
    cases.append(commit_case("fib8_f128_blowup2", "f128", 1, 2, [fib]))
    cases.append(commit_case("fib8_f64_blowup8", "f64", 1, 8, [fib]))
    F64, F128 = P.Field("f64"), P.Field("f128")
    rc = lambda F, n: [rnd.randrange(F.p) for _ in range(n)]  # noqa: E731
    cases.append(commit_case("rand16x3_f64_blowup4", "f64", 1, 4, [[rc(F64, 16) for _ in range(3)]]))
    cases.append(commit_case("rand8x9_f64_blowup2", "f64", 1, 2, [[rc(F64, 8) for _ in range(9)]]))  # 2 segments
    cases.append(commit_case("quad8x2_f64_blowup4", "f64", 2, 4, [[rc(F64, 16) for _ in range(2)]]))
    cases.append(commit_case("cube8x1_f64_blowup2", "f64", 3, 2, [[rc(F64, 24)]]))
    cases.append(commit_case("packed3_f64_8x2_blowup2", "f64", 1, 2, [[rc(F64, 8) for _ in range(2)] for _ in range(3)]))
    cases.append(commit_case("rand8x10_f128_blowup2", "f128", 1, 2, [[rc(F128, 8) for _ in range(10)]]))
    cases.append(commit_case("packed2_quad_f128_8x1_blowup2", "f128", 2, 2, [[rc(F128, 16)] for _ in range(2)]))
    # do_work trace x -> x^3 + 42 (examples/src/do_work/prover.rs:62-80), 10 identical-rule columns from starts 0..9
    dw = []
    for s in range(10):
        col, x = [], s
        for _ in range(8):
            col.append(x)
            x = (pow(x, 3, F128.p) + 42) % F128.p
        dw.append(col)
    cases.append(commit_case("do_work8x10_f128_blowup8", "f128", 1, 8, [dw]))
    cases.append(commit_case("rand16x2_f64_offset", "f64", 1, 2, [[rc(F64, 16) for _ in range(2)]], offset=49))
    return cases


def main():
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "blake3_kat.json"), "w") as f:
        json.dump(gen_blake3(), f, indent=1)
    with open(os.path.join(OUT, "field_kat.json"), "w") as f:
        json.dump(gen_field(), f, indent=1)
    with open(os.path.join(OUT, "lde_commit_small.json"), "w") as f:
        json.dump(gen_commit(), f)
    for n in os.listdir(OUT):
        print(n, os.path.getsize(os.path.join(OUT, n)))


if __name__ == "__main__":
    main()
