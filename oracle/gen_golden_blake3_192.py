"""Generates tests/golden/blake3_192.json -- fixtures for the Blake3_192 hasher (crypto/src/hash/blake/mod.rs:68-114: the
BLAKE3 output truncated to 24 bytes, merge of two digests = hash of their 48 bytes, merge_with_int = hash of 24 + 8 bytes).
Expected values: the official BLAKE3 C implementation bundled with LLVM + Python big integers (oracle/pyref.py), as in
oracle/gen_golden.py -- without the C oracle and without the HIP code.

    python oracle/gen_golden_blake3_192.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import gen_golden as GG  # noqa: E402
from oracle import pyref as P  # noqa: E402


def h192(data: bytes) -> bytes:
    return GG.blake3(data)[:24]


def merkle_nodes_192(leaves):
    n = len(leaves) // 2
    nodes = [bytes(24)] * (2 * n)
    for i in range(n):
        nodes[n + i] = h192(leaves[2 * i] + leaves[2 * i + 1])
    for i in range(n - 1, 0, -1):
        nodes[i] = h192(nodes[2 * i] + nodes[2 * i + 1])
    return nodes


def commit_case_192(name, field, ext, blowup, traces):
    c = GG.commit_case(name, field, ext, blowup, traces)   # values (polys, LDE) do not depend on the hasher
    F = P.Field(field)
    rows = [[[int(v) for v in row] for row in tr] for tr in c["lde"]]
    N = len(rows[0])
    leaves = [h192(b"".join(P.row_bytes(rows[t][j], F) for t in range(len(traces)))) for j in range(N)]
    nodes = merkle_nodes_192(leaves)
    c.update(leaves=[x.hex() for x in leaves], nodes=[x.hex() for x in nodes], root=nodes[1].hex(), digest_bytes=24)
    return c


def main():
    a, b = h192(b"left"), h192(b"right")
    out = dict(source="LLVM-bundled BLAKE3 (see blake3_kat.json), truncated to 24 bytes",
               kat=[dict(len=n, digest=h192(GG.pattern(n)).hex()) for n in (16, 64, 160, 1024, 1040, 4096)],
               merge=dict(left=a.hex(), right=b.hex(), digest=h192(a + b).hex()),
               merge_with_int=dict(seed=a.hex(), value=0x0123456789ABCDEF,
                                   digest=h192(a + (0x0123456789ABCDEF).to_bytes(8, "little")).hex()),
               trees=[])
    for n in (4, 8, 64):
        lv = [h192(i.to_bytes(4, "little")) for i in range(n)]
        out["trees"].append(dict(leaves=[x.hex() for x in lv], nodes=[x.hex() for x in merkle_nodes_192(lv)]))
    rnd = random.Random(192)
    F64, F128 = P.Field("f64"), P.Field("f128")
    rc = lambda F, n: [rnd.randrange(F.p) for _ in range(n)]  # noqa: E731
    fib = [[1, 2, 5, 13, 34, 89, 233, 610], [1, 3, 8, 21, 55, 144, 377, 987]]  # prover/src/trace/tests.rs:28-38
    out["commits"] = [
        commit_case_192("fib8_f64_blowup8_b192", "f64", 1, 8, [fib]),
        commit_case_192("packed3_f64_8x2_blowup2_b192", "f64", 1, 2, [[rc(F64, 8) for _ in range(2)] for _ in range(3)]),
        commit_case_192("rand8x10_f128_blowup2_b192", "f128", 1, 2, [[rc(F128, 8) for _ in range(10)]]),
        commit_case_192("rand16x9_f64_blowup4_b192", "f64", 1, 4, [[rc(F64, 16) for _ in range(9)]]),
    ]
    path = os.path.join(GG.OUT, "blake3_192.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
