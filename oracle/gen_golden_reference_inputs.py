"""Generates tests/golden/reference_inputs.json -- golden cases whose INPUTS are the literal inputs of the reference's
own tests; run once in the BUILD container (needs /opt/rocm LLVM for the official BLAKE3).

  * LEAVES4 / LEAVES8: the 32-byte leaves of /root/reference/crypto/src/merkle/tests.rs:13-65 (typed in below as the
    byte values that file lists) -> tree nodes, root and the single-leaf proofs whose SHAPE those tests assert
    (tests.rs:67-135: proof = [leaf, sibling leaf, sibling nodes bottom-up]).
  * the FRI test polynomial of /root/reference/fri/src/prover/tests.rs:58-69 (build_evaluations: coefficients
    0, 1, .., trace_length - 1 over f128, zero-extended to trace_length * lde_blowup and evaluated with fft::evaluate_poly)
    for trace_length 16, lde_blowup 8 -> the evaluations, and the first FRI layer (folding factor 4: transpose_slice,
    hash_values, Merkle tree) a FriProver would commit to.

The EXPECTED values are NOT the reference's (its tests assert structure, never digest bytes or field values, and it
cannot run here): they come from Python big integers (oracle/pyref.py, naive evaluation) and the official BLAKE3 C
implementation bundled with LLVM, exactly as oracle/gen_golden.py makes the other fixtures.

    python oracle/gen_golden_reference_inputs.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import pyref as P  # noqa: E402
from oracle.gen_golden import blake3  # noqa: E402  (LLVM's BLAKE3)

LEAVES4 = [
    [166, 168, 47, 140, 153, 86, 156, 86, 226, 229, 149, 76, 70, 132, 209, 109, 166, 193, 113, 197, 42, 116, 170, 144, 74,
     104, 29, 110, 220, 49, 224, 123],
    [243, 57, 40, 140, 185, 79, 188, 229, 232, 117, 143, 118, 235, 229, 73, 251, 163, 246, 151, 170, 14, 243, 255, 127, 175,
     230, 94, 227, 214, 5, 89, 105],
    [11, 33, 220, 93, 26, 67, 166, 154, 93, 7, 115, 130, 70, 13, 166, 45, 120, 233, 175, 86, 144, 110, 253, 250, 67, 108,
     214, 115, 24, 132, 45, 234],
    [47, 173, 224, 232, 30, 46, 197, 186, 215, 15, 134, 211, 73, 14, 34, 216, 6, 11, 217, 150, 90, 242, 8, 31, 73, 85, 150,
     254, 229, 244, 23, 231],
]
LEAVES8 = [
    [115, 29, 176, 48, 97, 18, 34, 142, 51, 18, 164, 235, 236, 96, 113, 132, 189, 26, 70, 93, 101, 143, 142, 52, 252, 33,
     80, 157, 194, 52, 209, 129],
    [52, 46, 37, 214, 24, 248, 121, 199, 229, 25, 171, 67, 65, 37, 98, 142, 182, 72, 202, 42, 223, 160, 136, 60, 38, 255,
     222, 82, 26, 27, 130, 203],
    [130, 43, 231, 0, 59, 228, 152, 140, 18, 33, 87, 27, 49, 190, 44, 82, 188, 155, 163, 108, 166, 198, 106, 143, 83, 167,
     201, 152, 106, 176, 242, 119],
    [207, 158, 56, 143, 28, 146, 238, 47, 169, 32, 166, 97, 163, 238, 171, 243, 33, 209, 120, 219, 17, 182, 96, 136, 13,
     90, 6, 27, 247, 242, 49, 111],
    [179, 64, 123, 119, 226, 139, 161, 127, 36, 251, 218, 88, 20, 217, 212, 85, 112, 85, 185, 193, 230, 181, 4, 22, 54,
     219, 135, 98, 235, 180, 182, 7],
    [101, 240, 19, 44, 43, 213, 31, 138, 39, 26, 82, 147, 255, 96, 234, 51, 105, 6, 233, 144, 255, 187, 242, 3, 157, 246,
     55, 175, 98, 121, 92, 175],
    [25, 96, 149, 179, 94, 8, 170, 214, 169, 135, 12, 212, 224, 157, 182, 127, 233, 93, 151, 214, 36, 183, 156, 212, 233,
     152, 125, 244, 146, 161, 75, 128],
    [247, 43, 130, 141, 234, 172, 61, 187, 109, 31, 56, 30, 14, 232, 92, 158, 48, 161, 108, 234, 170, 180, 233, 77, 200,
     248, 45, 152, 125, 11, 1, 171],
]


def tree_case(name, leaves):
    lv = [bytes(x) for x in leaves]
    assert all(len(x) == 32 for x in lv)
    nodes = P.merkle_nodes(lv, blake3)
    n = len(lv)
    proofs = {}
    for idx in range(n):  # MerkleTree::prove (merkle/mod.rs:192-212): leaf, sibling leaf, sibling nodes bottom-up
        path = [lv[idx], lv[idx ^ 1]]
        i = (idx + n) >> 1
        while i > 1:
            path.append(nodes[i ^ 1])
            i >>= 1
        proofs[str(idx)] = [x.hex() for x in path]
    return dict(name=name, leaves=[x.hex() for x in lv], nodes=[x.hex() for x in nodes], root=nodes[1].hex(), proofs=proofs)


def fri_case(trace_length=16, lde_blowup=8, folding=4):
    F = P.Field("f128")
    n = trace_length * lde_blowup
    coeffs = list(range(trace_length)) + [0] * (n - trace_length)
    w = F.root_of_unity(n.bit_length() - 1)
    evals = [P.poly_eval(coeffs, pow(w, i, F.p), F.p) for i in range(n)]  # fft::evaluate_poly: natural order, no offset
    rows = n // folding
    # transpose_slice (utils/core/src/lib.rs:206-227): result[i][j] = source[i + j * rows]
    transposed = [[evals[i + j * rows] for j in range(folding)] for i in range(rows)]
    leaves = [blake3(P.row_bytes(r, F)) for r in transposed]  # hash_values = hash_elements of the N values (fri/src/utils.rs:41-50)
    nodes = P.merkle_nodes(leaves, blake3)
    return dict(name="fri_prover_tests_build_evaluations", field="f128", trace_length=trace_length, lde_blowup=lde_blowup,
                folding=folding, evaluations=[str(v) for v in evals], transposed=[[str(v) for v in r] for r in transposed],
                leaves=[x.hex() for x in leaves], nodes=[x.hex() for x in nodes], root=nodes[1].hex())


def main():
    out = dict(
        note="inputs: literal test inputs of the reference (crypto/src/merkle/tests.rs:13-65, fri/src/prover/tests.rs:58-69); "
             "expected values: Python big integers + LLVM-bundled official BLAKE3 -- NOT outputs of the reference",
        trees=[tree_case("LEAVES4", LEAVES4), tree_case("LEAVES8", LEAVES8)],
        fri=fri_case())
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "reference_inputs.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
