"""Generates tests/golden/reference_kat.json -- the known-answer vectors the REFERENCE'S OWN TESTS hold for arithmetic that
the commitment path and its callers run: literal inputs AND literal expected outputs, typed in below as data with the file
and lines they stand in.  (Everything else the reference tests about this path is structural -- an FFT equals naive
evaluation, a root equals nested merges -- see tests/golden/README.md.)

  * f64 quadratic / cubic extension products (math/src/field/f64/tests.rs:251-280 `quad_mul`, :321-378 `cube_mul`):
    ExtensibleField<2/3>::mul -- what FRI folding (apply_drp), the out-of-domain evaluation, the DEEP composition and the
    constraint combination multiply with (csrc/fri_kernels.hpp ext_mul, oracle.c f64_ext_mul);
  * f128 elements_as_bytes (math/src/field/f128/tests.rs:165-181): the byte string hash_elements feeds to BLAKE3 for a row
    of f128 elements (crypto/src/hash/blake/mod.rs:39-49, 145-160) -- the digest of those bytes is added here from the
    official BLAKE3 (LLVM's copy), so that `wf_hash_rows` on the elements 1, 2, 3, 4 is pinned to reference bytes;
  * polynom::syn_div by (x - b) (math/src/polynom/tests.rs:178-207, f128): the division of the DEEP composition
    (composer/mod.rs:62-193 divides by x - z and x - z g with syn_div_in_place).

  * the composition polynomial's column split (prover/src/constraints/composition_poly.rs:109-123 `segment`, f128): the
    coefficients 0..15 in four columns of four.
  * transpose_slice (utils/core/src/lib.rs:198-205, the function's doc test): 0..7 in rows of two -- how FriProver lays out
    a layer's evaluations before hashing them (fri/src/prover/mod.rs:207-214).

Every literal is cross-checked below against Python integers (a typing error fails the run); the EXPECTED values written
to the fixture are the reference's literals, not the recomputation.

    python oracle/gen_golden_reference_kat.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.gen_golden import blake3  # noqa: E402  (LLVM's BLAKE3)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
M64 = 2**64 - 2**32 + 1
M128 = 2**128 - 45 * 2**40 + 1

# f64/tests.rs:251-280 -- QuadExtension<BaseElement>::new(a0, a1) * new(b0, b1) == new(e0, e1)
QUAD_MUL = [
    {"a": [3, 1], "b": [4, 2], "expected": [8, 12], "where": "f64/tests.rs:262-265 (within bounds)"},
    {"a": [3, M64 - 1], "b": [M64 - 3, 5], "expected": [1, 13], "where": "f64/tests.rs:267-272 (with overflow)"},
    {"a": [3, M64 - 1], "b": [10, M64 - 2], "expected": [26, 18446744069414584307], "where": "f64/tests.rs:274-280"},
]
# f64/tests.rs:321-378 -- CubeExtension
CUBE_MUL = [
    {"a": [3, 5, 2], "b": [320, 68, 3], "expected": [1111, 1961, 995], "where": "f64/tests.rs:331-347 (within bounds)"},
    {"a": [18446744069414584267, 18446744069414584309, 9223372034707292160],
     "b": [18446744069414584101, 420, 18446744069414584121],
     "expected": [14070, 18446744069414566571, 5970], "where": "f64/tests.rs:349-365 (with overflow)"},
    {"a": [18446744069414584266, 18446744069412558094, 5268562], "b": [18446744069414583589, 1226, 5346],
     "expected": [18446744065041672051, 25275910656, 21824696736], "where": "f64/tests.rs:367-383"},
]
# f128/tests.rs:165-181 -- BaseElement::elements_as_bytes(&[1, 2, 3, 4])
F128_BYTES_SOURCE = [1, 2, 3, 4]
F128_BYTES_EXPECTED = [1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                       0, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                       0, 0, 0, 0]
# polynom/tests.rs:178-207 (f128) -- syn_div(poly, 1, b) = poly / (x - b), remainder dropped, leading zeros removed
SYN_DIV = [
    {"poly": [6, 5, 1], "b": M128 - 3, "expected": [2, 1], "where": "polynom/tests.rs:181-190: (x + 2)(x + 3) / (x + 3)"},
    {"poly": [M128 - 42, 0, M128 - 12, 1], "b": 3, "expected": [M128 - 27, M128 - 9, 1],
     "where": "polynom/tests.rs:192-207: (x^3 - 12 x^2 - 42) / (x - 3), does not divide evenly"},
]


# composition_poly.rs:109-123 -- segment((0..16), 4, 4)
SEGMENT = {"values": list(range(16)), "trace_len": 4, "num_cols": 4,
           "expected": [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]}


# utils/core/src/lib.rs:198-205 -- transpose_slice::<u32, 2>(&[0, 1, .., 7])
TRANSPOSE = {"values": list(range(8)), "N": 2, "expected": [[0, 4], [1, 5], [2, 6], [3, 7]]}


def quad_mul(a, b):  # x^2 = x - 2 (f64/mod.rs:401-430)
    c0, c1, c2 = a[0] * b[0], a[0] * b[1] + a[1] * b[0], a[1] * b[1]
    return [(c0 - 2 * c2) % M64, (c1 + c2) % M64]


def cube_mul(a, b):  # x^3 = x + 1 (f64/mod.rs:432-472)
    c = [0] * 5
    for i in range(3):
        for j in range(3):
            c[i + j] += a[i] * b[j]
    return [(c[0] + c[3]) % M64, (c[1] + c[3] + c[4]) % M64, (c[2] + c[4]) % M64]


def syn_div(poly, b, p):  # quotient of poly / (x - b); the remainder is dropped
    q = [0] * (len(poly) - 1)
    carry = 0
    for i in range(len(poly) - 1, 0, -1):
        carry = (poly[i] + carry * b) % p
        q[i - 1] = carry
    return q


def main():
    for c in QUAD_MUL:
        assert quad_mul(c["a"], c["b"]) == c["expected"], c
    for c in CUBE_MUL:
        assert cube_mul(c["a"], c["b"]) == c["expected"], c
    want = b"".join(int(v).to_bytes(16, "little") for v in F128_BYTES_SOURCE)
    assert list(want) == F128_BYTES_EXPECTED
    for c in SYN_DIV:
        assert syn_div(c["poly"], c["b"], M128) == c["expected"], c
    assert [SEGMENT["values"][c * 4:(c + 1) * 4] for c in range(4)] == SEGMENT["expected"]
    rows = len(TRANSPOSE["values"]) // TRANSPOSE["N"]
    assert [[TRANSPOSE["values"][i + j * rows] for j in range(TRANSPOSE["N"])] for i in range(rows)] == TRANSPOSE["expected"]
    s = lambda v: [str(x) for x in v]  # noqa: E731  (decimal strings: JSON numbers stop at 2^53)
    out = {
        "_note": "known-answer vectors held by the reference's own tests; see oracle/gen_golden_reference_kat.py for the lines",
        "f64_quad_mul": [{"a": s(c["a"]), "b": s(c["b"]), "expected": s(c["expected"]), "where": c["where"]} for c in QUAD_MUL],
        "f64_cube_mul": [{"a": s(c["a"]), "b": s(c["b"]), "expected": s(c["expected"]), "where": c["where"]} for c in CUBE_MUL],
        "f128_elements_as_bytes": {"source": s(F128_BYTES_SOURCE), "expected_bytes": F128_BYTES_EXPECTED,
                                   "blake3_256_of_expected_bytes": blake3(bytes(F128_BYTES_EXPECTED)).hex(),
                                   "where": "f128/tests.rs:165-181; digest: official BLAKE3 of those bytes"},
        "f128_composition_segment": dict(SEGMENT, where="prover/src/constraints/composition_poly.rs:109-123"),
        "transpose_slice": dict(TRANSPOSE, where="utils/core/src/lib.rs:198-205 (doc test)"),
        "f128_syn_div": [{"poly": s(c["poly"]), "b": str(c["b"]), "expected": s(c["expected"]), "where": c["where"]} for c in SYN_DIV],
    }
    with open(os.path.join(OUT, "reference_kat.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.join(OUT, "reference_kat.json"))


if __name__ == "__main__":
    main()
