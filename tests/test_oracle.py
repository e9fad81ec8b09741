"""Pins the CPU oracle (runs without a GPU).

1. Golden vectors produced independently of the oracle (oracle/gen_golden.py: Python big integers + the official
   BLAKE3 C implementation).
2. The definitional / known-answer checks the reference's own tests make for this path, re-run on the oracle:
     math/src/field/f64/tests.rs:17-161, f128/tests.rs:19-117   field identities and edge cases
     math/src/fft/tests.rs:19-72                                  fft == polynom::eval_many, twiddles == permuted powers
     prover/src/matrix/tests.rs:13-40                             row-matrix LDE == per-column evaluation on the coset
     crypto/src/merkle/tests.rs:67-92                             tree == nested merges
     crypto/src/hash/blake/tests.rs:11-30                         hash is sensitive to zero padding
"""
import numpy as np
import pytest

import golden_util as G
from conftest import rand_f64, rand_f128

F64, F128 = 1, 2
P64 = 2**64 - 2**32 + 1
P128 = 2**128 - 45 * 2**40 + 1


# ------------------------------------------------------------------------------------------------ golden: BLAKE3
def test_blake3_golden(orc):
    g = G.load("blake3_kat.json")
    for k in g["kat"]:
        data = bytes(i % 251 for i in range(k["len"]))
        assert orc.blake3(data).hex() == k["digest"], k["len"]
    m = g["merge"]
    assert orc.merge(bytes.fromhex(m["left"]), bytes.fromhex(m["right"])).hex() == m["digest"]
    w = g["merge_with_int"]
    assert orc.merge_with_int(bytes.fromhex(w["seed"]), w["value"]).hex() == w["digest"]
    for t in g["trees"]:
        leaves = np.frombuffer(b"".join(bytes.fromhex(x) for x in t["leaves"]), dtype=np.uint8).reshape(-1, 32)
        assert G.hexrows(orc.build_merkle_nodes(leaves)) == t["nodes"]
        assert G.hexrows(orc.build_merkle_nodes(leaves, threads=4)) == t["nodes"]


def test_blake3_pyref_agrees(orc):
    from oracle import pyref
    g = G.load("blake3_kat.json")
    for k in g["kat"]:
        if k["len"] <= 8193:
            assert pyref.blake3_py(bytes(i % 251 for i in range(k["len"]))).hex() == k["digest"]


# ------------------------------------------------------------------------------------------------ golden: fields
def test_field_golden(orc):
    g = G.load("field_kat.json")
    L = orc.lib()
    f = g["f64"]
    for o in f["ops"]:
        a, b = int(o["a"]), int(o["b"])
        am, bm = L.orc_f64_new(a), L.orc_f64_new(b)
        assert am == int(o["mem_a"])
        assert am < P64
        assert L.orc_f64_as_int(am) == a
        assert L.orc_f64_as_int(L.orc_f64_add(am, bm)) == int(o["add"])
        assert L.orc_f64_as_int(L.orc_f64_sub(am, bm)) == int(o["sub"])
        assert L.orc_f64_as_int(L.orc_f64_mul(am, bm)) == int(o["mul"])
        assert L.orc_f64_as_int(L.orc_f64_inv(am)) == int(o["inv"])
    for n, r in f["roots_of_unity"].items():
        assert L.orc_f64_as_int(L.orc_f64_get_root_of_unity(int(n))) == int(r)
    f = g["f128"]
    for o in f["ops"]:
        a, b = int(o["a"]), int(o["b"])
        assert orc.f128_op("add", a, b) == int(o["add"])
        assert orc.f128_op("sub", a, b) == int(o["sub"])
        assert orc.f128_op("mul", a, b) == int(o["mul"])
        assert orc.f128_op("inv", a) == int(o["inv"])
    for n, r in f["roots_of_unity"].items():
        assert orc.f128_root_of_unity(int(n)) == int(r)


def test_f64_reference_known_answers(orc):
    """math/src/field/f64/tests.rs:17-161 (the assertions with literal operands)."""
    L = orc.lib()
    new, val = L.orc_f64_new, L.orc_f64_as_int
    add, sub, mul = L.orc_f64_add, L.orc_f64_sub, L.orc_f64_mul
    M = P64
    assert add(new(2), new(3)) == new(5)
    t = new(M - 1)
    assert add(t, new(1)) == new(0) and add(t, new(2)) == new(1)          # tests.rs:28-31
    assert sub(new(5), new(3)) == new(2) and sub(new(3), new(5)) == new(M - 2)  # :41-48
    assert mul(new(5), new(3)) == new(15)
    assert mul(t, t) == new(1) and mul(t, new(2)) == new(M - 2) and mul(t, new(4)) == new(M - 4)  # :74-78
    assert mul(new((M + 1) // 2), new(2)) == new(1)                        # :80-84
    assert val(new(2**64 - 1)) == (2**64 - 1) % M                           # :124-127
    assert val(new(0)) == val(new(M)) == 0                                  # :129-132
    assert L.orc_f64_inv(new(1)) == new(1) and L.orc_f64_inv(new(0)) == new(0)  # :117-120
    r32 = L.orc_f64_get_root_of_unity(32)
    assert val(r32) == 7277203076849721926 and L.orc_f64_exp(r32, 1 << 32) == new(1)  # :150-154
    assert L.orc_f64_get_root_of_unity(31) == L.orc_f64_exp(r32, 2)        # :156-159
    assert val(L.orc_f64_get_root_of_unity(6)) == 8  # omega_64 = 2^3


def test_f128_reference_known_answers(orc):
    """math/src/field/f128/tests.rs:19-117 (literal operands)."""
    M = P128
    op = orc.f128_op
    assert op("add", 2, 3) == 5 and op("add", M - 1, 1) == 0 and op("add", M - 1, 2) == 1
    assert op("sub", 5, 3) == 2 and op("sub", 3, 5) == M - 2
    assert op("mul", 5, 3) == 15 and op("mul", M - 1, M - 1) == 1
    assert op("mul", M - 1, 2) == M - 2 and op("mul", M - 1, 4) == M - 4
    assert op("mul", (M + 1) // 2, 2) == 1
    assert op("inv", 1) == 1 and op("inv", 0) == 0
    rng = np.random.default_rng(3)
    for v in orc.f128_to_ints(rand_f128(rng, 200)):
        assert op("mul", v, op("inv", v)) == (1 if v else 0)
    r40 = orc.f128_root_of_unity(40)
    assert r40 == 23953097886125630542083529559205016746 and pow(r40, 1 << 40, M) == 1


# ------------------------------------------------------------------------------------------------ golden: the path
@pytest.mark.parametrize("case", G.load("lde_commit_small.json"), ids=lambda c: c["name"])
def test_commit_golden(orc, case):
    field, fid, ext = case["field"], G.field_id(case["field"]), case["ext"]
    traces = [[G.to_mem(field, col) for col in tr] for tr in case["traces"]]
    for threads in (1, 3):
        got = orc.build_trace_commitment(fid, traces, ext, case["log2_trace_len"], case["log2_blowup"],
                                         int(case["offset"]), threads=threads)
        rw = orc.row_width(len(traces[0]), ext)
        for t in range(len(traces)):
            for c in range(len(traces[t])):
                assert np.array_equal(got["polys"][t][c], G.to_mem(field, case["polys"][t][c]))
            assert np.array_equal(got["lde"][t], G.lde_to_mem(field, case["lde"][t], rw))
        assert G.hexrows(got["leaves"]) == case["leaves"]
        assert G.hexrows(got["nodes"]) == case["nodes"]
        assert got["root"].hex() == case["root"]
    if len(traces) == 1:  # the constraint-commitment entry takes coefficient columns
        polys = [G.to_mem(field, col) for col in case["polys"][0]]
        got = orc.build_constraint_commitment(fid, polys, ext, case["log2_trace_len"], case["log2_blowup"],
                                              int(case["offset"]))
        assert got["root"].hex() == case["root"]


# ------------------------------------------------------------------------------------------------ reference-shaped checks
@pytest.mark.parametrize("field", [F64, F128])
@pytest.mark.parametrize("n", [4, 8, 16, 1024])
def test_fft_equals_naive_evaluation(orc, field, n):
    """math/src/fft/tests.rs:19-60: fft_in_place + permute == polynom::eval_many over the domain."""
    rng = np.random.default_rng(n)
    p = rand_f64(rng, n) if field == F64 else rand_f128(rng, n)
    tw = orc.get_twiddles(field, n)
    lg = n.bit_length() - 1
    if field == F64:
        g = orc.lib().orc_f64_get_root_of_unity(lg)
        one = orc.lib().orc_f64_new(1)
        dom = np.empty(n, dtype=np.uint64)
        acc = one
        for i in range(n):
            dom[i] = acc
            acc = orc.lib().orc_f64_mul(acc, g)
    else:
        g = orc.f128_root_of_unity(lg)
        dom = orc.f128_from_ints([pow(g, i, P128) for i in range(n)])
    want = orc.eval_many(field, p, dom)
    got = p.copy()
    orc.fft_in_place(field, got, n, 1, tw)
    orc.permute(field, got, n)
    assert np.array_equal(got, want)
    got2 = p.copy()
    orc.evaluate_poly(field, got2, n, 1, tw)
    assert np.array_equal(got2, want)
    # interpolate back (doctest fft/mod.rs:253-273)
    orc.interpolate_poly(field, got2, n, 1, orc.get_twiddles(field, n, inverse=True))
    assert np.array_equal(got2, p)


def test_twiddles_are_permuted_powers(orc):
    """math/src/fft/tests.rs:62-72."""
    n = 2048
    tw = orc.get_twiddles(F128, n)
    g = orc.f128_root_of_unity(11)
    want = orc.f128_from_ints([pow(g, i, P128) for i in range(n // 2)])
    orc.permute(F128, want, n // 2)
    assert np.array_equal(tw, want)
    assert tw.shape[0] == n // 2


def test_row_matrix_lde_equals_column_evaluation(orc):
    """prover/src/matrix/tests.rs:13-40: 64 polys of size 256, blowup 8, offset GENERATOR; every column of the
    row-major LDE equals evaluate_poly_with_offset of that column (and the naive evaluation on the coset)."""
    rng = np.random.default_rng(64)
    n, blowup, ncols = 256, 8, 64
    polys = [rand_f64(rng, n) for _ in range(ncols)]
    lde = orc.evaluate_polys_over(F64, polys, 1, 8, 3, 7)
    tw = orc.get_twiddles(F64, n)
    L = orc.lib()
    off = L.orc_f64_new(7)
    for c in range(ncols):
        col = orc.evaluate_poly_with_offset(F64, polys[c], n, 1, tw, off, blowup)
        assert np.array_equal(lde[:, c], col)
    # naive: P(offset * g^j) for a few columns
    g = L.orc_f64_get_root_of_unity(11)
    xs = np.empty(n * blowup, dtype=np.uint64)
    acc = off
    for j in range(n * blowup):
        xs[j] = acc
        acc = L.orc_f64_mul(acc, g)
    for c in (0, 17, 63):
        assert np.array_equal(orc.eval_many(F64, polys[c], xs), lde[:, c])
    # multi-threaded build is identical
    assert np.array_equal(orc.evaluate_polys_over(F64, polys, 1, 8, 3, 7, threads=4), lde)


def test_interpolate_with_offset_roundtrip(orc):
    """math/src/fft/mod.rs:339-361 doctest shape: evaluate on a coset (blowup 1 equivalent) and interpolate back."""
    rng = np.random.default_rng(5)
    n = 512
    for field, off in ((F64, 7), (F128, 3)):
        p = rand_f64(rng, n) if field == F64 else rand_f128(rng, n)
        tw = orc.get_twiddles(field, n)
        off_mem = orc.lib().orc_f64_new(off) if field == F64 else off
        ev = orc.evaluate_poly_with_offset(field, p, n, 1, tw, off_mem, 2)
        # even-indexed LDE points form the coset of size n with the same offset
        sub = np.ascontiguousarray(ev[0::2])
        orc.interpolate_poly_with_offset(field, sub, n, 1, orc.get_twiddles(field, n, inverse=True), off_mem)
        assert np.array_equal(sub, p)


def test_merkle_is_nested_merges(orc):
    """crypto/src/merkle/tests.rs:67-92."""
    rng = np.random.default_rng(8)
    leaves = rng.integers(0, 256, size=(8, 32), dtype=np.uint8)
    lv = [bytes(x) for x in leaves]
    m = orc.merge
    n4 = [m(lv[0], lv[1]), m(lv[2], lv[3]), m(lv[4], lv[5]), m(lv[6], lv[7])]
    n2 = [m(n4[0], n4[1]), m(n4[2], n4[3])]
    root = m(n2[0], n2[1])
    nodes = orc.build_merkle_nodes(leaves)
    assert [bytes(x) for x in nodes] == [bytes(32), root, n2[0], n2[1], n4[0], n4[1], n4[2], n4[3]]
    with pytest.raises(ValueError):
        orc.build_merkle_nodes(leaves[:1])   # merkle/mod.rs:118-120
    with pytest.raises(ValueError):
        orc.build_merkle_nodes(leaves[:6])   # :121-123
    big = rng.integers(0, 256, size=(4096, 32), dtype=np.uint8)
    assert np.array_equal(orc.build_merkle_nodes(big), orc.build_merkle_nodes(big, threads=8))  # concurrent.rs:84-90


def test_hash_padding_sensitivity(orc):
    """crypto/src/hash/blake/tests.rs:11-30."""
    assert orc.blake3(bytes([1, 2, 3])) != orc.blake3(bytes([1, 2, 3, 0]))
    e = np.array([orc.lib().orc_f64_new(v) for v in (1, 2, 3)], dtype=np.uint64)
    e0 = np.array([orc.lib().orc_f64_new(v) for v in (1, 2, 3, 0)], dtype=np.uint64)
    assert orc.hash_elements(F64, e) != orc.hash_elements(F64, e0)
    # hash_elements of f64 hashes canonical little-endian bytes (f64/mod.rs:605-610)
    assert orc.hash_elements(F64, e) == orc.blake3(b"".join(int(v).to_bytes(8, "little") for v in (1, 2, 3)))
    # f128 hashes the raw element bytes (blake/mod.rs:47-51)
    x = orc.f128_from_ints([5, 2**100 + 3])
    assert orc.hash_elements(F128, x) == orc.blake3((5).to_bytes(16, "little") + (2**100 + 3).to_bytes(16, "little"))


def test_oracle_rejects_bad_parameters(orc):
    col = [np.zeros(8, dtype=np.uint64)]
    with pytest.raises(ValueError):
        orc.build_trace_commitment(F64, [col], 1, 2, 1, 7)     # trace too short
    with pytest.raises(ValueError):
        orc.build_trace_commitment(F64, [col], 1, 3, 0, 7)     # blowup 1
    with pytest.raises(ValueError):
        orc.build_trace_commitment(F64, [col], 1, 3, 1, 0)     # zero offset
    with pytest.raises(ValueError):
        orc.build_trace_commitment(F128, [col], 3, 3, 1, 3)    # no cubic extension over f128


# ------------------------------------------------------------------------------------------------ FRI layer pieces
@pytest.mark.parametrize("N", [2, 4, 8, 16])
def test_apply_drp_equals_coefficient_folding(orc, N):
    """fri/src/folding/mod.rs:40-84 (doctest): DRP of the evaluations == evaluations of the folded polynomial."""
    from oracle import pyref as P
    import random
    F = P.Field("f64")
    p, rnd = F.p, random.Random(N)
    n, off = 64, 7
    poly = [rnd.randrange(p) for _ in range(n // 2)] + [0] * (n // 2)
    g = F.root_of_unity(6)
    ev = [P.poly_eval(poly, off * pow(g, i, p) % p, p) for i in range(n)]
    alpha = rnd.randrange(p)
    folded = [sum(pow(alpha, j, p) * poly[N * i + j] for j in range(N)) % p for i in range(n // N)]
    g2 = F.root_of_unity((n // N).bit_length() - 1)
    want = [P.poly_eval(folded, pow(off, N, p) * pow(g2, i, p) % p, p) for i in range(n // N)]
    mem = lambda v: np.array([F.to_mem(x) for x in v], dtype=np.uint64)  # noqa: E731
    tr = orc.transpose_slice(F64, mem(ev), n, 1, N)
    assert [int(v) for v in tr[:N]] == [F.to_mem(ev[j * (n // N)]) for j in range(N)]   # utils/core/src/lib.rs:206-227
    got = orc.apply_drp(F64, tr, n // N, 1, N, off, mem([alpha]))
    assert [F.from_mem(int(v)) for v in got] == want


def test_extension_products(orc):
    """f64 quadratic x^2-x+2 / cubic x^3-x-1 (f64/mod.rs:401-472), f128 quadratic x^2-x-1 (f128/mod.rs:273-279),
    checked against schoolbook polynomial arithmetic on Python integers; one/zero identities (f64/tests.rs:246-260)."""
    import random
    rnd = random.Random(9)
    p, q = P64, P128
    mem = lambda v: np.array([(x << 64) % p for x in v], dtype=np.uint64)  # noqa: E731
    val = lambda a: [int(x) * pow(2**64, -1, p) % p for x in a]  # noqa: E731
    for _ in range(100):
        a = [rnd.randrange(p) for _ in range(3)]
        b = [rnd.randrange(p) for _ in range(3)]
        c0, c1, c2 = a[0] * b[0], a[0] * b[1] + a[1] * b[0], a[1] * b[1]
        assert val(orc.ext_mul(F64, 2, mem(a[:2]), mem(b[:2]))) == [(c0 - 2 * c2) % p, (c1 + c2) % p]
        c = [0] * 5
        for i in range(3):
            for j in range(3):
                c[i + j] += a[i] * b[j]
        assert val(orc.ext_mul(F64, 3, mem(a), mem(b))) == [(c[0] + c[3]) % p, (c[1] + c[3] + c[4]) % p, (c[2] + c[4]) % p]
        x = [rnd.randrange(q) for _ in range(2)]
        y = [rnd.randrange(q) for _ in range(2)]
        got = orc.f128_to_ints(orc.ext_mul(F128, 2, orc.f128_from_ints(x), orc.f128_from_ints(y)))
        assert got == [(x[0] * y[0] + x[1] * y[1]) % q, (x[0] * y[1] + x[1] * y[0] + x[1] * y[1]) % q]
    r = mem([5, 9])
    assert np.array_equal(orc.ext_mul(F64, 2, r, mem([1, 0])), r)
    assert not orc.ext_mul(F64, 2, r, mem([0, 0])).any()


@pytest.mark.parametrize("field,ext", [(F64, 1), (F64, 2), (F64, 3), (F128, 1), (F128, 2)])
def test_deep_composition_is_the_sum_of_the_quotients(orc, field, ext):
    """DeepCompositionPoly (prover/src/composer/mod.rs:52-66,156-167): the composed polynomial D satisfies, at a random
    point x of E,  D(x) (x - z)(x - z g) = sum_i cc_i [(T_i(x) - T_i(z))(x - z g) + (T_i(x) - T_i(z g))(x - z)]
    + sum_i cc'_i (H_i(x) - H_i(z))(x - z g) -- schoolbook arithmetic on Python integers, independent of the oracle's
    field code."""
    import random
    rnd = random.Random(field * 10 + ext)
    p = P64 if field == F64 else P128

    def emul(a, b):
        c = [0] * (2 * ext - 1)
        for i in range(ext):
            for j in range(ext):
                c[i + j] += a[i] * b[j]
        if ext == 2 and field == F64:      # x^2 = x - 2
            c = [c[0] - 2 * c[2], c[1] + c[2]]
        elif ext == 2:                     # x^2 = x + 1
            c = [c[0] + c[2], c[1] + c[2]]
        elif ext == 3:                     # x^3 = x + 1, x^4 = x^2 + x
            c = [c[0] + c[3], c[1] + c[3] + c[4], c[2] + c[4]]
        return [v % p for v in c]

    eadd = lambda a, b: [(x + y) % p for x, y in zip(a, b)]  # noqa: E731
    esub = lambda a, b: [(x - y) % p for x, y in zip(a, b)]  # noqa: E731

    def horner(coeffs, x):  # coeffs: list of E
        acc = [0] * ext
        for c in reversed(coeffs):
            acc = eadd(emul(acc, x), c)
        return acc

    if field == F64:
        to_mem = lambda vals: np.array([(v << 64) % p for v in vals], dtype=np.uint64)  # noqa: E731
        from_mem = lambda a: [int(v) * pow(2**64, -1, p) % p for v in np.asarray(a).reshape(-1)]  # noqa: E731
    else:
        to_mem = orc.f128_from_ints
        from_mem = lambda a: orc.f128_to_ints(np.asarray(a).reshape(-1, 2))  # noqa: E731
    embed = lambda v: [v] + [0] * (ext - 1)  # noqa: E731
    n = 32
    rand_e = lambda: [rnd.randrange(p) for _ in range(ext)]  # noqa: E731
    # two packed traces: main columns over the base field, auxiliary ones over E; two composition columns
    tables_int = []
    for _ in range(2):
        main = [[embed(rnd.randrange(p)) for _ in range(n)] for _ in range(3)]
        aux = [[rand_e() for _ in range(n)] for _ in range(2 if ext > 1 else 0)]
        tables_int.append((main, aux))
    cons_int = [[rand_e() for _ in range(n)] for _ in range(2)]
    z = rand_e()
    cc_t = [[rand_e() for _ in range(len(m) + len(a))] for m, a in tables_int]
    cc_c = [rand_e() for _ in cons_int]
    tables = [[(to_mem([c[0] for c in col]), 1) for col in m] + [(to_mem([v for c in col for v in c]), ext) for col in a]
              for m, a in tables_int]
    cons = [to_mem([v for c in col for v in c]) for col in cons_int]
    d = orc.deep_compose(field, ext, n, tables, cons, to_mem(z), [to_mem(c) for tab in cc_t for c in tab], [to_mem(c) for c in cc_c])
    d_int = from_mem(d)
    d_e = [d_int[i * ext:(i + 1) * ext] for i in range(n)]
    assert d_e[n - 1] == [0] * ext and d_e[n - 2] != [0] * ext      # degree n - 2 (composer/mod.rs:151)
    if field == F64:  # the generator of the trace domain: the reference's get_root_of_unity(log2 n)
        g = int(orc.lib().orc_f64_get_root_of_unity(5)) * pow(2**64, -1, p) % p
    else:
        g = orc.f128_root_of_unity(5)
    assert pow(g, n, p) == 1 and pow(g, n // 2, p) == p - 1
    zg = emul(z, embed(g))
    for _ in range(3):
        x = rand_e()
        lhs = emul(horner(d_e, x), emul(esub(x, z), esub(x, zg)))
        rhs = [0] * ext
        for (m, a), cc in zip(tables_int, cc_t):
            for col, k in zip(m + a, cc):
                tx, tz, tzg = horner(col, x), horner(col, z), horner(col, zg)
                term = eadd(emul(esub(tx, tz), esub(x, zg)), emul(esub(tx, tzg), esub(x, z)))
                rhs = eadd(rhs, emul(k, term))
        for col, k in zip(cons_int, cc_c):
            rhs = eadd(rhs, emul(k, emul(esub(horner(col, x), horner(col, z)), esub(x, zg))))
        assert lhs == rhs


@pytest.mark.parametrize("field", [F64, F128])
def test_acc_column_divides_by_the_divisor(orc, field):
    """acc_column + get_inv_evaluation (prover/src/constraints/evaluation_table.rs:335-426): for every point x_i = offset g^i
    of the constraint evaluation domain, result[i] (x_i^a - b) = column[i] prod_k (x_i - e_k) -- on Python integers; and the
    zero of a divisor's numerator inside the domain leaves a zero (math::batch_inversion ignores zeros)."""
    import random
    rnd = random.Random(field)
    p = P64 if field == F64 else P128
    ce, ext, off = 64, 2, (7 if field == F64 else 3)
    if field == F64:
        to_mem = lambda vals: np.array([(v << 64) % p for v in vals], dtype=np.uint64)  # noqa: E731
        from_mem = lambda a: [int(v) * pow(2**64, -1, p) % p for v in np.asarray(a).reshape(-1)]  # noqa: E731
        g = int(orc.lib().orc_f64_get_root_of_unity(6)) * pow(2**64, -1, p) % p
    else:
        to_mem = orc.f128_from_ints
        from_mem = lambda a: orc.f128_to_ints(np.asarray(a).reshape(-1, 2))  # noqa: E731
        g = orc.f128_root_of_unity(6)
    for a, n_ex in [(16, 2), (1, 0), (4, 0), (64, 1)]:
        b = rnd.randrange(1, p)
        ex = [rnd.randrange(p) for _ in range(n_ex)]
        col = [rnd.randrange(p) for _ in range(ce * ext)]
        got = from_mem(orc.combine_evaluation_table(field, ext, [to_mem(col)], [(a, to_mem([b]), to_mem(ex) if ex else None)], off))
        for i in range(ce):
            x = off * pow(g, i, p) % p
            e = 1
            for v in ex:
                e = e * (x - v) % p
            for w in range(ext):
                assert got[i * ext + w] * (pow(x, a, p) - b) % p == col[i * ext + w] * e % p
    # b = (offset g^5)^1: the numerator vanishes at i = 5 -> the inverse there is taken as zero
    b = off * pow(g, 5, p) % p
    col = [rnd.randrange(1, p) for _ in range(ce)]
    got = from_mem(orc.combine_evaluation_table(field, 1, [to_mem(col)], [(1, to_mem([b]), None)], off))
    assert got[5] == 0 and all(v != 0 for i, v in enumerate(got) if i != 5)


# ------------------------------------------------------------------------------------ golden: the reference's own inputs
def test_reference_literal_inputs_golden(orc):
    """LEAVES4 / LEAVES8 of crypto/src/merkle/tests.rs:13-65 and the polynomial of fri/src/prover/tests.rs:58-69 as
    inputs; expected values from Python big integers + the official BLAKE3 (tests/golden/README.md)."""
    g = G.load("reference_inputs.json")
    for t in g["trees"]:
        leaves = np.frombuffer(b"".join(bytes.fromhex(x) for x in t["leaves"]), dtype=np.uint8).reshape(-1, 32)
        nodes = orc.build_merkle_nodes(leaves)
        assert G.hexrows(nodes) == t["nodes"] and bytes(nodes[1]).hex() == t["root"]
        n = len(t["leaves"])
        # the structure the reference's new_tree test asserts (tests.rs:67-92): root = nested hash_2x1
        lv = [bytes.fromhex(x) for x in t["leaves"]]
        level = lv
        while len(level) > 1:
            level = [orc.merge(level[2 * i], level[2 * i + 1]) for i in range(len(level) // 2)]
        assert level[0].hex() == t["root"]
        for idx in range(n):                                   # prove (tests.rs:94-135) and verify
            proof = orc.merkle_prove(nodes, leaves, idx)
            assert [x.hex() for x in proof] == t["proofs"][str(idx)]
            assert orc.merkle_verify(bytes(nodes[1]), idx, proof)
    f = g["fri"]
    tl, blowup, folding = f["trace_length"], f["lde_blowup"], f["folding"]
    n = tl * blowup
    p = np.zeros((n, 2), dtype=np.uint64)
    p[:tl, 0] = np.arange(tl, dtype=np.uint64)                  # build_evaluations: coefficients 0..trace_length-1
    ev = p.reshape(-1).copy()
    orc.evaluate_poly(orc.F128, ev, n, 1, orc.get_twiddles(orc.F128, n))
    assert orc.f128_to_ints(ev.reshape(-1, 2)) == [int(v) for v in f["evaluations"]]
    layer = orc.fri_layer_commit(orc.F128, ev, n, 1, folding)
    assert orc.f128_to_ints(layer["transposed"].reshape(-1, 2)) == [int(v) for r in f["transposed"] for v in r]
    assert G.hexrows(layer["leaves"]) == f["leaves"] and layer["root"].hex() == f["root"]


def test_blake3_simd_compression_equals_the_scalar_definition(orc):
    """oracle/blake3_ref.c computes one compression with the state's columns side by side in SSE vectors (the speed of the
    `blake3` crate the reference links); the scalar transcription of the specification stays as its cross-check."""
    rng = np.random.default_rng(2024)
    for _ in range(500):
        cv = rng.integers(0, 2**32, 8, dtype=np.uint32)
        blk = rng.integers(0, 2**32, 16, dtype=np.uint32)
        a, b = orc.blake3_compress_both(cv, blk, int(rng.integers(0, 2**63)), int(rng.integers(0, 65)), int(rng.integers(0, 16)))
        assert np.array_equal(a, b)


def test_phase_clock_of_the_commitment(orc):
    rng = np.random.default_rng(1)
    cols = [rng.integers(0, 2**62, size=1 << 10, dtype=np.uint64) for _ in range(4)]
    orc.build_trace_commitment(1, [cols], 1, 10, 3, 7)
    ph = orc.last_phase_ms()
    assert len(ph) == 4 and all(x >= 0 for x in ph) and sum(ph) > 0


def test_reference_held_known_answers(orc):
    """The known-answer vectors the reference's OWN tests hold (tests/golden/reference_kat.json; inputs and expected outputs
    are the literals of math/src/field/f64/tests.rs:251-280 / :321-378, math/src/field/f128/tests.rs:165-181 and
    math/src/polynom/tests.rs:178-207) against the oracle's extension products, canonical bytes + leaf hash, and syn_div."""
    g = G.load("reference_kat.json")
    ints = lambda v: [int(x) for x in v]  # noqa: E731
    for ext, key in ((2, "f64_quad_mul"), (3, "f64_cube_mul")):
        for c in g[key]:
            got = orc.f64_as_int(orc.ext_mul(F64, ext, orc.f64_new(ints(c["a"])), orc.f64_new(ints(c["b"]))))
            assert [int(x) for x in got] == ints(c["expected"]), c["where"]
    c = g["f128_elements_as_bytes"]
    elems = orc.f128_from_ints(ints(c["source"]))
    assert elems.tobytes() == bytes(c["expected_bytes"])            # the canonical u128s, little-endian: elements_as_bytes
    assert orc.hash_elements(F128, elems).hex() == c["blake3_256_of_expected_bytes"]
    for c in g["f128_syn_div"]:
        q = orc.f128_to_ints(orc.syn_div(F128, 1, orc.f128_from_ints(ints(c["poly"])), orc.f128_from_ints([int(c["b"])])))
        want = ints(c["expected"])
        assert q[:len(want)] == want and not any(q[len(want):]), c["where"]
    c = g["f128_composition_segment"]
    cols = orc.segment(F128, 1, orc.f128_from_ints(c["values"]), c["trace_len"], c["num_cols"])
    assert [orc.f128_to_ints(col) for col in cols] == c["expected"], c["where"]
    c = g["transpose_slice"]
    tr = orc.transpose_slice(F64, np.array(c["values"], dtype=np.uint64), len(c["values"]), 1, c["N"])
    assert tr.reshape(-1, c["N"]).tolist() == c["expected"], c["where"]

