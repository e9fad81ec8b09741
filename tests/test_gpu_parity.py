"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Mirrors the reference's own tests for this path:
  math/src/fft/tests.rs:19-60            fft == naive evaluation            -> test_fft_*
  prover/src/matrix/tests.rs:13-40       row-matrix LDE == per-column eval  -> test_evaluate_polys_over
  prover/src/trace/tests.rs:42-128       LDE / commitment of a trace        -> test_trace_commit
  crypto/src/merkle/tests.rs:67-92       tree == nested merges              -> test_merkle_build
"""
import numpy as np
import pytest

from conftest import rand_cols, rand_f64, rand_f128

pytestmark = pytest.mark.gpu

F64, F128 = 1, 2


def _rand(rng, field, n):
    return rand_f64(rng, n) if field == F64 else rand_f128(rng, n)


@pytest.mark.parametrize("field", [F64, F128])
@pytest.mark.parametrize("ext", [1, 2, 3])
@pytest.mark.parametrize("logn", [1, 2, 3, 4, 5, 7, 10, 11, 12, 13])
def test_fft_evaluate_and_interpolate(ctx, orc, field, ext, logn):
    if field == F128 and ext == 3:
        pytest.skip("f128 has no cubic extension (f128/mod.rs:296-314)")
    rng = np.random.default_rng(1000 * field + 100 * ext + logn)
    n = 1 << logn
    p = _rand(rng, field, n * ext)
    tw = orc.get_twiddles(field, n)
    inv_tw = orc.get_twiddles(field, n, inverse=True)

    want = p.copy()
    orc.evaluate_poly(field, want, n, ext, tw)
    got = ctx.fft_evaluate_poly(field, ext, p)
    assert np.array_equal(got, want)

    want_i = p.copy()
    orc.interpolate_poly(field, want_i, n, ext, inv_tw)
    got_i = ctx.fft_interpolate_poly(field, ext, p)
    assert np.array_equal(got_i, want_i)
    # round trip (fft/tests.rs style identity)
    assert np.array_equal(ctx.fft_interpolate_poly(field, ext, got), p)


@pytest.mark.parametrize("field,offset", [(F64, 7), (F64, 12345678901234567), (F128, 3), (F128, 2**100 + 17)])
@pytest.mark.parametrize("ext", [1, 2, 3])
@pytest.mark.parametrize("logn", [3, 6, 10, 12])
def test_fft_with_offset(ctx, orc, field, offset, ext, logn):
    if field == F128 and ext == 3:
        pytest.skip("f128 has no cubic extension (f128/mod.rs:296-314)")
    rng = np.random.default_rng(7 * logn + ext)
    n = 1 << logn
    p = _rand(rng, field, n * ext)
    tw = orc.get_twiddles(field, n)
    inv_tw = orc.get_twiddles(field, n, inverse=True)
    off_mem = orc.lib().orc_f64_new(offset) if field == F64 else offset

    want = p.copy()
    orc.interpolate_poly_with_offset(field, want, n, ext, inv_tw, off_mem)
    got = ctx.fft_interpolate_poly_with_offset(field, ext, p, offset)
    assert np.array_equal(got, want)

    for blowup in (2, 8):
        want_e = orc.evaluate_poly_with_offset(field, p, n, ext, tw, off_mem, blowup)
        got_e = ctx.fft_evaluate_poly_with_offset(field, ext, p, offset, blowup)
        assert np.array_equal(got_e, want_e)


CASES = [
    # field, ext, logR, logB, n_cols, n_traces
    (F64, 1, 3, 1, 1, 1),      # smallest legal trace
    (F64, 1, 4, 3, 2, 1),      # fib-like 2 columns
    (F64, 1, 8, 3, 8, 1),      # one full segment
    (F64, 1, 8, 3, 64, 1),     # matrix/tests.rs:13-40 shape (256 rows, 64 polys, blowup 8)
    (F64, 1, 10, 2, 3, 1),     # ragged width, single pass
    (F64, 1, 11, 3, 8, 1),     # two passes
    (F64, 1, 12, 3, 10, 1),    # two passes, ragged second segment
    (F64, 1, 13, 1, 17, 2),    # starkpack: two traces, three segments
    (F64, 2, 11, 3, 3, 1),     # quadratic extension columns
    (F64, 3, 11, 2, 2, 1),     # cubic extension columns
    (F64, 1, 6, 7, 5, 3),      # blowup 128, three packed traces
    (F128, 1, 3, 1, 1, 1),
    (F128, 1, 10, 3, 10, 1),   # do_work shape (examples/src/do_work: 10 columns)
    (F128, 1, 12, 3, 10, 2),   # two passes, packed
    (F128, 2, 11, 2, 3, 1),    # quadratic extension over f128
    (F128, 1, 7, 3, 10, 8),    # 8 packed traces: 1280-byte combined rows -> multi-chunk BLAKE3
    (F64, 1, 8, 1, 255, 1),    # MAX_TRACE_WIDTH columns (air/src/air/trace_info.rs:37): 32 segments, 2040-byte rows
    (F64, 1, 11, 7, 2, 1),     # blowup 128 on a two-pass transform, narrow matrix (coset-packed lanes)
    (F128, 2, 4, 1, 127, 2),   # 508 base columns per row x 2 traces: 16 KiB rows, 16 BLAKE3 chunks each
    # one segment, one trace, two passes: leaves come from the persistent fused last pass (k_seg_last_hash)
    (F128, 1, 12, 2, 3, 1),    # f128: 48-byte leaves out of 64-byte tile rows, LDE rows padded to 8 elements
    (F64, 1, 12, 7, 8, 1),     # blowup 128: 128 cosets per row block
    (F64, 1, 11, 1, 7, 1),     # blowup 2: two cosets
    (F64, 3, 11, 3, 2, 1),     # cubic extension: 6 base columns
    # two passes, rows longer than one BLAKE3 chunk: the persistent last pass leaves chunk chaining values
    (F64, 1, 12, 1, 200, 1),   # 1600-byte rows: 25 segments, chunks of 16 + 9 blocks
    (F64, 1, 12, 2, 255, 1),   # 2040 bytes: the last block of the second chunk is short (56 bytes)
    (F64, 1, 12, 1, 17, 9),    # 9 packed traces x 17 columns: 153 lanes, padded rows of 24 elements each
    (F64, 1, 12, 1, 3, 50),    # 50 narrow traces: whole-row stores + chunked hashing
    (F128, 1, 11, 1, 10, 20),  # 20 packed f128 do_work traces: 50 segments, 3200-byte rows (4 chunks)
    (F128, 1, 11, 2, 70, 1),   # 1120-byte rows: 18 segments, the second chunk has two blocks
    (F64, 2, 12, 1, 65, 1),    # 130 base columns (quadratic extension): 17 segments -> a one-block second chunk
    # few, very long rows: chunk chaining values merged by 16 lanes per row (k_hash_merge_chunks_par, 32..128 chunks)
    (F128, 1, 3, 3, 10, 512),  # the reference's do_work default width: 512 packed traces, 80 chunks per row
    (F128, 1, 3, 1, 10, 206),  # 33 chunks: an unpaired chaining value is carried up at several levels
    (F128, 1, 4, 1, 10, 250),  # 40 chunks
    (F64, 1, 4, 1, 255, 17),   # 34 chunks of f64 rows
    # single-pass plans with rows longer than a chunk and an EVEN number of f64 columns: k_hash_chunks_staged on 16-byte units of two
    # f64 elements (odd widths -- 255 above -- keep k_hash_chunks); rows / chunk groups that do not fill a wave's 16 rows x 4 chunks
    (F64, 1, 9, 1, 6, 40),     # 40 traces x 6 columns: 1920-byte rows, a short second chunk, blocks that straddle two traces' rows
    (F64, 1, 3, 2, 8, 40),     # 32 LDE rows x 3 chunks (2560 bytes): fewer chunks than a wave's four slots
    (F64, 2, 5, 1, 5, 13),     # quadratic extension: 10 base columns per trace, 1040-byte rows (one 16-byte unit in the second chunk)
    (F128, 1, 5, 2, 3, 30),    # f128: 1440-byte rows, three columns per trace (blocks straddle up to three traces)
]


@pytest.mark.parametrize("field,ext,logR,logB,n_cols,n_traces", CASES)
def test_trace_commit(ctx, orc, capi, field, ext, logR, logB, n_cols, n_traces):
    rng = np.random.default_rng(hash((field, ext, logR, logB, n_cols, n_traces)) % 2**32)
    R = 1 << logR
    traces = [rand_cols(rng, field, n_cols, R * ext) for _ in range(n_traces)]
    offset = 7 if field == F64 else 3
    want = orc.build_trace_commitment(field, traces, ext, logR, logB, offset)
    params = capi.make_params(field, ext, logR, logB, n_cols, n_traces)
    got = ctx.trace_commit(params, [c for t in traces for c in t])
    for t in range(n_traces):
        for c in range(n_cols):
            assert np.array_equal(got["polys"][t * n_cols + c], want["polys"][t][c]), f"poly {t},{c}"
        assert np.array_equal(got["lde"][t], want["lde"][t]), f"lde {t}"
    assert np.array_equal(got["leaves"], want["leaves"])
    assert np.array_equal(got["nodes"], want["nodes"])
    assert got["root"] == want["root"]


@pytest.mark.parametrize("field,ext,logR,logB,n_cols", [
    (F64, 1, 10, 3, 4), (F64, 2, 12, 3, 8), (F64, 3, 9, 3, 3), (F128, 2, 11, 3, 4), (F128, 1, 8, 4, 1)])
def test_constraint_commit(ctx, orc, capi, field, ext, logR, logB, n_cols):
    rng = np.random.default_rng(99 + logR + n_cols)
    polys = rand_cols(rng, field, n_cols, (1 << logR) * ext)
    offset = 7 if field == F64 else 3
    want = orc.build_constraint_commitment(field, polys, ext, logR, logB, offset)
    params = capi.make_params(field, ext, logR, logB, n_cols, 1)
    got = ctx.constraint_commit(params, polys)
    assert np.array_equal(got["lde"], want["lde"])
    assert np.array_equal(got["leaves"], want["leaves"])
    assert np.array_equal(got["nodes"], want["nodes"])
    assert got["root"] == want["root"]
    assert np.array_equal(ctx.evaluate_polys_over(params, polys), want["lde"])


@pytest.mark.parametrize("field", [F64, F128])
@pytest.mark.parametrize("row_elems", [1, 3, 8, 9, 64, 127, 128, 129, 200, 600])
def test_hash_rows(ctx, orc, field, row_elems):
    rng = np.random.default_rng(row_elems)
    n_rows = 37
    rows = _rand(rng, field, n_rows * row_elems)
    got = ctx.hash_rows(field, rows, n_rows, row_elems)
    r = rows.reshape(n_rows, -1)
    for i in range(n_rows):
        assert bytes(got[i]) == orc.hash_elements(field, r[i]), f"row {i}"


@pytest.mark.parametrize("log_leaves", [1, 2, 3, 8, 9, 10, 13, 17])
def test_merkle_build(ctx, orc, log_leaves):
    rng = np.random.default_rng(log_leaves)
    leaves = rng.integers(0, 256, size=(1 << log_leaves, 32), dtype=np.uint8)
    got = ctx.merkle_build(leaves)
    want = orc.build_merkle_nodes(leaves)
    assert np.array_equal(got, want)
    assert not got[0].any()


def test_error_codes(ctx, capi):
    """Preconditions the reference asserts on (SURVEY.md §8b 'Error convention') come back as status codes."""
    col = np.zeros(8, dtype=np.uint64)

    def code(**kw):
        args = dict(field=F64, ext_degree=1, log2_trace_len=3, log2_blowup=1, n_cols=1, n_traces=1)
        args.update(kw)
        off = args.pop("offset", None)
        p = capi.make_params(args["field"], args["ext_degree"], args["log2_trace_len"], args["log2_blowup"],
                             args["n_cols"], args["n_traces"], off)
        try:
            ctx.trace_commit(p, [col] * (args["n_cols"] * max(1, args["n_traces"])), want_lde=False, want_polys=False)
        except capi.WfError as e:
            return e.code
        return 0

    assert code() == 0
    assert code(field=9) == -10
    assert code(ext_degree=4) == -11
    assert code(field=F128, ext_degree=3) == -11
    assert code(log2_trace_len=2) == -12
    assert code(log2_blowup=0) == -13
    assert code(log2_blowup=8) == -13
    assert code(log2_trace_len=30, log2_blowup=3) == -14
    assert code(n_cols=0) == -15
    assert code(n_traces=0) == -16
    assert code(offset=0) == -17
    assert code(offset=2**64 - 2**32 + 1) == -17
    with pytest.raises(capi.WfError) as e:
        ctx.merkle_build(np.zeros((1, 32), dtype=np.uint8))
    assert e.value.code == -18
    with pytest.raises(capi.WfError) as e:
        ctx.merkle_build(np.zeros((6, 32), dtype=np.uint8))
    assert e.value.code == -18


@pytest.mark.parametrize("field,logR,logB,n_cols", [
    (F64, 21, 1, 2),    # digits [11, 10]: a 2^11-row tile is a whole 160 KiB of LDS, 1024 threads per work-group
    (F64, 22, 1, 1),    # segment kernels: three passes [8, 7, 7] instead of two full tiles; column kernels: [11, 11]
    (F64, 23, 1, 1),    # three passes [8, 8, 7]
    (F64, 21, 1, 8),    # full rows, one segment: leaves hashed by the persistent last pass (2^11-row strided tiles)
    (F64, 22, 1, 5),    # the same under a three-pass plan, 5 of 8 lanes used (40-byte leaves, padded rows)
    (F128, 20, 1, 1),   # f128 digits are capped at 10 bits: two full passes [10, 10] (80 KiB tiles, two work-groups per CU)
    (F128, 21, 1, 1),   # three passes [7, 7, 7]
    (F128, 20, 1, 6),   # [10, 10] with two segments (rows of 8 elements, 6 used): fused hashing over the segments
    (F128, 22, 1, 1),   # three passes [8, 7, 7]
])
def test_large_transform_plans(ctx, orc, capi, field, logR, logB, n_cols):
    rng = np.random.default_rng(logR)
    cols = rand_cols(rng, field, n_cols, 1 << logR)
    offset = 7 if field == F64 else 3
    want = orc.build_trace_commitment(field, [cols], 1, logR, logB, offset, threads=16)
    got = ctx.trace_commit(capi.make_params(field, 1, logR, logB, n_cols, 1), cols)
    assert np.array_equal(got["lde"][0], want["lde"][0])
    assert got["root"] == want["root"]
    # the stand-alone transforms use the column-layout kernels: check their multi-pass plans too
    tw_inv = orc.get_twiddles(field, 1 << logR, inverse=True)
    want_i = cols[0].copy()
    orc.interpolate_poly(field, want_i, 1 << logR, 1, tw_inv)
    assert np.array_equal(ctx.fft_interpolate_poly(field, 1, cols[0]), want_i)


def _random_configs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        field = int(rng.integers(1, 3))
        ext = int(rng.integers(1, 4 if field == F64 else 3))
        logR = int(rng.integers(3, 13))
        logB = int(rng.integers(1, 5))
        n_cols = int(rng.integers(1, 21))
        n_traces = int(rng.integers(1, 5))
        if (1 << (logR + logB)) * n_cols * ext * n_traces > (1 << 22):
            n_traces, n_cols = 1, min(n_cols, 8)
        offset = int(rng.integers(2, 2**62))
        out.append((field, ext, logR, logB, n_cols, n_traces, offset))
    return out


@pytest.mark.parametrize("cfg", _random_configs(48, 20240661), ids=lambda c: "-".join(str(x) for x in c[:6]))
def test_random_shapes(ctx, orc, capi, cfg):
    """Seeded sweep over shapes (field, extension, length, blowup, width, packed traces) and random domain offsets."""
    field, ext, logR, logB, n_cols, n_traces, offset = cfg
    rng = np.random.default_rng(offset % 2**32)
    traces = [rand_cols(rng, field, n_cols, (1 << logR) * ext) for _ in range(n_traces)]
    want = orc.build_trace_commitment(field, traces, ext, logR, logB, offset)
    params = capi.make_params(field, ext, logR, logB, n_cols, n_traces, offset)
    got = ctx.trace_commit(params, [c for t in traces for c in t])
    for t in range(n_traces):
        assert np.array_equal(got["lde"][t], want["lde"][t])
        for c in range(n_cols):
            assert np.array_equal(got["polys"][t * n_cols + c], want["polys"][t][c])
    assert np.array_equal(got["nodes"], want["nodes"])


@pytest.mark.parametrize("field,ext,logR,logB,n_cols,n_traces", [
    (F64, 1, 10, 3, 1, 1), (F64, 1, 10, 3, 2, 1), (F64, 1, 10, 3, 3, 1), (F64, 1, 10, 3, 4, 1), (F64, 1, 10, 3, 5, 1),
    (F64, 1, 10, 3, 7, 1), (F64, 2, 10, 3, 1, 1), (F64, 3, 10, 3, 1, 1), (F64, 1, 10, 1, 3, 1), (F64, 1, 10, 2, 1, 1),
    (F64, 1, 10, 3, 11, 1), (F64, 1, 10, 3, 3, 2), (F64, 1, 10, 3, 5, 3), (F64, 1, 13, 3, 2, 1),
    (F128, 1, 10, 3, 1, 1), (F128, 1, 10, 3, 2, 1), (F128, 1, 10, 3, 3, 1), (F128, 2, 10, 3, 1, 1), (F128, 1, 10, 3, 5, 1),
    (F128, 1, 10, 3, 3, 2),
    # narrow traces side by side (STARKPack): whole padded rows written by thread quads; the two-lane case stays packed
    (F64, 1, 10, 3, 1, 4), (F64, 1, 10, 3, 2, 2), (F64, 1, 10, 3, 1, 2), (F64, 1, 10, 3, 1, 3), (F64, 1, 10, 3, 2, 16),
    (F64, 1, 10, 3, 7, 3), (F64, 1, 12, 3, 3, 5), (F64, 2, 10, 3, 1, 5), (F128, 1, 10, 3, 1, 2), (F128, 1, 10, 3, 2, 5),
    (F128, 1, 10, 3, 7, 2)])
def test_padding_lanes_are_written(ctx, orc, capi, field, ext, logR, logB, n_cols, n_traces):
    """The zero padding of LDE rows (segments.rs:65-72) must not depend on what the caller's buffer held: the device
    form is run into an LDE buffer pre-filled with ones and compared in full, padding lanes included."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(hash((field, ext, logR, logB, n_cols, n_traces, "pad")) % 2**32)
    R, N = 1 << logR, 1 << (logR + logB)
    traces = [rand_cols(rng, field, n_cols, R * ext) for _ in range(n_traces)]
    offset = 7 if field == F64 else 3
    want = orc.build_trace_commitment(field, traces, ext, logR, logB, offset)
    params = capi.make_params(field, ext, logR, logB, n_cols, n_traces)
    flat = np.concatenate([np.ascontiguousarray(c).reshape(-1) for t in traces for c in t]).view(np.int64)
    d_trace = torch.from_numpy(flat.copy()).to(dev)
    d_polys = torch.empty_like(d_trace)
    want_lde = np.stack([np.ascontiguousarray(l) for l in want["lde"]])
    d_lde = torch.full((want_lde.size,), -1, dtype=torch.int64, device=dev)
    d_leaves = torch.empty((N, 32), dtype=torch.uint8, device=dev)
    d_nodes = torch.empty((N, 32), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.trace_commit_dev(params, d_trace.data_ptr(), d_polys.data_ptr(), d_lde.data_ptr(), d_leaves.data_ptr(),
                         d_nodes.data_ptr())
    torch.cuda.synchronize()
    got = d_lde.cpu().numpy().view(np.uint64).reshape(want_lde.shape)
    assert np.array_equal(got, want_lde)
    assert np.array_equal(d_leaves.cpu().numpy().reshape(-1), np.ascontiguousarray(want["leaves"]).view(np.uint8).reshape(-1))
