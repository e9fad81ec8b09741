"""GPU: one packed (STARKPack) commitment sharded by coset, as W ranks would compute it.
All "ranks" run one after the other on the single test GPU (the replicated-tree form: wf_trace_commit_shard_dev per
rank, leaf shards concatenated rank-major as an all-gather returns them, reordered, wf_merkle_build_dev); the result
must equal the unsharded commitment bit for bit.  The exchange itself and the distributed-tree form
(wf_trace_commit_sharded_dev) run with real ranks in tests/test_gpu_comm.py."""
import numpy as np
import pytest

from conftest import rand_cols

pytestmark = pytest.mark.gpu
F64, F128 = 1, 2


@pytest.mark.parametrize("field,logR,logB,n_cols,n_traces,world", [
    (F64, 11, 3, 8, 1, 8), (F64, 12, 3, 5, 3, 4), (F64, 8, 3, 10, 2, 2), (F128, 10, 2, 10, 2, 4), (F64, 10, 3, 8, 2, 1),
    # narrow matrices, one coset per rank (never coset-packed) and two (packed): who writes the rows' zero padding differs
    (F64, 10, 3, 3, 1, 8), (F64, 10, 3, 1, 1, 8), (F64, 10, 3, 2, 3, 8), (F64, 10, 3, 2, 1, 4), (F64, 10, 3, 1, 3, 4),
    (F128, 10, 3, 1, 1, 8), (F128, 10, 3, 3, 1, 4), (F128, 10, 3, 1, 2, 4), (F128, 10, 3, 5, 1, 8),
    # rows longer than one BLAKE3 chunk in a shard: chunk chaining values indexed by the shard's own rows
    (F64, 12, 3, 200, 1, 4), (F64, 12, 3, 40, 5, 2), (F128, 11, 3, 10, 20, 8)])
def test_coset_sharded_commitment(ctx, orc, capi, field, logR, logB, n_cols, n_traces, world):
    import torch
    from starkpack_winterfell_amd import shard
    rng = np.random.default_rng(world * 100 + logR)
    R, blowup = 1 << logR, 1 << logB
    traces = [rand_cols(rng, field, n_cols, R) for _ in range(n_traces)]
    offset = 7 if field == F64 else 3
    want = orc.build_trace_commitment(field, traces, 1, logR, logB, offset)
    params = capi.make_params(field, 1, logR, logB, n_cols, n_traces)
    w = 1 if field == F64 else 2
    rw = 8 * ((n_cols + 7) // 8)
    dev = torch.device("cuda", 0)
    host = np.concatenate([c.reshape(-1) for t in traces for c in t]).view(np.int64)
    d_trace = torch.from_numpy(host).to(dev)
    stream = torch.cuda.Stream(device=dev)
    shards = []
    with torch.cuda.stream(stream):
        for rank in range(world):
            c0, nc = shard.cosets_of_rank(blowup, rank, world)
            d_polys = torch.empty_like(d_trace)
            d_lde = torch.full((n_traces * R * nc * rw * w,), -1, dtype=torch.int64, device=dev)  # padding must be written
            d_leaves = torch.empty((R * nc, 32), dtype=torch.uint8, device=dev)
            ctx.trace_commit_shard_dev(params, c0, nc, d_trace.data_ptr(), d_polys.data_ptr(), d_lde.data_ptr(),
                                       d_leaves.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            # local row k*nc + cl of the shard is row k*blowup + c0 + cl of the full LDE
            lde = d_lde.cpu().numpy().view(np.uint64).reshape((n_traces, R, nc, rw) + ((2,) if w == 2 else ()))
            for t in range(n_traces):
                full = want["lde"][t].reshape((R, blowup, rw) + ((2,) if w == 2 else ()))
                assert np.array_equal(lde[t], full[:, c0:c0 + nc])
            shards.append(d_leaves)
        gathered = torch.cat(shards, dim=0)          # what all_gather_into_tensor returns (rank-major)
        per = blowup // world                        # natural order j = k * blowup + rank * per + local coset
        leaves = gathered.view(world, R, per, 32).permute(1, 0, 2, 3).reshape(R * blowup, 32).contiguous()
        nodes = torch.empty_like(leaves)
        ctx.merkle_build_dev(leaves.data_ptr(), R * blowup, nodes.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
    assert np.array_equal(leaves.cpu().numpy(), want["leaves"])
    assert np.array_equal(nodes.cpu().numpy(), want["nodes"])


def test_shard_argument_errors(ctx, capi):
    p = capi.make_params(F64, 1, 4, 2, 1, 1)
    for c0, nc in ((4, 1), (0, 0), (3, 2)):
        with pytest.raises(capi.WfError) as e:
            ctx.trace_commit_shard_dev(p, c0, nc, 8, 0, 8, 8)
        assert e.value.code == -19
