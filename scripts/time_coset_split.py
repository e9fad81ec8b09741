"""Experiment aid: does the evaluation run faster when the cosets are taken a few at a time (the strided pass's output of a
coset group then sits in the 256 MiB Infinity Cache when the last pass reads it)?  Uses wf_trace_commit_shard_dev (a coset
range per call) and the context's per-mark profile: the sum of the evaluation marks over the calls of one split against
the one-call form.  The interpolation is repeated by every call and is NOT part of the comparison.
    python scripts/time_coset_split.py <field 1|2> <log2 R> <log2 blowup> <n_cols>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starkpack_winterfell_amd.capi as capi

field, logR, logB, n_cols = (int(x) for x in sys.argv[1:5])
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
w = 1 if field == 1 else 2
R, B = 1 << logR, 1 << logB
gen = torch.Generator(device=dev); gen.manual_seed(1)
trace = torch.randint(0, 2**62, (n_cols * R * w,), dtype=torch.int64, device=dev, generator=gen)
polys = torch.empty_like(trace)
rw = 8 * ((n_cols + 7) // 8)
lde = torch.empty(R * B * rw * w, dtype=torch.int64, device=dev)
leaves = torch.empty((R * B, 32), dtype=torch.uint8, device=dev)
p = capi.make_params(field, 1, logR, logB, n_cols, 1)
s = torch.cuda.Stream(device=dev)
ctx.profile_enable(2)
for split in (1, 2, 4, 8, 1, 2, 4, 8):
    if split > B:
        continue
    cnt = B // split
    with torch.cuda.stream(s):
        for rep in range(3):
            if rep == 1:
                torch.cuda.synchronize(); ctx.profile_read()
            for part in range(split):
                ctx.trace_commit_shard_dev(p, part * cnt, cnt, trace.data_ptr(), polys.data_ptr(),
                                           lde.data_ptr() + part * cnt * R * rw * 8 * w, leaves.data_ptr() + part * cnt * R * 32, s.cuda_stream)
        torch.cuda.synchronize()
    acc = {}
    for k, v in ctx.profile_read():
        acc[k] = acc.get(k, 0.0) + v
    ev = sum(v for k, v in acc.items() if k.startswith("evaluate") or k.startswith("hash")) / 2
    print(f"cosets per call {cnt}: evaluation marks {ev:.4f} ms per commitment", {k: round(v / 2, 4) for k, v in acc.items() if not k.startswith("interp") and not k.startswith("layout")})
