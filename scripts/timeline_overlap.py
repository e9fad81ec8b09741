"""Summary of a rocprofv3 timeline (--hip-trace --memory-copy-trace --kernel-trace, csv): do host-to-device copies run
while kernels run?  Usage: python scripts/timeline_overlap.py <dir> > profiles/rNN_two_contexts.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict


def rows(pattern):
    out = []
    for f in glob.glob(os.path.join(sys.argv[1], "**", pattern), recursive=True):
        with open(f, newline="") as fh:
            out += list(csv.DictReader(fh))
    return out


def col(r, *names):
    for n in names:
        if n in r:
            return r[n]
    raise KeyError(names)


kern = [(int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), col(r, "Kernel_Name"), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), r.get("Thread_Id", "?"))
        for r in rows("*kernel_trace.csv")]
cop = [(int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), col(r, "Direction"), r.get("Stream_Id", "?")) for r in rows("*memory_copy_trace.csv")]
api = [(int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), col(r, "Function"), r.get("Thread_Id", "?")) for r in rows("*hip_api_trace.csv")]
kern.sort()
cop.sort()
if not kern:
    sys.exit("no kernel trace found")
t0 = min(kern[0][0], cop[0][0] if cop else kern[0][0])


def merged(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def overlap(a, b):
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e > s:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


big = [c for c in cop if "HOST_TO_DEVICE" in c[2].upper() and c[1] - c[0] > 50_000]
km = merged([(k[0], k[1]) for k in kern])
cm = merged([(c[0], c[1]) for c in big])
ksum, csum = sum(e - s for s, e in km), sum(e - s for s, e in cm)
print(f"kernels: {len(kern)} dispatches, busy {ksum / 1e6:.2f} ms; large H2D copies: {len(big)}, busy {csum / 1e6:.2f} ms; "
      f"copy time that runs under a kernel: {overlap(km, cm) / 1e6:.2f} ms ({100.0 * overlap(km, cm) / max(csum, 1):.0f} % of the copy time)")
print(f"span of the trace: {(max(k[1] for k in kern) - t0) / 1e6:.2f} ms")
streams = sorted({k[4] for k in kern} | {c[3] for c in cop})
print("streams:", streams, " queues:", sorted({k[3] for k in kern}))
print("\n60 events from the middle of the trace (ms from start; K = kernel, C = copy):")
ev = [(k[0], k[1], "K", f"q{k[3]} s{k[4]} " + k[2][:60]) for k in kern] + [(c[0], c[1], "C", f"s{c[3]} {c[2]} {(c[1] - c[0]) / 1e3:.0f} us") for c in cop if c[1] - c[0] > 20_000]
ev.sort()
mid = len(ev) // 2
for s, e, kind, what in ev[mid:mid + 60]:
    print(f"  {(s - t0) / 1e6:9.3f} .. {(e - t0) / 1e6:9.3f}  {kind}  {what}")
print("\nHIP API time per thread and function (ms, calls), top 12:")
agg = defaultdict(lambda: [0, 0])
for s, e, f, t in api:
    agg[(t, f)][0] += e - s
    agg[(t, f)][1] += 1
for (t, f), (ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  thread {t}  {f:34s} {ns / 1e6:9.2f} ms  {n:6d} calls")
