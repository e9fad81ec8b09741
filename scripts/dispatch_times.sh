#!/bin/bash
# Run on the GPU box: per-DISPATCH durations of the wf:: kernels of one shape (kernels launched twice per step under one
# name -- the two strided passes of a three-pass plan -- are told apart by their order).
#   scripts/dispatch_times.sh "<time_config args>" [library.so]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=/tmp/wfdisp; rm -rf $W; mkdir -p $W
[ -n "$2" ] && export WF_LDE_LIB=$ROOT/$2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $W/t -o t -- python3 $ROOT/scripts/time_config.py $1 > $W/t.out 2> $W/t.log || tail -3 $W/t.log
python3 - <<P
import csv, glob, collections
f = glob.glob("$W/t/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "wf::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = collections.defaultdict(list)
for r in rows:
    seq[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in seq.items():
    if "seg_" not in k: continue
    n = len(v)
    per_step = max(1, n // 7)  # time_config.py: 2 warm-up + 5 timed commitments
    print(f"{k[:80]:80s} n={n:3d} last step (ms):", [round(x, 3) for x in v[-per_step:]])
P
