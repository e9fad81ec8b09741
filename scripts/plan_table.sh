#!/bin/bash
# Run on the GPU box: the table "shape -> pass plan (wf_plan_digits) -> the kernels one commitment launches" of DESIGN.md section 4,
# read off a rocprofv3 kernel trace of scripts/time_config.py per shape (so it is what the launcher DOES, not a restatement of its rules).
#   scripts/plan_table.sh > gpurun_out/plan_table.md
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=/tmp/wfplan; rm -rf $W; mkdir -p $W
cd /tmp && export TMPDIR=/tmp
echo "| shape (field, rows x columns [x traces], blowup 8) | plan (log2 tile rows per pass) | kernels of one commitment, in launch order (launches) |"
echo "|---|---|---|"
i=0
while read -r label args; do
    [ -z "$label" ] && continue
    i=$((i + 1))
    rocprofv3 --kernel-trace --output-format csv -d $W/s$i -o s$i -- python3 $ROOT/scripts/time_config.py $args > $W/s$i.out 2> $W/s$i.log || { echo "| $label | FAILED | $(tail -1 $W/s$i.log) |"; continue; }
    python3 - "$label" "$args" $W/s$i <<'P'
import csv, glob, sys, os, re
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
label, args, d = sys.argv[1], sys.argv[2].split(), sys.argv[3]
field, ext, logr, logb, ncols, ntr = (int(x) for x in args)
import starkpack_winterfell_amd.capi as capi
S = 8 if field == 1 else 4
nseg = (ncols * ext * ntr + S - 1) // S
plan = capi.plan_digits(field, logr, nseg)
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "wf::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("wf::", "") for r in rows]
# one commitment = the launches between two k_cols_to_seg; take the last complete one
starts = [i for i, n in enumerate(names) if n.startswith("k_cols_to_seg")]
one = names[starts[-2]:starts[-1]] if len(starts) >= 2 else names
out, prev, cnt = [], None, 0
for n in one + [None]:
    if n == prev:
        cnt += 1
        continue
    if prev is not None:
        out.append(f"`{prev}`" + (f" x{cnt}" if cnt > 1 else ""))
    prev, cnt = n, 1
print(f"| {label} | {list(plan)} | " + ", ".join(out) + " |")
P
done <<'SHAPES'
f64_2^10x8 1 1 10 3 8 1
f64_2^14x8 1 1 14 3 8 1
f64_2^17x8 1 1 17 3 8 1
f64_2^18x8 1 1 18 3 8 1
f64_2^20x8_(cfg2) 1 1 20 3 8 1
f64_2^21x8 1 1 21 3 8 1
f64_2^22x8 1 1 22 3 8 1
f64_2^18x32 1 1 18 3 32 1
f64_2^20x64 1 1 20 3 64 1
f64_2^22x64_(cfg3) 1 1 22 3 64 1
f64_2^20x10_(tail-packed) 1 1 20 3 10 1
f64_2^20x8x8traces_(STARKPack) 1 1 20 3 8 8
f64_2^20x200_(rows>1chunk) 1 1 20 3 200 1
f64_quad_2^20x4 1 2 20 3 4 1
f64_2^20x1 1 1 20 3 1 1
f128_2^10x10x512traces_(dowork) 2 1 10 3 10 512
f128_2^18x10_(cfg5) 2 1 18 3 10 1
f128_2^20x10 2 1 20 3 10 1
f128_2^16x4 2 1 16 3 4 1
f128_quad_2^19x2 2 2 19 3 2 1
SHAPES
