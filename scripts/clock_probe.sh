#!/bin/bash
# Run on the GPU box: effective shader clock of the path's kernels vs. the register-only microbenchmark loops
# (MI355X_MICROARCH.md "DVFS give-back": clock = GRBM_GUI_ACTIVE / 8 / kernel wall time; reads high on dispatches < 0.3 ms).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=/tmp/wfclk; rm -rf $W; mkdir -p $W "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -I "$ROOT/starkpack-winterfell_amd/csrc" "$ROOT/scripts/microbench.hip" -o $W/mb || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $W/a -o a -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-transfers > $W/a.out 2> $W/a.log
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $W/b -o b -- $W/mb > $W/b.out 2> $W/b.log
python3 - "$W" <<'PY' | tee "$ROOT/gpurun_out/clock_probe.txt"
import csv, glob, sys, collections
W = sys.argv[1]
for tag in ("a", "b"):
    cc = glob.glob(f"{W}/{tag}/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(f"{W}/{tag}/**/*kernel_trace.csv", recursive=True)
    if not cc or not kt:
        print(tag, "missing output", cc, kt); continue
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for r in csv.DictReader(open(cc[0])):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
            continue
        ns, name = dur[r["Dispatch_Id"]]
        if ns < 40000:
            continue
        a = acc[name[:70]]
        a[0] += float(r["Counter_Value"]); a[1] += ns; a[2] += 1
    print("#", "bench.py kernels" if tag == "a" else "scripts/microbench.hip loops", "(dispatches >= 40 us)")
    for name, (g, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:70s} n={n:3d} avg {ns / n / 1e3:8.1f} us  clock {g / 8 / ns:5.2f} GHz")
PY
