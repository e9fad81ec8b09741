#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + the PMC passes behind profiles/ (separate runs, as the pool requires).
#   scripts/profile_round.sh <tag> [config]  -> gpurun_out/prof_<tag>/ : kernel_stats.csv, fetch.csv, write.csv, sq1.csv, sq2.csv, bench.json
#   config: cfg2 (default, the metric) | cfg3 | cfg5 | dowork  (bench.py --config)
TAG=${1:-r01}
CONFIG=${2:-cfg2}
STEPS=5; [ "$CONFIG" = cfg3 ] && STEPS=2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
W=/tmp/wfprof
rm -rf "$W"; mkdir -p "$OUT" "$W"
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprof args...
    local name=$1; shift
    rocprofv3 "$@" --output-format csv -d "$W/$name" -o "$name" -- python3 "$ROOT/bench.py" --config "$CONFIG" --steps $STEPS --warmup 1 --no-cpu-baseline --no-transfers > "$W/$name.out" 2> "$W/$name.log" || { echo "$name failed"; tail -5 "$W/$name.log"; }
}
run trace --kernel-trace --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq1 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES
run sq2 --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU
find "$W" -name "*.csv" -size -40M | while read f; do echo "$f $(wc -c < "$f")"; done
cp $(find "$W/trace" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv" 2>/dev/null
for n in fetch write sq1 sq2; do cp $(find "$W/$n" -name "*counter_collection.csv" | head -1) "$OUT/$n.csv" 2>/dev/null; done
cp "$W/trace.out" "$OUT/bench_under_trace.json"
python3 "$ROOT/bench.py" --config "$CONFIG" > "$OUT/bench.json" 2> "$OUT/bench.log"
ls -la "$OUT"
