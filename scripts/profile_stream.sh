#!/bin/bash
# Run on the GPU box: timelines (HIP API + memory copies + kernels) of a stream of cfg-2 commitments from PINNED host
# columns -- two host threads with a context each, and one context through wf_trace_commit_resident_async -- summarised by
# scripts/timeline_overlap.py into gpurun_out/two_contexts.txt; then the unprofiled timings of all variants.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/two_contexts.txt
: > "$OUT"
for v in two batch; do
    rm -rf /tmp/tl_$v
    rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --output-format csv -d /tmp/tl_$v -- python3 "$ROOT/scripts/two_contexts.py" trace pinned $v > /tmp/tl_$v.out 2> /tmp/tl_$v.log || tail -5 /tmp/tl_$v.log
    echo "==================== variant: $v (under the profiler) ====================" >> "$OUT"
    grep "roots agree" /tmp/tl_$v.out >> "$OUT"
    python3 "$ROOT/scripts/timeline_overlap.py" /tmp/tl_$v >> "$OUT" 2>&1
done
echo "==================== unprofiled, 12 commitments per variant ====================" >> "$OUT"
python3 "$ROOT/scripts/two_contexts.py" time 2>/dev/null | grep "roots agree" >> "$OUT"
tail -4 "$OUT"
