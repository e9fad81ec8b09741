#!/bin/bash
# run on the GPU box: the shapes of profiles/*_other_configs.txt through scripts/time_config.py -> gpurun_out/other_configs.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/other_configs.txt
: > $out
while read -r shape note; do
    [ -z "$shape" ] && continue
    line=$(timeout -k 10 300 python scripts/time_config.py ${shape//,/ } 2>&1 | tail -1 | sed -e "s/'interpolate/'int/g" -e "s/'evaluate/'ev/g" -e "s/'layout\./'/g")
    echo "$line   # $note" | tee -a $out
done <<'LIST'
2,1,10,3,10,512 the reference's do_work default, packed (examples/src/lib.rs:97-104)
1,1,20,3,8,8 eight packed traces under one tree
1,1,20,3,2,16 sixteen packed 2-column (Fibonacci-like) traces
1,1,20,3,4,4 four packed 4-column traces
1,1,20,3,200,1 wide trace: 1600-byte rows (two BLAKE3 chunks), fused chunk by chunk
1,1,18,3,255,1 MAX_TRACE_WIDTH columns at 2^18
2,1,14,3,10,32 32 packed f128 do_work traces of 2^14 steps
1,2,20,3,4,1 quadratic-extension columns, 8 base columns
1,2,20,3,2,1 4 base columns: coset-packed lanes
1,2,20,3,1,1 composition-poly shape: one E column
1,1,20,3,1,1 one column: 8 cosets per row
1,1,20,3,5,1 ragged width
1,1,16,3,2,1 Fibonacci-size
2,1,18,3,10,1 do_work at 2^18
2,1,20,3,4,1 f128 2^20 x 4
1,1,18,3,8,1 cfg 2 at 2^18
1,1,22,3,64,1 BASELINE configs[2], three passes
1,1,20,3,8,1 BASELINE configs[1] (the metric) under per-launch events
LIST
