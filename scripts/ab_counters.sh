#!/bin/bash
# Run on the GPU box: SQ counters of cfg 2's kernels for the product library and for other builds (same box).
#   [SHAPE="1 1 22 3 64 1"] scripts/ab_counters.sh product build/exp_x/libwf_lde.so ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=/tmp/wfab; rm -rf $W; mkdir -p $W
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
    i=$((i + 1))
    if [ "$v" = product ]; then unset WF_LDE_LIB; else export WF_LDE_LIB=$ROOT/$v; fi
    rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d $W/v$i -o v$i -- python3 $ROOT/scripts/time_config.py ${SHAPE:-1 1 20 3 8 1} > $W/v$i.out 2> $W/v$i.log || tail -3 $W/v$i.log
    rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $W/w$i -o w$i -- python3 $ROOT/scripts/time_config.py ${SHAPE:-1 1 20 3 8 1} > $W/w$i.out 2> $W/w$i.log || tail -3 $W/w$i.log
    f=$(find $W/v$i -name "*counter_collection.csv" | head -1); g=$(find $W/w$i -name "*counter_collection.csv" | head -1)
    echo "== $v"; python3 $ROOT/scripts/sq_from_pmc.py $f $g $W/v$i.json > /dev/null
    python3 - <<P
import json
d=json.load(open("$W/v$i.json"))
for k,c in d.items():
    if "seg_" in k or "merkle_level2" in k:
        print(f"{k[:78]:78s}", {n:round(x/1e6,2) for n,x in c.items()})
P
done
