#!/bin/bash
# Run on the GPU box: SQ instruction counters of cfg 2's kernels for the product library and for build/exp_base (same box).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=/tmp/wfab; rm -rf $W; mkdir -p $W
cd /tmp && export TMPDIR=/tmp
for v in product base; do
    if [ $v = base ]; then export WF_LDE_LIB=$ROOT/build/exp_base/libwf_lde.so; else unset WF_LDE_LIB; fi
    rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $W/$v -o $v -- python3 $ROOT/scripts/time_config.py 1 1 20 3 8 1 > $W/$v.out 2> $W/$v.log || tail -3 $W/$v.log
    f=$(find $W/$v -name "*counter_collection.csv" | head -1)
    echo "== $v"; python3 $ROOT/scripts/sq_from_pmc.py $f $f $W/$v.json > /dev/null
    python3 - <<P
import json
d=json.load(open("$W/$v.json"))
for k,c in d.items():
    if "seg_" in k or "merkle_level2" in k:
        print(f"{k[:70]:70s}", {n:round(x/1e6,2) for n,x in c.items()})
P
done
