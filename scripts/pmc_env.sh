#!/bin/bash
# Run on the GPU box: kernel durations + SQ counters of one time_config shape under a set of WF_EXP_* switches (exported here, so that
# the program after `--` is python itself).
#   scripts/pmc_env.sh "<time_config args>" "<VAR=1 VAR2=..>" [kernel-name filter]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ARGS=$1; SW=$2; FILT=${3:-wf::}
W=/tmp/wfpmc; rm -rf $W; mkdir -p $W
cd /tmp && export TMPDIR=/tmp
export WF_EXP_ENABLE=1
for kv in $SW; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $W/t -o t -- python3 $ROOT/scripts/time_config.py $ARGS > $W/t.out 2> $W/t.log || tail -3 $W/t.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $W/v -o v -- python3 $ROOT/scripts/time_config.py $ARGS > $W/v.out 2> $W/v.log || tail -3 $W/v.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $W/w -o w -- python3 $ROOT/scripts/time_config.py $ARGS > $W/w.out 2> $W/w.log || tail -3 $W/w.log
python3 - "$FILT" <<'P'
import csv, glob, sys, collections
filt = sys.argv[1]
st = glob.glob('/tmp/wfpmc/t/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(st[0])):
    if filt in r['Name']:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:9.1f} us")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('v', 'w'):
    for f in glob.glob(f'/tmp/wfpmc/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if filt in r['Kernel_Name']:
                acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    print(k[:90], {n: round(sum(v) / len(v) / 1e6, 3) for n, v in c.items()})
P
