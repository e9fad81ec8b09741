#!/bin/bash
# Run HERE (not on the GPU box), after `gpurun -- 'scripts/profile_round.sh <tag>_<config> <config>'` for each config: copies the summaries
# the round's record is made of from gpurun_out/prof_<tag>_<config>/ into profiles/ and regenerates profiles/traffic*.json.
#   scripts/collect_round.sh r05 cfg2 cfg3 cfg5 dowork
TAG=$1; shift
cd "$(dirname "$0")/.."
for cfg in "$@"; do
    d=gpurun_out/prof_${TAG}_${cfg}
    [ -d $d ] || { echo "no $d"; continue; }
    cp $d/kernel_stats.csv profiles/${TAG}_${cfg}_kernel_stats.csv
    grep '^{' $d/bench.json | tail -1 > profiles/${TAG}_${cfg}_bench.json
    python3 scripts/sq_from_pmc.py $d/sq1.csv $d/sq2.csv profiles/${TAG}_${cfg}_sq_counters.json
    python3 scripts/traffic_from_pmc.py $d/fetch.csv $d/write.csv $TAG $cfg > /dev/null && echo "traffic for $cfg written"
done
