#!/bin/bash
# Run on the GPU box: A/B of ENVIRONMENT switches (wf_tuning) on one box, interleaved.
#   scripts/ab_env.sh <rounds> "<time_config args>" "" "WF_EXP_FULL_TILES=1" ...   (sets WF_EXP_ENABLE=1 for every variant)
ROUNDS=$1; ARGS=$2; shift 2
cd "$(dirname "$0")/.."
for r in $(seq 1 $ROUNDS); do
    for v in "$@"; do
        printf "round %d %-34s " $r "[$v]"
        env WF_EXP_ENABLE=1 $v python scripts/time_config.py $ARGS 2>&1 | tail -1 | sed -e "s/.*traces=[0-9]*: //" -e "s/'layout[^,]*, //g" -e "s/'interpolate/'int/g" -e "s/'evaluate/'ev/g"
    done
done
