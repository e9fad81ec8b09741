#!/bin/bash
# Run on the GPU box: A/B of library builds on ONE box, interleaved (boxes of the pool differ by up to 8 %).
#   scripts/ab_interleaved.sh <rounds> "<time_config args>" product build/exp_x/libwf_lde.so ...
ROUNDS=$1; ARGS=$2; shift 2
cd "$(dirname "$0")/.."
for r in $(seq 1 $ROUNDS); do
    for v in "$@"; do
        if [ "$v" = product ]; then unset WF_LDE_LIB; else export WF_LDE_LIB=$PWD/$v; fi
        printf "round %d %-40s " $r "$v"
        python scripts/time_config.py $ARGS 2>&1 | tail -1 | sed -e "s/.*traces=[0-9]*: //" -e "s/'layout[^,]*, //g" -e "s/'interpolate/'int/g" -e "s/'evaluate/'ev/g"
    done
done
