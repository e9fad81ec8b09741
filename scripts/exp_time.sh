#!/bin/bash
# run on the GPU box: times cfg 2 with every experiment variant built by scripts/exp_variants.sh
cd "$(dirname "$0")/.."
for v in "" $(ls -d build/exp_* 2>/dev/null); do
    if [ -n "$v" ]; then export WF_LDE_LIB=$v/libwf_lde.so; fi
    printf "%-28s " "${v:-product}"
    python scripts/time_config.py 1 1 20 3 8 1 2>&1 | tail -1 | sed -e "s/.*traces=1: //" -e "s/'layout[^,]*, //g" -e "s/'interpolate/'int/g" -e "s/'evaluate/'ev/g"
done
