#!/bin/bash
# Run on the GPU box at the end of a round, on the final sources: the soaks whose summary goes to profiles/rNN_soaks.txt
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_soaks.txt
: > $out
run() { echo "# $*" >> $out; timeout -k 10 600 "$@" 2>&1 | tail -2 >> $out; echo "rc=$?" >> $out; }
run python scripts/random_soak.py 200 51
run python scripts/random_soak.py 40 52 15 22 28
run env WF_EXP_ENABLE=1 WF_EXP_PERSISTENT_ALWAYS=1 WF_EXP_MAX_DIGIT=7 python scripts/random_soak.py 150 53 10 17 24
run env WF_EXP_ENABLE=1 WF_EXP_NO_CHUNKED=1 python scripts/random_soak.py 120 54 3 14 23
run python scripts/mixed_soak.py 300 55
run python scripts/soak.py 2000
cat $out
