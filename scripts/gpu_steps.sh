#!/bin/bash
# Runs the given steps ("name::command") one after the other on the GPU box, each under its own timeout, output to
# gpurun_out/<name>.log.  An ordinary failure is recorded and the next step runs; a step that was KILLED by its timeout
# (124 / 137) ends the sequence: no further GPU step after a hang.
#   scripts/gpu_steps.sh <seconds-per-step> "name::command" ...
LIMIT=$1; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for step in "$@"; do
    name=${step%%::*}; cmd=${step#*::}
    echo "=== $name: $cmd"
    start=$(date +%s)
    timeout -k 10 "$LIMIT" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "=== $name rc=$rc in $(( $(date +%s) - start )) s"; tail -n 6 "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name was killed at its limit: stopping"; exit $rc; fi
done
exit 0
