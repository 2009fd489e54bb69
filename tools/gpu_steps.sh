#!/bin/bash
# Runs GPU steps one after another on the gpurun box, each under its own timeout, logging to gpurun_out/<tag>/.
# A step that is killed at its limit (rc 124 / 137) or dies on a signal ends the session: nothing further touches
# the GPU.  A step that merely FAILS (a test assertion) does not.
#   tools/gpu_steps.sh <tag> "<name>|<seconds>|<command>" ...
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
for step in "$@"; do
    name=${step%%|*}; rest=${step#*|}; secs=${rest%%|*}; cmd=${rest#*|}
    echo "== $name (limit ${secs}s): $cmd" | tee -a "$out/steps.log"
    start=$(date +%s)
    timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.log" 2>&1
    rc=$?
    echo "   rc=$rc after $(( $(date +%s) - start ))s" | tee -a "$out/steps.log"
    tail -n 3 "$out/$name.log" | sed 's/^/   | /'
    if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then
        echo "   step killed: stopping here" | tee -a "$out/steps.log"
        exit $rc
    fi
done
exit 0
