#!/bin/bash
# tools/sweep_env.sh KNOB "V1 V2 ..." -- command ...    runs the command once per value with NPM_TUNE=KNOB=V in its environment
knob=$1; vals=$2; shift 3
for v in $vals; do echo "== NPM_TUNE=$knob=$v"; NPM_TUNE="$knob=$v" "$@" || exit 1; done
