#!/bin/bash
# tools/sweep_knob.sh KNOB "V1 V2 ..." -- command ...    runs the command once per value with --tune KNOB=V appended
knob=$1; vals=$2; shift 3
for v in $vals; do echo "== knob $knob = $v"; "$@" --tune "$knob=$v" || exit 1; done
