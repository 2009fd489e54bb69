#!/bin/bash
# Samples shader clock (MHz) and socket power (W) of every amdgpu card from sysfs while a command runs:
#   tools/clock_sampler.sh <logfile> -- <command...>
LOG=$1; shift; shift
( while true; do
    line="$(date +%s.%N | cut -c1-14)"
    for h in /sys/class/drm/card*/device/hwmon/hwmon*; do
      f=$(cat "$h/freq1_input" 2>/dev/null); p=$(cat "$h/power1_input" 2>/dev/null)
      line="$line | $((f / 1000000)) MHz $((p / 1000000)) W"
    done
    echo "$line"
  done ) > "$LOG" 2>&1 &
SAMPLER=$!
"$@"
RC=$?
kill $SAMPLER
exit $RC
