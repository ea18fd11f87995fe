#!/bin/bash
# Samples rocm-smi (power, sclk, temperature) twice a second while a command runs.  usage: tools/power_probe.sh <out.log> <command...>
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"
( while true; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (edge|junction|hotspot)" | tr '\n' ' ' ; echo; sleep 0.5; done ) > "$OUT" &
SAMPLER=$!
"$@"
RC=$?
kill $SAMPLER
exit $RC
