#!/bin/bash
# Sample the GPU's power / clocks (rocm-smi, read-only) while a command runs: bash tools/power_watch.sh <out> <cmd…>
# Development aid for the "clock inertia" observations of DESIGN.md §6·7.
OUT=$1; shift
"$@" > "$OUT.cmd.log" 2>&1 &
PID=$!
: > "$OUT"
while kill -0 $PID 2>/dev/null; do
  rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)|Performance Level" | tr '\n' ';' >> "$OUT"
  echo >> "$OUT"
  sleep 0.3
done
wait $PID
tail -n 1 "$OUT.cmd.log"
