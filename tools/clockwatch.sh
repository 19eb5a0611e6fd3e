#!/bin/bash
# Developer tool (GPU box): what clock and power the chip holds while the forward runs.  Samples rocm-smi every 0.5 s next to
# tools/spin.py (the whole forward of one precision in a loop).  usage: bash tools/clockwatch.sh [precision ...]
for P in ${@:-bf16}; do
  python tools/spin.py $P 36 > /tmp/cw_spin.txt 2>/dev/null &
  BP=$!
  sleep 22
  for i in $(seq 1 8); do
    rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction|memory)" | tr -s " " | cut -c1-110 | paste -sd"|"
    sleep 0.5
  done
  wait $BP
  cat /tmp/cw_spin.txt
done
