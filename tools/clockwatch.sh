#!/bin/bash
# Developer tool (GPU box): what clock and power the chip holds while the forward runs.  Samples rocm-smi every 0.5 s next to
# a long `bench.py` resnet run.  usage: bash tools/clockwatch.sh [precision]
P=${1:-bf16}
python bench.py --precision $P --steps 400 --warmup 5 --no_cpu_baseline --no_wsi --no_simclr > /tmp/cw_bench.json 2>/dev/null &
BP=$!
sleep 25
for i in $(seq 1 10); do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|memory)" | tr -s " " | cut -c1-110 | paste -sd"|"
  sleep 0.5
done
wait $BP
python - <<'PY'
import json
r = json.loads(open("/tmp/cw_bench.json").read().strip().splitlines()[-1])
print("bench:", round(r["value"]), "patches/s", r["steps"], "steps")
PY
