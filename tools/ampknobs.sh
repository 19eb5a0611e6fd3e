#!/bin/bash
# Developer tool (GPU box): images/s of the mixed-precision SimCLR step (2 x 1024 views) for combinations of the weight-gradient
# kernel's knobs: taps per workgroup of the wide (128 x 128 tile) and the narrow (64 x 64) layers, workgroups per launch.
# usage: bash tools/ampknobs.sh            (edit the list below)
for cfg in "1 3 768" "1 9 768" "3 3 768" "1 3 1536" "1 3 512"; do
  set -- $cfg
  HIPAC_WG_NTAP_BIG=$1 HIPAC_WG_NTAP_SMALL=$2 HIPAC_WG_TARGET=$3 python bench.py --workload simclr --steps 3 --warmup 1 --train_precision fp16 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(r['value']), round(r['ms_per_step'],2))
"
done
