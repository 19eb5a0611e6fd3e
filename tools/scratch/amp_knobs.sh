for cfg in "1 3 256" "1 3 512" "1 3 640" "1 3 768"; do
  set -- $cfg
  HIPAC_WG_NTAP_BIG=$1 HIPAC_WG_NTAP_SMALL=$2 HIPAC_WG_TARGET=$3 python bench.py --workload simclr --steps 3 --warmup 1 --train_precision fp16 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(r['value']), round(r['ms_per_step'],2))
"
done
