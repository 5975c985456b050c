#!/bin/bash
# run-time A/B (no rebuild): tools/ab_env.sh "VAR=val" ...
for cfg in "$@"; do
  echo "=== cfg: $cfg"
  env $cfg timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('ms/step', round(r['ms_per_step'],4), {k: round(v['ms_per_step'],4) for k,v in r['roofline']['per_kernel'].items()})"
done
