#!/bin/bash
# same-box A/B of NerfModel.REUSE_COARSE (round 5): tools/ab_reuse.sh [config ...]   (default: 2)
# alternating processes, 3 pairs per configuration; prints ms/step, M ray-samples/s and the per-kernel times
mkdir -p gpurun_out/r05
for c in ${@:-2}; do
steps=30; [ "$c" = 3 ] && steps=5
for r in 1 2 3; do
for cfg in "HN_REUSE_COARSE=0" "HN_REUSE_COARSE=1"; do
  echo "=== config $c: $cfg"
  env $cfg timeout 300 python bench.py --config $c --steps $steps --warmup 3 --no-cpu-baseline --no-also 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('ms/step', round(r['ms_per_step'],4), round(r['value']/1e6,2), {k: round(v,4) for k,v in r['roofline']['machine_kernel_ms_per_step'].items()})"
done
done
done
