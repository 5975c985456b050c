#!/bin/bash
# timing-only builds of hn_wgrad_kernel (results WRONG for EXP != 0): which part of a job's end costs what
for cfg in "HN_WGRAD_EXP=0" "HN_WGRAD_EXP=2" "HN_WGRAD_EXP=3" "HN_WGRAD_EXP=1" "HN_WGRAD_EXP=0"; do
  echo "=== cfg: $cfg"
  env $cfg python -c "import hypernerf_torch_amd._lib as L; L.build(force=True)" || continue
  for r in 1 2; do
  env $cfg timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-calibration 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('ms/step', round(r['ms_per_step'],4), 'wgrad', round(r['roofline']['machine_kernel_ms_per_step']['hn_mlp_wgrad_batched'],4), 'other', round(r['roofline']['other_ms_per_step'],4))"
  done
done
python -c "import hypernerf_torch_amd._lib as L; L.build(force=True)"
