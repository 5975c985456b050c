#!/bin/bash
# A/B of two versions of hn_mlp.hip on one box: tools/ab_src.sh fileA fileB
for f in "$@"; do
  echo "=== $f"
  cp $f hypernerf-torch_amd/csrc/hn_mlp.hip
  python -c "import hypernerf_torch_amd._lib as L; L.build(force=True)" || continue
  for r in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('ms/step', round(r['ms_per_step'],4), {k: round(v['ms_per_step'],4) for k,v in r['roofline']['per_kernel'].items()})"
  done
done
