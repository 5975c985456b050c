#!/bin/bash
# A/B helper: tools/ab.sh "VAR=val VAR2=val" ... ; each argument is one build configuration (env macros)
for cfg in "$@"; do
  echo "=== cfg: $cfg"
  env $cfg python -c "import hypernerf_torch_amd._lib as L; L.build(force=True)" || continue
  env $cfg python tools/kbench.py bf16 2>&1 | grep -E "mlp_|SUM"
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c100-200
done
python -c "import hypernerf_torch_amd._lib as L; L.build(force=True)"
