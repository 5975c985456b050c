#!/bin/bash
# A/B helper: tools/ab.sh "VAR=val VAR2=val" ... ; each argument is one build configuration (env macros)
for cfg in "$@"; do
  echo "=== cfg: $cfg"
  env $cfg python -c "import hypernerf_torch_amd._lib as L; L.build(force=True)" || continue
  env $cfg timeout 120 python tools/kbench.py bf16 2>&1 | grep -E "mlp_(forward|backward)\[template_fine|SUM"
done
python -c "import hypernerf_torch_amd._lib as L; L.build(force=True)"
