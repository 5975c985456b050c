#!/bin/bash
# build-variant A/B of one pytest selection: tools/ab_test.sh "<pytest -k expr>" "VAR=val" "VAR2=val" ...
sel="$1"; shift
for cfg in "$@"; do
  echo "=== cfg: $cfg"
  env $cfg python -c "import hypernerf_torch_amd._lib as L; L.build(force=True)" || continue
  env $cfg timeout 600 python -m pytest tests -m gpu -q -s -k "$sel" 2>&1 | grep -E "rel L2|passed|failed|Error" | cut -c1-1500
done
python -c "import hypernerf_torch_amd._lib as L; L.build(force=True)"
