#!/bin/bash
# same-box A/B of PREBUILT libraries (built in the build container, shipped with the snapshot: no hipcc time on the GPU
# box), alternating, 3 pairs: tools/ab_prebuilt.sh libA.so libB.so [bench.py args]
A=$1; B=$2; shift 2
LIB=hypernerf-torch_amd/csrc/libhn_hip.so
cp $LIB /tmp/lib_keep.so
for r in 1 2 3; do
for cfg in "$A" "$B"; do
  echo "=== lib: $cfg"
  cp "$cfg" $LIB
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-calibration "$@" 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('ms/step', round(r['ms_per_step'],4), round(r['value']/1e6,2), {k: round(v,4) for k,v in r['roofline']['machine_kernel_ms_per_step'].items()}, 'other', round(r['roofline']['other_ms_per_step'],4))"
done
done
cp /tmp/lib_keep.so $LIB
