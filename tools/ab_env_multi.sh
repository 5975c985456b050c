#!/bin/bash
# same-box comparison of (prebuilt library, environment) pairs, round-robin: tools/ab_env_multi.sh "lib.so VAR=val" "lib2.so" ...
LIB=hypernerf-torch_amd/csrc/libhn_hip.so
cp $LIB /tmp/lib_keep.so
for r in $(seq 1 ${ROUNDS:-2}); do
for cfg in "$@"; do
  lib=${cfg%% *}; envs=""; [ "$lib" != "$cfg" ] && envs=${cfg#* }
  echo "=== $cfg"
  cp "$lib" $LIB
  env $envs timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-calibration $BENCH_ARGS 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r['roofline']['machine_kernel_ms_per_step']; print('ms/step', round(r['ms_per_step'],4), round(r['value']/1e6,2), 'wgrad', round(k['hn_mlp_wgrad_batched'],4), 'fwd', round(sum(v for n,v in k.items() if 'forward' in n),4), 'bwd', round(sum(v for n,v in k.items() if 'backward' in n),4), 'other', round(r['roofline']['other_ms_per_step'],4))"
done
done
cp /tmp/lib_keep.so $LIB
