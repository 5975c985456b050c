#!/bin/bash
# one-box sweep of the weight-gradient job sizing knobs (run-time env, no rebuild): tools/sweep_wgrad_jobs.sh
run() {
  env $1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-calibration 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$1', 'ms/step', round(r['ms_per_step'],4), 'wgrad', round(r['roofline']['machine_kernel_ms_per_step']['hn_mlp_wgrad_batched'],4), 'other', round(r['roofline']['other_ms_per_step'],4))"
}
run "HN_WGRAD_JOB_MB=5"
for mb in 3 4 6 8; do run "HN_WGRAD_JOB_MB=$mb"; done
run "HN_WGRAD_JOB_MB=5"
for tf in 0.0 0.2 0.6 0.8; do run "HN_WGRAD_TAIL_FRAC=$tf"; done
run "HN_WGRAD_JOB_MB=5"
for tp in 3 4; do run "HN_WGRAD_TAIL_PARTS=$tp"; done
for sc in "1.5,1,0.5" "3,1,0.5" "2.2,1.5,0.5" "2.2,1,1" "2.2,0.7,0.35"; do run "HN_WGRAD_JOB_SCALE=$sc"; done
run "HN_WGRAD_JOB_MB=5"
