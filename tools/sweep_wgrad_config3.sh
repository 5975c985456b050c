#!/bin/bash
# config 3 (16,384 rays): weight-gradient job sizes that grow with the launch, with / without the partial-slab flush
run() {
  env $1 timeout 300 python bench.py --config 3 --steps 5 --warmup 3 --repeats 3 --no-cpu-baseline --no-also --no-calibration 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); e=r['roofline']['eager_pass']['kernel_ms_per_step']; print('$1', 'ms/step', round(r['ms_per_step'],3), 'wgrad', round(r['roofline']['machine_kernel_ms_per_step']['hn_mlp_wgrad_batched'],3), 'reduce', round(e.get('hn_mlp_wgrad_reduce',0),3), 'other', round(r['roofline']['other_ms_per_step'],3))"
}
run "HN_WGRAD_JOBS_PER_CU=0"
run "HN_WGRAD_JOBS_PER_CU=16"
run "HN_WGRAD_JOBS_PER_CU=8"
run "HN_WGRAD_JOBS_PER_CU=4"
run "HN_WGRAD_JOBS_PER_CU=0 HN_WGRAD_PARTIALS=0"
run "HN_WGRAD_JOBS_PER_CU=8 HN_WGRAD_PARTIALS=0"
run "HN_WGRAD_JOBS_PER_CU=0"
run "HN_WGRAD_JOBS_PER_CU=8"
