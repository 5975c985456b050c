for mb in 5 8 12 16 24; do
  HN_WGRAD_JOB_MB=$mb timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --precision bf16s8 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('job_mb $mb', 'ms/step', round(r['ms_per_step'],4), 'M/s', round(r['value']/1e6,2), {k: round(v['ms_per_step'],4) for k,v in r['roofline']['per_kernel'].items()})"
done
