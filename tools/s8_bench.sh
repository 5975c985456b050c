for p in bf16 bf16s8 bf16 bf16s8; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --precision $p 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$p', 'ms/step', round(r['ms_per_step'],4), 'M/s', round(r['value']/1e6,2), {k: round(v['ms_per_step'],4) for k,v in r['roofline']['per_kernel'].items()}, 'loss', r['final_loss'], r['hbm']['stash_bytes_moved_per_step']/1e9)"
done
