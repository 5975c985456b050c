"""Print the essentials of a bench.py JSON line: python tools/bench_line.py FILE [FILE...]"""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    rl = d.get("roofline", {})
    print(f"{f}: {d['value'] / 1e6:.2f} M ray-samples/s  {d['ms_per_step']:.4f} ms/step  spread "
          f"{[round(x, 4) for x in d.get('ms_per_step_spread', [])]}  step MFMA frac {d.get('step_mfma_frac', 0):.3f}")
    for k, v in list(rl.get("kernel_ms_per_step", {}).items())[:10]:
        print(f"    {k:42s} {v:.4f} ms")
