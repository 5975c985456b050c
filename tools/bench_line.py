"""Print the essentials of a bench.py JSON line: python tools/bench_line.py FILE [FILE...]"""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    rl = d.get("roofline", {})
    print(f"{f}: {d['value'] / 1e6:.2f} M ray-samples/s  {d['ms_per_step']:.4f} ms/step  spread "
          f"{[round(x, 4) for x in d.get('ms_per_step_spread', [])]}  step MFMA frac {d.get('step_mfma_frac', 0):.3f}")
    for k, v in list(rl.get("machine_kernel_ms_per_step", rl.get("kernel_ms_per_step", {})).items())[:10]:
        print(f"    {k:42s} {v:.4f} ms")
    if "other_ms_per_step" in rl:
        print(f"    {'other (small launches + gaps)':42s} {rl['other_ms_per_step']:.4f} ms")
    cal = d.get("calibration")
    if cal:
        tr = cal.get("timed_region", {})
        print(f"    box: MFMA probe {cal['mfma_probe_tflops']:.0f} TFLOP/s @ {cal['mfma_probe_sclk_mhz']:.0f} MHz, stream probe "
              f"{cal['hbm_probe_tbps']:.2f} TB/s, step (soak) at {(cal.get('soak') or {}).get('power_w')} W / "
              f"{(cal.get('soak') or {}).get('sclk_mhz')} MHz; at the nominal clock "
              f"{cal.get('normalised', {}).get('ms_per_step_at_nominal_clock')} ms")
