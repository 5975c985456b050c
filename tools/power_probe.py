"""Board power and clocks while bench.py replays the config-2 step back to back (one MI355X):
    python tools/power_probe.py [--config 2] [--seconds 8]
Starts bench.py as a CHILD process (this process never touches the GPU), samples the amdgpu hwmon files
(power1_average / power1_input, power1_cap, freq1_input = sclk, freq2_input = mclk) every 50 ms while the child runs its
timed steps, and prints one JSON line: the cap, the mean / max power and the clock range over the busy samples.
Why: a same-box A/B of round 4 (profiles/r04_aux_tile_skip_ab.log) showed unchanged kernels running 1.5-2 % slower when the
kernel in front of them got shorter — the chip gives saved cycles back as clock; this says how close to the cap the step runs."""
import argparse, glob, json, os, subprocess, sys, time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hwmons():
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if os.path.exists(d + "/power1_cap") or os.path.exists(d + "/power1_average") or os.path.exists(d + "/power1_input"):
            out.append(d)
    return out


def rd(path):
    try:
        return int(open(path).read().strip())
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=8.0)
    a = ap.parse_args()
    hm = hwmons()
    steps = int(a.seconds / (0.0019 if a.config != 3 else 0.037))
    cmd = [sys.executable, os.path.join(R, "bench.py"), "--config", str(a.config), "--steps", str(steps), "--warmup", "20",
           "--repeats", "1", "--no-cpu-baseline", "--no-roofline", "--no-also"]
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    samples = []
    t0 = time.time()
    while child.poll() is None and time.time() - t0 < a.seconds + 240:
        row = {"t": round(time.time() - t0, 3)}
        for i, d in enumerate(hm):
            for key, f in (("power_uW", "power1_average"), ("power_in_uW", "power1_input"), ("sclk_Hz", "freq1_input"),
                           ("mclk_Hz", "freq2_input")):
                v = rd(f"{d}/{f}")
                if v is not None:
                    row[f"{key}_{i}"] = v
        samples.append(row)
        time.sleep(0.05)
    line = (child.stdout.read() or "").strip().splitlines()
    bench = json.loads(line[-1]) if line else {}
    caps = {i: rd(f"{d}/power1_cap") for i, d in enumerate(hm)}
    res = {"what": "tools/power_probe.py: hwmon samples (50 ms) while bench.py replays the step", "config": a.config,
           "hwmon": hm, "power_cap_W": {i: (c / 1e6 if c else None) for i, c in caps.items()},
           "ms_per_step": bench.get("ms_per_step"), "value": bench.get("value"), "n_samples": len(samples)}
    for i in range(len(hm)):
        pk = f"power_uW_{i}" if any(f"power_uW_{i}" in s for s in samples) else f"power_in_uW_{i}"
        pw = [s[pk] / 1e6 for s in samples if pk in s]
        if not pw:
            continue
        busy_thr = 0.6 * max(pw)
        busy = [s for s in samples if s.get(pk, 0) / 1e6 >= busy_thr]
        res[f"gpu{i}"] = {
            "power_W_idle_min": round(min(pw), 1), "power_W_busy_mean": round(sum(s[pk] for s in busy) / 1e6 / max(1, len(busy)), 1),
            "power_W_max": round(max(pw), 1), "busy_samples": len(busy),
            "sclk_MHz_busy_min_mean_max": [round(f(s.get(f"sclk_Hz_{i}", 0) for s in busy) / 1e6 if busy else 0) for f in (min, lambda g: sum(g) / max(1, len(busy)), max)],
            "mclk_MHz_busy": sorted({round(s.get(f"mclk_Hz_{i}", 0) / 1e6) for s in busy})}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
