#!/usr/bin/env python3
"""tools/ab.sh's formatter: one bench JSON line on stdin -> one summary line; `--summary FILE` -> medians per label."""
import json
import os
import statistics
import sys


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--summary":
        by = {}
        for ln in open(sys.argv[2]):
            parts = ln.split()
            if len(parts) > 3 and parts[1] == "ms/step":
                rec = by.setdefault(parts[0], {})
                for k, v in zip(parts[1::2], parts[2::2]):
                    try:
                        rec.setdefault(k, []).append(float(v))
                    except ValueError:
                        pass
        print("--- medians")
        base = None
        for label, rec in by.items():
            med = {k: statistics.median(v) for k, v in rec.items()}
            base = base or med
            rel = 100.0 * (med["ms/step"] / base["ms/step"] - 1.0)
            print(label, " ".join(f"{k} {med[k]:.4f}" for k in med), f"({rel:+.2f} % vs {next(iter(by))})", f"n={len(rec['ms/step'])}")
        return
    label = os.environ.get("AB_LABEL", "?").strip().replace(" ", "_")
    txt = sys.stdin.read().strip()
    try:
        r = json.loads(txt)
    except ValueError:
        print(label, "FAILED", txt[-200:])
        return
    k = r.get("roofline", {}).get("machine_kernel_ms_per_step", {})
    fwd = sum(v for n, v in k.items() if "forward" in n)
    bwd = sum(v for n, v in k.items() if "backward" in n)
    wg = sum(v for n, v in k.items() if "wgrad" in n)
    print(label, "ms/step", f"{r['ms_per_step']:.4f}", "M/s", f"{r['value'] / 1e6:.2f}", "wgrad", f"{wg:.4f}", "fwd", f"{fwd:.4f}",
          "bwd", f"{bwd:.4f}", "other", f"{r.get('roofline', {}).get('other_ms_per_step', float('nan')):.4f}",
          "loss", f"{r.get('final_loss', float('nan')):.5f}", flush=True)


if __name__ == "__main__":
    main()
