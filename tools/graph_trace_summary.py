#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace CSV of `bench.py` (HIP-graph replay): finds the replayed steps (the repeating
kernel sequence at the end of the trace), prints ONE step's timeline — start, duration, gap to the previous kernel — and,
averaged over the last replays, every kernel's duration, the sum of the gaps and the step period.
    python tools/graph_trace_summary.py kernel_trace.csv [bench_line.json]"""
import csv
import json
import statistics
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").strip()[:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    # a step ends with the optimizer launch (hn_adam_kernel, or the reduce launch that applies it)
    ends = [i for i, n in enumerate(names) if n.startswith("hn_adam_kernel") or ("hn_wgrad_reduce_kernel" in n and "true>" in n.replace(" ", "")[-6:])]
    if len(ends) < 4:
        ends = [i for i, n in enumerate(names) if n.startswith("hn_adam_kernel") or n.startswith("hn_wgrad_reduce_kernel")]
    # step = kernels between two consecutive optimizer launches; keep the trailing steps of equal length (the replays)
    steps = []
    for a, b in zip(ends[:-1], ends[1:]):
        steps.append(rows[a + 1:b + 1])
    steps = [s for s in steps if len(s) == len(steps[-1])][-8:]
    print(f"# {sys.argv[1]}: {len(rows)} kernel records, {len(steps)} replayed steps of {len(steps[-1])} kernels analysed")
    if len(sys.argv) > 2:
        try:
            line = [ln for ln in open(sys.argv[2]) if ln.startswith("{")][-1]
            j = json.loads(line)
            print(f"# bench line of the same run (under the profiler): {j['ms_per_step']:.4f} ms/step, build {j['build']['kernel_src_sha256']}")
        except Exception as e:      # noqa: BLE001
            print("# (no bench line:", e, ")")
    last = steps[-1]
    t0 = int(last[0]["Start_Timestamp"])
    prev_end = None
    print("# one replayed step:   start_us   dur_us   gap_us  kernel")
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        print(f"  {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:8.1f}  {short(r['Kernel_Name'])}")
        prev_end = e
    durs, gaps, period = {}, [], []
    for st in steps:
        pe = None
        for k, r in enumerate(st):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            durs.setdefault((k, short(r["Kernel_Name"])), []).append((e - s) / 1e3)
            if pe is not None:
                gaps.append((s - pe) / 1e3)
            pe = e
    for a, b in zip(steps[:-1], steps[1:]):
        period.append((int(b[0]["Start_Timestamp"]) - int(a[0]["Start_Timestamp"])) / 1e3)
    by_name = {}
    for (k, n), v in durs.items():
        by_name.setdefault(n, []).append(statistics.mean(v))
    print("# per kernel name, mean over the replays: launches/step, us per step")
    tot = 0.0
    for n, v in sorted(by_name.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {len(v):3d} {sum(v):9.1f}  {n}")
        tot += sum(v)
    machine = sum(sum(v) for n, v in by_name.items() if n.startswith(("hn_mlp_fwd", "hn_mlp_bwd", "hn_wgrad_kernel")))
    ng = len(steps[-1]) - 1
    print(f"# sum of kernel durations {tot:.1f} us/step (machine kernels {machine:.1f}, the others {tot - machine:.1f}); "
          f"gaps inside a step {sum(gaps) / len(steps):.1f} us ({ng} gaps, mean {statistics.mean(gaps):.2f}); "
          f"step period {statistics.mean(period) if period else float('nan'):.1f} us")


if __name__ == "__main__":
    main()
