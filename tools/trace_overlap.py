"""Reads a rocprofv3 --kernel-trace CSV and prints, for the last step of a bench run, every kernel's start / end
(us, relative) and which kernels overlap in time: the evidence for (or against) concurrent execution of the forked
weight-gradient launches.   python tools/trace_overlap.py <kernel_trace.csv> [n_last_kernels]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = 0
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    flag = "  <-- starts before the previous kernel ended" if s < prev_end - 0.5 else ""
    print(f"{s:10.1f} {e:10.1f} {e - s:9.1f} us  q={r.get('Queue_Id', '?'):>3} grid={r.get('Grid_Size', '?'):>8} {name}{flag}")
    prev_end = max(prev_end, e)
