"""Per-kernel summary of a rocprofv3 --kernel-trace result database (rocpd sqlite): python tools/prof_summary.py DB [steps]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3 from {kd} d "
                      f"join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"kernel,calls,calls_per_step,total_us,avg_us,percent   (total {tot:.1f} us, {tot / steps:.1f} us/step over {steps} steps)")
for r in rows:
    print(f"\"{r[0][:110]}\",{r[1]},{r[1] / steps:.2f},{r[2]:.1f},{r[3]:.2f},{100 * r[2] / tot:.2f}")
