"""Per-kernel count / average / total duration from a rocprofv3 rocpd database (the default output format of
`rocprofv3 --kernel-trace` on ROCm 7): python tools/rocpd_stats.py results.db [name-filter]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id "
     f"group by s.kernel_name order by 4 desc")
print("kernel,calls,avg_us,total_us")
for name, n, avg, tot in c.execute(q):
    if flt in name:
        print(f"\"{name[:110]}\",{n},{avg / 1e3:.2f},{tot / 1e3:.1f}")
