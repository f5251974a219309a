"""Developer tool: per-kernel per-step summary of a rocprofv3 results .db (kernel-trace)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2]) if len(sys.argv) > 2 else 35.0
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{r[0][:92]:92s} {r[1]:5d} {r[2]/1e3/steps:9.1f}us/step {r[3]/1e3:8.1f}us {100*r[2]/tot:5.1f}%")
print("total us/step", tot / 1e3 / steps)
