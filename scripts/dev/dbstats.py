"""Per-kernel averages of a rocprofv3 --kernel-trace results database: python scripts/dev/dbstats.py <dir or .db> [n]"""
import glob, os, sqlite3, sys
path = sys.argv[1]
dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True)
db = sqlite3.connect(dbs[0]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
nc = "display_name" if "display_name" in cols else "kernel_name"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in cur.execute(f"select s.{nc}, count(*), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.{nc} order by sum(d.end-d.start) desc limit {n}"):
    print(f"{r[2]/1e3:8.1f} us x{r[1]:5d}  {r[0][:100]}")
