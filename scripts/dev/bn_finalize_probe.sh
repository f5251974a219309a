#!/bin/bash
# bash scripts/dev/bn_finalize_probe.sh <tag>: per-launch durations of bn_finalize_fwd_kernel by tile count
set -e
TAG=${1:-bnprobe}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/prof -o bn -- python3 $R/scripts/dev/bn_finalize_probe.py > $OUT/probe.log 2> $OUT/probe.err
python3 - "$OUT" <<'PY'
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/prof/**/*_results.db", recursive=True)[0]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, d.end-d.start from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
fin = [r[1] / 1e3 for r in rows if "bn_finalize_fwd" in r[0]]
for i in range(0, len(fin), 10):
    c = fin[i:i + 10]
    print("launches %3d-%3d: median %.1f us  min %.1f  max %.1f" % (i, i + len(c) - 1, sorted(c)[len(c) // 2], min(c), max(c)))
PY
