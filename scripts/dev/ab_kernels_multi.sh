#!/bin/bash
# Per-kernel times of several library builds (scripts/dev/ab/lib_<name>.so, selected through PC_DEV_LIB) on ONE box, side by side.
#   bash scripts/dev/ab_kernels_multi.sh <tag> <name> [<name> ...]
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export PC_DEV_LIB=$R/scripts/dev/ab/lib_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$v -o $v -- python3 $R/bench.py --phase p2v --steps 30 --warmup 5 --no-cpu-baseline --no-sustained --no-large --no-dropout-legs > $OUT/bench_$v.json 2> $OUT/err_$v.log
done
unset PC_DEV_LIB
python3 - "$OUT" "$@" <<'PY'
import csv, glob, sys
out, names = sys.argv[1], sys.argv[2:]
def load(v):
    f = glob.glob("%s/prof_%s/**/*kernel_stats.csv" % (out, v), recursive=True)[0]
    return {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f))}
tabs = [load(v) for v in names]
keys = sorted(set().union(*tabs), key=lambda k: -tabs[0].get(k, 0))
print("%-90s" % "kernel" + "".join("%11s" % n[:10] for n in names))
for k in keys:
    if "at::" in k or "rocclr" in k: continue
    print("%-90s" % k[:90] + "".join("%11.1f" % t.get(k, 0) for t in tabs))
PY
