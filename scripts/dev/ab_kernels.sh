#!/bin/bash
# Per-kernel times of two library builds (scripts/dev/ab/lib_a.so / lib_b.so, selected through PC_DEV_LIB) on ONE box: a kernel
# trace of the Product2Vec phase with each, per-kernel average us side by side.   bash scripts/dev/ab_kernels.sh <tag> [bench flags]
set -e
TAG=${1:-abk}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in a b; do
  export PC_DEV_LIB=$R/scripts/dev/ab/lib_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$v -o $v -- python3 $R/bench.py --phase p2v --steps 30 --warmup 5 --no-cpu-baseline --no-sustained --no-large --no-dropout-legs "$@" > $OUT/bench_$v.json 2> $OUT/err_$v.log
done
unset PC_DEV_LIB
python3 - <<PY
import csv, glob
def load(v):
    f = glob.glob("$OUT/prof_%s/**/*kernel_stats.csv" % v, recursive=True)[0]
    return {r["Name"]: (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(open(f))}
a, b = load("a"), load("b")
print("%-100s %9s %9s %6s" % ("kernel", "a us", "b us", "calls"))
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0))[0] * a.get(k, (0, 0))[1])):
    if "at::" in k or "rocclr" in k: continue
    print("%-100s %9.1f %9.1f %6d" % (k[:100], a.get(k, (0, 0))[0], b.get(k, (0, 0))[0], max(a.get(k, (0, 0))[1], b.get(k, (0, 0))[1])))
PY
