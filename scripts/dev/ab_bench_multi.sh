#!/bin/bash
# A/B/C/... of several builds (scripts/dev/ab/lib_<v>.so, selected through PC_DEV_LIB) on ONE box, alternating runs.
#   bash scripts/dev/ab_bench_multi.sh <tag> <rounds> "<v1 v2 ...>" [extra bench flags]
set -e
TAG=${1:-abm}; ROUNDS=${2:-3}; VARS=${3:-"a b"}; shift 3 || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for i in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    PC_DEV_LIB=$R/scripts/dev/ab/lib_$v.so python3 $R/bench.py --phase p2v --steps 100 --warmup 20 --no-cpu-baseline --no-large --no-dropout-legs "$@" > $OUT/${v}_$i.json 2> $OUT/${v}_$i.err
    python3 - <<PY
import json
d=json.load(open("$OUT/${v}_$i.json"))
print("$v $i", d["ms_per_step"], (d.get("sustained") or {}).get("median"), d["roofline"]["avg_launch_us"], flush=True)
PY
  done
done
