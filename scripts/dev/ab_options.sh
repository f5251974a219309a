#!/bin/bash
# A/B of pc_set_option settings on ONE box, alternating (box-to-box spread of the pool is ~4 %).
#   bash scripts/dev/ab_options.sh <tag> <rounds> "<opts a>" "<opts b>" [extra bench flags]      e.g. "3=0" "3=1"
set -e
TAG=$1; ROUNDS=$2; A=$3; B=$4; shift 4 || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for i in $(seq 1 $ROUNDS); do
  for v in a b; do
    if [ $v = a ]; then O=$A; else O=$B; fi
    PC_BENCH_SET_OPTIONS=$O python3 $R/bench.py --phase p2v --steps 100 --warmup 20 --no-cpu-baseline --no-large --no-dropout-legs --no-dropin "$@" > $OUT/${v}_$i.json 2> $OUT/${v}_$i.err
    python3 - <<PY
import json
d=json.load(open("$OUT/${v}_$i.json"))
print("$v ($O) $i", d["ms_per_step"], (d.get("sustained") or {}).get("ms_per_step"), d["roofline"]["avg_launch_us"], flush=True)
PY
  done
done
