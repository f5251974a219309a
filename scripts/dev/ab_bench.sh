#!/bin/bash
# A/B of two builds of libpcompanion_hip.so on ONE box (box-to-box spread of the pool is ~4 %: only alternating runs on the
# same box compare).  scripts/dev/ab/lib_a.so / lib_b.so are built beforehand (e.g. HEAD vs working tree) and selected through PC_DEV_LIB
# (p_companion_amd/_lib.py) -- the product library in the tree is never overwritten; each round runs the
# Product2Vec phase of bench.py with a, then b.   bash scripts/dev/ab_bench.sh <tag> [rounds] [extra bench flags]
set -e
TAG=${1:-ab}; ROUNDS=${2:-3}; shift 2 || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for i in $(seq 1 $ROUNDS); do
  for v in a b; do
    PC_DEV_LIB=$R/scripts/dev/ab/lib_$v.so python3 $R/bench.py --phase p2v --steps 100 --warmup 20 --no-cpu-baseline --no-large --no-dropout-legs "$@" > $OUT/${v}_$i.json 2> $OUT/${v}_$i.err
    python3 - <<PY
import json
d=json.load(open("$OUT/${v}_$i.json"))
print("$v $i", d["ms_per_step"], (d.get("sustained") or {}).get("ms_per_step"), d["roofline"]["avg_launch_us"], flush=True)
PY
  done
done
