#!/bin/bash
# A/B of two library builds on the joint phase of bench.py (see ab_bench.sh): bash scripts/dev/ab_joint.sh <tag> <rounds> <bench flags...>
set -e
TAG=${1:-abj}; ROUNDS=${2:-3}; shift 2 || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for i in $(seq 1 $ROUNDS); do
  for v in a b; do
    PC_DEV_LIB=$R/scripts/dev/ab/lib_$v.so python3 $R/bench.py --phase joint --no-cpu-baseline --no-ref-types --no-dropout-legs "$@" > $OUT/${v}_$i.json 2> $OUT/${v}_$i.err
    python3 -c "
import json
d=json.load(open('$OUT/${v}_$i.json'))
print('$v $i', d['ms_per_step'], d['roofline']['device_ms_per_step'], d['config']['final_loss'], flush=True)"
  done
done
