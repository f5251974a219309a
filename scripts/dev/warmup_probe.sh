#!/bin/bash
# how much of the driver-flag headline (--steps 20 --warmup 5) is warm-up: the same 20 timed steps after 5 / 50 / 200 warm-up steps
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for w in 5 50 200; do
  python3 $R/bench.py --phase p2v --steps 20 --warmup $w --no-cpu-baseline --no-large --no-dropout-legs --no-sustained > /tmp/wp.json 2>/dev/null
  python3 -c "
import json
d=json.load(open('/tmp/wp.json'))
print('warmup $w:', d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'], flush=True)"
done
done
