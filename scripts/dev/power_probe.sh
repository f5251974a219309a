#!/bin/bash
# Developer probe (GPU box): socket power and shader clock while the P2V bench runs (rocm-smi polled until it exits).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/power
mkdir -p $OUT
: > $OUT/samples.txt
python3 $R/bench.py --phase p2v --steps 6000 --warmup 10 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power \(W\)|sclk|GPU use" | sed -E 's/.*: //' | tr '\n' ' ' >> $OUT/samples.txt
  echo >> $OUT/samples.txt
  sleep 0.3
done
wait $BP
