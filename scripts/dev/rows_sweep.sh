#!/bin/bash
# on the GPU box: the row-count sweep, then per-kernel durations at four row counts (one traced process each)
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/rows; export TMPDIR=/tmp
python scripts/dev/rows_sweep_probe.py > gpurun_out/rows/sweep.txt 2>&1
for R in 65536 66048 86400 98304; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/rows/t$R -o t -- python3 scripts/dev/rows_sweep_probe.py $R > gpurun_out/rows/t$R.log 2>&1
  f=$(find gpurun_out/rows/t$R -name '*kernel_stats.csv' | head -1)
  echo "== R=$R" >> gpurun_out/rows/kernels.txt
  head -12 "$f" >> gpurun_out/rows/kernels.txt
  rm -rf gpurun_out/rows/t$R
done
cat gpurun_out/rows/sweep.txt; cat gpurun_out/rows/kernels.txt
