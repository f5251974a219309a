#!/bin/bash
# Developer probe (GPU box): per-kernel table of the joint step at the reference's shipped configuration (T = 34800, DROPOUT = 0.1).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-jd}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/$TAG/prof -o p -- python3 $R/bench.py --phase joint --types 34800 --dropout ${2:-0.1} --steps 50 --warmup 10 --no-cpu-baseline --no-ref-types --no-dropout-legs > $R/gpurun_out/$TAG/bench.json 2> $R/gpurun_out/$TAG/err.log
python3 $R/scripts/prof_summary.py $(ls $R/gpurun_out/$TAG/prof/*/p_results.db $R/gpurun_out/$TAG/prof/p_results.db 2>/dev/null | head -1) 60 30 > $R/gpurun_out/$TAG.txt
cat $R/gpurun_out/$TAG.txt
