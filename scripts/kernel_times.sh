#!/bin/bash
# usage: run_prof.sh TAG  -> gpurun_out/TAG.txt per-kernel summary
set -e
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/$TAG/prof -o p -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/$TAG/bench.json 2> $R/gpurun_out/$TAG/err.log
python3 $R/scripts/prof_summary.py $(ls $R/gpurun_out/$TAG/prof/*/p_results.db $R/gpurun_out/$TAG/prof/p_results.db 2>/dev/null | head -1) 35 30 > $R/gpurun_out/$TAG.txt
