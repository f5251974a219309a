#!/bin/bash
# Developer probe (GPU box): per-kernel table of the joint step at any (types, dropout): bash scripts/dev/joint_kernel_times_any.sh <tag> <types> <dropout>
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-jk}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/$TAG/prof -o p -- python3 $R/bench.py --phase joint --types ${2:-100} --dropout ${3:-0.0} --steps 50 --warmup 10 --no-cpu-baseline --no-ref-types --no-dropout-legs > $R/gpurun_out/$TAG/bench.json 2> $R/gpurun_out/$TAG/err.log
python3 $R/scripts/prof_summary.py $(ls $R/gpurun_out/$TAG/prof/*/p_results.db $R/gpurun_out/$TAG/prof/p_results.db 2>/dev/null | head -1) 60 30 > $R/gpurun_out/$TAG.txt
cat $R/gpurun_out/$TAG.txt
