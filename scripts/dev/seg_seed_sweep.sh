#!/bin/bash
# Developer probe (GPU box, needs scripts/dev/ab/lib_t16.so = a -DPC_SORT_TIMING build): table_sort / table_segsum durations against the
# longest run of the step, over model seeds (the selection's favourite types differ): bash scripts/dev/seg_seed_sweep.sh <tag> <seeds...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-seeds}; shift
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
for seed in "$@"; do
  PC_PROBE_STEPS=40 PC_DEV_LIB=$R/scripts/dev/ab/lib_t16.so rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/s$seed -o p -- python3 $R/scripts/dev/sort_phase_times.py 0.1 $seed > $R/gpurun_out/$TAG/s$seed.txt 2> $R/gpurun_out/$TAG/s$seed.err
  echo "seed $seed: $(grep -o 'long c: longest wave [0-9]* clk, [0-9]* waves with work, longest run [0-9]* rows' $R/gpurun_out/$TAG/s$seed.txt)"
  python3 $R/scripts/prof_summary.py $(ls $R/gpurun_out/$TAG/s$seed/*/p_results.db $R/gpurun_out/$TAG/s$seed/p_results.db 2>/dev/null | head -1) 60 30 2>/dev/null | grep "table_s"
done
