set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/jp
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/jp/prof -o p -- python3 $R/bench.py --phase joint --steps 100 --warmup 20 --no-cpu-baseline > $R/gpurun_out/jp/bench.json 2> $R/gpurun_out/jp/err.log
python3 $R/scripts/prof_summary.py $(ls $R/gpurun_out/jp/prof/*/p_results.db $R/gpurun_out/jp/prof/p_results.db 2>/dev/null | head -1) 120 40 > $R/gpurun_out/jp.txt
