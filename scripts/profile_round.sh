#!/bin/bash
# Run on the GPU box (through gpurun): the round's measurement artefacts for bench.py's P2V workload.
#   bash scripts/profile_round.sh r01c   ->  gpurun_out/r01c/{bench.json, bench_under_rocprof.json, prof/, pmc_fetch/, pmc_write/}
# NO_PMC=1 skips the two counter passes (bench.py reads roofline.traffic from the COMMITTED profiles/<tag>_pmc_traffic.json:
# run the counter passes, make_profiles.py, commit, then re-run with NO_PMC=1 for bench lines that carry the fresh figure).
# Kernel trace and the two PMC counters are separate rocprofv3 passes (never --pmc with a trace of the HIP/HSA domains).
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 100 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/prof -o $TAG -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
if [ -z "$NO_PMC" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
fi
cat $OUT/bench.json
