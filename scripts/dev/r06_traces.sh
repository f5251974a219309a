#!/bin/bash
# The kernel traces of the final code that STAGE=1 of profile_round.sh does not take (or took with the dense drop-in leg mixed in):
# Product2Vec leg alone, joint at T = 34800 without / with DROPOUT = 0.1.   bash scripts/dev/r06_traces.sh <tag>
TAG=${1:-r06k}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof $OUT/prof_joint34800 $OUT/prof_joint34800d
rocprofv3 --kernel-trace --stats -d $OUT/prof -o $TAG -- python3 $R/bench.py --phase p2v --steps 30 --warmup 5 --no-cpu-baseline --no-sustained --no-large --no-dropout-legs --no-dropin > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
rocprofv3 --kernel-trace --stats -d $OUT/prof_joint34800 -o ${TAG}_joint34800 -- python3 $R/bench.py --phase joint --types 34800 --steps 25 --warmup 5 --no-cpu-baseline --no-sustained --no-ref-types --no-dropout-legs > $OUT/bench_joint34800_under_rocprof.json 2> $OUT/rocprof_joint34800.err
rocprofv3 --kernel-trace --stats -d $OUT/prof_joint34800d -o ${TAG}_joint34800d -- python3 $R/bench.py --phase joint --types 34800 --dropout 0.1 --steps 25 --warmup 5 --no-cpu-baseline --no-sustained --no-ref-types --no-dropout-legs > $OUT/bench_joint34800d_under_rocprof.json 2> $OUT/rocprof_joint34800d.err
ls $OUT
