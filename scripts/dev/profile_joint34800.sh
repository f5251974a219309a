#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_joint34800_fetch -- python3 $R/bench.py --phase joint --types 34800 --steps 5 --warmup 2 --no-cpu-baseline --no-sustained --no-ref-types --no-dropout-legs > /dev/null 2> $OUT/pmc_joint34800_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_joint34800_write -- python3 $R/bench.py --phase joint --types 34800 --steps 5 --warmup 2 --no-cpu-baseline --no-sustained --no-ref-types --no-dropout-legs > /dev/null 2> $OUT/pmc_joint34800_write.err
rocprofv3 --kernel-trace --stats -d $OUT/prof_joint34800 -o r05c_joint34800 -- python3 $R/bench.py --phase joint --types 34800 --steps 25 --warmup 5 --no-cpu-baseline --no-sustained --no-ref-types --no-dropout-legs > $OUT/bench_joint34800_under_rocprof.json 2> $OUT/rocprof_joint34800.err
echo done
