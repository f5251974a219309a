#!/bin/bash
# Developer helper (GPU box): the HBM counter passes of the joint step at T = 100 (with and without DROPOUT = 0.1) into gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r05d}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
J="--phase joint --steps 5 --warmup 2 --no-cpu-baseline --no-sustained --no-ref-types --no-dropout-legs"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_joint_fetch -- python3 $R/bench.py $J > /dev/null 2> $OUT/pmc_joint_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_joint_write -- python3 $R/bench.py $J > /dev/null 2> $OUT/pmc_joint_write.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_jointd_fetch -- python3 $R/bench.py $J --dropout 0.1 > /dev/null 2> $OUT/pmc_jointd_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_jointd_write -- python3 $R/bench.py $J --dropout 0.1 > /dev/null 2> $OUT/pmc_jointd_write.err
echo done
