#!/bin/bash
# Run on the GPU box: issue / wait breakdown of the Product2Vec step's kernels from the SQ counters (two passes of 8).
#   bash scripts/dev/sq_counters.sh tag  ->  gpurun_out/<tag>/sq{1,2}/   (aggregate with scripts/dev/sq_table.py <tag>)
set -e
TAG=${1:-sq}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--phase p2v --steps 5 --warmup 2 --no-cpu-baseline --no-sustained --no-large --no-dropout-legs"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq1 -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/sq1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $OUT/sq2 -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/sq2.err
ls $OUT/sq1 $OUT/sq2
