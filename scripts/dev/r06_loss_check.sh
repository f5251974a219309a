#!/bin/bash
TAG=${1:-r06n}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R && timeout -k 10 700 python -m pytest tests/test_gpu_streams.py tests/test_gpu_p2v_step.py tests/test_gpu_ops.py tests/test_gpu_epoch_goldens.py tests/test_gpu_fullsize.py tests/test_gpu_dropout.py tests/test_gpu_modules.py tests/test_gpu_attention.py tests/test_gpu_optimizer_and_errors.py tests/test_gpu_sharded.py -q -m gpu > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
bash scripts/dev/ab_options.sh ${TAG}_ab 3 "${AB_A:-4=1}" "${AB_B:-4=0}" 2>&1 | tail -6
