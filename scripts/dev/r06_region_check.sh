#!/bin/bash
# region probe + the Product2Vec-side tests + three driver-flag bench lines.   bash scripts/dev/r06_region_check.sh <tag>
TAG=${1:-r06r}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
python scripts/dev/region_probe.py 2>&1 | tail -4 | cut -c1-420
timeout -k 10 700 python -m pytest tests/test_gpu_p2v_step.py tests/test_gpu_loader.py tests/test_gpu_dropout.py tests/test_gpu_streams.py tests/test_gpu_epoch_goldens.py tests/test_gpu_train_drivers.py tests/test_gpu_optimizer_and_errors.py tests/test_gpu_sharded.py tests/test_gpu_soak.py tests/test_gpu_config4.py -q -m gpu > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
for i in 1 2 3; do
python3 bench.py --phase p2v --steps 20 --warmup 5 --no-cpu-baseline --no-large --no-dropout-legs --no-dropin > $OUT/p2v_$i.json 2> $OUT/p2v_$i.err
python3 -c "import json;d=json.load(open('$OUT/p2v_$i.json'));print('driver flags', d['ms_per_step'], d['sustained']['ms_per_step'])"
done
