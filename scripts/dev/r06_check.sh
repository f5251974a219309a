#!/bin/bash
# One box: the finalize probe, the Product2Vec-side GPU tests, a kernel trace of the headline leg and its bench line.
#   bash scripts/dev/r06_check.sh <tag>
TAG=${1:-r06e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
bash $R/scripts/dev/bn_finalize_probe.sh ${TAG}_bnprobe 2>&1 | tail -6
cd $R && timeout -k 10 700 python -m pytest tests/test_gpu_ops.py tests/test_gpu_p2v_step.py tests/test_gpu_streams.py tests/test_gpu_epoch_goldens.py tests/test_gpu_fullsize.py tests/test_gpu_dim256.py tests/test_gpu_config4.py -x -q -m gpu > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o $TAG -- python3 $R/bench.py --phase p2v --steps 30 --warmup 5 --no-cpu-baseline --no-sustained --no-large --no-dropout-legs --no-dropin > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
python3 $R/bench.py --phase p2v --steps 100 --warmup 20 --no-cpu-baseline --no-large --no-dropout-legs --no-dropin > $OUT/bench_p2v.json 2> $OUT/bench_p2v.err
python3 -c "import json;d=json.load(open('$OUT/bench_p2v.json'));print('p2v', d['ms_per_step'], d['sustained']['ms_per_step'], d['roofline']['gemm_tn_kernel'])"
