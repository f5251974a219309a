#!/bin/bash
# One box: the joint-step GPU tests, the T = 34800 leg's bench line and its kernel trace.   bash scripts/dev/r06_joint_check.sh <tag>
TAG=${1:-r06h}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R && timeout -k 10 600 python -m pytest tests/test_gpu_joint_fused.py tests/test_gpu_train_drivers.py tests/test_gpu_fullsize.py tests/test_gpu_table_gradients.py tests/test_gpu_dropout.py tests/test_gpu_epoch_goldens.py tests/test_gpu_optimizer_and_errors.py -q -m gpu > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
for i in 1 2; do
python3 $R/bench.py --phase joint --types 34800 --no-cpu-baseline --no-ref-types --no-dropout-legs > $OUT/bench_j34800_$i.json 2> $OUT/bench_j34800_$i.err
python3 -c "import json;d=json.load(open('$OUT/bench_j34800_$i.json'));print('T=34800', d['ms_per_step'], d['roofline']['device_ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_joint34800 -o ${TAG}_joint34800 -- python3 $R/bench.py --phase joint --types 34800 --steps 25 --warmup 5 --no-cpu-baseline --no-sustained --no-ref-types --no-dropout-legs > $OUT/bench_joint34800_under_rocprof.json 2> $OUT/rocprof_joint34800.err
python3 - "$OUT" <<'PY'
import glob, sqlite3, sys, statistics, re
db = sqlite3.connect(glob.glob(sys.argv[1] + "/prof_joint34800/**/*_results.db", recursive=True)[0]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, count(*), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 2 desc").fetchall()
for n, c, a in rows:
    if c >= 20: print("%5d  %7.1f us  %s" % (c, a / 1e3, re.sub(r"\(.*", "", n)[:70]))
PY
