#!/bin/bash
# The round's closing run on one box: the whole GPU suite, smoke(), then bench.py with the DRIVER's flags.
TAG=${1:-r06z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R && timeout -k 10 1000 python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1
tail -3 $OUT/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
SECONDS=0
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
echo "bench wall: $SECONDS s"
python3 - "$OUT" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench_driver_flags.json"))
print(d["value"], d["ms_per_step"], d["sustained"]["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic_source"])
print("joint", d["joint"]["ms_per_step"], d["joint_num_types_34800"]["ms_per_step"], d["joint_num_types_34800_dropout_0p1"]["ms_per_step"])
print("lc", {k: (v.get("ms_per_step") if isinstance(v, dict) else None) for k, v in d["large_catalogue"].items()})
PY
