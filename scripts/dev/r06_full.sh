#!/bin/bash
# One box: the whole GPU suite, then the default bench line.   bash scripts/dev/r06_full.sh <tag>
TAG=${1:-r06g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R && timeout -k 10 1000 python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1
tail -8 $OUT/gpu_tests.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 - "$OUT" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench.json"))
print(d["value"], d["ms_per_step"], d["sustained"]["ms_per_step"])
print("joint", d["joint"]["ms_per_step"], d["joint_num_types_34800"]["ms_per_step"], d["joint_num_types_34800_dropout_0p1"]["ms_per_step"], d["joint_dropout_0p1"]["ms_per_step"])
print("dropin", (d.get("dropin_dense") or {}).get("ms_per_step"), "p2vd", (d.get("p2v_dropout_0p1") or {}).get("ms_per_step"))
lc = d["large_catalogue"]
print("lc", {k: (v.get("ms_per_step") if isinstance(v, dict) and "ms_per_step" in v else None) for k, v in lc.items()})
hs = lc.get("hot_set") or {}
print("hot", {k: (v.get("ms_per_step"), (v.get("sharded_lookup") or {}).get("hot_rows_served_per_batch")) for k, v in hs.items() if isinstance(v, dict)})
print("tn", d["roofline"]["gemm_tn_kernel"])
PY
