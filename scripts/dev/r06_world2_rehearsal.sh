#!/bin/bash
# Two ranks on the ONE card of this box over gloo (PC_FORCE_DEVICE=0): bench.py's N > 1 branches INCLUDING the large legs --
# configs[3] (10 M products, sharded), configs[4] (100 M x 256 sharded over 2 "GPUs", Zipf negatives, replicated hot set) and the
# hot_set comparison -- which no N > 1 job had driven before round 6.  Not a timing: both ranks share one GPU.
TAG=${1:-r06w2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
PC_DIST_BACKEND=gloo PC_FORCE_DEVICE=0 timeout -k 10 900 python3 bench.py --gpus 2 --steps 4 --warmup 2 --phase p2v --no-cpu-baseline --no-sustained --no-dropout-legs > $OUT/bench_world2.json 2> $OUT/bench_world2.err
echo rc=$?
tail -c 1500 $OUT/bench_world2.err
python3 - "$OUT" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench_world2.json"))
print(d["n_gpus"], d["ms_per_step"], d["rccl"])
lc = d["large_catalogue"]
for k, v in lc.items():
    if isinstance(v, dict):
        print(k, {kk: v[kk] for kk in ("ms_per_step", "error", "sharded_lookup") if kk in v} if "ms_per_step" in v or "error" in v else {n: (x.get("ms_per_step"), (x.get("sharded_lookup") or {}).get("hot_rows_served_per_batch"), x.get("error")) for n, x in v.items() if isinstance(x, dict)})
PY
