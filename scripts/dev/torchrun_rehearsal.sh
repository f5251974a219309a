#!/bin/bash
# Developer probe (GPU box): the driver's N > 1 command line -- torch.distributed.run, one process per rank -- rehearsed with TWO ranks
# on the one card of the box (gloo; two ranks cannot share a GPU under RCCL), every leg of the default line included: the
# large_catalogue legs then run their N > 1 form (each rank generates ITS shard of the 10 M / 100 M-product table in HBM, the
# loader's all-to-all lookup crosses ranks).  Prints the line's keys and per-leg ms_per_step.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-torchrun2}; mkdir -p $OUT
export PC_DIST_BACKEND=gloo PC_FORCE_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) \
    $R/bench.py --gpus 2 --steps 10 --warmup 3 --no-sustained > $OUT/line.out 2> $OUT/line.err
echo "rc=$?"
python3 - <<PY
import json
lines=[l for l in open("$OUT/line.out").read().strip().splitlines() if l.startswith("{")]
d=json.loads(lines[-1])
print(len(lines), "json line(s); n_gpus", d["n_gpus"], d["rccl"])
print("p2v", d["ms_per_step"], "| joint", d["joint"].get("ms_per_step"), d["joint"].get("config",{}).get("launch","")[:40], "| joint34800", d["joint_num_types_34800"].get("ms_per_step"))
for k,v in d.get("large_catalogue",{}).items():
    print(k, v.get("ms_per_step"), v.get("error"), (v.get("sharded_lookup") or {}))
for k in ("p2v_dropout_0p1","joint_dropout_0p1","joint_num_types_34800_dropout_0p1"):
    print(k, d[k].get("ms_per_step"), d[k].get("error"))
PY
tail -5 $OUT/line.err
