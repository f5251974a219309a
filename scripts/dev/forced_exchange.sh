#!/bin/bash
# Developer probe (GPU box): the N > 1 code path on ONE rank (PC_DIST_FORCE=1: every exchange goes through RCCL) with the
# gradient exchange through the library's slot (native) and from Python hooks (hook): ms_per_step / host_enqueue_ms_per_step.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-forced}; mkdir -p $OUT
export PC_DIST_FORCE=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1
for v in native hook; do
  export MASTER_PORT=$((29600 + RANDOM % 300))
  python3 $R/bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --no-large --no-sustained --no-dropout-legs --exchange $v > $OUT/$v.json 2> $OUT/$v.err
  python3 - <<PY
import json
d=json.loads(open("$OUT/$v.json").read().strip().splitlines()[-1])      # (RCCL prints its version banner on stdout first)
print("$v", "p2v", d["ms_per_step"], d["host_enqueue_ms_per_step"], "| joint", d["joint"]["ms_per_step"], d["joint"]["host_enqueue_ms_per_step"],
      "| joint34800", d["joint_num_types_34800"]["ms_per_step"], d["joint_num_types_34800"]["host_enqueue_ms_per_step"], "|", d["rccl"]["exchange"], flush=True)
PY
done
