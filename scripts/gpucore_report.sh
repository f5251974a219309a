#!/bin/bash
# After a GPU fault ("GPU core dump created: gpucore.N" in the cwd): name the faulting kernel and its wave state.
#   bash scripts/gpucore_report.sh <dir with gpucore.*> > gpurun_out/gpucore.txt
D=${1:-.}
for c in $D/gpucore.*; do
    [ -f "$c" ] || continue
    echo "== $c"
    timeout 120 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "info agents" -ex "info threads" -ex "thread apply all bt 3" $(which python3) "$c" 2>&1 | grep -v "^\[New\|^warning" | head -200
done
