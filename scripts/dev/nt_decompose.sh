#!/bin/bash
# Developer experiment (GPU box): which part of the gemm_nt_kernel loop costs what.  Rebuilds the library with one
# -DPC_EXP_* knob at a time (WRONG numerical results, right instruction mix), traces 12 steps of the P2V bench and keeps
# the average duration of every gemm_nt_kernel instantiation in gpurun_out/decomp/summary.txt.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/decomp
mkdir -p $OUT
: > $OUT/summary.txt
cd /tmp && export TMPDIR=/tmp
for K in ${KNOBS:-BASE NO_SPLIT NO_LDSREAD NO_MFMA NO_DMA DMA_L2}; do
  TAG=$(echo $K | sed 's/ -DPC_EXP_/+/g')
  touch $R/p_companion_amd/csrc/gemm_nt.hip $R/p_companion_amd/csrc/gemm_tn.hip
  # (several knobs in one build: comma-separated, e.g. KNOBS="DMA_L2,STAGGER=8")
  if [ "$K" = "BASE" ]; then FL=""; else FL="-DPC_EXP_${K//,/ -DPC_EXP_}"; fi
  (cd $R && PC_EXTRA_HIPCC_FLAGS="$FL" python3 -m p_companion_amd.build > /tmp/build_decomp.log 2>&1)
  rm -rf /tmp/prof_decomp
  rocprofv3 --kernel-trace --stats -d /tmp/prof_decomp -o t -- python3 $R/bench.py --phase p2v --steps 12 --warmup 3 --no-cpu-baseline --no-sustained > /tmp/b_decomp.json 2> /tmp/b_decomp.err
  python3 - "$TAG" >> $OUT/summary.txt <<'PY'
import sqlite3, glob, sys
f = glob.glob('/tmp/prof_decomp/*.db') + glob.glob('/tmp/prof_decomp/*/*.db')
c = sqlite3.connect(f[0])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
for n, cnt, avg in c.execute(f"select s.kernel_name,count(*),avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%gemm_nt_kernel%' or s.kernel_name like '%gemm_tn8%' or s.kernel_name like '%tn_reduce_kernel%' group by s.kernel_name"):
    print(sys.argv[1], n[4:52].replace(' ', ''), cnt, round(avg, 1))
PY
  echo "done $TAG"
done
