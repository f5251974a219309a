#!/bin/bash
# Upper bound on what ANY change of joint_tile_kernel's matrix instructions can buy at T = 100 (the six-product bf16
# form): the joint leg with the production library, then with a -DPC_EXP_NO_MFMA build (mfma16() = a keep-alive of its operands:
# WRONG numbers, every load / LDS hand-off / barrier / reduction still there), then the production library again, on ONE box.
# The knob library is built in the box's scratch copy only (build() and tests/test_abi.py refuse it).
#   bash scripts/dev/joint_mfma_knockout.sh <tag>
TAG=${1:-r06ko}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
run() {
  python3 bench.py --phase joint --steps 300 --warmup 30 --no-cpu-baseline --no-ref-types --no-dropout-legs > $OUT/$1.json 2> $OUT/$1.err
  python3 -c "import json;d=json.load(open('$OUT/$1.json'));print('$1', d['ms_per_step'], d['roofline'].get('device_ms_per_step'), 'build_flags', d.get('build_flags'))"
}
trace() {
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_$1 -o $1 -- python3 $R/bench.py --phase joint --steps 200 --warmup 20 --no-cpu-baseline --no-sustained --no-ref-types --no-dropout-legs > $OUT/$1_under_rocprof.json 2> $OUT/$1_rocprof.err)
  python3 - "$OUT/prof_$1" "$1" <<'PY'
import glob, sqlite3, sys, re
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
for n, c, a in cur.execute(f"select s.kernel_name, count(*), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 2 desc").fetchall():
    if c >= 100: print("   %s %5d launches  %7.2f us  %s" % (sys.argv[2], c, a / 1e3, re.sub(r"\(.*", "", n)[:70]))
PY
}
run prod_1; trace prod
cp p_companion_amd/libpcompanion_hip.so $OUT/lib.prod.so
SECONDS=0
PC_EXTRA_HIPCC_FLAGS=-DPC_EXP_NO_MFMA python3 -m p_companion_amd.build --force > $OUT/knob_build.log 2>&1 || { tail -5 $OUT/knob_build.log; exit 1; }
echo "knob build: $SECONDS s"
run nomfma_1; trace nomfma; run nomfma_2
cp $OUT/lib.prod.so p_companion_amd/libpcompanion_hip.so; rm -f $OUT/lib.prod.so
run prod_2
