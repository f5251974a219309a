#!/bin/bash
# Run on the GPU box (through gpurun): the round's measurement artefacts for bench.py's workloads.
#   bash scripts/profile_round.sh r02a   ->  gpurun_out/r02a/{bench.json, bench_under_rocprof*.json, prof/, prof_joint/,
#                                            pmc_fetch/, pmc_write/, pmc_joint_fetch/, pmc_joint_write/}
# NO_PMC=1 skips the counter passes (bench.py reads roofline.traffic from the COMMITTED profiles/<tag>_*pmc_traffic.json:
# run the counter passes, make_profiles.py, commit, then re-run with NO_PMC=1 for bench lines that carry the fresh figure).
# NO_BENCH=1 skips the plain default bench line (the first step).
# Kernel trace and the two PMC counters are separate rocprofv3 passes (never --pmc with a trace of the HIP/HSA domains).
# STAGE=n runs one slice of the whole (a gpurun call is limited to 20 minutes): 1 = the plain bench line + the two kernel traces,
# 2 = the six counter passes of the default workloads, 3 = the DROPOUT = 0.1 legs (PD + JD), 4 = CFG3, 5 = BIG.
set -e
case "${STAGE:-}" in
  1) NO_PMC=1 ;;
  2) NO_BENCH=1; NO_TRACE=1 ;;
  3) NO_BENCH=1; NO_TRACE=1; SKIP_MAIN_PMC=1; PD=1; JD=1 ;;
  4) NO_BENCH=1; NO_TRACE=1; SKIP_MAIN_PMC=1; CFG3=1 ;;
  5) NO_BENCH=1; NO_TRACE=1; SKIP_MAIN_PMC=1; BIG=1 ;;
esac
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -z "$NO_BENCH" ]; then
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
fi
if [ -z "$NO_TRACE" ]; then
rocprofv3 --kernel-trace --stats -d $OUT/prof -o $TAG -- python3 $R/bench.py --phase p2v --steps 30 --warmup 5 --no-cpu-baseline --no-sustained --no-large --no-dropout-legs --no-dropin > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
rocprofv3 --kernel-trace --stats -d $OUT/prof_joint -o ${TAG}_joint -- python3 $R/bench.py --phase joint --steps 25 --warmup 5 --no-cpu-baseline --no-sustained --no-ref-types --no-dropout-legs > $OUT/bench_joint_under_rocprof.json 2> $OUT/rocprof_joint.err
fi
if [ -z "$NO_PMC" ] && [ -z "$SKIP_MAIN_PMC" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --phase p2v --steps 5 --warmup 2 --no-cpu-baseline --no-sustained --no-large --no-dropout-legs --no-dropin > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --phase p2v --steps 5 --warmup 2 --no-cpu-baseline --no-sustained --no-large --no-dropout-legs --no-dropin > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_joint_fetch -- python3 $R/bench.py --phase joint --steps 5 --warmup 2 --no-cpu-baseline --no-sustained --no-ref-types --no-dropout-legs > /dev/null 2> $OUT/pmc_joint_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_joint_write -- python3 $R/bench.py --phase joint --steps 5 --warmup 2 --no-cpu-baseline --no-sustained --no-ref-types --no-dropout-legs > /dev/null 2> $OUT/pmc_joint_write.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_joint34800_fetch -- python3 $R/bench.py --phase joint --types 34800 --steps 5 --warmup 2 --no-cpu-baseline --no-sustained --no-ref-types --no-dropout-legs > /dev/null 2> $OUT/pmc_joint34800_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_joint34800_write -- python3 $R/bench.py --phase joint --types 34800 --steps 5 --warmup 2 --no-cpu-baseline --no-sustained --no-ref-types --no-dropout-legs > /dev/null 2> $OUT/pmc_joint34800_write.err
fi
if [ -z "$NO_BENCH" ]; then cat $OUT/bench.json; fi
# PD=1: the two DROPOUT = 0.1 legs of the default line that had no counter passes (Product2Vec and the joint step at T = 100):
# HBM traffic of the same commands (bench.py reads profiles/<tag>_p2vd_pmc_traffic.json / <tag>_jointd_pmc_traffic.json)
if [ -n "$PD" ]; then
PDA="--phase p2v --dropout 0.1 --steps 5 --warmup 2 --no-cpu-baseline --no-sustained --no-large --no-dropout-legs --no-dropin"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_p2vd_fetch -- python3 $R/bench.py $PDA > /dev/null 2> $OUT/pmc_p2vd_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_p2vd_write -- python3 $R/bench.py $PDA > /dev/null 2> $OUT/pmc_p2vd_write.err
JDB="--phase joint --dropout 0.1 --steps 5 --warmup 2 --no-cpu-baseline --no-sustained --no-ref-types --no-dropout-legs"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_jointd_fetch -- python3 $R/bench.py $JDB > /dev/null 2> $OUT/pmc_jointd_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_jointd_write -- python3 $R/bench.py $JDB > /dev/null 2> $OUT/pmc_jointd_write.err
fi
# BIG=1: BASELINE configs[4] on ONE GPU (100 M products x 256, Zipf negatives; the catalogue is generated in HBM): kernel
# trace + the HBM-traffic counters + the L2 hit counters of the same command (the hot-row cache question)
if [ -n "$BIG" ]; then
BIGARGS="--phase p2v --products 100000000 --dim 256 --negatives zipf --no-cpu-baseline --no-sustained --no-large --no-dropout-legs"
python3 $R/bench.py $BIGARGS --steps 30 --warmup 10 > $OUT/bench_big.json 2> $OUT/bench_big.err
rocprofv3 --kernel-trace --stats -d $OUT/prof_big -o ${TAG}_big -- python3 $R/bench.py $BIGARGS --steps 20 --warmup 5 > $OUT/bench_big_under_rocprof.json 2> $OUT/rocprof_big.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_big_fetch -- python3 $R/bench.py $BIGARGS --steps 5 --warmup 2 > /dev/null 2> $OUT/pmc_big_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_big_write -- python3 $R/bench.py $BIGARGS --steps 5 --warmup 2 > /dev/null 2> $OUT/pmc_big_write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_big_l2 -- python3 $R/bench.py $BIGARGS --steps 5 --warmup 2 > /dev/null 2> $OUT/pmc_big_l2.err
# the same three counter passes with UNIFORM negatives: what the Zipf head changes
BIGU="--phase p2v --products 100000000 --dim 256 --negatives uniform --no-cpu-baseline --no-sustained --no-large --no-dropout-legs"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_bigu_fetch -- python3 $R/bench.py $BIGU --steps 5 --warmup 2 > /dev/null 2> $OUT/pmc_bigu_fetch.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_bigu_l2 -- python3 $R/bench.py $BIGU --steps 5 --warmup 2 > /dev/null 2> $OUT/pmc_bigu_l2.err
fi
# CFG3=1: BASELINE configs[3] on ONE GPU (10 M products x 128 through the row-sharded lookup chain, G = 1): kernel trace + HBM traffic
if [ -n "$CFG3" ]; then
C3="--phase p2v --products 10000000 --table sharded --no-cpu-baseline --no-sustained --no-large --no-dropout-legs"
rocprofv3 --kernel-trace --stats -d $OUT/prof_cfg3 -o ${TAG}_cfg3 -- python3 $R/bench.py $C3 --steps 20 --warmup 5 > $OUT/bench_cfg3_under_rocprof.json 2> $OUT/rocprof_cfg3.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_cfg3_fetch -- python3 $R/bench.py $C3 --steps 5 --warmup 2 > /dev/null 2> $OUT/pmc_cfg3_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_cfg3_write -- python3 $R/bench.py $C3 --steps 5 --warmup 2 > /dev/null 2> $OUT/pmc_cfg3_write.err
fi
# JD=1: the joint step at the reference's shipped hyper-parameters (NUM_TYPES = 34800, DROPOUT = 0.1): kernel trace + HBM traffic
if [ -n "$JD" ]; then
if [ -z "$NO_PMC" ]; then
JDA="--phase joint --types 34800 --dropout 0.1 --steps 5 --warmup 2 --no-cpu-baseline --no-sustained --no-ref-types --no-dropout-legs"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_joint34800d_fetch -- python3 $R/bench.py $JDA > /dev/null 2> $OUT/pmc_joint34800d_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_joint34800d_write -- python3 $R/bench.py $JDA > /dev/null 2> $OUT/pmc_joint34800d_write.err
fi
rocprofv3 --kernel-trace --stats -d $OUT/prof_joint34800d -o ${TAG}_joint34800d -- python3 $R/bench.py --phase joint --types 34800 --dropout 0.1 --steps 25 --warmup 5 --no-cpu-baseline --no-sustained --no-ref-types --no-dropout-legs > $OUT/bench_joint34800d_under_rocprof.json 2> $OUT/rocprof_joint34800d.err
fi
