#!/bin/bash
# Round-4 profiling recipe:  gpurun -- 'bash profiles/run_r04.sh TAG'  (encode / transcode legs from fresh stream state only: --no-warm)
# Kernel trace and every PMC counter group are collected in SEPARATE rocprofv3 runs (program directly after --).
# Copy gpurun_out/prof_TAG/summary_* to profiles/TAG_* afterwards.
set -o pipefail
TAG=${1:-r04}
FR=${2:-65536}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export AC3MI_BENCH_MILLION=0           # its tiles would be the largest grids: per-frame counts are taken on the 65 536-frame legs
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-checks --no-warm --frames $FR"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $B > $OUT/trace_bench.json 2> $OUT/trace.err
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $B > $OUT/fetch_bench.json 2> $OUT/fetch.err
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $B > $OUT/write_bench.json 2> $OUT/write.err
echo "write rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/pmc_mix_a -o mix -- $B > $OUT/mix_bench.json 2> $OUT/mix.err
echo "mix rc=$?"
python3 $R/profiles/summarize_pmc.py $OUT $TAG $FR
