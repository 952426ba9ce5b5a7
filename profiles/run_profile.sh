#!/bin/bash
# Profiling recipe used for profiles/*: run on the GPU box via
#   gpurun -- 'bash profiles/run_profile.sh r01'
# Kernel trace and each PMC counter are collected in SEPARATE rocprofv3 runs.
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace.err
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fetch_bench.json 2> $OUT/fetch.err
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/write_bench.json 2> $OUT/write.err
echo "write rc=$?"
find $OUT -name "*.csv" | head -20
