#!/bin/bash
# Kernel trace of cold / warm encoder passes:  gpurun -- 'bash profiles/run_enc_trace.sh TAG [FRAMES]'
set -o pipefail
TAG=${1:-enc}
FR=${2:-65536}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/profiles/encode_cold.py $FR 5 > $OUT/run.log 2> $OUT/trace.err
echo "trace rc=$?"
cat $OUT/run.log | tail -1
f=$(ls $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv && cut -d, -f1-4 $OUT/kernel_stats.csv | head -12
