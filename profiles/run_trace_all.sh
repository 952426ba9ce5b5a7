#!/bin/bash
# per-kernel durations of every engine kernel (primary bench + secondary encode/decode/transcode legs)
#   gpurun -- 'bash profiles/run_trace_all.sh TAG'
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/traceall_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/trace.err
python3 - <<PY
import csv, glob
rows=list(csv.reader(open(glob.glob("$OUT/trace/*kernel_stats.csv")[0])))
with open("$OUT/summary_kernel_stats.csv","w") as f:
    w=csv.writer(f)
    for r in rows:
        r[0]=r[0][:90]; w.writerow(r)
for r in rows[:12]: print(r[0][:60], r[1:5])
PY
