#!/bin/bash
# gpurun -- 'bash profiles/run_ab_trace.sh TAG S MODES...': kernel trace of profiles/decode_ab.py (kernels back to back);
# SCRIPT=profiles/encode_cold.py in the environment traces that script instead
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/${SCRIPT:-profiles/decode_ab.py} "$@" > $OUT/trace.log 2>&1
grep -E "mode|cold" $OUT/trace.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/trace/t_kernel_stats.csv")))
for r in rows:
    if "ac3mi" in r["Name"]:
        print("%-70s calls %3s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
