#!/bin/bash
# Kernel trace of any script, then each ac3mi kernel's launches in order with their durations:
#   gpurun -- 'bash profiles/run_trace_of.sh TAG profiles/SCRIPT.py ARGS...'
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
SCRIPT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python3 $SCRIPT "$@" > $OUT/run.log 2> $OUT/trace.err
echo "trace rc=$?"
cat $OUT/run.log
python3 - <<PY
import csv
rows = [r for r in csv.DictReader(open("$OUT/trace/trace_kernel_trace.csv")) if "ac3mi" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows:
    gap = (int(r["Start_Timestamp"]) - prev) / 1e3 if prev else 0.0
    prev = int(r["End_Timestamp"])
    print("%10.1f us  gap %10.1f  grid %9s  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, gap, r["Grid_Size_X"], r["Kernel_Name"][:70]))
PY
