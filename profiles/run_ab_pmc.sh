#!/bin/bash
# gpurun -- 'bash profiles/run_ab_pmc.sh TAG S MODES...': SQ counters of profiles/decode_ab.py's kernels, per frame
TAG=$1; shift
S=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/pmc -o m -- python3 $R/${SCRIPT:-profiles/decode_ab.py} "$@" > $OUT/pmc.log 2>&1
python3 - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open("$OUT/pmc/m_counter_collection.csv")):
    k = r["Kernel_Name"]
    if "ac3mi" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k, c in acc.items():
    d = n[k] * $S
    wc = c["SQ_WAVE_CYCLES"]
    print("%-60s valu %7.0f salu %7.0f lds %6.0f /frame  waves/frame %.2f  wait_any %.2f wait_inst %.2f active %.2f" % (
        k[:60], c["SQ_INSTS_VALU"] / d, c["SQ_INSTS_SALU"] / d, c["SQ_INSTS_LDS"] / d, c["SQ_WAVES"] / d,
        c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc))
PY
