#!/bin/bash
# instruction mix of the decode kernels on profiles/decode_ab.py (one rocprofv3 --pmc pass):  bash profiles/run_mix_ab.sh TAG [modes]
TAG=${1:-mix}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/mix_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/pmc_mix_a -o mix -- python3 $R/profiles/decode_ab.py 65536 "$@" > $OUT/ab.log 2> $OUT/ab.err
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("$OUT/pmc_mix_a/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"]
        if "decode" in k or "xform" in k: acc[k.replace("void ","").replace("ac3mi::","")[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    n=65536.0
    print("%-42s" % k, {c: round(sum(v)/len(v)/n) for c,v in d.items()})
PY
