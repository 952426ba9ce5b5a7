#!/bin/bash
# instruction mix / wait breakdown of every engine kernel during the secondary (encode/decode/transcode) legs
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcenc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/a -o a -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --frames ${2:-16384} > /dev/null 2> $OUT/a.err
python3 - <<PY
import csv, glob, collections
for fn in glob.glob("$OUT/a/*counter_collection.csv"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"][:52]
        if "ac3mi" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,d in acc.items():
        w=sum(d["SQ_WAVES"])/len(d["SQ_WAVES"])
        print(k, "waves", int(w), {c: round(sum(v)/len(v)/w) for c,v in d.items() if c!="SQ_WAVES"})
PY
