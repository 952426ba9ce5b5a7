#!/bin/bash
# Issue/wait breakdown of the bench kernel: one rocprofv3 --pmc pass per counter group (no tracing flags).
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/a -o a -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/a.err
echo "a rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $OUT/b -o b -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/b.err
echo "b rc=$?"
rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/c -o c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/c.err
echo "c rc=$?"
python3 - <<PY
import csv, glob, collections
for g in ("a","b","c"):
    for fn in glob.glob("$OUT/%s/*counter_collection.csv" % g):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(fn)):
            if "xform_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items(): print(g, k, sum(v)/len(v), len(v))
PY
tail -2 $OUT/c.err
