#!/bin/bash
# gpurun -- 'bash profiles/run_pmc_groups.sh TAG SCRIPT ARGS...': SQ counter groups (separate rocprofv3 --pmc runs) of a script's ac3mi kernels,
# summed per kernel over its launches; prints per-launch values / 65536 frames where it makes sense
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH"
G2="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"
G3="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -o m -- python3 $R/"$@" > $OUT/g$i.log 2>&1
  echo "group $i rc=$?"
done
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
for fn in glob.glob("$OUT/g*/m_counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "ac3mi" not in k: continue
        k = k.replace("void ", "").replace("ac3mi::", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, c in acc.items():
    print(k)
    wc = c["SQ_WAVE_CYCLES"] / max(n[k]["SQ_WAVE_CYCLES"], 1)
    for name in sorted(c):
        v = c[name] / n[k][name]
        print("   %-26s %14.0f per launch   %.3f of wave cycles" % (name, v, v / wc if wc else 0))
PY
