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
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $OUT/trace_bench.json 2> $OUT/trace.err
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/fetch_bench.json 2> $OUT/fetch.err
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/write_bench.json 2> $OUT/write.err
echo "write rc=$?"
find $OUT -name "*.csv" | head -20

# summaries for profiles/ (copy gpurun_out/prof_$TAG/summary_* into profiles/ after the run)
python3 - <<PY
import csv, json, glob
out="$OUT"
rows=list(csv.reader(open(glob.glob(out+"/trace/*kernel_stats.csv")[0])))
with open(out+"/summary_kernel_stats.csv","w") as f:
    w=csv.writer(f)
    for r in rows:
        r[0]=r[0][:110]; w.writerow(r)
vals={}
for name,fn in (("FETCH_SIZE",glob.glob(out+"/pmc_fetch/*counter_collection.csv")[0]),("WRITE_SIZE",glob.glob(out+"/pmc_write/*counter_collection.csv")[0])):
    vals[name]=[float(r["Counter_Value"]) for r in csv.DictReader(open(fn)) if "xform_kernel" in r["Kernel_Name"] and r["Counter_Name"]==name]
fetch=sum(vals["FETCH_SIZE"])/len(vals["FETCH_SIZE"])*1024; write=sum(vals["WRITE_SIZE"])/len(vals["WRITE_SIZE"])*1024
json.dump({"kernel":"ac3mi::xform_kernel<false, 4, false>","frames_per_launch":65536,"fetch_bytes_raw":fetch,"fetch_bytes_corrected_x2":2*fetch,
  "write_bytes":write,"hbm_bytes_per_launch":2*fetch+write,"algorithmic_bytes_per_launch":65536*79872,
  "note":"FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE as read; separate --pmc passes","raw":vals},
  open(out+"/summary_hbm_traffic.json","w"),indent=1)
print(open(out+"/summary_kernel_stats.csv").read()[:400])
PY
