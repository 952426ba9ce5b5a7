set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests/test_xform_gpu.py tests/test_stream_gpu.py tests/test_decode_gpu.py -m gpu -x -q > gpurun_out/ab/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/ab/pytest.log
for i in 1 2; do
AC3MI_XFORM_AHEAD=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-checks > gpurun_out/ab/b_ahead0_$i.json 2> gpurun_out/ab/err0_$i.txt; echo "rc=$?"
AC3MI_XFORM_AHEAD=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-checks > gpurun_out/ab/b_ahead1_$i.json 2> gpurun_out/ab/err1_$i.txt; echo "rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/b_ahead*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    e=d['extra']
    print(f, 'xform ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'decode', round(e['decode']['ms_per_pass'],3), 'stream', round(e['stream_layer']['ms_per_round'],2), e['stream_layer']['frames_per_s'])
PY
