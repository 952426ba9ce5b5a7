set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for c in 1 0 2 4 8 1 0; do echo "chunks=$c"; AC3MI_STREAM_CHUNKS=$c timeout -k 10 120 python profiles/stream_rate.py 8192 20 | tail -2; done
timeout -k 10 300 python -m pytest tests/test_stream_gpu.py -m gpu -x -q | tail -2
AC3MI_DECODE_MODE=3 timeout -k 10 300 bash profiles/run_r02.sh r02w3 65536 | tail -12
