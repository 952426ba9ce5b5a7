set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_stream_gpu.py tests/test_tools_gpu.py -m gpu -x -q | tail -2
AC3MI_STREAM_TRACE=1 timeout -k 10 120 python profiles/stream_rate.py 8192 4 2>&1 | tail -9
timeout -k 10 120 python profiles/stream_rate.py 8192 50 2>&1 | tail -5
timeout -k 10 120 python profiles/stream_rate.py 2048 50 2>&1 | tail -5
