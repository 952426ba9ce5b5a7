set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02z
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02z/pytest_default.log 2>&1; echo "pytest default rc=$?"
timeout -k 10 300 bash profiles/run_r02.sh r02z 65536 > gpurun_out/r02z/prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
timeout -k 10 200 bash profiles/run_r02.sh r02w 1024 > gpurun_out/r02z/prof_w.log 2>&1; echo "profw rc=$?"
cd $GRAFT_REPO_ROOT
timeout -k 10 100 ./profiles/hbm_calibrate > gpurun_out/r02z/hbm_calibration.txt 2>&1; echo "cal rc=$?"
timeout -k 10 300 python bench.py > gpurun_out/r02z/bench.json 2> gpurun_out/r02z/bench.err; echo "bench rc=$?"
AC3MI_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r02z/bench_rehearse2.json 2> gpurun_out/r02z/bench_rehearse2.err; echo "rehearse rc=$?"
tail -c 600 gpurun_out/r02z/pytest_default.log
cat gpurun_out/r02z/bench.json
