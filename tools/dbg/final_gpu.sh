set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py > gpurun_out/bench_r02.json 2> gpurun_out/bench_r02.err || { tail -20 gpurun_out/bench_r02.err; exit 1; }
cat gpurun_out/bench_r02.json
