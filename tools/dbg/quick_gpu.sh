set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 200 --timeout-method thread -k "frames_bit_exact or full_size or golden or edge or big_and" > gpurun_out/quick_tests.log 2>&1 || { tail -30 gpurun_out/quick_tests.log; exit 1; }
tail -3 gpurun_out/quick_tests.log
timeout -k 10 200 python bench.py --steps 8 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['value'])"
