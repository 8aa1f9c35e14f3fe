#!/bin/bash
# times the headline (8 frames per call) with every library variant in tools/dbg/variants/
cd $GRAFT_REPO_ROOT
for f in tools/dbg/variants/libptx_*.so; do
  PTX_LIB_PATH=$PWD/$f timeout -k 10 120 python bench.py --steps 8 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', round(d['ms_per_step'],2))" || exit 1
done
