#!/bin/bash
# SQ instruction counters of the headline (4 frames, tools/perf.py) for every library in tools/dbg/variants/ and the in-tree build
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export PERF_TRAV=3 PERF_FRAMES=4
for f in $R/path-tracing_amd/csrc/libptx_hip.so $R/tools/dbg/variants/libptx_*.so; do
  n=$(basename $f .so)
  OUT=$R/gpurun_out/pmcv_$n
  rm -rf $OUT; mkdir -p $OUT
  PTX_LIB_PATH=$f timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $OUT/p1 -o pmc -- python3 $R/tools/perf.py > $OUT/p1.log 2>&1 || { echo "$n failed"; tail -5 $OUT/p1.log; exit 1; }
  grep "ms/frame" $OUT/p1.log
  python3 - $OUT/p1/pmc_results.db <<'PY'
import sqlite3, sys
d = sqlite3.connect(sys.argv[1])
r = {}
for name, kern, mx in d.execute("select counter_name, kernel_name, max(value) from counters_collection group by counter_name, kernel_name"):
    if "k_render" in kern: r[name] = mx / 4
print("   per frame:", {k: f"{v:.4g}" for k, v in sorted(r.items())}, "lane util", f"{r['SQ_THREAD_CYCLES_VALU'] / 64 / r['SQ_INSTS_VALU']:.3f}")
PY
done
