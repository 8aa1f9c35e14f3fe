#!/bin/bash
# One rocprofv3 --pmc pass per argument over a short headline run (tools/perf.py with PERF_TRAV=1).
# Usage: bash tools/pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PERF_TRAV=${PERF_TRAV:-3} PERF_FRAMES=${PERF_FRAMES:-4}
i=0
for PMC in "$@"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/p$i -o pmc -- python3 $R/tools/perf.py > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
  echo "pass $i done: $PMC"
done
