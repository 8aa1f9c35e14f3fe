#!/bin/bash
# rocprofv3 passes over the headline bench (run on the GPU box from the repo root): kernel trace + stats first,
# then PMC counters in separate runs (never combined with tracing domains other than kernel-trace), then the
# calibration pass over the VALU issue microbenchmark (what do the SQ counters read for a stream of known rate?).
# Usage: bash tools/profile.sh <tag> [extra bench args]        -> gpurun_out/prof_<tag>/; summarise with tools/prof_report.py
set -o pipefail
TAG=${1:-r02}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps ${PROF_STEPS:-32} --warmup 1 --no-cpu-baseline --no-extras $@"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
echo "trace pass done"
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES" \
  "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU" \
  "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
  "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE" \
  "TA_BUSY_avr TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/pmc$i -o pmc -- $BENCH > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i ($PMC) failed"; tail -5 $OUT/pmc$i.log; exit 1; }
  echo "pmc pass $i done: $PMC"
done
if [ -x $R/tools/dbg/valu_issue ]; then
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/calib -o pmc -- $R/tools/dbg/valu_issue 2000 calib > $OUT/calib.json 2> $OUT/calib.log || { echo "calibration pass failed"; tail -5 $OUT/calib.log; exit 1; }
  echo "calibration pass done"
fi
