#!/bin/bash
# Counters of the headline (4 frames in one call, tools/perf.py) for every library in tools/dbg/variants/: two rocprofv3 --pmc passes each
# (SQ instruction counters; texture-address path), summarised into gpurun_out/pmc_variants.json
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export PERF_TRAV=3 PERF_FRAMES=4
for f in $R/tools/dbg/variants/libptx_*.so; do
  n=$(basename $f .so)
  OUT=$R/gpurun_out/pmcv_$n
  rm -rf $OUT; mkdir -p $OUT
  i=0
  for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
    i=$((i+1))
    PTX_LIB_PATH=$f timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/p$i -o pmc -- python3 $R/tools/perf.py > $OUT/p$i.log 2>&1 || { echo "$n pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
  done
  grep "ms/frame" $OUT/p1.log
done
python3 - $R <<'PY'
import glob, json, os, re, sqlite3, sys
R = sys.argv[1]
out = {}
for d in sorted(glob.glob(os.path.join(R, "gpurun_out", "pmcv_libptx_*"))):
    name = os.path.basename(d)[len("pmcv_libptx_"):]
    r = {}
    for p in sorted(glob.glob(os.path.join(d, "p*", "pmc_results.db"))):
        db = sqlite3.connect(p)
        for cn, kern, mx in db.execute("select counter_name, kernel_name, max(value) from counters_collection group by counter_name, kernel_name"):
            if "k_render" in kern: r[cn] = mx / 4          # the timed 4-frame launch is the largest dispatch
    m = re.search(r"ms/frame=([0-9.]+)", open(os.path.join(d, "p1.log")).read())
    r["ms_per_frame_under_profiler"] = float(m.group(1)) if m else None
    if "SQ_INSTS_VALU" in r: r["lane_utilisation"] = r["SQ_THREAD_CYCLES_VALU"] / 64 / r["SQ_INSTS_VALU"]
    if "TA_TA_BUSY_sum" in r: r["ta_busy_frac"] = r["TA_TA_BUSY_sum"] / 256 / (r["GRBM_GUI_ACTIVE"] / 8)
    out[name] = r
    print(name, {k: (f"{v:.4g}" if isinstance(v, float) else v) for k, v in r.items()})
json.dump({"what": "headline, 4 frames in one ptx_render_frames call (tools/perf.py), counters per frame of the path-tracing kernel", "variants": out},
          open(os.path.join(R, "gpurun_out", "pmc_variants.json"), "w"), indent=1)
PY
