"""Summarises the rocprofv3 databases of tools/profile.sh (gpurun_out/prof_<tag>/) for profiles/:
     python tools/prof_report.py <tag> <frames per bench run, warm-up included>
writes profiles/<tag>_kernel_stats.txt (kernel-trace --stats table + per-launch PMC sums of the path-tracing kernel + the
calibration of the SQ counters on the VALU issue microbenchmark) and profiles/<tag>_pmc.json (per-frame counter values that
bench.py's roofline block reads).  HBM bytes as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE
from separate passes, both in KB, FETCH_SIZE doubled on gfx950."""
import glob, json, os, re, sqlite3, sys

# the timed path-tracing kernel: any pipeline, COUNT = false (bench.py's count=1 pass for `roofline.actual` is left out)
IS_RENDER = re.compile(r"k_render\w*<[^>]*false>")

tag, frames = sys.argv[1], int(sys.argv[2])
base = os.path.join("gpurun_out", "prof_" + tag)
lines = []
db = sqlite3.connect(os.path.join(base, "trace", "trace_results.db"))
lines.append("== kernel stats (rocprofv3 --kernel-trace --stats; durations in ns) ==")
lines.append(f"{'kernel':<70} {'calls':>6} {'total_ns':>16} {'avg_ns':>16} {'min_ns':>14} {'max_ns':>14}")
kern_ns, kern_calls, prim_ms, kern_max = 0.0, 0, None, 0.0
for r in db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall():
    lines.append(f"{r[0][:70]:<70} {r[1]:>6} {r[2]:>16.0f} {r[3]:>16.0f} {r[4]:>14.0f} {r[5]:>14.0f}")
    if IS_RENDER.search(r[0]):
        kern_ns, kern_calls, kern_max = float(r[2]), int(r[1]), float(r[5])
    if "k_primary" in r[0]:
        prim_ms = float(r[3]) / 1e6
lines.append("")
lines.append(f"== PMC counters (separate --pmc passes), path-tracing kernel dispatches only: sum over the run's {frames} frames ==")
tot = {}
for p in sorted(glob.glob(os.path.join(base, "pmc*", "pmc_results.db"))):
    d = sqlite3.connect(p)
    for name, kern, n, s in d.execute("select counter_name, kernel_name, count(*), sum(value) from counters_collection group by counter_name, kernel_name"):
        if IS_RENDER.search(kern):
            tot[name] = (n, s)
            lines.append(f"{name:<34} launches={n:<4} sum={s:.6g}   per_frame={s / frames:.6g}")
calib = []
cp = os.path.join(base, "calib", "pmc_results.db")
if os.path.exists(cp):
    d = sqlite3.connect(cp)
    rows = d.execute("select dispatch_id, kernel_name, counter_name, sum(value) from counters_collection group by dispatch_id, kernel_name, counter_name order by dispatch_id").fetchall()
    by = {}
    for did, kern, name, v in rows:
        by.setdefault((did, kern), {})[name] = v
    try:
        micro = json.load(open(os.path.join(base, "calib.json")))["results"]
    except Exception:
        micro = []
    lines.append("")
    lines.append("== calibration: SQ counters over tools/dbg/valu_issue streams of known issue rate (timed dispatches only) ==")
    timed = [(k, c) for k, c in sorted(by.items()) if c.get("SQ_INSTS_VALU", 0) > 1e8]          # skips the 64-iteration warm-ups
    for (key, c), m in zip(timed, micro):
        iv = c["SQ_INSTS_VALU"]
        row = {"op": m["op"], "waves_per_simd": m["waves_per_simd"], "measured_cycles_per_wave_instr_per_simd": m["cycles_per_wave_instr_per_simd"],
               "SQ_ACTIVE_INST_VALU_x4_per_inst": 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0) / iv, "SQ_BUSY_CYCLES_per_inst": c.get("SQ_BUSY_CYCLES", 0) / iv,
               "SQ_WAVE_CYCLES_x4_per_inst": 4.0 * c.get("SQ_WAVE_CYCLES", 0) / iv, "GRBM_GUI_ACTIVE": c.get("GRBM_GUI_ACTIVE", 0), "SQ_INSTS_VALU": iv}
        calib.append(row)
        lines.append(f"{m['op']:<14} W={m['waves_per_simd']}  measured {m['cycles_per_wave_instr_per_simd']:.2f} cyc/instr/SIMD | 4*SQ_ACTIVE_INST_VALU/inst = {row['SQ_ACTIVE_INST_VALU_x4_per_inst']:.2f}"
                     f" | 4*SQ_WAVE_CYCLES/inst = {row['SQ_WAVE_CYCLES_x4_per_inst']:.2f} | SQ_BUSY_CYCLES/inst = {row['SQ_BUSY_CYCLES_per_inst']:.3f}")
os.makedirs("profiles", exist_ok=True)
open(os.path.join("profiles", f"{tag}_kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
per_frame = {k: v[1] / frames for k, v in tot.items()}
out = {"source": f"gpurun_out/prof_{tag}: rocprofv3 --kernel-trace [--stats | --pmc <one group per pass>] -- python3 bench.py --steps {frames - 1} --warmup 1 --no-cpu-baseline",
       "frames": frames, "kernel_launches": kern_calls, "kernel_ns_total": kern_ns, "kernel_ms_per_frame": kern_ns / 1e6 / frames if frames else None,
       # the timed launch alone (the run's largest: frames - 1 frames behind a one-frame warm-up launch): what bench.py's stale check compares
       "kernel_ms_per_frame_timed_launch": kern_max / 1e6 / (frames - 1) if frames > 1 and kern_calls == 2 else None,
       "per_frame": per_frame, "calibration": calib, "one_off": {"k_primary_ms": prim_ms}}
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    out["hbm"] = {"correction": "FETCH_SIZE x2 (gfx950), KB -> B; WRITE_SIZE KB -> B", "fetch_bytes_per_frame": 2.0 * per_frame["FETCH_SIZE"] * 1024.0,
                  "write_bytes_per_frame": per_frame["WRITE_SIZE"] * 1024.0, "hbm_bytes_per_frame": (2.0 * per_frame["FETCH_SIZE"] + per_frame["WRITE_SIZE"]) * 1024.0}
if "SQ_INSTS_VALU" in per_frame and "SQ_THREAD_CYCLES_VALU" in per_frame:
    out["valu"] = {"wave_instr_per_frame": per_frame["SQ_INSTS_VALU"], "lane_utilisation": per_frame["SQ_THREAD_CYCLES_VALU"] / (64.0 * per_frame["SQ_INSTS_VALU"])}
if "TA_TA_BUSY_sum" in per_frame and "GRBM_GUI_ACTIVE" in per_frame:
    g = per_frame["GRBM_GUI_ACTIVE"] / 8.0
    out["ta"] = {"note": "sums over the 256 TA/TCP instances (one per CU) divided by 256 x busy cycles; GRBM_GUI_ACTIVE is itself a sum over the 8 XCDs (it reads 8 x kernel time x clock), hence / 8",
                 "gpu_busy_cycles_per_frame": g,
                 "busy_frac": per_frame["TA_TA_BUSY_sum"] / (256.0 * g),
                 "addr_stalled_by_tc_frac": per_frame.get("TA_ADDR_STALLED_BY_TC_CYCLES_sum", 0.0) / (256.0 * g),
                 "data_stalled_by_tc_frac": per_frame.get("TA_DATA_STALLED_BY_TC_CYCLES_sum", 0.0) / (256.0 * g),
                 "tcp_tagconflict_stall_frac": per_frame.get("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", 0.0) / (256.0 * g),
                 "tcp_pending_stall_frac": per_frame.get("TCP_PENDING_STALL_CYCLES_sum", 0.0) / (256.0 * g)}
json.dump(out, open(os.path.join("profiles", f"{tag}_pmc.json"), "w"), indent=1)
print("\n".join(lines))
