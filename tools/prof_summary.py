"""Summarises rocprofv3 rocpd databases (gpurun_out/prof_<tag>/) into a text file for profiles/."""
import glob, os, sqlite3, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
base = os.path.join("gpurun_out", "prof_" + tag)
lines = []
db = sqlite3.connect(os.path.join(base, "trace", "trace_results.db"))
lines.append("== kernel stats (rocprofv3 --kernel-trace --stats; durations in ns) ==")
lines.append(f"{'kernel':<70} {'calls':>6} {'total_ns':>16} {'avg_ns':>16} {'min_ns':>14} {'max_ns':>14}")
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
for r in rows:
    lines.append(f"{r[0][:70]:<70} {r[1]:>6} {r[2]:>16.0f} {r[3]:>16.0f} {r[4]:>14.0f} {r[5]:>14.0f}")
lines.append("")
lines.append("== per-dispatch PMC counters (separate --pmc passes), k_render dispatches only: per-launch average ==")
for p in sorted(glob.glob(os.path.join(base, "pmc*", "pmc_results.db"))):
    d = sqlite3.connect(p)
    cols = [c[1] for c in d.execute("pragma table_info(counters_collection)")]
    try:
        q = d.execute("select counter_name, kernel_name, count(*), sum(value) from counters_collection group by counter_name, kernel_name").fetchall()
    except Exception as e:
        lines.append(f"{p}: {e}; columns {cols}")
        continue
    for name, kern, n, tot in q:
        if "k_render" in kern:
            lines.append(f"{name:<34} launches={n:<4} avg_per_launch={tot / n:.6g}")
print("\n".join(lines))
