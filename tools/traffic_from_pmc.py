"""HBM traffic of the path-tracing kernel from the FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh.

Usage: python tools/traffic_from_pmc.py <tag> <frames per bench run> > profiles/<round>_traffic.json
Corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3): both counters are in KB; on gfx950 FETCH_SIZE
reports half the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is taken as is.  Separate --pmc passes."""
import glob, json, os, sqlite3, sys

tag, frames = sys.argv[1], int(sys.argv[2])
base = os.path.join("gpurun_out", "prof_" + tag)
tot = {}
for p in sorted(glob.glob(os.path.join(base, "pmc*", "pmc_results.db"))):
    d = sqlite3.connect(p)
    for name, kern, n, s in d.execute("select counter_name, kernel_name, count(*), sum(value) from counters_collection group by counter_name, kernel_name"):
        if "k_render" in kern and name in ("FETCH_SIZE", "WRITE_SIZE"):
            tot[name] = (n, s)
(nf, f), (nw, w) = tot["FETCH_SIZE"], tot["WRITE_SIZE"]
fetch_b, write_b = 2.0 * f * 1024.0, w * 1024.0     # bytes over all k_render launches of the run (warm-up frames included)
print(json.dumps({"source": f"gpurun_out/prof_{tag}: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) over bench.py",
                  "launches": nf, "frames": frames, "fetch_size_kb_sum": f, "write_size_kb_sum": w,
                  "correction": "FETCH_SIZE x2 (gfx950), KB -> B; WRITE_SIZE KB -> B",
                  "hbm_bytes_per_frame": (fetch_b + write_b) / frames, "fetch_bytes_per_frame": fetch_b / frames,
                  "write_bytes_per_frame": write_b / frames}, indent=1))
