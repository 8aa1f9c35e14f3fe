"""Counts the work of the REFERENCE algorithm (unpruned threaded traversal, every ray traced: options
traversal=0, memo=0, count=1) per frame for BASELINE.json's configs, on the GPU.  These are the Nv/Nt/Nh/Nl
figures the roofline uses.  The GPU literal path is bit-checked against the CPU oracle's
counters at small sizes by tests/test_gpu_parity.py.  Run on the GPU box:
    python -u tools/make_workload_counts.py   -> gpurun_out/workload_counts.json (copy to tests/golden/)
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import ptx_import
m = ptx_import.load()

FRAMES = {"C1": 1, "C2": 16, "C3": 32, "C4": 64, "headline": 64, "C5": 2}
out = {}
ctx = m.Context(0)
for ctx_opt in (("traversal", 0), ("memo", 0), ("count", 1)):
    ctx.set_option(*ctx_opt)
for cfg, nfr in FRAMES.items():
    c = m.CONFIGS[cfg]
    s = m.load_scene(c["scene"])
    p = m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"])
    ctx.upload_scene(s); ctx.set_params(p)
    frames = []
    t0 = time.time()
    for f in range(nfr):
        ctx.reset_counters(); ctx.reset_accum()
        ctx.render_frames([1000 + f])
        k = ctx.counters()
        frames.append({n: k[n] for n in ("samples", "rays", "node_visits", "tri_tests", "ray_hits", "light_attempts", "pixel_frames")})
    out[cfg] = dict(config=c, timestamps_from=1000, frames=frames)
    tot = {n: sum(fr[n] for fr in frames) for n in frames[0]}
    print(cfg, nfr, "frames", round(time.time() - t0, 1), "s", "rays/sample", tot["rays"] / tot["samples"], "Nv/ray", tot["node_visits"] / tot["rays"],
          "Nt/ray", tot["tri_tests"] / tot["rays"], flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "workload_counts.json"), "w"))
