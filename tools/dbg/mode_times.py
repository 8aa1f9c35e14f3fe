"""ms per frame (8 frames per call) of the option sets quoted in DESIGN.md section 7."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
import ptx_import
m = ptx_import.load()
for cfg in (sys.argv[1:] or ["headline", "C2", "C3", "C4"]):
    c = m.CONFIGS[cfg]
    s = m.load_scene(c["scene"]); p = m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"])
    ctx = m.Context(0); ctx.upload_scene(s); ctx.set_params(p)
    for label, opts in [("default", {}), ("anyhit=0", {"anyhit": 0}), ("prune=1", {"prune": 1}), ("prune=1 anyhit=0", {"prune": 1, "anyhit": 0}), ("traversal=2", {"traversal": 2})]:
        for k, v in {"traversal": 3, "prune": 2, "anyhit": 1, **opts}.items():
            ctx.set_option(k, v)
        ctx.reset_accum(); ctx.render_frames([999]); ctx.synchronize()
        t0 = time.time(); ctx.render_frames(list(range(1000, 1008))); ctx.synchronize(); dt = (time.time() - t0) / 8
        print(f"{cfg:9s} {label:18s} {dt*1000:7.1f} ms/frame", flush=True)
