"""ms/frame of single-frame and 8-frame launches of the headline against option pool_chunk."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
import ptx_import
m = ptx_import.load()
c = m.CONFIGS["headline"]
s = m.load_scene(c["scene"]); p = m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"])
ctx = m.Context(0); ctx.upload_scene(s); ctx.set_params(p)
for chunk in (16, 64, 256, 1024, 4096):
    ctx.set_option("pool_chunk", chunk)
    for n in (1, 8):
        ctx.reset_accum(); ctx.render_frames([999]); ctx.synchronize()
        t0 = time.time(); ctx.render_frames(list(range(1000, 1000 + n))); ctx.synchronize(); dt = time.time() - t0
        print(f"pool_chunk={chunk:5d} frames/launch={n} ms/frame={dt*1000/n:.1f}", flush=True)
