"""ms/frame of a config against frames per launch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
import numpy as np
import ptx_import
m = ptx_import.load()
cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
c = m.CONFIGS[cfg]
s = m.load_scene(c["scene"]); p = m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"])
ctx = m.Context(0); ctx.upload_scene(s); ctx.set_params(p)
ref = {}
for order in (0,):
    for n in [int(x) for x in os.environ.get('LAUNCH_SIZES', '1,4,8,32,64').split(',')]:
        ctx.reset_accum(); ctx.render_frames([999]); ctx.synchronize()     # warm (and, with order = 1, the cost map)
        ctx.reset_accum()
        t0 = time.time(); ctx.render_frames(list(range(1000, 1000 + n))); ctx.synchronize(); dt = time.time() - t0
        img = ctx.read_sum()
        if order == 0: ref[n] = img
        else: assert (img.view(np.uint32) == ref[n].view(np.uint32)).all()
        print(f"{cfg} frames/launch={n:3d} ms/frame={dt*1000/n:.1f} total={dt*1000:.1f} ms", flush=True)
    # a loop of one-frame calls that only synchronises at the end (the reference's frame loop, main.js:400-405, without its sleeps)
    for overlap in (0, 1):
        ctx.set_option("overlap", overlap)
        ctx.reset_accum(); ctx.render_frames([999]); ctx.synchronize()
        ctx.reset_accum()
        n = 32
        t0 = time.time()
        for f in range(n):
            ctx.render_frames([1000 + f], first_frame_number=f)
        ctx.synchronize(); dt = time.time() - t0
        assert (ctx.read_sum().view(np.uint32) == ref[32].view(np.uint32)).all() if 32 in ref else True
        print(f"{cfg} {n} calls of 1 frame, overlap={overlap}: ms/frame={dt*1000/n:.1f}", flush=True)
