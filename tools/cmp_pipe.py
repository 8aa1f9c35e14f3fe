"""Bit-compares the accumulation buffer of pipeline A vs pipeline B (default 1 vs 2) on a config, and times both."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import numpy as np
import ptx_import
m = ptx_import.load()
cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
pipes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2").split(",")]
nfr = int(os.environ.get("PERF_FRAMES", "2"))
c = m.CONFIGS[cfg]
s = m.load_scene(c["scene"]); p = m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"])
ctx = m.Context(0); ctx.upload_scene(s); ctx.set_params(p)
imgs = {}
for trav in [int(x) for x in os.environ.get("PERF_TRAV", "3").split(",")]:
    for pipe in pipes:
        ctx.set_option("traversal", trav); ctx.set_option("pipeline", pipe)
        ctx.reset_accum(); ctx.render_frames([999]); ctx.synchronize()
        ctx.reset_accum(); ctx.reset_counters()
        t0 = time.time(); ctx.render_frames(list(range(1000, 1000 + nfr))); ctx.synchronize(); dt = time.time() - t0
        k = ctx.counters(); imgs[(trav, pipe)] = ctx.read_sum()
        print(f"{cfg} trav={trav} pipeline={pipe} ms/frame={dt*1000/nfr:.1f} Msamples/s={k['samples']/dt/1e6:.1f} rays={k['rays']} traced={k['rays_traced']}", flush=True)
ks = sorted(imgs)
for k in ks[1:]:
    a, b = imgs[ks[0]].view(np.uint32), imgs[k].view(np.uint32)
    print(f"   {ks[0]} == {k}: {bool((a == b).all())}  (differing words: {int((a != b).sum())})", flush=True)
