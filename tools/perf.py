"""Quick headline timing of the library named by PTX_LIB_PATH (default: in-tree build): 2 frames of pixar_room
1920x1080x16spp, 8 bounces, literal and pruned.  Prints one line per mode; checks literal == pruned."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import numpy as np
import ptx_import
m = ptx_import.load()
cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
c = m.CONFIGS[cfg]
s = m.load_scene(c["scene"]); p = m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"])
ctx = m.Context(0); ctx.upload_scene(s); ctx.set_params(p)
for kv in filter(None, os.environ.get("PERF_OPTS", "").split(",")):
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
nfr = int(os.environ.get("PERF_FRAMES", "2"))
ref = np.load(os.path.join(ROOT, "tests", "golden", "frames.npz"))
imgs = {}
for trav in [int(x) for x in os.environ.get("PERF_TRAV", "0,3").split(",")]:
    ctx.set_option("traversal", trav)
    ctx.reset_accum(); ctx.render_frames([999]); ctx.synchronize()   # warm
    ctx.reset_accum(); ctx.reset_counters()
    t0 = time.time(); ctx.render_frames(list(range(1000, 1000 + nfr))); ctx.synchronize(); dt = time.time() - t0
    k = ctx.counters(); imgs[trav] = ctx.read_sum()
    print(f"{os.environ.get('PTX_LIB_PATH','default'):<40} {cfg} trav={trav} opts={os.environ.get('PERF_OPTS','')} ms/frame={dt*1000/nfr:.1f} Msamples/s={k['samples']/dt/1e6:.1f} Mrays/s={k['rays']/dt/1e6:.0f}", flush=True)
ks = sorted(imgs)
for k in ks[1:]:
    print(f"   trav {ks[0]} == trav {k}:", bool((imgs[ks[0]].view(np.uint32) == imgs[k].view(np.uint32)).all()), flush=True)
