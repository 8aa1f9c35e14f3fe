"""Headline (or config argv[1]) ms per frame, 8 frames per call, under a list of option sets: argv[2:] like hot_nodes=0 or prune=1,anyhit=0."""
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
nfr = int(os.environ.get("PERF_FRAMES", "8"))
base = None
for opts in (sys.argv[2:] or [""]):
    ctx = m.Context(0); ctx.upload_scene(s); ctx.set_params(p)
    for kv in filter(None, opts.split(",")):
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    ctx.render_frames([999]); ctx.synchronize(); ctx.reset_accum()
    best = 1e9
    for rep in range(2):
        ctx.reset_accum()
        t0 = time.time(); ctx.render_frames(list(range(1000, 1000 + nfr))); ctx.synchronize(); best = min(best, time.time() - t0)
    img = ctx.read_sum()
    same = "" if base is None else f" same_bits_as_first={bool((img.view(np.uint32) == base.view(np.uint32)).all())}"
    if base is None:
        base = img
    print(f"{os.path.basename(os.environ.get('PTX_LIB_PATH', 'default')):<28} {cfg} [{opts}] {best * 1000 / nfr:.2f} ms/frame{same}", flush=True)
    ctx.close()
