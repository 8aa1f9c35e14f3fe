"""First-light GPU check: parity vs the CPU oracle + a first timing. Run on the GPU box:
   python -u tools/gpu_check.py
"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401 (one HIP runtime)
import numpy as np
import ptx_import
m = ptx_import.load()
from oracle import orc

out = {}
ctx = m.Context(0)
for name, w, h in [("my_cornell_6", 64, 64), ("pixar_room", 96, 54), ("cornell", 32, 32), ("bedroom2_v5", 64, 36), ("teapot", 64, 36)]:
    s = m.load_scene(name)
    p = m.config_params(name, w, h, sample_count=4, max_path_length=4)
    ctx.upload_scene(s)
    ctx.set_params(p)
    for trav in (0, 1, 2, 3):
        ctx.set_option("traversal", trav)
        t_g, tri_g = ctx.primary_hits()
        t_o, tri_o, _ = orc.primary_hits(s, p)
        ok_prim = bool((tri_g == tri_o).all() and (t_g.view(np.uint32) == t_o.view(np.uint32)).all())
        d_g = ctx.primary_dirs(); d_o = orc.primary_dirs(p)
        ok_dirs = bool((d_g.view(np.uint32) == d_o.view(np.uint32)).all())
        res = {"prim": ok_prim, "dirs": ok_dirs}
        for memo in (0, 1):
            ctx.set_option("memo", memo)
            ctx.reset_accum(); ctx.reset_counters()
            ts = m.scene.timestamps(3)
            ctx.render_frames(ts)
            g = ctx.read_sum()
            o, oc = orc.render_frames(s, p, ts)
            nd = int((g.view(np.uint32) != o.view(np.uint32)).any(axis=2).sum())
            c = ctx.counters()
            res[f"memo{memo}_diff_pixels"] = nd
            res[f"memo{memo}_rays_eq"] = (c["rays"] == oc["rays"], c["samples"] == oc["samples"], c["ray_hits"] == oc["ray_hits"], c["light_attempts"] == oc["light_attempts"])
        print(name, "traversal", trav, res, flush=True)
        out[f"{name}_t{trav}"] = res

# timing: headline frame(s)
s = m.load_scene("pixar_room")
p = m.config_params("pixar_room", 1920, 1080, sample_count=16, max_path_length=8)
ctx.upload_scene(s); ctx.set_params(p); ctx.set_option("memo", 1)
imgs = {}
for trav in (0, 3):
    ctx.set_option("traversal", trav)
    ctx.reset_accum(); ctx.reset_counters()
    ts = m.scene.timestamps(2)
    t0 = time.time(); ctx.render_frames(ts); ctx.synchronize(); dt = time.time() - t0
    c = ctx.counters()
    imgs[trav] = ctx.read_sum()
    print("headline traversal", trav, "wall", dt, "kernel_ms", c["kernel_ms"], "Msamples/s", c["samples"] / dt / 1e6, "Mrays/s", c["rays"] / dt / 1e6,
          "rays/sample", c["rays"] / c["samples"], "traced/ref", c["rays_traced"] / c["rays"], flush=True)
    out[f"headline_t{trav}"] = dict(wall=dt, **c)
nd = int((imgs[0].view(np.uint32) != imgs[3].view(np.uint32)).any(axis=2).sum())
print("headline literal-vs-pruned differing pixels:", nd, "of", 1920 * 1080, flush=True)
out["headline_diff_pixels"] = nd
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gpu_check.json"), "w"), indent=1, default=str)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import imgio
imgio.write_png(os.path.join(ROOT, "gpurun_out", "headline_2frames.png"), imgio.linear_to_display(imgs[1] / 2))
