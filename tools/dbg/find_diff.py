"""Which pixel, which ray: frames of a config where the default walk and the unpruned walk differ (tools/whole_job_check.py found them).
For every such frame: the differing pixels under default / anyhit=0 / provable / unpruned, then the pixel's rays from the oracle
(orc.trace_pixel) traced on the GPU under the margin rule and without pruning; the rays whose closest hit differs are described.

    python tools/dbg/find_diff.py C5 52 107 ...
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import numpy as np  # noqa: E402
import ptx_import  # noqa: E402
from oracle import orc  # noqa: E402  (diagnostic tool, not product)

m = ptx_import.load()
cfg = sys.argv[1]
frames = [int(x) for x in sys.argv[2:]]
c = m.CONFIGS[cfg]
s = m.load_scene(c["scene"])
p = m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"])
ctx = m.Context(0)
ctx.upload_scene(s)
ctx.set_params(p)
MODES = {"default": dict(traversal=3, prune=2, anyhit=1), "no_anyhit": dict(traversal=3, prune=2, anyhit=0),
         "provable": dict(traversal=3, prune=1, anyhit=1), "unpruned": dict(traversal=2, prune=2, anyhit=1),
         "ref_tree_margin": dict(traversal=1, prune=2, anyhit=1),
         "pipeline0": dict(traversal=3, prune=2, anyhit=1, pipeline=0), "pipeline1": dict(traversal=3, prune=2, anyhit=1, pipeline=1)}
report = []
for f in frames:
    img = {}
    for name, opts in MODES.items():
        ctx.set_option("pipeline", 2)
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.reset_accum()
        ctx.render_frames([1000 + f], first_frame_number=f)
        img[name] = np.ascontiguousarray(ctx.read_sum()).copy()
    ref = img["unpruned"].view(np.uint32)
    for name in MODES:
        bad = np.argwhere((img[name].view(np.uint32) != ref).any(axis=2))
        print(f"frame {f} {name}: {len(bad)} differing pixels {bad[:4].tolist()}", flush=True)
    bad = np.argwhere((img["default"].view(np.uint32) != ref).any(axis=2))
    for py, px in bad[:4]:
        py, px = int(py), int(px)
        print(f"  pixel ({px},{py}) default {img['default'][py, px, :3]} unpruned {img['unpruned'][py, px, :3]}", flush=True)
        rgb, o, d, t, tri = orc.trace_pixel(s, p, 1000 + f, px, py)
        print(f"  oracle rgb {rgb} rays {len(t)}; equals unpruned: {bool((rgb.view(np.uint32) == img['unpruned'][py, px, :3].view(np.uint32)).all())}", flush=True)
        res = {}
        for name in ("default", "unpruned", "provable"):
            for k, v in MODES[name].items():
                ctx.set_option(k, v)
            res[name] = ctx.trace_closest(o, d)
        tu, ku = res["unpruned"]
        assert (tu.view(np.uint32) == t.view(np.uint32)).all() and (ku == tri).all(), "GPU unpruned != oracle"
        for name in ("default", "provable"):
            tm, km = res[name]
            dif = np.nonzero((tm.view(np.uint32) != tu.view(np.uint32)) | (km != ku))[0]
            print(f"  closest hits under {name} differing from unpruned: {len(dif)} of {len(t)} rays", flush=True)
            for r in dif[:4]:
                r = int(r)
                k_true, k_got = int(ku[r]), int(km[r])
                info = {"config": cfg, "frame": f, "pixel": [px, py], "ray_index": r, "mode": name, "o": o[r].tolist(), "d": d[r].tolist(),
                        "t_unpruned": float(tu[r]), "tri_unpruned": k_true, "t_pruned": float(tm[r]), "tri_pruned": k_got}
                for label, k in (("lost", k_true), ("kept", k_got)):
                    if k < 0:
                        continue
                    v = s.coords.reshape(-1, 9)[k].reshape(3, 3).astype(np.float64)
                    e1, e2 = v[1] - v[0], v[2] - v[0]
                    n = np.cross(e1, e2)
                    dd = d[r].astype(np.float64)
                    info[label] = {"tri": k, "v": v.tolist(), "edge_max": float(max(np.linalg.norm(e1), np.linalg.norm(e2), np.linalg.norm(v[2] - v[1]))),
                                   "area2": float(np.linalg.norm(n)), "det": float(np.dot(dd, np.cross(e2, e1))),
                                   "cos_to_normal": float(abs(np.dot(dd, n)) / (np.linalg.norm(n) * np.linalg.norm(dd))),
                                   "t_exact": float(np.dot(v[0] - o[r].astype(np.float64), n) / np.dot(dd, n)),
                                   "emission": s.emissions.reshape(-1, 3)[k].tolist()}
                print("   ", json.dumps(info), flush=True)
                report.append(info)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(report, open(os.path.join(ROOT, "gpurun_out", f"find_diff_{cfg}.json"), "w"), indent=1)
