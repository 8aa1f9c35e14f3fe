"""ptx_selfcheck_pruning with 1e10 adversarial rays per scene (8 families x 1.25e9) under the default rule and the provable rule."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
import ptx_import
m = ptx_import.load()
ctx = m.Context(0)
out = {}
per = int(os.environ.get("RAYS_PER_FAMILY", "1250000000"))
for name in ("pixar_room", "teapot", "bedroom2_v5", "my_cornell_6", "cornell"):
    ctx.upload_scene(m.load_scene(name))
    for prune in (2, 1):
        ctx.set_option("traversal", 3); ctx.set_option("prune", prune)
        t0 = time.time()
        res = []
        for fam in range(8):
            r = ctx.selfcheck_pruning(fam, per, seed=777000 + fam)
            res.append(r["mismatches"])
        out[f"{name} prune={prune}"] = {"rays_per_family": per, "mismatches_per_family": res, "seconds": round(time.time() - t0, 1)}
        print(name, "prune", prune, res, f"{time.time() - t0:.0f} s", flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "selfcheck_big.json"), "w"), indent=1)
