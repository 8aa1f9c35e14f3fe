"""ptx_selfcheck_pruning at 100 x the size the tests use: 1.25e9 adversarial rays per family (12 families: 1.5e10 per scene and mode)
under the default and the provable rule, on both trees and both arithmetic paths.  Writes gpurun_out/selfcheck_big.json.
    python tools/dbg/selfcheck_big.py [scene ...]      RAYS_PER_FAMILY=... to change the size"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
import ptx_import
m = ptx_import.load()
ctx = m.Context(0)
out = {}
per = int(os.environ.get("RAYS_PER_FAMILY", "1250000000"))
combos = (("pipeline 2 arithmetic, leaf-SAH tree", dict(pipeline=2, traversal=3)), ("pipeline 0 arithmetic, leaf-SAH tree", dict(pipeline=0, traversal=3)),
          ("pipeline 2 arithmetic, reference tree", dict(pipeline=2, traversal=1)))
for name in sys.argv[1:] or ("pixar_room", "teapot", "bedroom2_v5", "my_cornell_6", "cornell"):
    s = m.load_scene(name)
    ctx.upload_scene(s)
    for label, opts in combos:
        for prune in (2, 1):
            for k, v in opts.items():
                ctx.set_option(k, v)
            ctx.set_option("prune", prune)
            t0 = time.time()
            mis, vio, emi = [], 0, 0
            for fam in range(12 if s.L else 8):
                r = ctx.selfcheck_pruning(fam, per, seed=555000 + fam)
                mis.append(r["mismatches"]); vio += r["anyhit_violations"]; emi += r["emitter_hits"]
            out[f"{name} | {label} | prune={prune}"] = {"rays_per_family": per, "mismatches_per_family": mis, "emitter_hits": emi, "anyhit_violations": vio,
                                                          "seconds": round(time.time() - t0, 1)}
            print(name, "|", label, "| prune", prune, mis, "emitter hits", emi, "any-hit violations", vio, f"{time.time() - t0:.0f} s", flush=True)
            json.dump(out, open(os.path.join(ROOT, "gpurun_out", "selfcheck_big.json"), "w"), indent=1)
