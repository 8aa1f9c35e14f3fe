import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np, ptx_import
from oracle import orc
m = ptx_import.load(); ctx = m.Context(0)
s = m.load_scene("bedroom2_v5"); ctx.upload_scene(s)
ctx.set_option("traversal", 3); ctx.set_option("prune", 2)
r = ctx.selfcheck_pruning(6, 1250000000, seed=777006)
print(r["mismatches"])
o, d, te, tp, ke, kp = r["first"][0]
o = np.asarray(o, dtype=np.float32); d = np.asarray(d, dtype=np.float32)
print("o", o.tolist(), "d", d.tolist(), "signbits", np.signbit(d).tolist(), te, tp, ke, kp)
print("oracle", orc.trace_closest(s, o, d)[:2])
for pipe in (2, 0):
    for opts in (dict(traversal=3, prune=2), dict(traversal=2, prune=2), dict(traversal=3, prune=1), dict(traversal=1, prune=2), dict(traversal=0, prune=2), dict(traversal=3, prune=2, margin_lat=0)):
        ctx.set_option("margin_lat", 3000)
        ctx.set_option("pipeline", pipe)
        for k, v in opts.items(): ctx.set_option(k, v)
        t, tri = ctx.trace_closest(o, d)
        print("pipeline", pipe, opts, float(t[0]), int(tri[0]))
