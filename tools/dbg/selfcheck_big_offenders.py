import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np, ptx_import
m = ptx_import.load(); ctx = m.Context(0)
for name, fam in (("bedroom2_v5", 6), ("teapot", 3)):
    s = m.load_scene(name); ctx.upload_scene(s)
    diag = float(np.linalg.norm(s.coords.reshape(-1, 3).max(0) - s.coords.reshape(-1, 3).min(0)))
    ctx.set_option("traversal", 3); ctx.set_option("prune", 2)
    r = ctx.selfcheck_pruning(fam, 1250000000, seed=777000 + fam)
    print(name, "family", fam, r["mismatches"], "diag", diag)
    for o, d, te, tp, ke, kp in r["first"]:
        d = np.asarray(d, dtype=np.float64); o = np.asarray(o, dtype=np.float64)
        inv = 1.0 / np.maximum(np.abs(d), 1e-30)
        print(f"   max|1/d| {inv.max():10.1f} |o| {np.linalg.norm(o):8.2f} t_exact {te:.6g} t_pruned {tp:.6g} dt {tp-te:.3g} lat {3e-3*diag*inv.max():.3g} tri {ke}->{kp} d {d}")
