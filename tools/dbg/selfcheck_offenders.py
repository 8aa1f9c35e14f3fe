"""Offending rays of ptx_selfcheck_pruning under the margin rule at several margin_lat values: direction, max|1/d_k|, distances."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
import numpy as np
import ptx_import
m = ptx_import.load()
ctx = m.Context(0)
for name in sys.argv[1:] or ["teapot"]:
    s = m.load_scene(name)
    ctx.upload_scene(s)
    diag = float(np.linalg.norm(s.coords.reshape(-1, 3).max(0) - s.coords.reshape(-1, 3).min(0)))
    for lat in (0, 100, 300, 1000):
        ctx.set_option("traversal", 3); ctx.set_option("prune", 2); ctx.set_option("margin_lat", lat)
        for fam in range(8):
            r = ctx.selfcheck_pruning(fam, 12_500_000, seed=20260 + fam)
            if not r["mismatches"]:
                continue
            print(f"{name} diag {diag:.2f} margin_lat {lat} family {fam}: {r['mismatches']} mismatches", flush=True)
            for o, d, te, tp, ke, kp in r["first"][:16]:
                d = np.asarray(d, dtype=np.float64); o = np.asarray(o, dtype=np.float64)
                inv = 1.0 / np.maximum(np.abs(d), 1e-30)
                print(f"   max|1/d| {inv.max():10.1f}  |o| {np.linalg.norm(o):8.2f}  t_exact {te:.6g} t_pruned {tp:.6g} dt {tp - te:.3g}  needed ppm >= {1e6 * (tp - te) / (diag * inv.max()):.0f}?  d {d}", flush=True)
