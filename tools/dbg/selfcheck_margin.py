"""Margin rule (prune = 2) under ptx_selfcheck_pruning, all families: mismatches per scene with the library named by PTX_LIB_PATH."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
import ptx_import
m = ptx_import.load()
ctx = m.Context(0)
for name in ("pixar_room", "teapot", "bedroom2_v5", "my_cornell_6", "cornell"):
    ctx.upload_scene(m.load_scene(name))
    ctx.set_option("traversal", 3); ctx.set_option("prune", 2)
    res = [ctx.selfcheck_pruning(fam, 12_500_000, seed=20260 + fam)["mismatches"] for fam in range(8)]
    print(os.path.basename(os.environ.get("PTX_LIB_PATH", "default")), name, "mismatches per family", res, flush=True)
