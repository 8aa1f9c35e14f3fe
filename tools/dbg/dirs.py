import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, numpy as np, ptx_import
m = ptx_import.load()
from oracle import orc
ctx = m.Context(0)
s = m.load_scene("my_cornell_6"); p = m.config_params("my_cornell_6", 64, 64, 4, 4)
ctx.upload_scene(s); ctx.set_params(p)
dg = ctx.primary_dirs(); do = orc.primary_dirs(p)
diff = dg.view(np.int32).astype(np.int64) - do.view(np.int32).astype(np.int64)
print("ulp diff hist", np.unique(diff, return_counts=True))
idx = np.argwhere(diff != 0)[:5]
for y, x, c in idx: print(y, x, c, float(dg[y, x, c]).hex(), float(do[y, x, c]).hex())
print(p)
