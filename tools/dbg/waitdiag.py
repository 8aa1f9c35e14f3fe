import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ptx_import
m = ptx_import.load()
c = m.CONFIGS["headline"]
s = m.load_scene(c["scene"]); p = m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"])
ctx = m.Context(0); ctx.upload_scene(s); ctx.set_params(p); ctx.set_option("count", 1)
ctx.render_frames([1000]); k = ctx.counters(); d = k["diag"]
nph = d[9]
print("walk phases", nph, "per phase at entry: waiting-for-TRI lanes %.1f, waiting-for-SHADE lanes %.1f, dead lanes %.1f, walk util over steps %.3f" % (d[4]/nph, d[5]/nph, d[3]/nph, d[0]/max(1,d[1])))
