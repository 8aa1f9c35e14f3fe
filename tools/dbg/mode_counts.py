"""Executed node visits / triangle tests per traced ray (count = 1, one frame) under option sets: argv[1] config, argv[2:] option sets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
import ptx_import
m = ptx_import.load()
cfg = sys.argv[1]
c = m.CONFIGS[cfg]
s = m.load_scene(c["scene"]); p = m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"])
for opts in (sys.argv[2:] or [""]):
    ctx = m.Context(0); ctx.upload_scene(s); ctx.set_params(p); ctx.set_option("count", 1)
    for kv in filter(None, opts.split(",")):
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    ctx.render_frames([1000]); ctx.synchronize()
    k = ctx.counters()
    d = k["diag"]
    print(f"{cfg} [{opts}] nodes/ray {k['node_visits']/k['rays_traced']:.2f} tris/ray {k['tri_tests']/k['rays_traced']:.2f} walk util {d[0]/max(1,d[1]):.2f} tri util {d[2]/max(1,d[3]):.2f} shade util {d[4]/max(1,d[5]):.2f} cycles w/t/s {d[6]/1e9:.0f}/{d[7]/1e9:.0f}/{d[8]/1e9:.0f}", flush=True)
    ctx.close()
