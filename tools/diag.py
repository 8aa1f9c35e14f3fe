import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import ptx_import
m = ptx_import.load()
c = m.CONFIGS["headline"]
s = m.load_scene(c["scene"]); p = m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"])
ctx = m.Context(0); ctx.upload_scene(s); ctx.set_params(p); ctx.set_option("count", 1)
for kv in filter(None, os.environ.get("PERF_OPTS", "").split(",")):
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for trav in [int(x) for x in os.environ.get("PERF_TRAV", "1,3").split(",")]:
    ctx.set_option("traversal", trav); ctx.reset_accum(); ctx.reset_counters()
    ctx.render_frames([1000]); k = ctx.counters(); d = k["diag"]
    print(f"trav={trav} Nv/traced={k['node_visits']/k['rays_traced']:.1f} Nt/traced={k['tri_tests']/k['rays_traced']:.1f} walk_util={d[0]/max(1,d[1]):.3f} tri_util={d[2]/max(1,d[3]):.3f} trace_util={d[4]/max(1,d[5]):.3f}"
          f" walk_slots/ray={d[1]/k['rays_traced']:.0f} tri_slots/ray={d[3]/k['rays_traced']:.0f}", flush=True)
    if d[9]: print("   cycles/iteration walk %.0f (n=%.3g) tri %.0f (n=%.3g) shade %.0f (n=%.3g); share of cycles: %.2f %.2f %.2f" % (d[6]/d[9], d[9], d[7]/max(1,d[10]), d[10], d[8]/max(1,d[11]), d[11], d[6]/(d[6]+d[7]+d[8]), d[7]/(d[6]+d[7]+d[8]), d[8]/(d[6]+d[7]+d[8])), flush=True)
