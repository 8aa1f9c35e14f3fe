"""Per-phase cycles and lane utilisation of the wave-scheduled pipelines on the headline (count = 1 build, 2 frames)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
import ptx_import
m = ptx_import.load()
c = m.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "headline"]
s = m.load_scene(c["scene"]); p = m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"])
ctx = m.Context(0); ctx.upload_scene(s); ctx.set_params(p)
for kv in filter(None, os.environ.get("PTX_OPTS", "").split(",")):      # e.g. PTX_OPTS=prune=1,emit_list=0
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for pipeline in (2,):
    ctx.set_option("pipeline", pipeline)
    for count in (0, 1):
        ctx.set_option("count", count)
        ctx.reset_accum(); ctx.reset_counters(); ctx.render_frames([999]); ctx.synchronize(); ctx.reset_counters()
        t0 = time.time(); ctx.render_frames([1000, 1001, 1002, 1003]); ctx.synchronize(); dt = (time.time() - t0) / 4
        k = ctx.counters()
        print(f"pipeline {pipeline} count {count}: {dt*1000:.1f} ms/frame", flush=True)
        if count:
            print({key: v for key, v in k.items()}, flush=True)
