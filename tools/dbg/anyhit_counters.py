"""Node visits / triangle tests / time per frame with the any-hit rule for light-visibility rays off and on."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
import ptx_import
m = ptx_import.load()
for cfg in (sys.argv[1:] or ["headline"]):
    c = m.CONFIGS[cfg]
    s = m.load_scene(c["scene"]); p = m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"])
    ctx = m.Context(0); ctx.upload_scene(s); ctx.set_params(p)
    for prune in (2, 1):
        ctx.set_option("prune", prune)
        for anyhit in (0, 1):
            ctx.set_option("anyhit", anyhit)
            for count in (0, 1):
                ctx.set_option("count", count)
                ctx.reset_accum(); ctx.reset_counters(); ctx.render_frames([999]); ctx.synchronize(); ctx.reset_counters()
                t0 = time.time(); ctx.render_frames(list(range(1000, 1008))); ctx.synchronize(); dt = (time.time() - t0) / 8
                k = ctx.counters()
                if count:
                    g = k["diag"]
                    print(f"   node visits/ray {k['node_visits']/k['rays_traced']:.2f} tri tests/ray {k['tri_tests']/k['rays_traced']:.2f} | util walk {g[0]/g[1]:.2f} tri {g[2]/g[3]:.2f} shade {g[4]/g[5]:.2f} | cyc {g[6]:.3g} {g[7]:.3g} {g[8]:.3g}", flush=True)
                else:
                    print(f"{cfg} prune {prune} anyhit {anyhit}: {dt*1000:.1f} ms/frame", flush=True)
    ctx.destroy() if hasattr(ctx, "destroy") else None
