"""Node-visit heat of the default renderer (needs a -DPTX_HEAT build: tools/dbg/build_variants.sh heat "-DPTX_HEAT").
How much of the walk's node traffic would the K hottest node records per octant layout serve?  (LDS-staged top of the tree.)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
import numpy as np
import ptx_import
m = ptx_import.load()
cfgname = sys.argv[1] if len(sys.argv) > 1 else "headline"
c = m.CONFIGS[cfgname]
s = m.load_scene(c["scene"]); p = m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"])
ctx = m.Context(0); ctx.upload_scene(s); ctx.set_params(p)
ctx.render_frames([1000]); ctx.synchronize()
L = ctx._lib
n = 1 << 20
heat = np.zeros(n, np.uint32); nodes = np.zeros((n, 8), np.float32); tot = C.c_uint32(0); per = C.c_uint32(0)
rc = L.ptx_dbg_heat(C.c_void_p(ctx._h), heat.ctypes.data_as(C.c_void_p), n, nodes.ctypes.data_as(C.c_void_p), C.byref(tot), C.byref(per))
assert rc == 0
tot, per = tot.value, per.value
heat = heat[:tot].astype(np.float64); nodes = nodes[:tot]
fc = nodes[:, 7].view(np.int32) if False else nodes.view(np.int32)[:, 7]
print(f"{cfgname}: {tot} node records, {per} per layout; visits {heat.sum():.4g}")
lay = heat.reshape(-1, per)
print("visits per layout:", [f"{x:.3g}" for x in lay.sum(1)])
inner = (fc < 0).reshape(-1, per)
print("share of visits on inner nodes:", float((lay * inner).sum() / lay.sum()))
for K in (16, 32, 64, 128, 256, 512, 1024, 2048):
    cov = sum(np.sort(l)[::-1][:K].sum() for l in lay[:8]) / lay.sum()
    covp = sum(l[:K].sum() for l in lay[:8]) / lay.sum()
    # a hot set shared by all octants: nodes are the same set in every octant layout but at different indices; rank by box identity is
    # not available here, so report per-layout top-K and pre-order prefix only
    print(f"K={K:5d} per octant: hottest-K cover {cov:.3f} of all visits ({K*32*8/1024:.0f} KB LDS), pre-order prefix {covp:.3f}")
np.save(os.path.join(ROOT, "gpurun_out", f"heat_{cfgname}.npy"), lay)
