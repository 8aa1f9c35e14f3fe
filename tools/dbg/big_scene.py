"""One-off scale check: a random 150k-triangle scene (clustered blobs + floor + one emitter) flattened by the literal bvh.js
restatement, rendered small by the HIP path and compared with the CPU oracle.  python tools/dbg/big_scene.py make|check"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
F = os.path.join(ROOT, "build", "big_scene.npz")

def make(T=150000):
    from oracle import bvh_oracle
    sys.setrecursionlimit(100000)
    rng = np.random.default_rng(5)
    centres = rng.random((400, 3)) * [8, 4, 8] - [4, 2, 4]
    c = centres[rng.integers(400, size=T)] + rng.standard_normal((T, 3)) * 0.15
    tri = c[:, None, :] + rng.standard_normal((T, 3, 3)) * 0.03
    tri[-2] = [[-6, 5, -6], [6, 5, -6], [6, 5, 6]]; tri[-1] = [[-6, 5, -6], [6, 5, 6], [-6, 5, 6]]      # emitter quad above
    tri[-4] = [[-6, -3, -6], [6, -3, 6], [6, -3, -6]]; tri[-3] = [[-6, -3, -6], [-6, -3, 6], [6, -3, 6]]  # floor
    tri = tri.astype(np.float32)
    t0 = time.time()
    b = bvh_oracle.build_bvh(tri.reshape(T, 9).astype(np.float64).tolist())
    print("python bvh build", round(time.time() - t0, 1), "s, nodes", b["node_count"], flush=True)
    e1 = tri[:, 1] - tri[:, 0]; e2 = tri[:, 2] - tri[:, 0]
    n = np.cross(e1, e2); n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-20)
    em = np.zeros((T, 3), np.float32); em[-2:] = 50.0
    np.savez_compressed(F, coords=tri.reshape(-1), normals=n.astype(np.float32).reshape(-1), colors=np.full(3 * T, 0.7, np.float32), emissions=em.reshape(-1),
                        light_tri=np.array([T - 2, T - 1], np.int32), node_min=np.asarray(b["node_min"], np.float32).reshape(-1),
                        node_max=np.asarray(b["node_max"], np.float32).reshape(-1), node_miss=np.asarray(b["node_miss"], np.int32),
                        node_first=np.asarray(b["node_first"], np.int32), node_count=np.asarray(b["node_cnt"], np.int32),
                        inorder_tri=np.asarray(b["inorder_tri"], np.int32))

def check():
    import torch, ptx_import
    m = ptx_import.load()
    from oracle import orc
    z = np.load(F)
    T = z["inorder_tri"].size; N = z["node_miss"].size
    s = m.scene.Scene(T=T, N=N, L=2, **{k: np.ascontiguousarray(z[k]) for k in z.files})
    s.validate()
    p = m.scene.Params(width=96, height=64, cam_src=[0, 0, 14], cam_up=[0, 1, 0], cam_right=[1, 0, 0], cam_left_bottom=[-0.5, -0.5, 13], max_path_length=5, sample_count=4)
    ctx = m.Context(0)
    t0 = time.time(); ctx.upload_scene(s); print("upload (leaf-SAH build incl.)", round(time.time() - t0, 2), "s", flush=True)
    ctx.set_params(p)
    ts = [1000, 1001]
    t0 = time.time(); want, oc = orc.render_frames(s, p, ts); print("oracle", round(time.time() - t0, 1), "s; rays", oc["rays"], flush=True)
    for pipe in (0, 2):
        for trav in (0, 3):
            ctx.set_option("pipeline", pipe); ctx.set_option("traversal", trav); ctx.reset_accum(); ctx.reset_counters()
            t0 = time.time(); ctx.render_frames(ts); ctx.synchronize(); dt = time.time() - t0
            got = ctx.read_sum()
            print("pipeline", pipe, "traversal", trav, "bit-exact:", bool((got.view(np.uint32) == want.view(np.uint32)).all()), "mean", float(got.mean()), "ms", round(dt * 1e3, 1), flush=True)

if __name__ == "__main__":
    make() if sys.argv[1] == "make" else check()
