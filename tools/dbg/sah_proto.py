"""Prototype: rebuild a SAH tree over the REFERENCE LEAVES (their exact f32 boxes) and measure the work of the
literal threaded traversal on it with the CPU oracle."""
import sys, time, copy
sys.path.insert(0, '/root/repo')
import numpy as np
import ptx_import; m = ptx_import.load()
from oracle import orc
sys.setrecursionlimit(100000)

def rebuild(s):
    leaves = np.nonzero(s.node_first != -2)[0]
    bmin = s.node_min.reshape(-1, 3)[leaves].astype(np.float32); bmax = s.node_max.reshape(-1, 3)[leaves].astype(np.float32)
    cnt = s.node_count[leaves].astype(np.float64); first = s.node_first[leaves]
    cen = (bmin.astype(np.float64) + bmax) * 0.5
    def area(lo, hi):
        d = (hi - lo).astype(np.float64); return 2 * (d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0])
    nodes = []  # (min, max, first, count, left, right)
    def build(ids):
        lo = bmin[ids].min(0); hi = bmax[ids].max(0)
        me = len(nodes); nodes.append(None)
        if len(ids) == 1:
            nodes[me] = (lo, hi, int(first[ids[0]]), int(cnt[ids[0]]), -1, -1); return me
        best = (np.inf, None, None)
        for ax in range(3):
            order = ids[np.argsort(cen[ids, ax], kind='stable')]
            lmin = np.minimum.accumulate(bmin[order], 0); lmax = np.maximum.accumulate(bmax[order], 0)
            rmin = np.minimum.accumulate(bmin[order][::-1], 0)[::-1]; rmax = np.maximum.accumulate(bmax[order][::-1], 0)[::-1]
            cl = np.cumsum(cnt[order]); cr = np.cumsum(cnt[order][::-1])[::-1]
            cost = area(lmin[:-1], lmax[:-1]) * cl[:-1] + area(rmin[1:], rmax[1:]) * cr[1:]
            j = int(np.argmin(cost))
            if cost[j] < best[0]: best = (cost[j], order, j + 1)
        _, order, k = best
        l = build(order[:k]); r = build(order[k:])
        nodes[me] = (lo, hi, -2, 0, l, r); return me
    build(np.arange(len(leaves)))
    N = len(nodes)
    # nodes were appended in pre-order already (me allocated before children, left built before right)
    end = [0] * N
    def sub_end(i):
        n = nodes[i]
        if n[4] < 0: end[i] = i + 1; return i + 1
        sub_end(n[4]); e = sub_end(n[5]); end[i] = e; return e
    sub_end(0)
    t = copy.copy(s)
    t.N = N
    t.node_min = np.array([n[0] for n in nodes], dtype=np.float32).ravel(); t.node_max = np.array([n[1] for n in nodes], dtype=np.float32).ravel()
    t.node_first = np.array([n[2] for n in nodes], dtype=np.int32); t.node_count = np.array([n[3] for n in nodes], dtype=np.int32)
    t.node_miss = np.array([e if e < N else -1 for e in end], dtype=np.int32)
    return t

for name in sys.argv[1:] or ['pixar_room']:
    s = m.load_scene(name); p = m.config_params(name, 320, 180, 16, 8)
    t0 = time.time(); t = rebuild(s); print(name, 'rebuild', round(time.time() - t0, 1), 's N', s.N, '->', t.N)
    a, ca = orc.render_frame(s, p, 1000); b, cb = orc.render_frame(t, p, 1000)
    print('  ref tree Nv/ray %.1f Nt/ray %.1f | leaf-SAH tree Nv/ray %.1f Nt/ray %.1f | differing pixels %d, rays equal %s' % (
        ca['node_visits'] / ca['rays'], ca['tri_tests'] / ca['rays'], cb['node_visits'] / cb['rays'], cb['tri_tests'] / cb['rays'],
        int((a.view(np.uint32) != b.view(np.uint32)).any(axis=2).sum()), ca['rays'] == cb['rays']))

if True:
    s = m.load_scene('pixar_room'); t = rebuild(s)
    rng = np.random.default_rng(5)
    lo, hi = s.node_min[:3], s.node_max[:3]
    n = 400000
    o = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    ta, ia, ca = orc.trace_closest(s, o, d); tb, ib, cb = orc.trace_closest(t, o, d)
    dt = ta.view(np.uint32) != tb.view(np.uint32); di = ia != ib
    print('rays', n, 't differs', int(dt.sum()), 'tri differs', int(di.sum()), 'tri differs with equal t', int((di & ~dt).sum()), 'Nt', ca['tri_tests'], cb['tri_tests'])
    bad = np.nonzero(di)[0][:5]
    for k in bad: print(k, ta[k], tb[k], ia[k], ib[k])
