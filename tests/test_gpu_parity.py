"""Parity of the HIP path (through the C ABI, libptx_hip.so) against the CPU oracle -- needs an MI355X.

Bar: bit-exact.  Triangle ids are integers; the float results (t, directions, radiance sums) are required to be
bit-identical too because both sides implement the same strict-binary32 arithmetic contract.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCENES = ["cornell", "my_cornell_6", "bedroom2_v5", "teapot", "pixar_room"]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def setup(ctx, ptx, name, w, h, spp=4, bounces=4, **opts):
    s = ptx.load_scene(name)
    p = ptx.config_params(name, w, h, spp, bounces)
    ctx.upload_scene(s)
    ctx.set_params(p)
    for k, v in {**dict(traversal=3, prune=2, memo=1, count=0, frame_batch=0, pipeline=2, tasks_per_wave=128, anyhit=1), **opts}.items():
        ctx.set_option(k, v)
    ctx.reset_accum()
    ctx.reset_counters()
    return s, p


@pytest.mark.parametrize("name", SCENES)
@pytest.mark.parametrize("size", [(96, 54), (40, 64), (33, 33)])
@pytest.mark.parametrize("traversal", [0, 1, 2, 3])
def test_primary_dirs_and_hits_bit_exact(gpu_ctx, ptx, orc, name, size, traversal):
    s, p = setup(gpu_ctx, ptx, name, *size, traversal=traversal)
    assert (bits(gpu_ctx.primary_dirs()) == bits(orc.primary_dirs(p))).all()
    t_g, tri_g = gpu_ctx.primary_hits()
    t_o, tri_o, _ = orc.primary_hits(s, p)
    assert (tri_g == tri_o).all()
    assert (bits(t_g) == bits(t_o)).all()


@pytest.mark.parametrize("cfg", ["C1", "C2", "C3", "C4", "headline", "C5"])
def test_primary_hits_of_every_configuration_at_full_size(gpu_ctx, ptx, orc, cfg):
    """north_star's first bar where it counts: triangle id and t bits of EVERY pixel of every BASELINE configuration (up to
    3840 x 2160) against the oracle's literal walk, under every traversal option and both pruning rules (tools/full_size_primary_hits.py)."""
    c = ptx.CONFIGS[cfg]
    s, p = setup(gpu_ctx, ptx, c["scene"], c["width"], c["height"])
    t_o, tri_o, _ = orc.primary_hits(s, p)
    for traversal, prune in ((3, 2), (3, 1), (2, 2), (1, 2), (0, 2)):
        gpu_ctx.set_option("traversal", traversal)
        gpu_ctx.set_option("prune", prune)
        t_g, tri_g = gpu_ctx.primary_hits()
        assert (tri_g == tri_o).all() and (bits(t_g) == bits(t_o)).all(), (cfg, traversal, prune)
    gpu_ctx.set_option("traversal", 3); gpu_ctx.set_option("prune", 2)


@pytest.mark.parametrize("cfg", ["C2", "C3", "C4"])
def test_whole_frame_at_full_size_equals_the_oracle(gpu_ctx, ptx, orc, cfg):
    """Every pixel of one whole frame at the configuration's own size and the four reference counters, library defaults against the
    oracle (5 - 10 s of host cores each; the headline and C5 the same way by tools/full_frame_vs_oracle.py, 50 and 200 s:
    profiles/r03_full_frames_vs_oracle.json)."""
    c = ptx.CONFIGS[cfg]
    s, p = setup(gpu_ctx, ptx, c["scene"], c["width"], c["height"], spp=c["sample_count"], bounces=c["max_path_length"])
    gpu_ctx.render_frames([1003], first_frame_number=3)
    got, gc = gpu_ctx.read_sum(), gpu_ctx.counters()
    want, oc = orc.render_frame(s, p, 1003, nthreads=min(16, orc.max_threads()))
    assert (bits(got) == bits(want)).all()
    for key in ("samples", "rays", "ray_hits", "light_attempts"):
        assert gc[key] == oc[key], key


def random_rays(s, n, seed):
    rng = np.random.default_rng(seed)
    lo, hi = s.node_min[:3], s.node_max[:3]
    o = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    # axis-parallel rays (inf slabs), rays starting ON triangles (self-hit epsilon), rays from outside the scene
    d[: n // 10] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, n // 10)] * rng.choice([-1, 1], (n // 10, 1)).astype(np.float32)
    tri = rng.integers(0, s.T, n // 10)
    c = s.coords.reshape(-1, 3, 3)[tri]
    w = rng.dirichlet([1, 1, 1], n // 10).astype(np.float32)
    o[n // 10: n // 10 + n // 10] = (c * w[:, :, None]).sum(axis=1)
    o[-n // 10:] = (lo - (hi - lo) * rng.random((n // 10, 3))).astype(np.float32)
    return o, d


@pytest.mark.parametrize("name", SCENES)
@pytest.mark.parametrize("traversal", [0, 1, 2, 3])
@pytest.mark.parametrize("pipeline", [0, 2])          # 0: 48-B records + scalar test; 2: pair records, packed math, ordered slabs
def test_trace_closest_bit_exact(gpu_ctx, ptx, orc, name, traversal, pipeline):
    s, _ = setup(gpu_ctx, ptx, name, 8, 8, traversal=traversal, pipeline=pipeline)
    o, d = random_rays(s, 20000, 7)
    t_g, tri_g = gpu_ctx.trace_closest(o, d)
    t_o, tri_o, c = orc.trace_closest(s, o, d)
    assert (tri_g == tri_o).all() and (bits(t_g) == bits(t_o)).all()
    assert (tri_o >= 0).sum() > 1000                                    # the test really hits geometry
    k = gpu_ctx.counters()
    assert k["rays"] == 20000 and k["ray_hits"] == c["ray_hits"]
    if traversal == 0:                                                  # literal traversal does the reference's work exactly
        assert k["node_visits"] == c["node_visits"] and k["tri_tests"] == c["tri_tests"]
    elif traversal == 2:                                                # leaf-SAH tree: exactly the same leaves are hit
        assert k["tri_tests"] == c["tri_tests"]
    else:
        assert k["tri_tests"] <= c["tri_tests"]


def epsilon_scene(ptx):
    """A hand-made scene around the shader's epsilons: a quad of two triangles sharing an edge, the same quad
    again (coincident duplicates: ties on t go to the earlier inorder slot), a triangle whose doubled area is ~1e-4 (the
    absolute determinant cut-off of glsl:156 makes it visible only head-on) and one far below it (never visible).
    Two leaves under one root; slots in an order that differs from the triangle numbering."""
    tri = np.array([[[0, 0, 0], [1, 0, 0], [1, 1, 0]], [[0, 0, 0], [1, 1, 0], [0, 1, 0]],             # quad z = 0
                    [[0, 0, 0], [1, 0, 0], [1, 1, 0]], [[0, 0, 0], [1, 1, 0], [0, 1, 0]],             # the same quad again
                    [[2, 0, 0], [2.0101, 0, 0], [2, 0.0100, 0]],                                      # 2A = 1.01e-4
                    [[2, 1, 0], [2.005, 1, 0], [2, 1.005, 0]],                                        # 2A = 2.5e-5: invisible
                    [[-1, -1, -2], [3, -1, -2], [1, 3, -2]]], dtype=np.float32)                       # big backdrop z = -2
    T = len(tri)
    order = np.array([3, 1, 4, 0, 2, 6, 5], dtype=np.int32)                                            # inorder slots -> triangle ids
    lo = [tri[order[:4]].reshape(-1, 3).min(0), tri[order[4:]].reshape(-1, 3).min(0)]
    hi = [tri[order[:4]].reshape(-1, 3).max(0), tri[order[4:]].reshape(-1, 3).max(0)]
    nmin = np.array([np.minimum(lo[0], lo[1]), lo[0], lo[1]], dtype=np.float32)
    nmax = np.array([np.maximum(hi[0], hi[1]), hi[0], hi[1]], dtype=np.float32)
    z = np.zeros(3 * T, dtype=np.float32)
    return ptx.scene.Scene(T=T, N=3, L=0, coords=np.ascontiguousarray(tri.reshape(-1)), normals=np.tile(np.float32([0, 0, 1]), T), colors=z + 1, emissions=z,
                           light_tri=np.zeros(0, dtype=np.int32), node_min=nmin.reshape(-1), node_max=nmax.reshape(-1),
                           node_miss=np.array([-1, 2, -1], dtype=np.int32), node_first=np.array([-2, 0, 4], dtype=np.int32),
                           node_count=np.array([0, 4, 3], dtype=np.int32), inorder_tri=order)


@pytest.mark.parametrize("pipeline", [0, 2])
def test_epsilon_rays(gpu_ctx, ptx, orc, pipeline):
    s = epsilon_scene(ptx)
    s.validate()
    rng = np.random.default_rng(11)
    o, d = [], []
    def ray(origin, target):
        v = np.asarray(target, dtype=np.float64) - np.asarray(origin, dtype=np.float64)
        o.append(origin); d.append(v / np.linalg.norm(v))
    for _ in range(3000):                                       # the shared edge / the diagonal / the outer edges and corners, exactly
        a = rng.random()
        ray([0.3, 0.2, 1.5], [a, a, 0]); ray([a, a, 0.75], [a, a, 0]); ray([a, 0.1, 2.0], [a, 0, 0]); ray([0.5, 0.5, 1.0], [round(a), round(rng.random()), 0])
    for _ in range(3000):                                       # t around 1e-4: origins a hair above the quad
        x, y = rng.random(2)
        h = np.float32(1e-4) * (0.5 + rng.random())
        ray([x, y, h], [x + 1e-6 * rng.standard_normal(), y, -1])
    for _ in range(6000):                                       # |a| around 1e-4: the small triangles from every tilt
        c = rng.random(2) * 0.004 + [2.001, 0.001] if rng.random() < 0.7 else rng.random(2) * 0.002 + [2.0005, 1.0005]
        th, ph = np.arccos(rng.random()), 2 * np.pi * rng.random()
        dirv = np.array([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)])
        ray(np.array([c[0], c[1], 0]) + dirv * (0.5 + rng.random()), [c[0], c[1], 0])
    for ax in range(3):                                         # axis-parallel rays: 1/d = inf, NaN slabs on bounding planes
        for _ in range(300):
            p0 = rng.random(3) * [3, 2, 3] - [0.5, 0.5, 1.0]
            p0[rng.integers(3)] = rng.choice([0.0, 1.0, 2.0, -2.0, -1.0, 3.0])
            dv = np.zeros(3); dv[ax] = rng.choice([-1.0, 1.0])
            o.append(p0); d.append(dv)
    o = np.asarray(o, dtype=np.float32); d = np.asarray(d, dtype=np.float32)
    t_o, tri_o, _ = orc.trace_closest(s, o, d)
    small = (tri_o == 4).sum()
    # duplicates: triangle 3 (slot 0) always beats its twin 1 (slot 1); the ~1e-4 triangle shows only head-on; the 2.5e-5 one never
    assert 0 < small < 4200 and (tri_o == 5).sum() == 0 and (tri_o == 1).sum() == 0 and (tri_o == 0).sum() > 500 and (tri_o == 3).sum() > 500
    gpu_ctx.upload_scene(s)
    for traversal in (0, 1, 2, 3):
        gpu_ctx.set_option("traversal", traversal); gpu_ctx.set_option("pipeline", pipeline)
        t_g, tri_g = gpu_ctx.trace_closest(o, d)
        assert (tri_g == tri_o).all() and (bits(t_g) == bits(t_o)).all(), traversal


def test_trace_closest_empty_and_single(gpu_ctx, ptx, orc):
    s, _ = setup(gpu_ctx, ptx, "my_cornell_6", 8, 8)
    t, tri = gpu_ctx.trace_closest(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert t.size == 0 and tri.size == 0
    t, tri = gpu_ctx.trace_closest([[100, 100, 100]], [[1, 0, 0]])
    assert tri[0] == -1 and t[0] == np.float32(1e38)


@pytest.mark.parametrize("name,size,spp,bounces", [("cornell", (48, 48), 4, 4), ("my_cornell_6", (64, 64), 4, 5), ("bedroom2_v5", (64, 36), 3, 4),
                                                   ("teapot", (64, 36), 3, 4), ("pixar_room", (96, 54), 4, 8), ("pixar_room", (27, 45), 2, 12)])
@pytest.mark.parametrize("traversal,memo,pipeline", [(0, 0, 0), (0, 1, 0), (1, 0, 0), (1, 1, 0), (2, 1, 0), (3, 1, 0),
                                                     (0, 0, 1), (0, 1, 1), (1, 0, 1), (1, 1, 1), (2, 0, 1), (3, 1, 1),
                                                     (0, 0, 2), (0, 1, 2), (1, 0, 2), (1, 1, 2), (2, 0, 2), (3, 1, 2)])
def test_frames_bit_exact_and_counters(gpu_ctx, ptx, orc, name, size, spp, bounces, traversal, memo, pipeline):
    s, p = setup(gpu_ctx, ptx, name, *size, spp=spp, bounces=bounces, traversal=traversal, memo=memo, count=1, pipeline=pipeline)
    ts = ptx.scene.timestamps(3)
    gpu_ctx.render_frames(ts)
    got = gpu_ctx.read_sum()
    want, oc = orc.render_frames(s, p, ts)
    assert (bits(got) == bits(want)).all()
    assert want.max() > 0                                               # not a black image
    k = gpu_ctx.counters()
    for key in ("samples", "rays", "ray_hits", "light_attempts"):
        assert k[key] == oc[key], key
    assert k["pixel_frames"] == 3 * p.width * p.height
    if traversal == 0 and memo == 0:
        assert k["node_visits"] == oc["node_visits"] and k["tri_tests"] == oc["tri_tests"] and k["rays_traced"] == oc["rays"]
    if memo == 1:
        assert k["rays_traced"] < k["rays"]


def test_golden_frame_fixture(gpu_ctx, ptx):
    """Committed golden vectors (tests/golden/frames.npz, written by tests/golden/make_golden_frames.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "frames.npz"))
    for key in [k for k in g.files if k.endswith("_sum")]:
        name, w, h, spp, bounces, nfr = key[:-4].rsplit("-", 5)
        s, p = setup(gpu_ctx, ptx, name, int(w), int(h), spp=int(spp), bounces=int(bounces))
        gpu_ctx.render_frames(ptx.scene.timestamps(int(nfr)))
        assert (bits(gpu_ctx.read_sum()) == bits(g[key])).all(), key
        _, tri = gpu_ctx.primary_hits()
        assert (tri == g[key[:-4] + "_tri"]).all(), key


def test_edge_cases(gpu_ctx, ptx, orc):
    # L = 0: black image, no light attempts (cornell through the faithful loader)
    s, p = setup(gpu_ctx, ptx, "cornell_faithful", 24, 24)
    gpu_ctx.render_frames([1000, 1001])
    assert not gpu_ctx.read_sum().any() and gpu_ctx.counters()["light_attempts"] == 0
    # 1x1 image, 1 sample, 1 bounce; odd timestamps (negative, > 16 bits: only the low 16 bits seed, glsl:520)
    for (w, h, spp, b, ts) in [(1, 1, 1, 1, [0]), (7, 5, 1, 1, [-1]), (9, 13, 2, 2, [0x12345, 70000]), (16, 16, 5, 1, [65535])]:
        s, p = setup(gpu_ctx, ptx, "my_cornell_6", w, h, spp=spp, bounces=b)
        gpu_ctx.render_frames(ts)
        want, _ = orc.render_frames(s, p, ts)
        assert (bits(gpu_ctx.read_sum()) == bits(want)).all(), (w, h, spp, b, ts)
    # max_path_length = 0: the bounce loop never runs (glsl:408) -> black, but samples are counted
    s, p = setup(gpu_ctx, ptx, "my_cornell_6", 8, 8, spp=3, bounces=0)
    gpu_ctx.render_frames([1000])
    assert not gpu_ctx.read_sum().any() and gpu_ctx.counters()["samples"] == 8 * 8 * 3
    # n_frames = 0 is a no-op
    gpu_ctx.render_frames([])
    assert gpu_ctx.frames_done == 1
    # two leaves that share their slots (legal arrays no builder produces): every pipeline setting still matches the oracle
    import copy
    s = copy.deepcopy(ptx.load_scene("my_cornell_6"))
    leaves = np.nonzero(s.node_first != -2)[0]
    s.node_first[leaves[3]] = s.node_first[leaves[2]]; s.node_count[leaves[3]] = s.node_count[leaves[2]]
    p = ptx.config_params("my_cornell_6", 24, 24, 3, 4)
    gpu_ctx.upload_scene(s); gpu_ctx.set_params(p)
    want, _ = orc.render_frames(s, p, [1000, 1001])
    for pipeline in (0, 1, 2):
        for traversal in (0, 3):
            gpu_ctx.set_option("pipeline", pipeline); gpu_ctx.set_option("traversal", traversal)
            gpu_ctx.reset_accum(); gpu_ctx.render_frames([1000, 1001])
            assert (bits(gpu_ctx.read_sum()) == bits(want)).all(), (pipeline, traversal)


def _with_emitters(ptx, name, emit):
    """The scene with the emission of some triangles replaced: emit = {triangle: (r, g, b)}; the light list is rebuilt as
    utils/triangleMapper.js:71 does (any channel > 0)."""
    import copy
    s = copy.deepcopy(ptx.load_scene(name))
    em = s.emissions.reshape(-1, 3).copy()
    for tri, e in emit.items():
        em[tri] = e
    s.emissions = np.ascontiguousarray(em.reshape(-1), dtype=np.float32)
    s.light_tri = np.ascontiguousarray(np.nonzero((em > 0).any(axis=1))[0], dtype=np.int32)
    s.L = int(s.light_tri.size)
    s.validate()
    return s


@pytest.mark.gpu
@pytest.mark.parametrize("prune", [1, 2])
def test_anyhit_light_visibility(gpu_ctx, ptx, orc, prune):
    """Light-visibility rays stop at the first conclusive blocker (option anyhit, default on): same sums and reference
    counters as with the option off and as the oracle -- on the BASELINE scenes, with emitters in several far-apart
    clusters, with more emitters than boxes, and with a non-light triangle of non-zero (negative) emission, which blocks
    like a light of another emission (glsl:353-363) and must never count as a conclusive blocker."""
    far = {0: (50, 50, 50), 1: (50, 50, 50), 40: (20, 30, 40), 41: (20, 30, 40), 100: (5, 0, 0), 77: (0, 9, 0)}
    many = {t: (50, 50, 50) for t in range(0, 122, 5)}
    most = {t: (7, 7, 7) for t in range(122) if t % 6}                  # > 64 emitters: the chunked pre-merge of the boxes
    cases = [("my_cornell_6", None, (48, 48), 4, 5), ("pixar_room", None, (64, 36), 3, 6), ("bedroom2_v5", None, (48, 27), 3, 4),
             ("my_cornell_6", far, (40, 40), 3, 5), ("my_cornell_6", many, (40, 40), 3, 4), ("my_cornell_6", most, (32, 32), 2, 4),
             ("my_cornell_6", {60: (-5, 0, 0), 61: (0, 0, -1)}, (40, 40), 4, 5)]
    for name, emit, size, spp, bounces in cases + [("my_cornell_6", "dark light", (40, 40), 3, 5)]:
        s = ptx.load_scene(name) if emit is None else _with_emitters(ptx, name, emit if isinstance(emit, dict) else {})
        if emit == "dark light":                   # the light list names a triangle of emission (0,0,0): a blocker then compares EQUAL to the light (glsl:359) and the
            # attempt counts as visible -- shade_step still makes that comparison, the rule only supplies "emission (0,0,0)"
            s.light_tri = np.ascontiguousarray(np.append(s.light_tri, 70), dtype=np.int32); s.L = int(s.light_tri.size); s.validate()
        p = ptx.config_params(name, *size, spp, bounces)
        gpu_ctx.upload_scene(s); gpu_ctx.set_params(p)
        ts = ptx.scene.timestamps(2)
        want, oc = orc.render_frames(s, p, ts)
        assert want.max() > 0
        visits = {}
        for anyhit in (0, 1):
            for k, v in dict(traversal=3, prune=prune, memo=1, count=1, pipeline=2, anyhit=anyhit, frame_batch=0).items():
                gpu_ctx.set_option(k, v)
            gpu_ctx.reset_accum(); gpu_ctx.reset_counters()
            gpu_ctx.render_frames(ts)
            assert (bits(gpu_ctx.read_sum()) == bits(want)).all(), (name, emit is not None, anyhit)
            c = gpu_ctx.counters()
            for key in ("samples", "rays", "ray_hits", "light_attempts"):
                assert c[key] == oc[key], (name, key, anyhit)
            visits[anyhit] = c["node_visits"]
        assert visits[1] <= visits[0]
        if prune == 2 and emit is None:
            assert visits[1] < 0.95 * visits[0], (name, visits)        # it does end rays early
    gpu_ctx.set_option("count", 0); gpu_ctx.set_option("prune", 2)


def test_boxes_that_do_not_bound(gpu_ctx, ptx, orc):
    """Legal arrays no builder produces: leaf boxes that do not contain their triangles and an inner box smaller than its
    children.  The reference walks them literally (glsl:207-249: a leaf is tested iff every box on the way to it is hit), so
    pruning, the any-hit rule and the blocker cache -- which all assume boxes that bound what is beneath them -- must switch
    themselves off (ptx_upload_scene checks both properties): every option set still equals the oracle."""
    import copy
    for variant in ("leaf", "inner", "both"):
        s = copy.deepcopy(ptx.load_scene("my_cornell_6"))
        mn, mx = s.node_min.reshape(-1, 3).copy(), s.node_max.reshape(-1, 3).copy()
        leaves = np.nonzero(s.node_first != -2)[0]
        inner = np.nonzero(s.node_first == -2)[0]
        if variant in ("leaf", "both"):
            for i in leaves[::3]:
                mx[i] = mn[i] + 0.35 * (mx[i] - mn[i])                      # the triangles stick out of their leaf's box
        if variant in ("inner", "both"):
            for i in inner[1::2]:
                c = 0.5 * (mn[i] + mx[i]); mn[i] = c + 0.6 * (mn[i] - c); mx[i] = c + 0.6 * (mx[i] - c)   # smaller than its children
        s.node_min = np.ascontiguousarray(mn.reshape(-1), dtype=np.float32); s.node_max = np.ascontiguousarray(mx.reshape(-1), dtype=np.float32)
        s.validate()
        p = ptx.config_params("my_cornell_6", 40, 40, 3, 5)
        gpu_ctx.upload_scene(s); gpu_ctx.set_params(p)
        ts = ptx.scene.timestamps(2)
        want, oc = orc.render_frames(s, p, ts)
        assert want.max() > 0
        for pipeline in (0, 1, 2):
            for traversal in (0, 1, 3):
                for prune in (1, 2):
                    for k, v in dict(pipeline=pipeline, traversal=traversal, prune=prune, anyhit=1, memo=1).items():
                        gpu_ctx.set_option(k, v)
                    gpu_ctx.reset_accum(); gpu_ctx.reset_counters(); gpu_ctx.render_frames(ts)
                    assert (bits(gpu_ctx.read_sum()) == bits(want)).all(), (variant, pipeline, traversal, prune)
                    c = gpu_ctx.counters()
                    assert c["rays"] == oc["rays"] and c["ray_hits"] == oc["ray_hits"]
    gpu_ctx.set_option("prune", 2); gpu_ctx.set_option("pipeline", 2); gpu_ctx.set_option("traversal", 3)


def test_task_list_shapes(gpu_ctx, ptx, orc):
    """Pipeline 2's task list (tile-major, frame-minor, chunks drawn from a counter): ragged image sizes, frame counts that do
    not divide anything, chunk sizes that are not multiples of a tile -- the sum never changes."""
    for (w, h, nfr) in [(37, 19, 5), (8, 8, 7), (70, 9, 3)]:
        s, p = setup(gpu_ctx, ptx, "my_cornell_6", w, h, spp=2, bounces=3)
        ts = ptx.scene.timestamps(nfr)
        want, _ = orc.render_frames(s, p, ts)
        for chunk in (64, 1, 37, 640, 65536):
            gpu_ctx.set_option("pool_chunk", chunk)
            gpu_ctx.reset_accum()
            gpu_ctx.render_frames(ts)
            assert (bits(gpu_ctx.read_sum()) == bits(want)).all(), (w, h, nfr, chunk)
    gpu_ctx.set_option("pool_chunk", 64)


def collapse_tree(ptx, s, max_depth):
    """The same scene with every subtree below `max_depth` folded into one big leaf (leaf ranges are contiguous because the
    inorder list is the pre-order concatenation of the leaves): exercises leaf sizes the reference builder never makes."""
    n = s.N
    depth = np.zeros(n, dtype=np.int64)
    end = np.where(s.node_miss < 0, n, s.node_miss)                 # pre-order index after the node's subtree
    for i in range(n):
        if s.node_first[i] == -2:                                   # children: i + 1 and miss[i + 1]
            depth[i + 1] = depth[i] + 1
            depth[s.node_miss[i + 1]] = depth[i] + 1
    keep = [i for i in range(n) if depth[i] <= max_depth]
    new_index = {old: k for k, old in enumerate(keep)}
    first, count, miss = [], [], []
    for i in keep:
        leaves = [j for j in range(i, end[i]) if s.node_first[j] != -2]
        if s.node_first[i] == -2 and depth[i] < max_depth:
            first.append(-2); count.append(0)
        else:
            first.append(int(s.node_first[leaves[0]])); count.append(int(sum(s.node_count[j] for j in leaves)))
        miss.append(-1 if s.node_miss[i] < 0 else new_index[int(s.node_miss[i])])
    k = np.asarray(keep)
    return ptx.scene.Scene(T=s.T, N=len(keep), L=s.L, coords=s.coords, normals=s.normals, colors=s.colors, emissions=s.emissions,
                           light_tri=s.light_tri, node_min=np.ascontiguousarray(s.node_min.reshape(-1, 3)[k].reshape(-1)),
                           node_max=np.ascontiguousarray(s.node_max.reshape(-1, 3)[k].reshape(-1)), node_miss=np.asarray(miss, dtype=np.int32),
                           node_first=np.asarray(first, dtype=np.int32), node_count=np.asarray(count, dtype=np.int32), inorder_tri=s.inorder_tri)


@pytest.mark.parametrize("name,max_depth", [("my_cornell_6", 0), ("my_cornell_6", 2), ("my_cornell_6", 3), ("bedroom2_v5", 5), ("cornell", 1)])
@pytest.mark.parametrize("pipeline", [0, 1, 2])
def test_big_and_uneven_leaves(gpu_ctx, ptx, orc, name, max_depth, pipeline):
    """Leaves of 1..127 triangles (odd and even counts, a single-leaf tree): the pair records and the multi-round TRI loop."""
    s = collapse_tree(ptx, ptx.load_scene(name), max_depth)
    s.validate()
    assert s.node_count.max() > 5 and s.node_count.max() <= 127
    p = ptx.config_params(name, 40, 28, 3, 4)
    gpu_ctx.upload_scene(s); gpu_ctx.set_params(p)
    ts = ptx.scene.timestamps(2)
    want, oc = orc.render_frames(s, p, ts)
    assert want.max() > 0
    for traversal in (0, 1, 2, 3):
        for k, v in dict(traversal=traversal, memo=1, count=1, pipeline=pipeline, frame_batch=0).items():
            gpu_ctx.set_option(k, v)
        gpu_ctx.reset_accum(); gpu_ctx.reset_counters()
        gpu_ctx.render_frames(ts)
        assert (bits(gpu_ctx.read_sum()) == bits(want)).all(), traversal
        c = gpu_ctx.counters()
        assert c["rays"] == oc["rays"] and c["ray_hits"] == oc["ray_hits"]
    gpu_ctx.set_option("count", 0)


@pytest.mark.parametrize("pipeline", [0, 1, 2])
def test_axis_aligned_rays(gpu_ctx, ptx, orc, pipeline):
    """Rays with exactly zero direction components (1/d = inf, 0 * inf = NaN slabs): the middle row and column of an
    axis-aligned camera with an odd image size, traced in the kernel itself (memo = 0).  Pipeline 2 must send them through
    the selecting slab test, not the entry-first shortcut."""
    s = ptx.load_scene("my_cornell_6")
    p = ptx.scene.Params(width=5, height=5, cam_src=[12.4, 0, 0], cam_up=[0, 1, 0], cam_right=[0, 0, -1], cam_left_bottom=[11.4, -0.5, 0.5],
                         max_path_length=4, sample_count=6)
    d = orc.primary_dirs(p)
    assert (d[2, :, 1] == 0).all() and (d[:, 2, 2] == 0).all()
    gpu_ctx.upload_scene(s); gpu_ctx.set_params(p)
    ts = ptx.scene.timestamps(3)
    want, _ = orc.render_frames(s, p, ts)
    assert want.max() > 0
    for traversal in (0, 1, 2, 3):
        for memo in (0, 1):
            for k, v in dict(traversal=traversal, memo=memo, count=0, pipeline=pipeline, frame_batch=0).items():
                gpu_ctx.set_option(k, v)
            gpu_ctx.reset_accum()
            gpu_ctx.render_frames(ts)
            assert (bits(gpu_ctx.read_sum()) == bits(want)).all(), (traversal, memo)


def test_rays_inside_bounding_planes(gpu_ctx, ptx, orc):
    """Rays that start ON a bounding plane of some node and run inside it (e.g. a ceiling point towards the ceiling light of
    my_cornell_6: y = 4, d.y = 0): 0 * inf = NaN slabs, decided by the second-operand rule of min/max and therefore by the
    reference's own hierarchy -- every traversal mode must send them through the reference tree (slab_node_ref) and agree
    with the oracle."""
    for name in ("my_cornell_6", "cornell", "teapot"):
        s = ptx.load_scene(name)
        gpu_ctx.upload_scene(s)
        rng = np.random.default_rng(11)
        n = 6000
        nmin, nmax = s.node_min.reshape(-1, 3), s.node_max.reshape(-1, 3)
        o = (nmin[0] + (nmax[0] - nmin[0]) * rng.random((n, 3))).astype(np.float32)
        d = rng.normal(size=(n, 3))
        axis = rng.integers(0, 3, n)
        node = rng.integers(0, s.N, n)
        plane = np.where(rng.random(n) < 0.5, nmin[node, axis], nmax[node, axis])
        o[np.arange(n), axis] = plane
        d[np.arange(n), axis] = 0.0
        d[np.arange(n // 6), (axis[: n // 6] + 1) % 3] = 0.0              # two zero components
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        if name == "my_cornell_6":
            o[-1000:, 1] = 4.0; d[-1000:, 1] = 0.0                        # the ceiling plane itself
            d[-1000:] = (d[-1000:] / np.linalg.norm(d[-1000:], axis=1, keepdims=True)).astype(np.float32)
        wt, wtri, _ = orc.trace_closest(s, o, d)
        assert (wtri >= 0).mean() > 0.05
        for pipeline in (0, 2):
            for traversal in (0, 1, 2, 3):
                gpu_ctx.set_option("pipeline", pipeline); gpu_ctx.set_option("traversal", traversal)
                t, tri = gpu_ctx.trace_closest(o, d)
                assert (tri == wtri).all() and (bits(t) == bits(wt)).all(), (name, pipeline, traversal, int((tri != wtri).sum()))
    gpu_ctx.set_option("pipeline", 2); gpu_ctx.set_option("traversal", 3)


def test_accumulation_semantics(gpu_ctx, ptx, orc):
    s, p = setup(gpu_ctx, ptx, "my_cornell_6", 40, 24, spp=2, bounces=3)
    ts = ptx.scene.timestamps(7)
    want, _ = orc.render_frames(s, p, ts)
    for batch in (1, 3, 0):                                             # frame-batch size never changes the sum
        gpu_ctx.set_option("frame_batch", batch)
        gpu_ctx.reset_accum()
        gpu_ctx.render_frames(ts)
        assert (bits(gpu_ctx.read_sum()) == bits(want)).all(), batch
    gpu_ctx.reset_accum()                                               # two calls == one call
    gpu_ctx.render_frames(ts[:4]); gpu_ctx.render_frames(ts[4:], first_frame_number=4)
    assert gpu_ctx.frames_done == 7
    assert (bits(gpu_ctx.read_sum()) == bits(want)).all()
    lin = gpu_ctx.read_linear()
    assert (bits(lin) == bits((want / np.float32(7)).astype(np.float32))).all()
    # display pass: pow(c, 1/2.2), unorm8, flip (fragmentShaderOutput.glsl:35, main.js:464-473); pow is the contract's
    # operation sequence (oracle orc_pow, pinned by ref_prim/probe_pow.npz) -> equal to the oracle's, and within one code
    # value of the exactly rounded power
    px = gpu_ctx.read_rgba8(apply_gamma=True, flip_y=True)
    y = float(np.float32(1.0) / np.float32(2.2))
    g = np.array([orc.pow_(float(v), y) for v in np.clip(lin, 0, 1).reshape(-1)], dtype=np.float32).reshape(lin.shape)
    assert (px[..., :3] == np.floor(np.clip(g, 0, 1) * np.float32(255) + np.float32(0.5)).astype(np.uint8)[::-1]).all()
    ref = np.floor(np.clip(lin.astype(np.float64), 0, 1) ** (1 / 2.2) * 255 + 0.5)[::-1]
    assert np.abs(px[..., :3].astype(np.int32) - ref.astype(np.int32)).max() <= 1 and (px[..., 3] == 255).all()
    raw = gpu_ctx.read_rgba8(apply_gamma=False, flip_y=False)
    assert (raw[..., :3] == np.floor(np.clip(lin, 0, 1) * np.float32(255) + np.float32(0.5)).astype(np.uint8)).all()
    # reference-compat check: with ONE frame the RGBA8 image equals the reference's render target (blend factor 1)
    gpu_ctx.reset_accum(); gpu_ctx.render_frames(ts[:1])
    one, _ = orc.render_frame(s, p, int(ts[0]))
    assert (gpu_ctx.read_rgba8(False, False) == orc.blend_rgba8(p, one, None, 0)).all()


def test_checkpoint_restore(gpu_ctx, ptx):
    """ptx_write_sum_f32: a render resumed from {sum, frames_done} equals the uninterrupted one bit for bit."""
    s, p = setup(gpu_ctx, ptx, "my_cornell_6", 40, 24, spp=3, bounces=4)
    ts = ptx.scene.timestamps(5)
    gpu_ctx.render_frames(ts)
    want = gpu_ctx.read_sum()
    gpu_ctx.reset_accum(); gpu_ctx.render_frames(ts[:2])
    ck, n = gpu_ctx.read_sum(), gpu_ctx.frames_done
    gpu_ctx.reset_accum()
    assert gpu_ctx.frames_done == 0
    gpu_ctx.write_sum(ck, n)
    assert gpu_ctx.frames_done == 2
    gpu_ctx.render_frames(ts[2:], first_frame_number=2)
    assert gpu_ctx.frames_done == 5 and (bits(gpu_ctx.read_sum()) == bits(want)).all()
    with pytest.raises(ValueError):
        gpu_ctx.write_sum(ck[:3], 1)


def test_device_export_import_for_allreduce(gpu_ctx, ptx):
    import torch
    s, p = setup(gpu_ctx, ptx, "my_cornell_6", 32, 16, spp=2, bounces=2)
    gpu_ctx.render_frames([1000, 1001])
    base = gpu_ctx.read_sum()
    t = torch.zeros(16, 32, 4, dtype=torch.float32, device="cuda:0")
    gpu_ctx.accum_export_device(t.data_ptr())
    torch.cuda.synchronize()
    assert (bits(t[..., :3].cpu().numpy()) == bits(base)).all()
    t *= 2
    torch.cuda.synchronize()
    gpu_ctx.accum_import_device(t.data_ptr(), 4)
    assert gpu_ctx.frames_done == 4 and (bits(gpu_ctx.read_sum()) == bits(base * 2)).all()


def test_error_behaviour(ptx):
    b = ptx.binding
    lib = ptx.load_library()
    assert lib.ptx_create((C.c_int * 3)(0, 0, 1), 3) is None and b"distinct" in lib.ptx_last_error(None)   # neither all distinct nor all equal
    assert lib.ptx_create((C.c_int * 1)(0), 0) is None and lib.ptx_create(None, 1) is None
    with pytest.raises(ptx.PtxError):
        ptx.Context(0).read_sum()                                     # read before set_params: a clear error
    with pytest.raises(ptx.PtxError):
        ptx.Context(99)
    ctx = ptx.Context(0)
    with pytest.raises(ptx.PtxError) as e:
        ctx.render_frames([1000])
    assert e.value.status == b.PTX_ERR_STATE
    s = ptx.load_scene("my_cornell_6")
    import copy
    leaf = int(np.nonzero(s.node_first != -2)[0][0])
    for field, idx, val in [("node_miss", 5, 5), ("node_miss", 5, 2), ("node_miss", 0, s.N), ("node_first", leaf, s.T), ("node_count", leaf, 200),
                            ("inorder_tri", 0, s.T), ("inorder_tri", 1, -1), ("light_tri", 0, s.T)]:
        bad = copy.deepcopy(s)
        getattr(bad, field)[idx] = val
        with pytest.raises(ptx.PtxError) as e:
            ctx.upload_scene(bad)
        assert e.value.status == b.PTX_ERR_INVALID, (field, idx, val)
    ctx.upload_scene(s)
    with pytest.raises(ptx.PtxError):
        ctx.set_option("nonsense", 1)
    p = ptx.config_params("my_cornell_6", 8, 8)
    p.sample_count = 0
    with pytest.raises(ptx.PtxError):
        ctx.set_params(p)
    ctx.close()


# ---------------------------------------------------------------- BASELINE.json's full sizes: properties + oracle row samples

@pytest.mark.parametrize("cfg", ["C2", "C3", "C4", "headline", "C5"])
def test_full_size_properties(gpu_ctx, ptx, orc, cfg):
    c = ptx.CONFIGS[cfg]
    s, p = setup(gpu_ctx, ptx, c["scene"], c["width"], c["height"], spp=c["sample_count"], bounces=c["max_path_length"])
    ts = ptx.scene.timestamps(2)
    # (1) primary hits: literal == pruned over the whole image, and == oracle on sampled rows
    gpu_ctx.set_option("traversal", 0); t0, tri0 = gpu_ctx.primary_hits()
    for trav in (1, 2, 3):
        gpu_ctx.set_option("traversal", trav); t1, tri1 = gpu_ctx.primary_hits()
        assert (tri0 == tri1).all() and (bits(t0) == bits(t1)).all(), trav
    # (2) one frame: the default (leaf-SAH tree + pruning + memo, wave pipeline) == the reference's every-ray algorithm
    #     (literal traversal of the reference tree, no memo, one ray at a time), whole image, bit for bit
    gpu_ctx.set_option("traversal", 0); gpu_ctx.set_option("memo", 0); gpu_ctx.set_option("pipeline", 0)
    gpu_ctx.reset_accum(); gpu_ctx.render_frames(ts[:1]); lit = gpu_ctx.read_sum()
    gpu_ctx.set_option("traversal", 2); gpu_ctx.set_option("memo", 1); gpu_ctx.set_option("pipeline", 1)
    gpu_ctx.reset_accum(); gpu_ctx.render_frames(ts[:1]); sah = gpu_ctx.read_sum()
    assert (bits(lit) == bits(sah)).all()
    gpu_ctx.set_option("traversal", 3)
    gpu_ctx.reset_accum(); gpu_ctx.render_frames(ts[:1]); w3 = gpu_ctx.read_sum()
    assert (bits(lit) == bits(w3)).all()
    gpu_ctx.set_option("traversal", 2); gpu_ctx.set_option("pipeline", 2)              # the default pipeline (shared leaf queue)
    gpu_ctx.reset_accum(); gpu_ctx.render_frames(ts[:1]); sah = gpu_ctx.read_sum()
    assert (bits(lit) == bits(sah)).all()
    gpu_ctx.set_option("traversal", 3)
    gpu_ctx.reset_accum(); gpu_ctx.render_frames(ts[:1]); fa = gpu_ctx.read_sum()
    assert (bits(lit) == bits(fa)).all()
    # (3) linearity of the accumulation: frames {a,b} together == a + b
    gpu_ctx.reset_accum(); gpu_ctx.render_frames(ts[1:]); fb = gpu_ctx.read_sum()
    gpu_ctx.reset_accum(); gpu_ctx.render_frames(ts); both = gpu_ctx.read_sum()
    assert (bits(both) == bits((fa + fb).astype(np.float32))).all()
    assert 0 <= both.min() and both.max() <= 2.0                         # each frame is clamped to [0,1] (glsl:527)
    # (4) the oracle on an evenly spread row sample of the same full-size frame
    step = c["height"] // 6
    rows, _ = orc.render_rows(s, p, int(ts[0]), 3, c["height"], step)
    ys = list(range(3, c["height"], step))
    assert (bits(fa[ys]) == bits(rows[ys])).all()
    _, tri_o, _ = orc.primary_hits(s, ptx.config_params(c["scene"], c["width"], c["height"])) if cfg == "C2" else (None, None, None)
    if tri_o is not None:
        assert (tri_o == tri1).all()


@pytest.mark.parametrize("pipeline", [0, 1, 2])
def test_overlapped_one_frame_calls_equal_one_call(gpu_ctx, ptx, orc, pipeline):
    """Successive ptx_render_frames calls alternate between two launch slots (option "overlap"): a loop of one-frame calls that
    never synchronises (the reference's frame loop, main.js:400-405) must leave the same bits as one call with all frames --
    the sum stays in frame order -- also through the 8-bit compat blend, a mid-loop read-back, and the no-alloc entry point."""
    name = "my_cornell_6"
    s, p = setup(gpu_ctx, ptx, name, 72, 40, spp=3, bounces=4, pipeline=pipeline, compat_rgba8=2)
    ts = ptx.scene.timestamps(9)
    gpu_ctx.set_option("overlap", 0)
    gpu_ctx.render_frames(ts)
    want, want8 = gpu_ctx.read_sum(), gpu_ctx.read_compat_rgba8()
    oracle, _ = orc.render_frames(s, p, ts)
    assert (bits(want) == bits(oracle)).all()
    for overlap in (1, 0):
        gpu_ctx.set_option("overlap", overlap)
        gpu_ctx.reset_accum()
        for f, t in enumerate(ts):
            if f == 6:
                gpu_ctx.render_frames_noalloc([t], first_frame_number=f)       # slot 0 + the static scratch, behind slot 1's work
            else:
                gpu_ctx.render_frames([t], first_frame_number=f)
            if f == 3:
                assert gpu_ctx.frames_done == 4 and (bits(gpu_ctx.read_sum()) == bits(orc.render_frames(s, p, ts[:4])[0])).all()
        assert (bits(gpu_ctx.read_sum()) == bits(want)).all() and (gpu_ctx.read_compat_rgba8() == want8).all()
        gpu_ctx.reset_accum()
        gpu_ctx.render_frames(ts[:5]); gpu_ctx.render_frames(ts[5:], first_frame_number=5)            # uneven calls
        assert (bits(gpu_ctx.read_sum()) == bits(want)).all()
    gpu_ctx.set_option("overlap", 1); gpu_ctx.set_option("compat_rgba8", 0)
