"""Best-t pruning (option "traversal" bit 0) against the unpruned traversal -- the reference never prunes
(shaders/fragmentShaderPathTracing.glsl:207-249), so pruning must be harmless for EVERY ray.

Three statements, each with its test (DESIGN.md "Pruning and exactness"):
  * option "prune" = 1 is provable; this is its machine check: ptx_selfcheck_pruning generates adversarial rays on the device
    and traces each with and without pruning, >= 1e8 rays per BASELINE scene over eight families (grazing hits on large
    triangles, rays skimming bounding-box faces, rays in a triangle's own plane, (nearly) axis-parallel rays, rays towards
    vertices and edges, ...): (t bits, triangle id) agree for every one; so do whole frames through every pipeline.
  * the default margin rule (option "prune" = 2) has no proof.  Round 2's form of it (option "margin_lat" = 0) is NOT exact: the
    same generator finds rays whose closest hit it loses (family 3, grazing a large triangle from afar, and on bedroom2_v5 most
    other families) -- kept as a known-answer test of the checker -- and so did a render: C5's whole job (256 frames of 3840 x 2160,
    12 bounces, 726 G rays, tools/whole_job_check.py) came out with six pixel-frames of 2.1e9 different from the unpruned walk, all
    of them rays that skim the lower edge of one lamp panel of pixar_room with a direction component below 4e-3.  Round 3's form
    (lateral allowance, ptx_kernels.h PTX_MARGIN_LAT_PPM) gets those six frames right and leaves the generator nothing to find in
    1e8 rays per scene; both are tested here.
  * the whole headline job (64 frames, 44 G rays) is compared here between the default, the provable and the unpruned walk, every
    pixel of every frame bit for bit (the whole jobs of the other BASELINE configurations: profiles/r03_whole_job_check.json; one
    frame of each in tests/test_gpu_parity.py::test_full_size_properties); and primary rays are never pruned at all."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCENES = ["cornell", "my_cornell_6", "bedroom2_v5", "teapot", "pixar_room"]
FAMILIES = range(8)
RAYS_PER_FAMILY = 12_500_000          # x 8 families = 1e8 rays per scene


def describe(r):
    return "\n".join(f"  o={o} d={d} t_exact={te!r} t_pruned={tp!r} tri_exact={ke} tri_pruned={kp}" for o, d, te, tp, ke, kp in r["first"])


@pytest.fixture
def ctx(gpu_ctx):
    yield gpu_ctx
    for k, v in dict(pipeline=2, traversal=3, prune=2, count=0, margin_lat=3000).items():      # the library defaults
        gpu_ctx.set_option(k, v)


@pytest.mark.parametrize("name", SCENES)
def test_1e8_adversarial_rays_agree_with_and_without_provable_pruning(ptx, ctx, name):
    ctx.upload_scene(ptx.load_scene(name))
    for k, v in dict(pipeline=2, traversal=3, prune=1).items():      # leaf-SAH tree, provable pruning
        ctx.set_option(k, v)
    total = hits = 0
    for fam in FAMILIES:
        r = ctx.selfcheck_pruning(fam, RAYS_PER_FAMILY, seed=20260 + fam)
        assert r["rays"] == RAYS_PER_FAMILY
        assert r["mismatches"] == 0, f"{name} family {fam}: {r['mismatches']} of {r['rays']} rays differ\n" + describe(r)
        total += r["rays"]
        hits += r["hits"]
    assert total >= 100_000_000 and hits > total // 4       # the rays really meet geometry


@pytest.mark.parametrize("name", SCENES)
def test_anyhit_light_bound_certificate(ptx, ctx, name):
    """The any-hit rule for light-visibility rays (option "anyhit") can only go wrong by ending a ray on a blocker although an
    emitter is hit nearer; for that, light_bound must exclude an emitter at the distance where it IS the closest hit.
    ptx_selfcheck_pruning checks exactly that for every ray whose unpruned closest hit is an emitter; families 8 .. 11 aim at
    the emitters (towards points on them, grazing them from afar, skimming their boxes, at their vertices and edges).
    Provable rule: no violation in >= 5e7 rays per scene.  Margin rule: the violations it does find are counted -- the same
    caveat as for pruning (rays constructed to graze an emitter), stated in include/ptx.h; none on the rendered frames
    (tests/test_gpu_parity.py::test_anyhit_light_visibility and every parity test, which run with the rule on)."""
    s = ptx.load_scene(name)
    if s.L == 0:
        pytest.skip("no emitters")
    ctx.upload_scene(s)
    for prune in (1, 2):
        for k, v in dict(pipeline=2, traversal=3, prune=prune).items():
            ctx.set_option(k, v)
        emitter_hits = violations = 0
        for fam in list(range(8, 12)) * 2 + [0, 1]:
            r = ctx.selfcheck_pruning(fam, 5_000_000, seed=4242 + fam + 100 * prune + emitter_hits % 97)
            emitter_hits += r["emitter_hits"]
            violations += r["anyhit_violations"]
        assert emitter_hits > 1_000_000, (name, prune, emitter_hits)      # the rays really end on emitters
        if prune == 1:
            assert violations == 0, (name, violations, emitter_hits)
        else:
            print(f"{name}: margin rule, {violations} of {emitter_hits} adversarial emitter hits excluded by the light bound")
            assert violations < emitter_hits // 1000, (name, violations, emitter_hits)


@pytest.mark.parametrize("name", SCENES)
@pytest.mark.parametrize("pipeline", [0, 2])               # 48-B records + scalar test / pair records + packed math
def test_reference_tree_and_other_arithmetic(ptx, ctx, name, pipeline):
    ctx.upload_scene(ptx.load_scene(name))
    ctx.set_option("pipeline", pipeline)
    ctx.set_option("prune", 1)
    for traversal in (1, 3):
        ctx.set_option("traversal", traversal)
        for fam in FAMILIES:
            r = ctx.selfcheck_pruning(fam, 1_000_000, seed=77 + fam)
            assert r["mismatches"] == 0, f"{name} traversal {traversal} family {fam}\n" + describe(r)


@pytest.mark.parametrize("pipeline", [0, 1, 2])
def test_provable_pruning_frames_bit_exact(ptx, orc, ctx, pipeline):
    for name, traversal in (("my_cornell_6", 1), ("pixar_room", 3), ("bedroom2_v5", 3), ("teapot", 1)):
        s, p = ptx.load_scene(name), ptx.config_params(name, 64, 40, 3, 5)
        ctx.upload_scene(s)
        ctx.set_params(p)
        for k, v in dict(pipeline=pipeline, traversal=traversal, prune=1).items():
            ctx.set_option(k, v)
        ctx.reset_accum()
        ts = ptx.scene.timestamps(2)
        ctx.render_frames(ts)
        want, _ = orc.render_frames(s, p, ts)
        assert (ctx.read_sum().view(np.uint32) == want.view(np.uint32)).all(), (name, traversal)


def test_default_rule_on_the_whole_headline_job_64_frames(ptx, ctx):
    """The north_star job itself -- pixar_room 1920x1080, 64 frames x 16 spp, 8 bounces, 44 G reference rays -- frame by frame
    (a differing frame cannot hide in a sum): the default margin rule, the provable rule and the unpruned walk give the same
    bits in every pixel of every frame."""
    c = ptx.CONFIGS["headline"]
    ctx.upload_scene(ptx.load_scene(c["scene"]))
    ctx.set_params(ptx.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"]))
    ctx.set_option("pipeline", 2)
    rays = 0
    for f0 in range(0, c["frames"], 8):
        imgs = {}
        for mode, opts in (("default", dict(traversal=3, prune=2)), ("provable", dict(traversal=3, prune=1)), ("unpruned", dict(traversal=2, prune=2))):
            for k, v in opts.items():
                ctx.set_option(k, v)
            per_frame = []
            for f in range(f0, f0 + 8):
                ctx.reset_accum()
                ctx.reset_counters()
                ctx.render_frames([1000 + f])
                per_frame.append(ctx.read_sum())
                rays += ctx.counters()["rays"] if mode == "default" else 0
            imgs[mode] = per_frame
        for k in range(8):
            assert (imgs["default"][k].view(np.uint32) == imgs["unpruned"][k].view(np.uint32)).all(), f0 + k
            assert (imgs["provable"][k].view(np.uint32) == imgs["unpruned"][k].view(np.uint32)).all(), f0 + k
    assert rays > 40_000_000_000


def test_primary_rays_are_never_pruned(ptx, orc, ctx):
    """The per-pixel primary-hit cache comes from the unpruned walk whatever the options say (k_primary<0>)."""
    name = "pixar_room"
    s, p = ptx.load_scene(name), ptx.config_params(name, 96, 54)
    ctx.upload_scene(s)
    ctx.set_params(p)
    ctx.set_option("count", 0)
    t_o, tri_o, _ = orc.primary_hits(s, p)
    for traversal, prune in ((3, 2), (3, 1), (1, 2), (2, 2)):
        ctx.set_option("traversal", traversal)
        ctx.set_option("prune", prune)
        t, tri = ctx.primary_hits()
        assert (tri == tri_o).all() and (t.view(np.uint32) == t_o.view(np.uint32)).all()


def test_round_2_margin_is_not_exact_and_the_checker_sees_it(ptx, ctx):
    """margin_lat = 0 (round 2's rule) loses grazing hits on large triangles: the generator must find some (else the zeros above
    and below mean nothing), and they stay rare even among adversarial rays."""
    found = 0
    for name in ("teapot", "pixar_room", "bedroom2_v5"):
        ctx.upload_scene(ptx.load_scene(name))
        for k, v in dict(traversal=3, prune=2, margin_lat=0).items():
            ctx.set_option(k, v)
        r = ctx.selfcheck_pruning(3, RAYS_PER_FAMILY, seed=20263)
        found += r["mismatches"]
        assert r["mismatches"] < r["rays"] // 5_000
    assert found > 0


@pytest.mark.parametrize("name", SCENES)
def test_default_margin_with_lateral_allowance_leaves_the_generator_nothing(ptx, ctx, name):
    """The default rule (margin_lat = 3000 ppm of the scene's diagonal per unit of max|1/d_k|): 0 differing rays in 1e8 adversarial
    rays per scene, the same rays that break round 2's rule.  Not a proof (prune = 1 has one) -- the measure of what is left."""
    ctx.upload_scene(ptx.load_scene(name))
    for k, v in dict(pipeline=2, traversal=3, prune=2).items():
        ctx.set_option(k, v)
    for fam in FAMILIES:
        r = ctx.selfcheck_pruning(fam, RAYS_PER_FAMILY, seed=20260 + fam)
        assert r["mismatches"] == 0, f"{name} family {fam}: {r['mismatches']} of {r['rays']} rays differ\n" + describe(r)


def test_c5_frames_that_round_2s_margin_got_wrong(ptx, ctx):
    """Known answers from a real job: frames 52, 107, 125, 162, 168, 170 of C5 (pixar_room 3840 x 2160, 12 bounces) each have ONE pixel
    where round 2's margin rule and the unpruned walk differ -- a light ray towards the lower edge of a lamp panel, |d_y| < 4e-3,
    accepted by the shader's triangle test 3e-5 .. 2e-4 beside the neighbouring triangle, whose leaf box the ray enters 0.017 .. 0.055
    later (tools/dbg/find_diff.py).  The default rule renders all six like the unpruned walk; with margin_lat = 0 the two frames
    whose ray fails in every pipeline still show the difference (frame 52's depends on the order the leaves are tested in)."""
    c = ptx.CONFIGS["C5"]
    ctx.upload_scene(ptx.load_scene(c["scene"]))
    ctx.set_params(ptx.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"]))
    ctx.set_option("pipeline", 2)

    def frame(f, **opts):
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.reset_accum()
        ctx.render_frames([1000 + f], first_frame_number=f)
        return np.ascontiguousarray(ctx.read_sum()).view(np.uint32)

    for f, pixel in ((52, None), (107, (2075, 183)), (125, None), (162, None), (168, None), (170, (1408, 522))):
        want = frame(f, traversal=2, prune=2, margin_lat=3000)
        assert (frame(f, traversal=3, prune=2, margin_lat=3000) == want).all(), f
        if pixel is not None:
            bad = np.argwhere((frame(f, traversal=3, prune=2, margin_lat=0) != want).any(axis=2))
            assert bad.tolist() == [list(pixel)], (f, bad.tolist())


def test_provable_anyhit_with_the_emitter_list(ptx, orc, ctx):
    """Round 3: under the provable rule, scenes with at most 32 emissive triangles bound each emitter's determinant for the ray
    at hand (option "emit_list", default on; ptx_kernels.h light_bound) instead of by the shader's 1e-4 cut-off -- which on
    pixar_room (a lamp of 19 triangles with edges of a scene unit) had switched the any-hit rule off.  Same bits and reference
    counters as the oracle with the list on and off; with it, light rays do end early on pixar_room."""
    name = "pixar_room"
    s = ptx.load_scene(name)
    p = ptx.config_params(name, 96, 54, 4, 6)
    ts = ptx.scene.timestamps(2)
    want, oc = orc.render_frames(s, p, ts)
    ctx.upload_scene(s); ctx.set_params(p)
    visits = {}
    for emit_list in (0, 1):
        for k, v in dict(pipeline=2, traversal=3, prune=1, count=1, emit_list=emit_list).items():
            ctx.set_option(k, v)
        ctx.reset_accum(); ctx.reset_counters()
        ctx.render_frames(ts)
        assert (np.ascontiguousarray(ctx.read_sum()).view(np.uint32) == want.view(np.uint32)).all(), emit_list
        c = ctx.counters()
        for key in ("samples", "rays", "ray_hits", "light_attempts"):
            assert c[key] == oc[key], (key, emit_list)
        visits[emit_list] = c["node_visits"]
    ctx.set_option("emit_list", 1)
    assert visits[1] < 0.8 * visits[0], visits


def test_whole_job_checker_on_the_smallest_job(ptx, ctx, monkeypatch, tmp_path):
    """tools/whole_job_check.py (the tool behind profiles/r03_whole_job_check.json) on C1, the one-frame cornell job: runs, finds no
    differing frame under any mode, against either reference."""
    import importlib.util
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("whole_job_check", os.path.join(root, "tools", "whole_job_check.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    for extra in ([], ["--mode", "provable"], ["--mode", "unpruned", "--ref-literal"]):
        monkeypatch.setattr(sys, "argv", ["whole_job_check.py", "C1"] + extra)
        assert tool.main() == 0
    out = json.load(open(os.path.join(root, "gpurun_out", "whole_job_check_C1.json")))
    assert out["frames_compared"] == 1 and out["differing_frames"] == [] and out["reference_rays"] > 5_000_000


def test_margin_lat_option_range(ptx, ctx):
    for bad in (-1, 1_000_001):
        with pytest.raises(ptx.PtxError):
            ctx.set_option("margin_lat", bad)
    ctx.set_option("margin_lat", 0)
    ctx.set_option("margin_lat", 1_000_000)


def test_axis_parallel_ray_that_starts_in_two_bounding_planes(ptx, orc, ctx):
    """Known answer found by 1e10 adversarial rays (round 3): d = (-1, 0, 0) from a point that lies in a y plane AND a z plane of
    the boxes above its target.  0 * inf = NaN in two slabs makes the shader's slab test report tNear = tFar = +inf, a hit; a best-t
    limit of 3e38 "before the first hit" cut that box and the walk returned a hit 2.4 farther.  Every mode and pipeline must give the
    literal walk's answer (a +inf limit until there is a hit, and always for rays with an infinite 1/d component)."""
    s = ptx.load_scene("bedroom2_v5")
    ctx.upload_scene(s)
    o = np.array([[-2.4672701358795166, -2.413858413696289, 5.96727180480957]], dtype=np.float32)
    d = np.array([[-1.0, 0.0, 0.0]], dtype=np.float32)
    t_o, tri_o, _ = orc.trace_closest(s, o, d)
    assert int(tri_o[0]) == 1979 and abs(float(t_o[0]) - 2.0480468) < 1e-6
    for pipeline in (0, 1, 2):
        for opts in (dict(traversal=3, prune=2), dict(traversal=3, prune=2, margin_lat=0), dict(traversal=1, prune=2), dict(traversal=3, prune=1),
                     dict(traversal=2, prune=2), dict(traversal=0, prune=2)):
            ctx.set_option("margin_lat", 3000)
            ctx.set_option("pipeline", pipeline)
            for k, v in opts.items():
                ctx.set_option(k, v)
            t, tri = ctx.trace_closest(o, d)
            assert int(tri[0]) == int(tri_o[0]) and t.view(np.uint32)[0] == t_o.view(np.uint32)[0], (pipeline, opts, float(t[0]), int(tri[0]))
    # the generator's family of axis-parallel rays at the size that found it
    for k, v in dict(pipeline=2, traversal=3, prune=2, margin_lat=3000).items():
        ctx.set_option(k, v)
    r = ctx.selfcheck_pruning(6, 1_250_000_000, seed=777006)
    assert r["mismatches"] == 0, describe(r)
